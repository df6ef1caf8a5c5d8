"""Handle lifecycle on the MI355X: a long-running caller (the reference's ROS node builds a new
PyOMap on every map message, /root/reference/scripts/ros_interface.py:202-223, and the two-player
simulator rebuilds its map and CDDT table on every tick, scripts/two_player/rcs_two_player.py:110-121)
must get back every byte of HBM and pinned memory a destroyed handle held."""
import gc

import numpy as np
import pytest

from pyracecarsimulator_amd import _lib, maps, range_libc
from pyracecarsimulator_amd.followgap import PyFollowGap
from pyracecarsimulator_amd.racecar import CarBatch
from pyracecarsimulator_amd.scan_simulator import ScanSimulator2D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(need_gpu):
    yield


def _free_bytes(torch):
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _one_life(g, poses, B, fov, with_big_tables):
    """Everything a caller can create, used once, then dropped."""
    omap = range_libc.PyOMap(g)
    kinds = [(range_libc.PyRayMarchingGPU, ()), (range_libc.PyRayMarching, ()),
             (range_libc.PyBresenhamsLine, ()), (range_libc.PyCDDTCast, (108,))]
    if with_big_tables:
        kinds.append((range_libc.PyGiantLUTCast, (180,)))
    outs = []
    for cls, args in kinds:
        m = cls(omap, 120, *args)
        out = np.empty(len(poses) * B, np.float32)
        m.calc_range_fan(poses, out, fov, B)
        rays = np.repeat(poses[:4], 8, axis=0).astype(np.float32)
        out2 = np.empty(len(rays), np.float32)
        m.calc_range_many(rays, out2)
        outs.append(out[:16].copy())
        m.close()
    # a map update with methods alive (tables rebuilt in place), then the other handle kinds
    m = range_libc.PyCDDTCast(omap, 120, 108)
    occ2 = g.occ.copy()
    occ2[5:9, 5:9] = 1
    omap.update(occ2)
    out = np.empty(len(poses) * B, np.float32)
    m.calc_range_fan(poses, out, fov, B)
    sim = ScanSimulator2D(B, fov, 0.0, batch_size=len(poses))
    sim.setMap(omap, 120, g.resolution, g.origin)
    sim.setRaytracingMethod("RMGPU")
    sim.scanMany(poses.astype(np.float64))
    fg = PyFollowGap(10, 15.0, 0.4189, fov / B)
    fg.eval(out[:B].copy(), B)
    car = CarBatch(device=0)
    pin = _lib.pinned_zeros((len(poses) * B,), np.float32)
    m.calc_range_fan(poses, pin, fov, B)
    del pin, car, fg, sim
    m.close()
    omap.close()
    return outs


def test_create_use_destroy_cycles_return_all_device_memory():
    torch = pytest.importorskip("torch")
    g = maps.make_maze(160, cell=20, wall=2, p=0.4, seed=5, origin=(-1.0, 0.5, 0.2))
    B, fov = 271, 4.71
    poses = maps.sample_free_poses(g, 96, 11)
    first = _one_life(g, poses, B, fov, True)          # warm-up: runtime pools, code objects, torch context
    _one_life(g, poses, B, fov, True)
    gc.collect()
    base = _free_bytes(torch)
    for i in range(40):
        again = _one_life(g, poses, B, fov, i % 8 == 0)
        for a, b in zip(first, again):
            assert np.array_equal(a, b)                # and a fresh handle computes the same bits
    gc.collect()
    leaked = base - _free_bytes(torch)
    # the HIP runtime may keep a pool block or two; a leaked table would be tens of MB after 40 lives
    assert leaked <= 8 << 20, "%.1f MB of device memory not returned after 40 create/destroy cycles" % (leaked / 2**20)


def test_map_update_in_place_does_not_grow_memory():
    """The two-player tick: same map handle, rl_map_update + CDDT rebuild + scan, thousands of times."""
    torch = pytest.importorskip("torch")
    g = maps.make_maze(200, cell=25, wall=2, p=0.4, seed=8)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyCDDTCast(omap, 150, 108)
    rm = range_libc.PyRayMarchingGPU(omap, 150)
    B, fov = 360, 6.2
    poses = maps.sample_free_poses(g, 8, 3)
    out = np.empty(len(poses) * B, np.float32)
    rng = np.random.default_rng(0)

    def tick():
        occ = g.occ.copy()
        r, c = rng.integers(10, 180, 2)
        occ[r:r + 6, c:c + 10] = 1                     # the opponent's car drawn into the grid
        omap.update(occ)
        m.calc_range_fan(poses, out, fov, B)
        rm.calc_range_fan(poses, out, fov, B)

    for _ in range(20):
        tick()
    base = _free_bytes(torch)
    for _ in range(300):
        tick()
    leaked = base - _free_bytes(torch)
    assert leaked <= 4 << 20, "%.1f MB grown over 300 ticks" % (leaked / 2**20)
