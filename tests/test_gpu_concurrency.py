"""Concurrency promises of the C ABI (include/scanlib.h "Rules of the boundary"), on the MI355X:
one method handle used from several streams at once (bench.py's pipelined steps), one handle
shared by threads (the reference's rospy timer + subscriber callbacks,
/root/reference/scripts/ros_interface.py:107-115,142,189), a map updated while scans run."""
import threading

import numpy as np
import pytest

from pyracecarsimulator_amd import maps, range_libc
from pyracecarsimulator_amd.pipeline import concurrent_streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(need_gpu):
    yield


@pytest.mark.parametrize("opts", [
    {},                                            # defaults: small map -> records derived in the march kernel
    {"inline_map_kb": 0},                          # big-map policy: keys-only binning launch + inline march
    {"inline_map_kb": 0, "order_inline": 0, "stripe_max": 0},   # binning launch writes records
    {"inline_map_kb": 0, "bin_multi_min": 64},     # grid-wide binning kernels (hist / keys scratch)
    {"grid_mult": 4},                              # half-machine grids: two launches co-resident
    {"grid_mult": 3, "slots": 2},                  # bench.py's pipelined configuration: two rays per lane
])
def test_interleaved_batches_on_concurrent_streams_are_bit_exact(oracle_mod, opts):
    """Batches enqueued round robin on 3 concurrent streams through ONE handle: each stream's
    launch context has its own pose records / tile order, so overlapping launches cannot corrupt
    one another; every batch equals the oracle bit for bit."""
    torch = pytest.importorskip("torch")
    g = maps.make_maze(600, cell=40, wall=3, p=0.45, seed=31, origin=(-4.0, 2.0, 0.3))
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    B, fov = 1081, 4.71
    m = range_libc.PyRayMarchingGPU(omap, 300)
    for k, v in opts.items():
        m.set_option(k, v)
    streams = concurrent_streams(3)
    assert len(streams) >= 2, "this box runs no two streams concurrently"
    sizes = [3000, 2000, 700]                     # a different batch per stream
    batches = [maps.sample_free_poses(g, n, 40 + i, dt=om.dt) for i, n in enumerate(sizes)]
    want = [om.rm_fan(p, fov, B, step_coeff=1.0, nthreads=oracle_mod.max_threads())[0] for p in batches]
    d_poses = [torch.from_numpy(p).cuda() for p in batches]
    d_out = [torch.zeros(len(p) * B, dtype=torch.float32, device="cuda") for p in batches]
    torch.cuda.synchronize()
    for rep in range(12):
        for i in range(len(batches)):
            s = streams[i % len(streams)]
            m.calc_range_fan_device(d_poses[i].data_ptr(), len(batches[i]), fov, B, d_out[i].data_ptr(),
                                    stream=s.cuda_stream)
    torch.cuda.synchronize()
    for i in range(len(batches)):
        assert np.array_equal(d_out[i].cpu().numpy(), want[i]), i
    # grouped crash test on two streams at once (per-stream crash marks)
    edge = oracle_mod.edge_distances(B, -fov / 2, fov / B, 0.275, 0.2032, 0.3302)
    d_edge = torch.from_numpy(edge).cuda()
    firsts = [torch.zeros(len(p) // 100, dtype=torch.int32, device="cuda") for p in batches]
    for rep in range(6):
        for i in range(len(batches)):
            s = streams[i % len(streams)]
            m.check_collision_groups_device(d_poses[i].data_ptr(), len(batches[i]) // 100, 100, fov, B,
                                            d_edge.data_ptr(), 0.001, firsts[i].data_ptr(),
                                            d_out[i].data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    for i, p in enumerate(batches):
        exp = [oracle_mod.is_crashed(want[i][k * 100 * B:(k + 1) * 100 * B], B, 100, edge, 0.001)
               for k in range(len(p) // 100)]
        assert firsts[i].cpu().numpy().tolist() == exp, i


def test_more_streams_than_launch_contexts(oracle_mod):
    """Two streams more than a handle has launch contexts (rl_launch_contexts): contexts are handed over
    after a device synchronisation, results stay exact."""
    torch = pytest.importorskip("torch")
    from pyracecarsimulator_amd import _lib
    n_streams = _lib.lib().rl_launch_contexts() + 2
    g = maps.make_maze(300, cell=30, wall=2, p=0.5, seed=5)
    om = oracle_mod.OracleMap.from_gridmap(g, 200)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarching(omap, 200)
    m.set_option("variant", 1)
    m.set_option("inline_map_kb", 0)
    B, fov = 360, 6.0
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    batches = [maps.sample_free_poses(g, 600 + 50 * i, 70 + i, dt=om.dt) for i in range(n_streams)]
    d_poses = [torch.from_numpy(p).cuda() for p in batches]
    d_out = [torch.zeros(len(p) * B, dtype=torch.float32, device="cuda") for p in batches]
    for rep in range(3):
        for i, s in enumerate(streams):
            m.calc_range_fan_device(d_poses[i].data_ptr(), len(batches[i]), fov, B, d_out[i].data_ptr(),
                                    stream=s.cuda_stream)
    torch.cuda.synchronize()
    for i, p in enumerate(batches):
        assert np.array_equal(d_out[i].cpu().numpy(), om.rm_fan(p, fov, B, nthreads=4)[0]), i


def test_threads_share_one_method_handle(oracle_mod):
    """The rospy pattern: a timer thread calling scan() (one pose, 4-arg calc_range_many) and a
    callback thread calling scanMany() / the fused crash test on the SAME range method object, a few
    thousand calls, every result checked against the oracle."""
    g = maps.load_colombia()
    mrx, B, fov = 300, 1081, 4.71
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    poses = maps.sample_free_poses(g, 64, 9, dt=om.dt)
    want = om.rm_fan(poses, fov, B, step_coeff=1.0, nthreads=4)[0].reshape(len(poses), B)
    edge = oracle_mod.edge_distances(B, -fov / 2, fov / B, 0.275, 0.2032, 0.3302)
    errors = []

    def scan_thread(n_calls):
        ins = np.zeros((B, 3), np.float32)
        outs = np.zeros(B, np.float32)
        try:
            for k in range(n_calls):
                p = k % len(poses)
                ins[0] = poses[p]
                m.calc_range_many(ins, outs, fov, B)          # scan(): scripts/scan_simulator.py:103-106
                if not np.array_equal(outs, want[p]):
                    errors.append(("scan", k))
                    return
        except Exception as e:                                  # noqa: BLE001
            errors.append(("scan", repr(e)))

    def many_thread(n_calls):
        nb = 8
        ins = np.zeros((nb * B, 3), np.float32)
        outs = np.zeros(nb * B, np.float32)
        try:
            for k in range(n_calls):
                p0 = (3 * k) % (len(poses) - nb)
                ins[::B] = poses[p0:p0 + nb]
                m.calc_range_many(ins, outs, fov, B)          # scanMany(): scripts/scan_simulator.py:130-133
                if not np.array_equal(outs, want[p0:p0 + nb].ravel()):
                    errors.append(("scanMany", k))
                    return
                if k % 4 == 0:                                  # checkCollisionMany on the same handle
                    code = m.check_collision_many(poses[p0:p0 + nb], fov, B, edge, 0.001)
                    if code != oracle_mod.is_crashed(want[p0:p0 + nb].ravel(), B, nb, edge, 0.001):
                        errors.append(("crash", k))
                        return
        except Exception as e:                                  # noqa: BLE001
            errors.append(("scanMany", repr(e)))

    threads = [threading.Thread(target=scan_thread, args=(3000,)),
               threading.Thread(target=many_thread, args=(1500,)),
               threading.Thread(target=scan_thread, args=(3000,))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive()
    assert not errors, errors[:3]


def test_map_update_thread_races_scan_threads(oracle_mod):
    """A map callback thread swaps the occupancy between two grids (rl_map_update) while two scan
    threads keep scanning through methods of that map: every scan must equal the oracle on ONE of the
    two grids, never a mixture of old and new tables."""
    g = maps.make_maze(256, cell=32, wall=3, p=0.5, seed=8)
    occ_a = g.occ.copy()
    occ_b = g.occ.copy()
    occ_b[60:200, 120:126] = 1
    occ_b[100:106, 20:230] = 1
    mrx, B, fov = 200, 360, 6.2
    om_a = oracle_mod.OracleMap(occ_a, g.resolution, g.origin, mrx)
    om_b = oracle_mod.OracleMap(occ_b, g.resolution, g.origin, mrx)
    omap = range_libc.PyOMap(g)
    methods = [range_libc.PyRayMarching(omap, mrx), range_libc.PyRayMarchingGPU(omap, mrx)]
    coeff = [0.999, 1.0]
    poses = maps.sample_free_poses(g, 24, 3, dt=np.minimum(om_a.dt, om_b.dt))
    # (PyRayMarching computes the upstream-literal arithmetic, PyRayMarchingGPU the canonical one)
    want = [[(o.rm_fan_libm if c == 0.999 else o.rm_fan)(poses, fov, B, step_coeff=c)[0] for o in (om_a, om_b)] for c in coeff]
    assert not np.array_equal(want[0][0], want[0][1])
    stop = threading.Event()
    errors = []
    seen = [set(), set()]

    def updater():
        k = 0
        while not stop.is_set():
            omap.update(occ_b if k % 2 == 0 else occ_a)
            k += 1

    def scanner(i):
        out = np.empty(len(poses) * B, np.float32)
        try:
            for k in range(400):
                methods[i].calc_range_fan(poses, out, fov, B)
                if np.array_equal(out, want[i][0]):
                    seen[i].add("a")
                elif np.array_equal(out, want[i][1]):
                    seen[i].add("b")
                else:
                    errors.append((i, k))
                    return
        except Exception as e:                                  # noqa: BLE001
            errors.append((i, repr(e)))

    up = threading.Thread(target=updater)
    sc = [threading.Thread(target=scanner, args=(i,)) for i in range(2)]
    up.start()
    for t in sc:
        t.start()
    for t in sc:
        t.join(300)
    stop.set()
    up.join(60)
    assert not errors, errors[:3]
    assert seen[0] == {"a", "b"} or seen[1] == {"a", "b"}     # the updates really interleaved with scans
