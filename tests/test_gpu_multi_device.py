"""In-process multi-device handles (rl_map_create_multi, include/scanlib.h): ONE process, one handle, the
host-pointer batch cut into contiguous pose blocks, one per device — what a drop-in caller of
ScanSimulator2D.scanMany / checkCollisionMany (/root/reference/scripts/scan_simulator.py:113-135,
scripts/racecar_simulator_v2.py:146-167, one Python process: scripts/mcts.py:237) needs to use the other
GPUs of the node.  With ONE visible MI355X (the gpurun box) the device list names device 0 several times (N
contexts, N worker threads, N streams on one GPU); with two or more visible devices the SAME tests run over
DISTINCT devices (up to 8: ``_devs``) — hipSetDevice discipline, portable pinned blocks written by kernels of
several devices, per-device worker threads — without a code change.  Every result must be bit-identical to the
single-device call — ranges, noise (global ray ids), crash indices — and to the CPU oracle."""

import numpy as np
import pytest

from pyracecarsimulator_amd import _lib, maps, range_libc
from pyracecarsimulator_amd import racecar as RC

pytestmark = pytest.mark.gpu

FOV, B, MRX = 4.71, 1081, 300


def _devs(k=3):
    """Device list of a multi-device fixture: >= 2 visible devices -> distinct devices, min(visible, 8) of them for
    the main fixture (k = 3) and min(visible, k) otherwise; one visible device -> device 0 named k times."""
    n = _lib.lib().rl_device_count()
    if n >= 2:
        return list(range(min(n, 8 if k == 3 else k)))
    return [0] * k


def _distinct():
    return _lib.lib().rl_device_count() >= 2


@pytest.fixture(scope="module", autouse=True)
def _gpu(need_gpu):
    yield


@pytest.fixture(scope="module")
def world():
    g = maps.make_maze(640, cell=40, wall=3, p=0.45, seed=23, origin=(-3.0, 2.0, 0.3))
    one = range_libc.PyOMap(g, device=0)
    multi = range_libc.PyOMap(g, device=_devs(3))
    yield g, one, multi
    multi.close()
    one.close()


def _split(m, per_device=64):
    """A multi-device method that really cuts these (small) test batches: the default brings a device in per 512 poses."""
    m.set_option("multi_min_poses", per_device)
    return m


def _edge():
    return RC.edge_distances(B, -FOV / 2.0, FOV / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])


def test_multi_map_shape_and_replicas(world):
    g, one, multi = world
    L = _lib.lib()
    nd = len(_devs(3))
    assert L.rl_map_n_devices(one._h) == 1 and L.rl_map_n_devices(multi._h) == nd
    assert L.rl_map_rows(multi._h) == g.rows and L.rl_map_cols(multi._h) == g.cols and L.rl_map_device(multi._h) == 0
    assert L.rl_map_replica(multi._h, nd) is None and L.rl_map_replica(multi._h, nd - 1) is not None
    if _distinct():          # every replica lives on its own device
        assert [L.rl_map_device(L.rl_map_replica(multi._h, i)) for i in range(nd)] == _devs(3)
    assert np.array_equal(multi.distance_transform(), one.distance_transform())


@pytest.mark.parametrize("cls", [range_libc.PyRayMarchingGPU, range_libc.PyRayMarching, range_libc.PyBresenhamsLine])
def test_fan_blocks_equal_the_single_device_scan_and_the_oracle(world, oracle_mod, cls):
    g, one, multi = world
    m1, mm = cls(one, MRX), _split(cls(multi, MRX))
    assert mm.n_devices == len(_devs(3)) and m1.n_devices == 1
    om = oracle_mod.OracleMap.from_gridmap(g, MRX)
    assert mm.get_info("multi_min_poses") == 64 and cls(multi, MRX).get_info("multi_min_poses") == 512     # (default)
    for n in (1, 5, 63, 200, 1000):                      # below multi_min_poses x devices: fewer blocks (1, 1, 1, 3, 3)
        poses = maps.sample_free_poses(g, n, 100 + n)
        a, b = np.empty(n * B, np.float32), np.full(n * B, -7.0, np.float32)
        m1.calc_range_fan(poses, a, FOV, B)
        mm.calc_range_fan(poses, b, FOV, B)
        assert np.array_equal(a, b), (cls.__name__, n)
        if cls is range_libc.PyBresenhamsLine:
            want = om.bl_fan(poses, FOV, B, nthreads=oracle_mod.max_threads())[0]
        elif cls is range_libc.PyRayMarching:            # range_libc's CPU RayMarching: the upstream-literal arithmetic
            want = om.rm_fan_libm(poses, FOV, B, step_coeff=0.999)[0]
        else:
            want = om.rm_fan(poses, FOV, B, step_coeff=1.0, nthreads=oracle_mod.max_threads())[0]
        assert np.array_equal(b, want)
    # diagnostics travel per block too
    poses = maps.sample_free_poses(g, 333, 9)
    a, b = np.empty(333 * B, np.float32), np.empty(333 * B, np.float32)
    ha, hb = np.empty((333 * B, 2), np.int32), np.empty((333 * B, 2), np.int32)
    sa, sb = np.empty(333 * B, np.uint16), np.empty(333 * B, np.uint16)
    m1.calc_range_fan(poses, a, FOV, B, hit_cells=ha, steps=sa)
    mm.calc_range_fan(poses, b, FOV, B, hit_cells=hb, steps=sb)
    assert np.array_equal(a, b) and np.array_equal(ha, hb) and np.array_equal(sa, sb)
    mm.close()
    m1.close()


def test_reference_call_forms_on_a_multi_device_handle(world, oracle_mod):
    """The fork's sparse 4-argument calc_range_many (pose p in row p * num_rays) and upstream's 2-argument
    per-ray form through a multi-device handle; pinned result blocks are written by every device directly."""
    g, one, multi = world
    m1, mm = range_libc.PyRayMarchingGPU(one, MRX), _split(range_libc.PyRayMarchingGPU(multi, MRX))
    n = 700
    poses = maps.sample_free_poses(g, n, 4)
    ins = np.zeros((n * B, 3), np.float32)
    ins[::B] = poses
    a = np.empty(n * B, np.float32)
    b = _lib.pinned_zeros(n * B, np.float32)              # rl_host_alloc: zero-copy stores from every device
    m1.calc_range_many(ins, a, FOV, B)
    mm.calc_range_many(ins, b, FOV, B)
    assert np.array_equal(a, b)
    rays = np.ascontiguousarray(np.repeat(poses, 300, axis=0)[:200000])
    rays[:, 2] += np.linspace(-2.0, 2.0, len(rays), dtype=np.float32)
    ra, rb = np.empty(len(rays), np.float32), np.empty(len(rays), np.float32)
    m1.calc_range_many(rays, ra)
    mm.calc_range_many(rays, rb)
    assert np.array_equal(ra, rb)
    # device-pointer entry points refuse the multi handle and point at the replicas
    with pytest.raises(_lib.ScanLibError) as e:
        mm.calc_range_fan_device(1, 1, FOV, B, 1)
    assert "rl_method_replica" in str(e.value)
    assert mm.replica(len(_devs(3)) - 1).n_devices == 1
    with pytest.raises(IndexError):
        mm.replica(3)
    mm.close()
    m1.close()


@pytest.mark.parametrize("cls,td", [(range_libc.PyCDDTCast, 108), (range_libc.PyGiantLUTCast, 180)])
def test_table_methods_on_a_multi_device_map(world, oracle_mod, cls, td):
    """The bandwidth-bound variants build their table on EVERY replica (lazily, on the first block a device gets)
    and answer like the single-device handle and the oracle; options set on the handle reach every replica."""
    g, one, multi = world
    small = maps.make_maze(96, cell=12, wall=2, p=0.5, seed=3)
    o1, om_ = range_libc.PyOMap(small, device=0), range_libc.PyOMap(small, device=_devs(2))
    m1, mm = cls(o1, 60, td), cls(om_, 60, td)
    mm.set_option("multi_min_poses", 8)
    assert mm.get_info("multi_min_poses") == 8 and mm.get_info("n_devices") == 2 and m1.get_info("n_devices") == 1
    mm.set_option("grid_mult", 5)
    assert mm.replica(0).get_info("grid_mult") == 5 == mm.replica(1).get_info("grid_mult")
    orc = oracle_mod.OracleMap.from_gridmap(small, 60)
    poses = maps.sample_free_poses(small, 100, 8)
    a, b = np.empty(100 * 360, np.float32), np.empty(100 * 360, np.float32)
    m1.calc_range_fan(poses, a, 6.0, 360)
    mm.calc_range_fan(poses, b, 6.0, 360)
    want = orc.cddt_fan(td, poses, 6.0, 360) if cls is range_libc.PyCDDTCast else orc.lut_fan(orc.lut_build(td), poses, 6.0, 360)
    assert np.array_equal(a, b) and np.array_equal(b, want)
    assert mm.calc_range(float(poses[0, 0]), float(poses[0, 1]), 0.3) == m1.calc_range(float(poses[0, 0]), float(poses[0, 1]), 0.3)
    # a map update reaches every replica's table
    occ2 = small.occ.copy()
    occ2[40:44, 10:80] = 1
    o1.update(occ2)
    om_.update(occ2)
    m1.calc_range_fan(poses, a, 6.0, 360)
    mm.calc_range_fan(poses, b, 6.0, 360)
    assert np.array_equal(a, b)
    for o in (mm, m1, om_, o1):
        o.close()


def test_noise_is_keyed_by_the_global_ray_id_across_device_blocks(world):
    g, one, multi = world
    m1, mm = range_libc.PyRayMarchingGPU(one, MRX), _split(range_libc.PyRayMarchingGPU(multi, MRX))
    n = 640
    poses = maps.sample_free_poses(g, n, 11)
    for off in (0, 12345678901):
        m1.set_noise(0.02, 77, off)
        mm.set_noise(0.02, 77, off)
        a, b = np.empty(n * B, np.float32), np.empty(n * B, np.float32)
        m1.calc_range_fan(poses, a, FOV, B)
        mm.calc_range_fan(poses, b, FOV, B)
        assert np.array_equal(a, b)
    m1.set_noise(0.0)
    clean = np.empty(n * B, np.float32)
    m1.calc_range_fan(poses, clean, FOV, B)
    assert 0.015 < float((b - clean).std()) < 0.025
    mm.close()
    m1.close()


@pytest.mark.parametrize("cls", [range_libc.PyRayMarchingGPU, range_libc.PyCDDTCast])
def test_crash_indices_are_global(world, cls):
    g, one, multi = world
    args = (MRX, 108) if cls is range_libc.PyCDDTCast else (MRX,)
    m1, mm = cls(one, *args), _split(cls(multi, *args))
    edge = _edge()
    rng = np.random.default_rng(5)
    n = 900
    poses = maps.sample_free_poses(g, n, 21)
    # put a few poses right at a wall so that some roll-outs crash, at known places in different blocks
    occ_r, occ_c = np.nonzero(g.occ)
    for idx in (17, 420, 421, 899):
        k = rng.integers(len(occ_r))
        poses[idx, 0] = g.origin[0] + (occ_c[k] + 0.5) * g.resolution * np.cos(g.origin[2]) - (occ_r[k] + 0.5) * g.resolution * np.sin(g.origin[2])
        poses[idx, 1] = g.origin[1] + (occ_c[k] + 0.5) * g.resolution * np.sin(g.origin[2]) + (occ_r[k] + 0.5) * g.resolution * np.cos(g.origin[2])
    ra, rb = np.empty(n * B, np.float32), np.empty(n * B, np.float32)
    fa = m1.check_collision_many(poses, FOV, B, edge, 0.001, ranges=ra)
    fb = mm.check_collision_many(poses, FOV, B, edge, 0.001, ranges=rb)
    assert fa == fb and np.array_equal(ra, rb)
    assert fa == RC.is_crashed(ra, B, n, edge, 0.001)
    # no crash at all in a slice: -(n + 1)
    free = maps.sample_free_poses(g, 300, 31)
    keep = np.array([i for i in range(300) if RC.is_crashed(_scan(m1, free[i:i + 1]), B, 1, edge, 0.001) < 0][:256])
    assert mm.check_collision_many(free[keep], FOV, B, edge, 0.001) == -(len(keep) + 1) == \
        m1.check_collision_many(free[keep], FOV, B, edge, 0.001)
    for group in (1, 9, 100, 300):
        ga = m1.check_collision_groups(poses, group, FOV, B, edge, 0.001)
        gb = mm.check_collision_groups(poses, group, FOV, B, edge, 0.001)
        assert np.array_equal(ga, gb), group
        want = np.array([RC.is_crashed(ra[q * group * B:(q + 1) * group * B], B, group, edge, 0.001)
                         for q in range(n // group)], np.int32)
        assert np.array_equal(gb, want)
    mm.close()
    m1.close()


def _scan(m, poses):
    out = np.empty(len(poses) * B, np.float32)
    m.calc_range_fan(poses, out, FOV, B)
    return out


def test_rollout_chain_over_device_blocks(world):
    """rl_car_rollout / rl_car_rollout_check with the roll-outs cut over the devices: same poses, same crash
    indices, same final states as one device."""
    g, one, multi = world
    from pyracecarsimulator_amd import workloads as W
    w = W.Workload("t", g, 1, B, FOV, MRX, "RMGPU", 1)
    states, actions = W.rollout_inputs(w, 300, 3)
    c1, cm = RC.CarBatch(device=0), RC.CarBatch(device=_devs(3))
    p1, s1, v1 = c1.rollout(states, actions)
    pm, sm, vm = cm.rollout(states, actions)
    assert np.array_equal(p1, pm) and np.array_equal(s1, sm) and np.array_equal(v1, vm)
    m1, mm = range_libc.PyRayMarchingGPU(one, MRX), _split(range_libc.PyRayMarchingGPU(multi, MRX))
    edge = _edge()
    f1, o1, w1 = c1.rollout_check(m1, states, actions, FOV, B, edge, 0.001)
    fm, om_, wm = cm.rollout_check(mm, states, actions, FOV, B, edge, 0.001)
    assert np.array_equal(f1, fm) and np.array_equal(o1, om_) and np.array_equal(w1, wm)
    assert (f1 >= 0).any() and (f1 < 0).any()            # some roll-outs crash, some do not
    with pytest.raises(_lib.ScanLibError):
        c1.rollout_check(mm, states, actions, FOV, B, edge, 0.001)      # single car, multi method
    for o in (mm, m1, cm, c1):
        o.close()


def test_scan_simulator_and_map_update_on_a_multi_device_map(world):
    """The reference's façade unchanged on a multi-device map: ScanSimulator2D.scanMany (aliasing kept) and the
    two-player style map update reach every replica."""
    from pyracecarsimulator_amd import ScanSimulator2D
    g, one, multi = world
    sims = []
    for omap in (one, multi):
        sim = ScanSimulator2D(B, FOV, 0.01, batch_size=512)
        sim.setMap(omap, MRX, g.resolution, g.origin)
        sim.setRaytracingMethod("RMGPU")
        if omap is multi:
            _split(sim.scan_method)
        sims.append(sim)
    poses = maps.sample_free_poses(g, 512, 77)
    a = sims[0].scanMany(poses).copy()
    b = sims[1].scanMany(poses)
    assert b is sims[1].output_vector_many and np.array_equal(a, b)
    assert np.array_equal(sims[0].scan(*poses[3]).copy(), sims[1].scan(*poses[3]))
    occ2 = g.occ.copy()
    occ2[200:260, 300:304] = 1
    try:
        one.update(occ2)
        multi.update(occ2)
        a = sims[0].scanMany(poses).copy()
        b = sims[1].scanMany(poses).copy()
        assert np.array_equal(a, b)
        assert np.array_equal(multi.distance_transform(), one.distance_transform())
    finally:
        one.update(g.occ)
        multi.update(g.occ)
    for s_ in sims:
        s_.scan_method.close()


def test_multi_device_handle_from_several_threads(world):
    """A multi-device handle shared by threads (the reference's rospy callbacks share one object,
    scripts/ros_interface.py:115,142,189): calls serialise on the handle, results stay right."""
    import threading
    g, one, multi = world
    m1, mm = range_libc.PyRayMarchingGPU(one, MRX), _split(range_libc.PyRayMarchingGPU(multi, MRX))
    poses = [maps.sample_free_poses(g, 400, 500 + t) for t in range(3)]
    want = [_scan(m1, p) for p in poses]
    errs = []

    def work(t):
        try:
            for _ in range(6):
                got = _scan(mm, poses[t])
                if not np.array_equal(got, want[t]):
                    errs.append(t)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs, errs
    mm.close()
    m1.close()


@pytest.mark.parametrize("cls", [range_libc.PyRayMarchingGPU, range_libc.PyRayMarching, range_libc.PyCDDTCast])
def test_device_resident_exchange_lands_in_the_consumer_gpu(world, oracle_mod, cls):
    """rl_calc_range_fan_multi_device / rl_check_collision_groups_multi_device (round 6): every device marches its pose
    block into its own HBM and sends it — chunk by chunk under the next chunk's march, hipMemcpyPeerAsync — to ONE
    consumer device; the caller (one process: scripts/mcts.py:237) finds every range / every roll-out's crash index in
    that GPU's memory.  Bit-identical to the single-device scan, with noise (global ray ids), for every consumer index,
    chunk count and batch size (fewer blocks than devices included).  One visible GPU: device 0 several times (the peer
    copy degenerates to a device-to-device copy); several: distinct devices."""
    torch = pytest.importorskip("torch")
    g, one, multi = world
    args = (112,) if cls is range_libc.PyCDDTCast else ()
    m1, mm = cls(one, MRX, *args), _split(cls(multi, MRX, *args))
    nd = mm.n_devices
    devs = _devs(3)
    for n, chunks, consumer in ((1000, 0, 0), (1000, 3, nd - 1), (200, 1, 1 % nd), (37, 7, nd - 1), (1, 4, 0)):
        poses = maps.sample_free_poses(g, n, 500 + n)
        poses[n // 2] = [np.nan, 0.0, 0.0]
        for noise in (0.0, 0.01):
            for m in (m1, mm):
                m.set_noise(noise, seed=11, ray_offset=12345)
            want = np.empty(n * B, np.float32)
            m1.calc_range_fan(poses, want, FOV, B)
            with torch.cuda.device(devs[consumer]):
                d_out = torch.full((n * B,), -7.0, dtype=torch.float32, device="cuda:%d" % devs[consumer])
                torch.cuda.synchronize()
                mm.calc_range_fan_multi_device(poses, d_out.data_ptr(), FOV, B, consumer=consumer, chunks=chunks)
                got = d_out.cpu().numpy()
            assert np.array_equal(got, want, equal_nan=True), (cls.__name__, n, chunks, consumer, noise, int((got != want).sum()))
    for m in (m1, mm):
        m.set_noise(0.0)
    with pytest.raises(_lib.ScanLibError):
        mm.calc_range_fan_multi_device(poses, 8, FOV, B, consumer=nd)         # not a replica of this handle
    with pytest.raises(_lib.ScanLibError):
        m1.calc_range_fan_multi_device(poses, 8, FOV, B, consumer=0)          # not a multi-device handle
    if cls is not range_libc.PyCDDTCast:
        # the fused crash test: one int32 per roll-out lands in the consumer's memory
        edge = _edge()
        for n_groups, grp, consumer in ((20, 50, 0), (7, 100, nd - 1), (1, 64, 1 % nd)):
            poses = maps.sample_free_poses(g, n_groups * grp, 900 + grp)
            want = m1.check_collision_groups(poses, grp, FOV, B, edge, 0.001)
            with torch.cuda.device(devs[consumer]):
                d_first = torch.full((n_groups,), 12345, dtype=torch.int32, device="cuda:%d" % devs[consumer])
                torch.cuda.synchronize()
                mm.check_collision_groups_multi_device(poses, grp, FOV, B, edge, 0.001, d_first.data_ptr(), consumer=consumer)
                assert d_first.cpu().numpy().tolist() == want.tolist(), (cls.__name__, n_groups, grp, consumer)
    mm.close()
    m1.close()
