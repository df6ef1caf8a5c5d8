"""The N>1 path with the PRODUCT in it, on the one GPU of the test box: two ranks (cuda:0 each, gloo)
scan their pose blocks with libscan_amd.so and all-gather the ranges; what every rank holds must be
the oracle's scan of the whole batch in global pose order (SURVEY.md §8e "1-GPU output == N-GPU
gathered output").  Also: ``bench.py --gpus 2`` started bare launches its own ranks."""
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


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_two_ranks_gather_the_oracle_scan_in_global_pose_order(oracle_mod, tmp_path):
    n_total, B = 600, 1081
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path), str(n_total), str(B)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    g = maps.make_maze(512, cell=40, wall=3, p=0.45, seed=17, origin=(1.0, -2.0, 0.25))
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    from pyracecarsimulator_amd import range_libc
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, 300)
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
        for rank in range(2):
            got = np.load(os.path.join(str(tmp_path), "rank%d_step%d.npy" % (rank, k)))
            assert got.shape == one.shape
            assert np.array_equal(got, one), "rank %d step %d: gathered ranges differ from the unsharded scan" % (rank, k)
            assert np.abs(got - clean).max() < 0.2 and np.abs((got - clean).std() - 0.02) < 2e-3


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
    assert "all-gather ranges" in d["config"]["gather"]
    assert d["gather_bytes_per_step"] == 4 * 512 * 1081 * 2
    assert d["crash_mode"]["value"] > 0


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
