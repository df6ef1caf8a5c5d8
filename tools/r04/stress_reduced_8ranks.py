"""Diagnostics: the 8-rank (one device, gloo) crash / steer worker of tests/test_gpu_dist.py, repeated; prints every
(rank, slot, row) whose gathered values differ from the unsharded result, block by block."""
import os, socket, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pyracecarsimulator_amd import maps, range_libc, racecar as RC

def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

world, mode, reps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
n_total, B, GROUP = 640, 1081, 20
g = maps.make_maze(512, cell=40, wall=3, p=0.45, seed=17, origin=(1.0, -2.0, 0.25))
omap = range_libc.PyOMap(g); m = range_libc.PyRayMarchingGPU(omap, 300)
edge = RC.edge_distances(B, -4.71 / 2.0, 4.71 / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
want = [m.check_collision_groups(maps.sample_free_poses(g, n_total, 5 + k), GROUP, 4.71, B, edge, 0.001) for k in range(2)]
env = dict(os.environ); env["PYTHONPATH"] = ROOT; env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for rep in range(reps):
    d = tempfile.mkdtemp()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), d, str(n_total), str(B), mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode:
        print("rep", rep, "rc", r.returncode, r.stderr[-800:]); continue
    bad = []
    per = len(want[0]) // world
    for rank in range(world):
        for k, rows in ((0, 1), (1, 2)):
            for j in range(rows):
                got = np.load(os.path.join(d, "rank%d_slot%d_row%d.npy" % (rank, k, j)))
                if not np.array_equal(got, want[k]):
                    blocks = [b for b in range(world) if not np.array_equal(got[b * per:(b + 1) * per], want[k][b * per:(b + 1) * per])]
                    bad.append((rank, k, j, blocks, got[:8].tolist()))
    print("rep", rep, "bad:", bad)
