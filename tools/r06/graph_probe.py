"""Does a hipGraph of a 20-step pipelined burst (four slot streams, bin + march per step) beat launching it step by step?
cfg2, the schedule bench.py uses (grid_mult 3, two rays per lane).  Prints wall-clock medians per burst and checks that the
replayed graph leaves the same ranges."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyracecarsimulator_amd import range_libc, workloads, pipeline

w = workloads.CONFIGS["cfg2"]()
B, n = w.num_rays, 4096
omap = range_libc.PyOMap(w.gmap)
meth = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
meth.set_option("grid_mult", 3); meth.set_option("slots", 2)
dt = omap.distance_transform()
P = 4
streams = pipeline.concurrent_streams(P)
batches = [workloads.rank_poses(w, n, 0, 1, dt=dt, seed=w.pose_seed + 7919 * k, device=0) for k in range(P)]
d_poses = [torch.from_numpy(b).cuda() for b in batches]
outs = [torch.empty(n * B, dtype=torch.float32, device="cuda") for _ in range(P)]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def step(k):
    s = streams[k % P]
    meth.calc_range_fan_device(d_poses[k % P].data_ptr(), n, w.fov, B, outs[k % P].data_ptr(), stream=s.cuda_stream)

def direct():
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(K): step(k)
    torch.cuda.synchronize(); return time.perf_counter() - t

for _ in range(3): direct()
ref = [o.clone() for o in outs]
td = sorted(direct() for _ in range(25))
print("direct : %d steps median %.1f us (%.2f us per step), min %.1f" % (K, td[12] * 1e6, td[12] * 1e6 / K, td[0] * 1e6))

g = torch.cuda.CUDAGraph()
main = torch.cuda.Stream()
try:
    with torch.cuda.graph(g, stream=main):
        for s in streams: s.wait_stream(main)
        for k in range(K): step(k)
        for s in streams: main.wait_stream(s)
except Exception as e:
    print("capture failed:", repr(e)[:300]); sys.exit(0)
for o in outs: o.zero_()
def replay():
    torch.cuda.synchronize(); t = time.perf_counter()
    g.replay()
    torch.cuda.synchronize(); return time.perf_counter() - t
for _ in range(3): replay()
print("graph leaves the same ranges:", all(torch.equal(a, b) for a, b in zip(outs, ref)))
tg = sorted(replay() for _ in range(25))
print("graph  : %d steps median %.1f us (%.2f us per step), min %.1f" % (K, tg[12] * 1e6, tg[12] * 1e6 / K, tg[0] * 1e6))
