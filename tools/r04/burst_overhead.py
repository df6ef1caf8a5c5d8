"""Diagnostics: what a timed burst of bench.py costs beyond its kernels — the fixed host/runtime latencies of the
bracket (event records, first launch, device synchronisation) at 0, 1, 4, 20 steps.  Run twice: plain and with
HSA_ENABLE_INTERRUPT=0 (polled completion signals)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pyracecarsimulator_amd import range_libc, workloads
from pyracecarsimulator_amd.distributed import ShardedScan
from pyracecarsimulator_amd.pipeline import concurrent_streams

dev = torch.device("cuda", 0)
w = workloads.cfg2()
omap = range_libc.PyOMap(w.gmap, device=0)
meth = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
n, B = 4096, 1081
streams = concurrent_streams(4)
meth.set_option("grid_mult", 3); meth.set_option("slots", 2)
for kv in sys.argv[1:]:
    k, v = kv.split("="); meth.set_option(k, int(v))
dt = omap.distance_transform()
d_poses = [torch.from_numpy(workloads.make_poses(w, dt=dt, n_poses=n, seed=2 + 7919 * k)).to(dev) for k in range(4)]
scan = ShardedScan(n, B, dev, n_chunks=1, gather=False, streams=streams)
scan.bind(meth, [t.data_ptr() for t in d_poses], w.fov)
print("plan:", meth.plan_fan(n, B)["name"], meth.plan_fan(n, B)["binning"], "HSA_ENABLE_INTERRUPT=", os.environ.get("HSA_ENABLE_INTERRUPT"))
for _ in range(20): scan.step()
scan.finish(); torch.cuda.synchronize()

def burst(steps):
    e0 = torch.cuda.Event(enable_timing=True); ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
    for e in [e0] + ends: e.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    for _ in range(steps): scan.step()
    t1 = time.perf_counter()
    scan.finish(ends)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    dev_ms = max(e0.elapsed_time(e) for e in ends)
    return (t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, dev_ms * 1e3
for steps in (0, 1, 2, 4, 8, 20, 40, 100):
    rs = np.array([burst(steps) for _ in range(41)])
    med = np.median(rs, axis=0)
    print("steps %3d: wall %7.1f us (enqueue %6.1f, event records %5.1f, synchronize %6.1f)  device span %7.1f us  | per step %6.2f us" % (
        steps, med[0], med[1], med[2], med[3], med[4], med[0] / max(steps, 1)))
