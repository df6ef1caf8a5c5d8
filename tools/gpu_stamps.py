#!/usr/bin/env python3
"""Diagnostic: per-wave start/end stamps of the stream kernel (where does a launch's time go?)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyracecarsimulator_amd import range_libc, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--poses", type=int, default=4096)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
w = workloads.cfg2(a.poses)
omap = range_libc.PyOMap(w.gmap)
poses = workloads.make_poses(w, dt=omap.distance_transform())
n, B = len(poses), w.num_rays
d_poses = torch.from_numpy(poses).cuda()
d_out = torch.empty(n * B, dtype=torch.float32, device="cuda")
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
for kv in a.opt:
    k, v = kv.split("="); m.set_option(k, int(v))
m.set_option("debug_stamps", 1)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    m.calc_range_fan_device(d_poses.data_ptr(), n, w.fov, B, d_out.data_ptr(), stream=st)
torch.cuda.synchronize()
s = m.debug_stamps()
t0 = s[:, 0].astype(np.int64); t1 = s[:, 1].astype(np.int64)
it = (s[:, 2] & 0xFFFFFFFF).astype(np.int64); sv = (s[:, 2] >> 32).astype(np.int64)
K = ((s[:, 3] >> 8) & 0xFFFFFF).astype(np.int64)
base = t0.min()
us = lambda x: (x - base) / 100.0
print("waves", len(s), "kernel span %.1f us (last end - first start)" % us(t1.max()))
print("start: p50 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(us(t0), [50, 99, 100])))
print("end  : p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(us(t1), [10, 50, 90, 99, 100])))
dur = (t1 - t0) / 100.0
print("wave duration: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(dur, [10, 50, 90, 100])))
print("services/wave mean %.1f ; chunks per workgroup mean %.2f" % (sv.mean(), K.mean()))
band = (s[:, 3] & 0xFF).astype(np.int64)
for b in range(int(band.max()) + 1):
    k = band == b
    print(" band %d: waves %d end p50 %.1f max %.1f us, chunks/WG mean %.1f" % (b, k.sum(), np.median(us(t1[k])), us(t1[k]).max(), K[k].mean()))

# drain phase (diagnostics build of the kernel: hit cells / step counts requested): how long after
# its stream ran dry does a wave live, and how many dependent samples is its longest ray chain?
hits = np.empty((n * B, 2), np.int32); steps = np.empty(n * B, np.uint16); out = np.empty(n * B, np.float32)
m.calc_range_fan(poses, out, w.fov, B, hits, steps)
s = m.debug_stamps()
t0 = s[:, 0].astype(np.int64); t1 = s[:, 1].astype(np.int64)
td = (s[:, 3] >> 32).astype(np.int64)           # drain start - wave start, 10 ns ticks
ds = (s[:, 2] & 0xFFFFFFFF).astype(np.int64)    # longest per-lane sample chain after drain start
ok = (td > 0) & (ds > 0)
drain_us = ((t1 - t0) - td)[ok] / 100.0
print("AUX launch: span %.1f us; drain phase per wave: p50 %.1f p90 %.1f max %.1f us; chain p50 %d p90 %d max %d samples"
      % ((t1.max() - t0.min()) / 100.0, *np.percentile(drain_us, [50, 90, 100]), *np.percentile(ds[ok], [50, 90, 100])))
long = ok & (ds >= 30)
ns = ((t1 - t0) - td)[long] * 10.0 / ds[long]
print("ns per dependent sample in drain (waves with chains >= 30): p10 %.0f p50 %.0f p90 %.0f   (n=%d)"
      % (*np.percentile(ns, [10, 50, 90]), long.sum()))
late = long & (t1 >= np.percentile(t1, 99))
if late.any():
    print("  ... of the last 1 %% of waves to finish: p50 %.0f ns, chains p50 %d (n=%d)"
          % (np.median(((t1 - t0) - td)[late] * 10.0 / ds[late]), np.median(ds[late]), late.sum()))
