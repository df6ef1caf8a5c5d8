#!/usr/bin/env python3
"""Diagnostic: which waves end a lone cfg2 launch, and when did their workgroup's stream run dry?
(per-wave stamps of the production one-ray-per-lane kernel: start, end, drain start)"""
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
td = (s[:, 3] >> 32).astype(np.int64)
base = t0.min()
end = (t1 - base) / 100.0
dstart = np.where(td > 0, (t0 + td - base) / 100.0, np.nan)
ddur = end - dstart
print("kernel span %.1f us; wave end p50 %.1f p90 %.1f p99 %.1f max %.1f" % (end.max(), *np.percentile(end, [50, 90, 99, 100])))
print("drain start (stream dry) per wave: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.nanpercentile(dstart, [10, 50, 90, 99, 100])))
print("drain duration per wave: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.nanpercentile(ddur, [50, 90, 99, 100])))
order = np.argsort(-end)[:12]
print("the 12 waves that end last: end / drain start / drain duration (us), workgroup")
for i in order:
    print("  %6.1f  %6.1f  %6.1f   wg %d wave %d" % (end[i], dstart[i], ddur[i], i // 16, i % 16))
wg_end = end.reshape(-1, 16).max(axis=1)
wg_dry = np.nanmin(dstart.reshape(-1, 16), axis=1)
print("workgroups: stream dry p50 %.1f p90 %.1f max %.1f us; end p50 %.1f p90 %.1f max %.1f us" % (
    *np.nanpercentile(wg_dry, [50, 90, 100]), *np.percentile(wg_end, [50, 90, 100])))
