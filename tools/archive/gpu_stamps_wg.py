#!/usr/bin/env python3
"""Diagnostic: when do the workgroups of one cfg2 launch run dry, and how long do their waves drain?
(AUX launch of the stream kernel with debug stamps; 10 ns ticks.)  usage: gpu_stamps_wg.py [opt=value ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyracecarsimulator_amd import range_libc, workloads

w = workloads.cfg2(4096)
omap = range_libc.PyOMap(w.gmap)
poses = workloads.make_poses(w, dt=omap.distance_transform())
n, B = len(poses), w.num_rays
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
for kv in sys.argv[1:]:
    k, v = kv.split("="); m.set_option(k, int(v))
m.set_option("debug_stamps", 1)
hits = np.empty((n * B, 2), np.int32); steps = np.empty(n * B, np.uint16); out = np.empty(n * B, np.float32)
for _ in range(3):
    m.calc_range_fan(poses, out, w.fov, B, hits, steps)
s = m.debug_stamps()
t0 = s[:, 0].astype(np.int64); t1 = s[:, 1].astype(np.int64)
base = t0.min()
td = (s[:, 3] >> 32).astype(np.int64)            # drain start - wave start
ds = (s[:, 2] & 0xFFFFFFFF).astype(np.int64)     # longest chain after drain start
dry = (t0 - base + td) / 100.0                   # absolute time the wave learned the stream is dry
end = (t1 - base) / 100.0
W = 16
nwg = len(s) // W
dry_wg = dry.reshape(nwg, W); end_wg = end.reshape(nwg, W); ds_wg = ds.reshape(nwg, W)
print("launch span %.1f us, %d workgroups" % (end.max(), nwg))
print("stream dry (first wave of a workgroup to notice): p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us"
      % tuple(np.percentile(dry_wg.min(1), [10, 50, 90, 99, 100])))
print("last wave of a workgroup to notice:                p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us"
      % tuple(np.percentile(dry_wg.max(1), [10, 50, 90, 99, 100])))
print("workgroup end:                                      p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us"
      % tuple(np.percentile(end_wg.max(1), [10, 50, 90, 99, 100])))
late = np.argsort(end_wg.max(1))[-8:]
for g in late:
    k = np.argmax(end_wg[g])
    print("  wg %4d: dry %.1f..%.1f, ends %.1f; its last wave: dry at %.1f, chain %d samples, %.0f ns per sample"
          % (g, dry_wg[g].min(), dry_wg[g].max(), end_wg[g].max(), dry_wg[g, k], ds_wg[g, k],
             (end_wg[g, k] - dry_wg[g, k]) * 1000.0 / max(ds_wg[g, k], 1)))
sm = steps.reshape(n, B).astype(np.int64)
print("samples per ray: mean %.2f p99 %d max %d; per pose: mean min %.1f p50 %.1f p90 %.1f max %.1f"
      % (sm.mean(), np.percentile(sm, 99), sm.max(), sm.mean(1).min(), np.median(sm.mean(1)), np.percentile(sm.mean(1), 90), sm.mean(1).max()))
