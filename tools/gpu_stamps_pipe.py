#!/usr/bin/env python3
"""Diagnostic: where does a workgroup's life go when launches are PIPELINED (bench.py's schedule: 4 batches in
flight on 4 concurrent streams, three rays per lane, grid_mult 3)?  Per-wave stamps of the LAST launch on each
stream: kernel entry -> prologue done -> stream dry -> end."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyracecarsimulator_amd import range_libc, workloads
from pyracecarsimulator_amd.pipeline import concurrent_streams

ap = argparse.ArgumentParser()
ap.add_argument("--poses", type=int, default=4096)
ap.add_argument("--pipeline", type=int, default=4)
ap.add_argument("--grid-mult", type=int, default=3)
ap.add_argument("--slots", type=int, default=3)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
w = workloads.cfg2(a.poses)
omap = range_libc.PyOMap(w.gmap)
dt = omap.distance_transform()
P = a.pipeline
batches = [workloads.make_poses(w, dt=dt, seed=w.pose_seed + 7919 * k) for k in range(P)]
n, B = a.poses, w.num_rays
d_poses = [torch.from_numpy(b).cuda() for b in batches]
d_out = [torch.empty(n * B, dtype=torch.float32, device="cuda") for _ in range(P)]
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
m.set_option("grid_mult", a.grid_mult)
m.set_option("slots", a.slots)
for kv in a.opt:
    k, v = kv.split("="); m.set_option(k, int(v))
streams = concurrent_streams(P)
for _ in range(8):
    for k in range(P):
        m.calc_range_fan_device(d_poses[k].data_ptr(), n, w.fov, B, d_out[k].data_ptr(), stream=streams[k % len(streams)].cuda_stream)
torch.cuda.synchronize()
m.set_option("debug_stamps", 1)
for i in range(a.steps):
    k = i % P
    m.calc_range_fan_device(d_poses[k].data_ptr(), n, w.fov, B, d_out[k].data_ptr(), stream=streams[k % len(streams)].cuda_stream)
torch.cuda.synchronize()
s = m.debug_stamps()            # the last launch (its stream's context)
print(m.last_plan()["name"], "grid", m.last_plan()["grid"], "waves", len(s))
ent, end, pro, dry = (s[:, i].astype(np.int64) for i in range(4))
base = ent.min()
us = lambda x: (x - base) / 100.0
life = (end - ent) / 100.0
print("launch span %.1f us (first entry -> last end); wave entry p50 %.1f p90 %.1f max %.1f us" % (us(end.max()), *np.percentile(us(ent), [50, 90, 100])))
print("wave lifetime p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(life, [10, 50, 90, 100])))
print("prologue (entry -> records in LDS) p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile((pro - ent) / 100.0, [50, 90, 100])))
has = dry > 0
main = np.where(has, (dry - pro) / 100.0, (end - pro) / 100.0)
drain = np.where(has, (end - dry) / 100.0, 0.0)
print("main phase (prologue done -> stream dry) p50 %.1f p90 %.1f us; drain (stream dry -> end) p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (
    *np.percentile(main, [50, 90]), *np.percentile(drain, [50, 90, 99, 100])))
wg_end = end.reshape(-1, 16).max(axis=1); wg_ent = ent.reshape(-1, 16).min(axis=1)
wg_first_done = end.reshape(-1, 16).min(axis=1)
wg_dry = np.where(has.reshape(-1, 16).any(axis=1), np.where(has, dry, end.max() * 2).reshape(-1, 16).min(axis=1), wg_end)
print("workgroups %d: lifetime p50 %.1f p90 %.1f us; stream dry -> workgroup end p50 %.1f p90 %.1f max %.1f us; "
      "first wave done -> last wave done p50 %.1f p90 %.1f us" % (
          len(wg_end), *np.percentile((wg_end - wg_ent) / 100.0, [50, 90]),
          *np.percentile((wg_end - wg_dry) / 100.0, [50, 90, 100]), *np.percentile((wg_end - wg_first_done) / 100.0, [50, 90])))
tot_slot_time = (wg_end - wg_ent).sum() / 100.0
idle = ((wg_end[:, None] - end.reshape(-1, 16))).sum() / 100.0 / 16
print("wave-slot time held after the wave itself ended (waiting for the workgroup's last wave): %.1f %% of the workgroups' lifetime"
      % (100.0 * idle / tot_slot_time))
print("share of a workgroup's life: prologue %.1f %%, stream dry -> end %.1f %%" % (
    100.0 * np.median((pro - ent).reshape(-1, 16).max(axis=1) / (wg_end - wg_ent)), 100.0 * np.median((wg_end - wg_dry) / (wg_end - wg_ent))))
