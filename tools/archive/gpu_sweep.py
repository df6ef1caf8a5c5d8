#!/usr/bin/env python3
"""Tuning sweep on the GPU box: kernel time of the cfg2 workload (or --workload) for a grid of
kernel options.  Prints one line per configuration (ms from HIP events, Mrays/s)."""
import argparse
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from pyracecarsimulator_amd import range_libc, workloads  # noqa: E402


def time_cfg(meth, d_poses, n, fov, B, d_out, reps=20, warm=3):
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warm):
        meth.calc_range_fan_device(d_poses.data_ptr(), n, fov, B, d_out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        meth.calc_range_fan_device(d_poses.data_ptr(), n, fov, B, d_out.data_ptr(), stream=stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--poses", type=int, default=0)
    ap.add_argument("--grid", default="full")
    a = ap.parse_args()
    w = workloads.CONFIGS[a.workload]()
    if a.poses:
        w.n_poses = a.poses
    omap = range_libc.PyOMap(w.gmap)
    dt = omap.distance_transform()
    poses = workloads.make_poses(w, dt=dt)
    n, B = len(poses), w.num_rays
    d_poses = torch.from_numpy(poses).cuda()
    d_out = torch.empty(n * B, dtype=torch.float32, device="cuda")
    meth = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
    ref = None
    rows = []
    if a.grid == "full":
        combos = [dict(variant=0, grid_mult=8)]
        combos += [dict(variant=1, low_water=lw, wg_threads=nt, grid_mult=8)
                   for nt, lw in itertools.product((256, 512, 1024), (0, 16, 24, 32, 40, 48))]
        combos += [dict(variant=1, low_water=32, wg_threads=1024, grid_mult=gm) for gm in (4, 6, 7, 16)]
        combos += [dict(variant=1, low_water=32, wg_threads=1024, grid_mult=8, sort_poses=0),
                   dict(variant=1, low_water=32, wg_threads=1024, grid_mult=8, sort_poses=1, xcd_bands=1),
                   dict(variant=1, low_water=32, wg_threads=1024, grid_mult=8, xcd_bands=16),
                   dict(variant=1, low_water=32, wg_threads=1024, grid_mult=8, xcd_bands=8)]
    elif a.grid == "inline":
        combos = [dict(variant=1, inline_prep=i, low_water=lw, xcd_bands=b)
                  for i, lw, b in itertools.product((0, 1), (16, 24, 32), (8, 1))]
    elif a.grid == "prio":
        combos = [dict(variant=1, low_water=lw, wg_threads=1024, grid_mult=8, drain_prio=dp)
                  for lw, dp in itertools.product((16, 24), (0, 1, 0, 1))]
    elif a.grid == "tiled":
        combos = [dict(variant=1, tiled=t, low_water=lw) for lw, t in itertools.product((24, 32), (0, 1, 0, 1))]
    elif a.grid == "runs":
        combos = [dict(variant=1, inline_prep=0, run_log2=r) for r in (0, 1, 2, 3, 4, 5, 0, 2, 4)]
    elif a.grid == "nosort":
        combos = [dict(variant=1), dict(variant=1, inline_max=1000000), dict(variant=1, inline_max=1000000, xcd_bands=1),
                  dict(variant=1, inline_max=512, sort_poses=0), dict(variant=1, inline_max=512, sort_poses=0, xcd_bands=1),
                  dict(variant=1, inline_max=512, sort_poses=1, xcd_bands=8)]
    elif a.grid == "lw":
        combos = [dict(variant=1, low_water=lw) for lw in (0, 2, 4, 6, 8, 10, 12, 16, 4, 8, 12)]
    elif a.grid == "stripe":
        combos = [dict(variant=1, stripe_max=v) for v in (0, 8192, 0, 8192)]
    elif a.grid == "oi":
        combos = [dict(variant=1, order_inline=v) for v in (0, 1, 0, 1)]
    elif a.grid == "gm":
        combos = [dict(variant=1, grid_mult=g_, low_water=lw) for g_ in (4, 6, 7, 8, 10, 12, 16) for lw in (12,)]
        combos += [dict(variant=1, grid_mult=8, low_water=lw) for lw in (6, 8, 16, 20)]
    elif a.grid == "small":
        combos = [dict(variant=0, grid_mult=8)]
        combos += [dict(variant=1, low_water=lw, wg_threads=nt, grid_mult=8)
                   for nt, lw in itertools.product((256, 1024), (16, 32))]
    else:
        combos = [json.loads(a.grid)]
    for c in combos:
        for k, v in c.items():
            meth.set_option(k, v)
        med, mn = time_cfg(meth, d_poses, n, w.fov, B, d_out)
        got = d_out.cpu().numpy()
        if ref is None:
            ref = got.copy()
        same = bool(np.array_equal(ref, got))
        rows.append((med, c))
        print("%8.4f ms (min %8.4f)  %9.1f Mrays/s  same=%s  %s" %
              (med, mn, n * B / med / 1e3, same, json.dumps(c)), flush=True)
    best = min(rows, key=lambda r: r[0])
    print("BEST %.4f ms %s" % (best[0], json.dumps(best[1])))


if __name__ == "__main__":
    main()
