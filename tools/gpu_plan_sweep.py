#!/usr/bin/env python3
"""Evidence for the launch planner's thresholds (csrc/launch_plan.h): lone-launch time of the ray-marching
fan over pose counts 64 .. 65536 on three maps, for the plan the library picks and for the neighbouring
plans it could have picked (options that move a batch across a case boundary).  One line per point:
map, poses, variant, record source / binning of the plan, median us of 15 launches, Grays/s."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyracecarsimulator_amd import maps, range_libc, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--maps", default="colombia,maze2049,maze4096")
ap.add_argument("--counts", default="64,128,256,512,1024,2048,2560,3000,4096,6000,8191,8192,12000,16384,32768,65536")
a = ap.parse_args()
B, fov, mrx = 1081, 4.71, 300
VARIANTS = [
    ("default", {}),
    ("derive-in-LDS, caller order", {"inline_max": 1 << 30, "stripe_max": 0, "order_inline": 0}),
    ("derive-in-LDS, row stripes", {"inline_max": 0, "stripe_max": 8192, "order_inline": 0}),
    ("keys-only binning + LDS", {"inline_max": 0, "stripe_max": 0, "order_inline": 1}),
    ("one-workgroup binning, records", {"inline_prep": 0, "bin_multi_min": 1 << 30}),
    ("grid-wide binning, records", {"inline_prep": 0, "bin_multi_min": 64}),
    ("two rays per lane", {"slots": 2}),
]
DEFAULTS = {"inline_max": 512, "stripe_max": 2560, "order_inline": 1, "inline_prep": 1, "bin_multi_min": 8192, "slots": 0}
for name in a.maps.split(","):
    g = {"colombia": maps.load_colombia, "maze2049": lambda: workloads.cfg2().gmap,
         "maze4096": lambda: workloads.cfg5().gmap}[name]()
    omap = range_libc.PyOMap(g)
    dt = omap.distance_transform()
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    st = torch.cuda.current_stream().cuda_stream
    for n in [int(x) for x in a.counts.split(",")]:
        poses = maps.sample_free_poses(g, n, 11, 2.0, dt)
        d_p = torch.from_numpy(poses).cuda()
        d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
        ref = None
        seen = set()
        for vname, opts in VARIANTS:
            for k, v in DEFAULTS.items():
                m.set_option(k, v)
            for k, v in opts.items():
                m.set_option(k, v)
            pl = m.plan_fan(n, B)
            key = (pl["name"], pl["grid"], pl["binning"], pl["record_source"], pl["run_log2"])
            if key in seen:
                continue
            seen.add(key)
            ts = []
            for _ in range(3):
                m.calc_range_fan_device(d_p.data_ptr(), n, fov, B, d_o.data_ptr(), stream=st)
            torch.cuda.synchronize()
            for _ in range(15):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                m.calc_range_fan_device(d_p.data_ptr(), n, fov, B, d_o.data_ptr(), stream=st)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            got = d_o.clone()
            if ref is None:
                ref = got
            same = bool(torch.equal(ref, got))
            us = float(np.median(ts))
            print("%-9s %6d  %-32s src %d bin %-13s slots %d grid %4d  %8.1f us  %7.1f Grays/s  same=%s" % (
                name, n, vname, pl["record_source"], pl["binning"], pl["slots"], pl["grid"], us, n * B / us / 1e3, same),
                flush=True)
    m.close()
