#!/usr/bin/env python3
"""Diagnostic: table build / rebuild times on the device (map update, CDDT, GiantLUT)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyracecarsimulator_amd import range_libc, workloads, maps
for name, g in (("colombia", maps.load_colombia()), ("maze2049", workloads.cfg2().gmap), ("maze4096", workloads.cfg5().gmap)):
    omap = range_libc.PyOMap(g)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); omap.update(g.occ); ts.append(time.perf_counter() - t)
    print("%s %dx%d: map update (upload + EDT + bit-pack), host wall: %.3f ms" % (name, g.rows, g.cols, 1e3 * min(ts)))
    m = range_libc.PyCDDTCast(omap, 300, 112)
    ins = maps.sample_free_poses(g, 1081, 1); outs = np.empty(1081, np.float32)
    t = time.perf_counter(); m.calc_range_many(ins, outs); t1 = time.perf_counter() - t
    omap.update(g.occ)
    t = time.perf_counter(); m.calc_range_many(ins, outs); t2 = time.perf_counter() - t
    t = time.perf_counter(); m.calc_range_many(ins, outs); t3 = time.perf_counter() - t
    print("   CDDT theta_disc=112: first build+scan %.2f ms, rebuild+scan %.2f ms, scan only %.3f ms" % (1e3 * t1, 1e3 * t2, 1e3 * t3))
