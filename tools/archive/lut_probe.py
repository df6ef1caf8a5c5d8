#!/usr/bin/env python3
"""Diagnostic: GiantLUT fan-query time vs spatial spread of the poses (TLB / DRAM-page effects)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyracecarsimulator_amd import range_libc, workloads
w = workloads.cfg3()
omap = range_libc.PyOMap(w.gmap)
dt = omap.distance_transform()
m = range_libc.PyGiantLUTCast(omap, w.max_range_px, w.theta_disc)
B = w.num_rays
st = torch.cuda.current_stream().cuda_stream
for rows in (2000, 500, 100, 20, 2):
    g2 = type(w.gmap)(w.gmap.occ.copy(), w.gmap.resolution, w.gmap.origin, "sub")
    sub = g2.occ.copy(); sub[rows:, :] = 1           # poses only from the first `rows` rows
    g2.occ = sub
    from pyracecarsimulator_amd import maps
    poses = maps.sample_free_poses(g2, 65536, 4, 2.0, dt)
    for sort in (0, 1):
        p = poses
        if sort:
            cell = np.floor((p[:, 1] - w.gmap.origin[1]) / 0.05).astype(np.int64) * 2000 + np.floor((p[:, 0] - w.gmap.origin[0]) / 0.05).astype(np.int64)
            p = np.ascontiguousarray(p[np.argsort(cell)])
        d_p = torch.from_numpy(p).cuda(); d_o = torch.empty(len(p) * B, dtype=torch.float32, device="cuda")
        for _ in range(3): m.calc_range_fan_device(d_p.data_ptr(), len(p), w.fov, B, d_o.data_ptr(), stream=st)
        torch.cuda.synchronize(); ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m.calc_range_fan_device(d_p.data_ptr(), len(p), w.fov, B, d_o.data_ptr(), stream=st); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print("pose rows < %4d (LUT span %6.0f MB) sorted=%d: %.4f ms" % (rows, rows * 2000 * 2884 / 1e6, sort, np.median(ts)), flush=True)
