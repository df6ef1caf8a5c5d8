#!/usr/bin/env python3
"""Diagnostic: what does fusing the per-roll-out crash test into the scan cost?  Device-resident
launch sequences, HIP-event timed: plain scan vs rl_check_collision_groups_device (ranges kept /
not written), for a few group sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyracecarsimulator_amd import range_libc, workloads, racecar as RC

for P, group in ((4096, 128), (4096, 4096), (4096, 1), (32768, 128)):
    w = workloads.cfg2(P)
    omap = range_libc.PyOMap(w.gmap); dt = omap.distance_transform()
    poses = workloads.make_poses(w, dt=dt); B = w.num_rays
    m = range_libc.PyRayMarchingGPU(omap, 300)
    if os.environ.get("SCAN_SLOTS"):
        m.set_option("slots", int(os.environ["SCAN_SLOTS"]))
    d_poses = torch.from_numpy(poses).cuda(); d_out = torch.empty(P * B, dtype=torch.float32, device="cuda")
    try:
        edge = RC.edge_distances(B, -w.fov / 2, w.fov / B, 0.275, 0.2032, 0.3302)
    except AttributeError:                                   # an older build under SCANLIB_SO (A/B run)
        edge = np.full(B, 0.15)
    d_edge = torch.from_numpy(edge).cuda(); d_first = torch.empty(P // group, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def t(fn, n=40):
        for _ in range(5): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3

    a = t(lambda: m.calc_range_fan_device(d_poses.data_ptr(), P, w.fov, B, d_out.data_ptr(), stream=st))
    b = t(lambda: m.check_collision_groups_device(d_poses.data_ptr(), P // group, group, w.fov, B, d_edge.data_ptr(), 0.001, d_first.data_ptr(), d_out.data_ptr(), stream=st))
    c = t(lambda: m.check_collision_groups_device(d_poses.data_ptr(), P // group, group, w.fov, B, d_edge.data_ptr(), 0.001, d_first.data_ptr(), 0, stream=st))
    print("P=%d groups of %d: plain %.1f us | fused crash, ranges kept %.1f us (+%.1f %%) | no ranges %.1f us | crashed groups %d/%d"
          % (P, group, a, b, 100 * (b - a) / a, c, int((d_first.cpu().numpy() >= 0).sum()), P // group), flush=True)
