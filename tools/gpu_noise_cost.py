#!/usr/bin/env python3
"""What the counter-based Gaussian noise (row a15, cfg5) costs: the cfg5 shard (32 768 poses x 720 beams, 4096^2
maze) with noise off / on, lone launches, HIP-event timed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyracecarsimulator_amd import range_libc, workloads

w = workloads.cfg5()
omap = range_libc.PyOMap(w.gmap)
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
poses = workloads.make_poses(w, n_poses=n)
d_p = torch.from_numpy(poses).cuda()
d_o = torch.empty(n * w.num_rays, dtype=torch.float32, device="cuda")
for std in (0.0, w.noise_std, 0.0, w.noise_std):
    m.set_noise(std, w.noise_seed, 0)
    for _ in range(3):
        m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, w.num_rays, d_o.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, w.num_rays, d_o.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("noise std %.3f: %.4f ms per launch, %.0f Mrays/s  (%s)" % (std, ms, n * w.num_rays / ms / 1e3, m.last_plan()["name"]))
