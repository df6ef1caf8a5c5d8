"""cfg2 lone launch (and the pipelined shape's grid) over run_log2: how long a workgroup stays on one run of
consecutive 64-ray blocks decides how evenly the static split spreads the heavy-tailed block costs
(CPU replay: per-workgroup samples max / mean 1.20 at single blocks, 1.31 at runs of 8)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = workloads.cfg2(n)
omap = range_libc.PyOMap(w.gmap)
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
B = w.num_rays
dt = omap.distance_transform()
poses = workloads.make_poses(w, dt=dt, n_poses=n)
d_p = torch.from_numpy(poses).cuda()
d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
d_ref = torch.empty(n * B, dtype=torch.float32, device="cuda")
m.set_option("slots", 1); m.set_option("grid_mult", 8)
m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_ref.data_ptr()); torch.cuda.synchronize()
m.set_option("timing", 2)
for gm in (8, 3):
    for slots in (2,):
        for rl in (-1, 0, 1, 2, 3, 4):
            m.set_option("slots", slots); m.set_option("grid_mult", gm); m.set_option("run_log2", rl)
            ks = []
            for _ in range(60):
                m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
                ks.append(m.last_kernel_ms())
            torch.cuda.synchronize()
            pl = m.last_plan()
            ok = bool(torch.equal(d_o, d_ref))
            ks = np.array(ks[8:]) * 1e3
            print("cfg2 %5d poses slots %d grid_mult %d run_log2 %2d -> plan run_log2 %d grid %4d: kernel %6.1f us (p10 %.1f p90 %.1f) %s" % (
                n, slots, gm, rl, pl["run_log2"], pl["grid"], np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90),
                "bit-equal" if ok else "DIFFERS"))
