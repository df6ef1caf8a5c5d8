"""Lone cfg2 launch (4096 poses x 1081 beams) over grid sizes beyond the persistent 2 workgroups per CU: with more
workgroups than slots the hardware dispatcher hands a freed slot to the next pending workgroup — dynamic balancing at
workgroup granularity for free — against more prologues and more ragged workgroup ends.  Kernel-only time (library
events around the march kernel), median of 40."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
w = workloads.CONFIGS[wl]()
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
for slots in (1, 2):
    for gm in (4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64):
        m.set_option("slots", slots); m.set_option("grid_mult", gm)
        ks = []
        for _ in range(45):
            m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
            ks.append(m.last_kernel_ms())
        torch.cuda.synchronize()
        pl = m.last_plan()
        ok = bool(torch.equal(d_o, d_ref))
        ks = np.array(ks[5:]) * 1e3
        print("%s %5d poses  slots %d grid_mult %2d -> grid %4d lds %6d k_max %4d : kernel %6.1f us (p10 %.1f p90 %.1f)  %s  %s" % (
            wl, n, slots, gm, pl["grid"], pl["lds_bytes"], pl["k_max"], np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90),
            "bit-equal" if ok else "DIFFERS", pl["name"][-28:]))
