"""The reference's own batch sizes (one scan, a 200-pose roll-out, up to 1024 poses) on cfg2's map: lone launch,
kernel-only, one / two rays per lane x group drain 0 / 4 / 8 — the tail of such a launch is one long chain."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads
w = workloads.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]()
omap = range_libc.PyOMap(w.gmap)
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
B = w.num_rays
dt = omap.distance_transform()
for n in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,16,64,200,512,1024".split(","))]:
    poses = workloads.make_poses(w, dt=dt, n_poses=max(n, 2))[:n]
    d_p = torch.from_numpy(np.ascontiguousarray(poses)).cuda()
    d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
    d_ref = torch.empty(n * B, dtype=torch.float32, device="cuda")
    m.set_option("slots", 1); m.set_option("group_drain", 0); m.set_option("timing", 0)
    m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_ref.data_ptr()); torch.cuda.synchronize()
    m.set_option("timing", 2)
    for slots, gd in ((1, 0), (2, 0), (2, 4), (2, 8), (2, 16), (2, 0), (2, 8)):
        m.set_option("slots", slots); m.set_option("group_drain", gd)
        ks = []
        for _ in range(80):
            m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
            ks.append(m.last_kernel_ms())
        torch.cuda.synchronize()
        ks = np.array(ks[10:]) * 1e3
        print("%5d poses slots %d group_drain %d: %6.1f us (p10 %.1f p90 %.1f) %s %s" % (
            n, slots, gd, np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90),
            "bit-equal" if bool(torch.equal(d_o, d_ref)) else "DIFFERS", m.last_plan()["name"][-30:]), flush=True)
