"""lone cfg2 launches (kernel-only) of whatever library SCANLIB_SO selects: 4096 / 1024 poses, two rays per lane"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads
w = workloads.cfg2()
omap = range_libc.PyOMap(w.gmap)
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
B = w.num_rays
dt = omap.distance_transform()
for n in (4096, 1024):
    poses = workloads.make_poses(w, dt=dt, n_poses=n)
    d_p = torch.from_numpy(poses).cuda()
    d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
    m.set_option("timing", 2)
    for slots in (1, 2):
        m.set_option("slots", slots)
        for k, v in [kv.split("=") for kv in sys.argv[1:]]:
            m.set_option(k, int(v))
        ks = []
        for _ in range(60):
            m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
            ks.append(m.last_kernel_ms())
        torch.cuda.synchronize()
        ks = np.array(ks[8:]) * 1e3
        print("%s n %d slots %d %s: %.1f us" % (os.environ.get("SCANLIB_SO", "cur"), n, slots, sys.argv[1:], np.median(ks)), flush=True)
