"""Would tile-ordered poses help the theta-major CDDT search?  cfg3's 65 536 random poses as they are, and the same poses
sorted on the host by map tile (64 cells), through the same launch: device time of the whole step (HIP events) + kernel trace."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads
w = workloads.CONFIGS["cfg3"]()
g = w.gmap
omap = range_libc.PyOMap(g)
dt = omap.distance_transform()
poses = workloads.make_poses(w, dt=dt)
n, B = len(poses), w.num_rays
ox, oy, yaw = g.origin
c, s = np.cos(-yaw), np.sin(-yaw)
x = (poses[:, 0] - ox) / g.resolution
y = (poses[:, 1] - oy) / g.resolution
gx, gy = c * x - s * y, s * x + c * y
for td in (112,):
    m = range_libc.PyCDDTCast(omap, w.max_range_px, td)
    m.set_option("timing", 1)
    d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
    res = {}
    for name, key in (("caller's order (random)", None), ("sorted by 64-cell tile", (gy.astype(np.int64) >> 6) * 64 + (gx.astype(np.int64) >> 6)),
                      ("sorted by 16-cell tile", (gy.astype(np.int64) >> 4) * 256 + (gx.astype(np.int64) >> 4))):
        p = poses if key is None else np.ascontiguousarray(poses[np.argsort(key, kind="stable")])
        d_p = torch.from_numpy(p).cuda()
        ks = []
        for _ in range(25):
            m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
            ks.append(m.last_kernel_ms())
        torch.cuda.synchronize()
        print("theta_disc %d, %-26s: %.1f us per step (prep + search + fan; median of 20), %s" % (td, name, np.median(ks[5:]) * 1e3, m.last_plan()["name"]), flush=True)
