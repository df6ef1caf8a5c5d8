"""cfg3 CDDT (65 536 poses x 1081 beams, theta_disc 112 / 108): the theta-major search kernel of round 4 (every 8-lane
group prepares its own look-ups) against round 5's (look-ups prepared once per pose, picked up with ds_bpermute):
whole step and kernel-only time, serial launches; outputs compared bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads
w = workloads.cfg3()
omap = range_libc.PyOMap(w.gmap)
dt = omap.distance_transform()
n, B = 65536, w.num_rays
poses = workloads.make_poses(w, dt=dt, n_poses=n)
d_p = torch.from_numpy(poses).cuda()
d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
d_ref = torch.empty(n * B, dtype=torch.float32, device="cuda")
for td in (112, 108):
    m = range_libc.PyCDDTCast(omap, w.max_range_px, td)
    m.set_option("cddt_search", 0)
    m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_ref.data_ptr()); torch.cuda.synchronize()
    m.set_option("timing", 1)
    for search in (0, 1, 2, 1, 2):
        m.set_option("cddt_search", search)
        d_o.fill_(-1.0)
        ks = []
        for _ in range(30):
            m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
            ks.append(m.last_kernel_ms())
        torch.cuda.synchronize()
        ks = np.array(ks[5:]) * 1e3
        print("cfg3 CDDT theta_disc %d search kernel %d: step %.1f us (p10 %.1f p90 %.1f) = %.1f Grays/s  %s  grid %d" % (
            td, search, np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90), n * B / np.median(ks) / 1e3,
            "bit-equal" if bool(torch.equal(d_o, d_ref)) else "DIFFERS", m.last_plan()["grid"]), flush=True)
    m.close()
