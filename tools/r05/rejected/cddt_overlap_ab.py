"""cfg3 CDDT (65 536 poses x 1081 beams, theta_disc 112): the two-kernel theta-major form in one piece against S pose
slices whose fan stage runs on a second stream beside the next slice's search (option cddt_overlap = S); whole call, HIP
events around it, serial launches; outputs compared bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads
w = workloads.cfg3()
omap = range_libc.PyOMap(w.gmap)
dt = omap.distance_transform()
B = w.num_rays
for n in (65536, 32768):
    poses = workloads.make_poses(w, dt=dt, n_poses=n)
    d_p = torch.from_numpy(poses).cuda()
    d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
    d_ref = torch.empty(n * B, dtype=torch.float32, device="cuda")
    m = range_libc.PyCDDTCast(omap, w.max_range_px, 112)
    m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_ref.data_ptr()); torch.cuda.synchronize()
    for S in (0, 2, 3, 4, 6, 8, 0, 4):
        m.set_option("cddt_overlap", S)
        d_o.fill_(-1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ks = []
        for _ in range(30):
            e0.record()
            m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            e1.record(); e1.synchronize()
            ks.append(e0.elapsed_time(e1))
        ks = np.array(ks[5:]) * 1e3
        print("cfg3 CDDT theta_disc 112, %d poses, %d slices: call %.1f us (p10 %.1f p90 %.1f) = %.1f Grays/s  %s" % (
            n, S, np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90), n * B / np.median(ks) / 1e3,
            "bit-equal" if bool(torch.equal(d_o, d_ref)) else "DIFFERS"), flush=True)
    m.close()
