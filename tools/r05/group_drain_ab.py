"""Group drain (2 / 4 lanes per ray, 8 / 16 samples per round trip) A/B: lone launches of cfg2 / cfg2 at other batch
sizes / cfg4 shard / cfg5 shard, kernel-only time, option group_drain 0 / 1, every output bit-equal to a
one-ray-per-lane launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads

def one(wl, n, gms=(8,)):
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
    m.set_option("timing", 2); m.set_option("slots", 2)
    for gm in gms:
        for gd in (0, 2, 4, 8, 16, 0):
            m.set_option("group_drain", gd); m.set_option("grid_mult", gm)
            d_o.fill_(-1.0)
            ks = []
            for _ in range(60):
                m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
                ks.append(m.last_kernel_ms())
            torch.cuda.synchronize()
            ok = bool(torch.equal(d_o, d_ref))
            ks = np.array(ks[8:]) * 1e3
            print("%s %6d poses grid_mult %d group_drain %d: march %7.1f us (p10 %.1f p90 %.1f) %s" % (
                wl, n, gm, gd, np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90), "bit-equal" if ok else "DIFFERS"), flush=True)
    omap.close() if hasattr(omap, "close") else None

one("cfg2", 4096, (8,))
one("cfg2", 1024)
one("cfg2", 200)
