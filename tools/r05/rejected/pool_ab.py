"""Band pool A/B (round 5, lead (a)): lone launches of cfg2 / other batch sizes and maps, kernel-only time (HIP events
around the march), option pool = 0 / 10 / 15 / 25 / 40, every output bit-equal to a one-ray-per-lane launch; SCANLIB_SO
selects the library (the build without the pool code for the regression check)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads

def one(wl, n, pools, gm=8):
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
    m.set_option("timing", 2); m.set_option("slots", 2); m.set_option("grid_mult", gm)
    for pool in pools:
        try:
            m.set_option("pool", pool)
        except Exception:
            if pool:
                continue
        d_o.fill_(-1.0)
        ks = []
        for _ in range(60):
            m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
            ks.append(m.last_kernel_ms())
        torch.cuda.synchronize()
        ok = bool(torch.equal(d_o, d_ref))
        ks = np.array(ks[8:]) * 1e3
        print("%s %s %6d poses grid_mult %d pool %2d: march %7.1f us (p10 %.1f p90 %.1f) %s %s" % (
            os.environ.get("SCANLIB_SO", "cur")[-12:], wl, n, gm, pool, np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90),
            "bit-equal" if ok else "DIFFERS", m.last_plan()["name"][-28:]), flush=True)

P = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,10,15,25,40,0".split(","))]
one("cfg2", 4096, P)
one("cfg2", 4096, P[:3], gm=3)
one("cfg2", 2048, P)
one("cfg2", 8192, P)
one("cfg5", 8192, P[:4])
one("cfg4", 4096, P[:4])
