"""lone launches (kernel-only, library events around the march) of cfg2 / cfg5-shard / cfg4 shapes for a list of option sets:
python tools/r06/ab_lone.py "tail_pct=0" "tail_pct=25" "tail_pct=25,tail_wg_pct=100" ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads
sets = [dict((k, int(v)) for k, v in (kv.split("=") for kv in s.split(",") if kv)) for s in sys.argv[1:]] or [{}]
for wname, n_list in (("cfg2", (4096, 8192, 32768)), ("cfg5", (32768,)), ("cfg4", (131072,))):
    w = workloads.CONFIGS[wname]()
    omap = range_libc.PyOMap(w.gmap)
    dt = omap.distance_transform()
    B = w.num_rays
    for n in n_list:
        poses = workloads.make_poses(w, dt=dt, n_poses=n)
        d_p = torch.from_numpy(poses).cuda()
        d_o = torch.empty(n * B, dtype=torch.float32, device="cuda")
        ref = None
        for rep in range(2):
            for opts in sets:
                m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
                m.set_option("timing", 2)
                for k, v in opts.items():
                    m.set_option(k, v)
                ks = []
                for _ in range(40 if n <= 8192 else 12):
                    m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
                    ks.append(m.last_kernel_ms())
                torch.cuda.synchronize()
                out = d_o.clone()
                if ref is None:
                    ref = out
                same = bool(torch.equal(out, ref))
                pl = m.last_plan()
                print("%s n %6d %-40s grid %4d gen1 %4d code %d: %7.1f us  (min %.1f)  same %s" % (
                    wname, n, opts, pl["grid"], pl["gen1"], pl["code"], np.median(ks[4:]) * 1e3, np.min(ks[4:]) * 1e3, same), flush=True)
                m.close()
    omap.close()
