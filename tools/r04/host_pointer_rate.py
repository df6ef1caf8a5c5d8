"""The PCIe-inclusive rate of the host-pointer API (never bench.py's `value`): rl_calc_range_fan with NumPy poses in,
ranges out, cfg2 map, batches of 200 ... 65536 poses x 1081 beams — result buffer pageable (staged D2H copy) vs in a
pinned block of the library (rl_host_alloc: the kernel stores straight over PCIe), and through a multi-device handle
naming device 0 three times (three contexts sharing the ONE link of this box: what the split itself costs)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyracecarsimulator_amd import _lib, range_libc, workloads

w = workloads.cfg2()
B = 1081
omap = range_libc.PyOMap(w.gmap)
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
multi = range_libc.PyOMap(w.gmap, device=[0, 0, 0])
mm = range_libc.PyRayMarchingGPU(multi, w.max_range_px)
dt = omap.distance_transform()
for n in (200, 512, 1024, 2048, 4096, 16384, 65536):
    poses = workloads.make_poses(w, dt=dt, n_poses=n)
    pageable = np.empty(n * B, np.float32)
    pinned = _lib.pinned_zeros(n * B, np.float32)
    row = []
    mm.set_option("multi_min_poses", 64)
    for meth, name, out, oname, direct in ((m, "one device", pageable, "pageable", None), (m, "one device", pinned, "pinned, kernel stores", 1 << 30),
                                           (m, "one device", pinned, "pinned, DMA", 0), (m, "one device", pinned, "pinned, DMA, no slice overlap", -1),
                                           (mm, "3 contexts", pinned, "pinned, default", None)):
        if True:
            if meth is m:
                meth.set_option("overlap_min_rays", 0 if direct == -1 else 1 << 24)
            if direct is not None:
                meth.set_option("direct_max_rays", max(direct, 0))
            elif meth is m:
                meth.set_option("direct_max_rays", 1 << 20)
            for _ in range(3): meth.calc_range_fan(poses, out, w.fov, B)
            ts = []
            for _ in range(12):
                t = time.perf_counter(); meth.calc_range_fan(poses, out, w.fov, B); ts.append(time.perf_counter() - t)
            t = float(np.median(ts))
            row.append("%s/%s %.0f us = %.1f Grays/s (%.1f GB/s of ranges)" % (name, oname, t * 1e6, n * B / t / 1e9, n * B * 4 / t / 1e9))
    assert np.array_equal(pageable, pinned)
    print("%6d poses: " % n + " | ".join(row))
