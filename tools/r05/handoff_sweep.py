"""Hand-off march (round 5): lone cfg2 launch, kernel-only time (events around main + leftover kernel), over
handoff 0/1, handoff_cap, grid_mult (more workgroups than resident slots tile without a ragged end per generation
when dry workgroups leave at once) and the leftover launch's workgroup size; every output checked bit for bit
against a one-ray-per-lane launch."""
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
m.set_option("timing", 2); m.set_option("slots", 2)

def run(tag, **opts):
    for k, v in opts.items():
        m.set_option(k, v)
    d_o.fill_(-1.0)
    ks = []
    for _ in range(50):
        m.calc_range_fan_device(d_p.data_ptr(), n, w.fov, B, d_o.data_ptr())
        ks.append(m.last_kernel_ms())
    torch.cuda.synchronize()
    ok = bool(torch.equal(d_o, d_ref))
    ks = np.array(ks[8:]) * 1e3
    pl = m.last_plan()
    print("%s %5d poses %-44s grid %4d: march %6.1f us (p10 %.1f p90 %.1f) %s" % (
        wl, n, tag, pl["grid"], np.median(ks), np.percentile(ks, 10), np.percentile(ks, 90), "bit-equal" if ok else "DIFFERS"), flush=True)

run("handoff 0 gm 8", handoff=0, grid_mult=8)
for cap in (8, 16, 32, 64):
    run("handoff 1 cap %d gm 8" % cap, handoff=1, handoff_cap=cap, drain_cap=64, grid_mult=8)
for gm in (6, 10, 12, 16, 24):
    for cap in (16, 32):
        run("handoff 1 cap %d gm %d" % (cap, gm), handoff=1, handoff_cap=cap, grid_mult=gm)
for wg in (64, 128):
    run("handoff 1 cap 16 gm 8 leftover wg %d" % wg, handoff=1, handoff_cap=16, grid_mult=8, handoff_wg=wg)
for gm in (10, 12, 16):
    run("handoff 0 gm %d" % gm, handoff=0, grid_mult=gm)
m.set_option("handoff_wg", 256)
for st in (4, 16):
    run("handoff 1 cap 16 gm 8 drain_stretch %d" % st, handoff=1, handoff_cap=16, grid_mult=8, drain_stretch=st)
