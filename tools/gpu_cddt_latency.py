#!/usr/bin/env python3
"""CDDT on the GPU box: (a) the two-player per-tick path (scripts/two_player/rcs_two_player.py:110-124:
stamp the opponent into the map, rebuild PyOMap + PyCDDTCast, one 1081-ray scan) as the host sees it;
(b) the fan kernel on a cfg2-size batch, per-bin kernel vs per-ray kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyracecarsimulator_amd import maps, range_libc, workloads

g = maps.load_colombia()
omap = range_libc.PyOMap(g)
m = range_libc.PyCDDTCast(omap, 300, 112)
dt = omap.distance_transform()
pose = maps.sample_free_poses(g, 1, 3, 2.0, dt)
ins = np.zeros((1081, 3), np.float32)
ins[:, :2] = pose[0, :2]
ins[:, 2] = pose[0, 2] + np.linspace(-4.71 / 2, 4.71 / 2, 1081, dtype=np.float32)
outs = np.zeros(1081, np.float32)
occ = [g.occ.copy(), g.occ.copy()]
occ[1][200:206, 150:158] = 1
for k in range(10):
    omap.update(occ[k & 1]); m.calc_range_many(ins, outs)
N = 300
t_up, t_scan = [], []
for k in range(N):
    t0 = time.perf_counter()
    omap.update(occ[k & 1])
    t1 = time.perf_counter()
    m.calc_range_many(ins, outs)                       # table rebuild (enqueue only) + scan
    t2 = time.perf_counter()
    t_up.append(t1 - t0); t_scan.append(t2 - t1)
med = lambda v: float(np.median(v)) * 1e6
print("colombia theta_disc 112 (median of %d ticks; means in brackets): map update %.1f us [%.1f] | CDDT rebuild "
      "+ 1081-ray scan %.1f us [%.1f] | tick total %.1f us"
      % (N, med(t_up), np.mean(t_up) * 1e6, med(t_scan), np.mean(t_scan) * 1e6, med(np.add(t_up, t_scan))), flush=True)
ts = []
for k in range(N):
    t0 = time.perf_counter()
    m.calc_range_many(ins, outs)
    ts.append(time.perf_counter() - t0)
print("  scan alone (table current): %.1f us [%.1f, max %.0f]" % (med(ts), np.mean(ts) * 1e6, max(ts) * 1e6), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "--tick-only":
    sys.exit(0)
w = workloads.cfg2()
omap2 = range_libc.PyOMap(w.gmap)
dt2 = omap2.distance_transform()
poses = workloads.make_poses(w, dt=dt2)
n, B = len(poses), w.num_rays
d_poses = torch.from_numpy(poses).cuda()
d_out = torch.empty(n * B, dtype=torch.float32, device="cuda")
ref = None
for td in (108, 720):
    m2 = range_libc.PyCDDTCast(omap2, 300, td)
    for bins in (1, 0):
        m2.set_option("cddt_bins", bins)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            m2.calc_range_fan_device(d_poses.data_ptr(), n, w.fov, B, d_out.data_ptr(), stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            m2.calc_range_fan_device(d_poses.data_ptr(), n, w.fov, B, d_out.data_ptr(), stream=s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        got = d_out.cpu().numpy()
        if bins == 1:
            ref = got
        print("cfg2 CDDT theta_disc %d, %s kernel: %.1f us/batch  %.0f Mrays/s  %s"
              % (td, "per-bin" if bins else "per-ray", ms * 1e3, n * B / ms / 1e3,
                 "" if bins else "identical=%s" % np.array_equal(ref, got)), flush=True)
