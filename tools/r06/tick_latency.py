#!/usr/bin/env python3
"""The two-player tick on colombia (scripts/two_player/rcs_two_player.py:105-124: the other car's outline laid over the map,
tables rebuilt, one 1080-beam CDDT scan) as the host sees it: the grid re-uploaded every tick (rl_map_update, rounds 2-5)
against the outline sent as cell indices (rl_map_stamp_cells, round 6), and through the two_player facade."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pyracecarsimulator_amd import maps, range_libc
from pyracecarsimulator_amd.two_player import ScanSimulator2D as TwoPlayerScan

g = maps.load_colombia()
omap = range_libc.PyOMap(g)
m = range_libc.PyCDDTCast(omap, 300, 112)
dt = omap.distance_transform()
pose = maps.sample_free_poses(g, 1, 3, 2.0, dt)
B = 1080
ins = np.zeros((B, 3), np.float32)
ins[:, :2] = pose[0, :2]
ins[:, 2] = pose[0, 2] + np.linspace(-4.71 / 2, 4.71 / 2, B, dtype=np.float32)
outs = np.zeros(B, np.float32)
# the other car's outline: a 6 x 8-cell block somewhere free (two places, alternating ticks)
rr, cc = np.nonzero(dt >= 9.0)
occ = [g.occ.copy(), g.occ.copy()]
for o, k in zip(occ, (len(rr) // 3, 2 * len(rr) // 3)):
    o[rr[k] - 3:rr[k] + 3, cc[k] - 4:cc[k] + 4] = 1
cells = [np.flatnonzero((o != 0).reshape(-1) & (g.occ == 0).reshape(-1)).astype(np.int32) for o in occ]
assert all(len(c) == 48 for c in cells), [len(c) for c in cells]
med = lambda v: float(np.median(v)) * 1e6
N = 300
res = {}
for name, tick in (("grid re-uploaded (rl_map_update, %d B)" % g.occ.size, lambda k: omap.update(occ[k & 1])),
                   ("outline as cell indices (rl_map_stamp_cells, %d B)" % (4 * len(cells[0])), lambda k: omap.stamp_cells(cells[k & 1]))):
    for k in range(10):
        tick(k); m.calc_range_many(ins, outs)
    t_up, t_scan = [], []
    for k in range(N):
        t0 = time.perf_counter(); tick(k); t1 = time.perf_counter()
        m.calc_range_many(ins, outs); t2 = time.perf_counter()
        t_up.append(t1 - t0); t_scan.append(t2 - t1)
    res[name] = outs.copy()
    print("colombia, CDDT theta_disc 112, %s: map tables %.1f us | CDDT rebuild + %d-beam scan %.1f us | tick %.1f us (median of %d)"
          % (name, med(t_up), B, med(t_scan), med(np.add(t_up, t_scan)), N), flush=True)
sim = TwoPlayerScan(B, 4.71, 0.01)
maps_ = [maps.GridMap(o, g.resolution, g.origin, name="tick") for o in occ]
for k in range(10):
    sim.build(maps_[k & 1], 300, 112); sim.scan(*[float(v) for v in pose[0]])
ts = []
for k in range(N):
    t0 = time.perf_counter()
    sim.build(maps_[k & 1], 300, 112)
    sim.scan(*[float(v) for v in pose[0]])
    ts.append(time.perf_counter() - t0)
print("through two_player.ScanSimulator2D.build(map_msg) + scan(pose) (the grid re-uploaded): %.1f us per tick (median of %d)"
      % (med(ts), N), flush=True)
sim.build(maps.GridMap(g.occ, g.resolution, g.origin, name="base"), 300, 112)
ts = []
for k in range(N):
    t0 = time.perf_counter()
    sim.build_with_outline(cells[k & 1])
    sim.scan(*[float(v) for v in pose[0]])
    ts.append(time.perf_counter() - t0)
print("through build_with_outline(%d cells) + scan(pose): %.1f us per tick" % (len(cells[0]), med(ts)), flush=True)

# the same two ticks on a 2049^2 map (4.2 MB of grid per re-upload)
from pyracecarsimulator_amd import workloads
w = workloads.cfg2()
g2 = w.gmap
omap2 = range_libc.PyOMap(g2)
m2 = range_libc.PyCDDTCast(omap2, 300, 112)
dt2 = omap2.distance_transform()
pose2 = maps.sample_free_poses(g2, 1, 3, 2.0, dt2)
ins[:, :2] = pose2[0, :2]
ins[:, 2] = pose2[0, 2] + np.linspace(-4.71 / 2, 4.71 / 2, B, dtype=np.float32)
rr, cc = np.nonzero(dt2 >= 9.0)
occ2 = [g2.occ.copy(), g2.occ.copy()]
for o, k in zip(occ2, (len(rr) // 3, 2 * len(rr) // 3)):
    o[rr[k] - 3:rr[k] + 3, cc[k] - 4:cc[k] + 4] = 1
cells2 = [np.flatnonzero((o != 0).reshape(-1) & (g2.occ == 0).reshape(-1)).astype(np.int32) for o in occ2]
for name, tick in (("grid re-uploaded (rl_map_update, %d B)" % g2.occ.size, lambda k: omap2.update(occ2[k & 1])),
                   ("outline as cell indices (rl_map_stamp_cells, %d B)" % (4 * len(cells2[0])), lambda k: omap2.stamp_cells(cells2[k & 1]))):
    for k in range(5):
        tick(k); m2.calc_range_many(ins, outs)
    t_up, t_scan = [], []
    for k in range(60):
        t0 = time.perf_counter(); tick(k); t1 = time.perf_counter()
        m2.calc_range_many(ins, outs); t2 = time.perf_counter()
        t_up.append(t1 - t0); t_scan.append(t2 - t1)
    print("maze 2049^2, CDDT theta_disc 112, %s: map tables %.1f us | CDDT rebuild + %d-beam scan %.1f us | tick %.1f us (median of 60)"
          % (name, med(t_up), B, med(t_scan), med(np.add(t_up, t_scan))), flush=True)
