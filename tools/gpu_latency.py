#!/usr/bin/env python3
"""Diagnostic: host-visible latency of the reference's two calls — scan() (1 pose, sim tick) and
scanMany() (200-pose MCTS roll-out, params.yaml:126) — through the whole stack (Python shim, C ABI,
H2D, kernels, D2H)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyracecarsimulator_amd import ScanSimulator2D, maps, range_libc, workloads
from pyracecarsimulator_amd import racecar as RC

for name, g in (("colombia", maps.load_colombia()), ("maze2049", workloads.cfg2().gmap)):
    omap = range_libc.PyOMap(g)
    dt = omap.distance_transform()
    for beams in (1080, 1081):
        sim = ScanSimulator2D(beams, 4.71, 0.01, batch_size=200)
        sim.setMap(omap, 300, g.resolution, g.origin)
        sim.setRaytracingMethod("RMGPU")
        poses = maps.sample_free_poses(g, 200, 3, 2.0, dt)
        for _ in range(20):
            sim.scanMany(poses); sim.scan(*[float(v) for v in poses[0]])
        t = time.perf_counter()
        for _ in range(200): sim.scanMany(poses)
        tm = (time.perf_counter() - t) / 200
        t = time.perf_counter()
        for _ in range(500): sim.scan(float(poses[1, 0]), float(poses[1, 1]), float(poses[1, 2]))
        ts = (time.perf_counter() - t) / 500
        edge = RC.edge_distances(beams, -4.71 / 2, 4.71 / beams, 0.275, 0.2032, 0.3302)
        m = sim.scan_method
        t = time.perf_counter()
        for _ in range(200): m.check_collision_many(poses, 4.71, beams, edge, 0.001)
        tc = (time.perf_counter() - t) / 200
        print("%s %d beams: scan() %.1f us | scanMany(200) %.1f us (%.0f Mrays/s host-visible) | fused scan+crash(200) %.1f us"
              % (name, beams, ts * 1e6, tm * 1e6, 200 * beams / tm / 1e6, tc * 1e6), flush=True)
