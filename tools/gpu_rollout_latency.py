#!/usr/bin/env python3
"""Diagnostic: host-visible time of the MCTS roll-out chain (scripts/mcts.py:202-245) for R roll-outs
per call: 200 x {control, updatePosition} -> 200 poses -> scan -> first crashed pose, all on the
device (rl_car_rollout_check), colombia map."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyracecarsimulator_amd import maps, range_libc, racecar as RC

g = maps.load_colombia()
omap = range_libc.PyOMap(g); dt = omap.distance_transform()
m = range_libc.PyRayMarchingGPU(omap, 300)
cars = RC.CarBatch()
B, fov = 1081, 4.71
edge = RC.edge_distances(B, -fov / 2, fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
rng = np.random.default_rng(3)
for R in (1, 16, 256, 4096):
    start = maps.sample_free_poses(g, R, 5, 4.0, dt)
    states = np.zeros((R, 11)); states[:, :3] = start; states[:, 3] = 1.0
    actions = np.stack([rng.uniform(0, 7, (R, 20)), rng.uniform(-0.4189, 0.4189, (R, 20))], -1)
    for _ in range(3):
        first, _, _ = cars.rollout_check(m, states, actions, fov, B, edge, 0.001)
    n = 20 if R < 4096 else 5
    t = time.perf_counter()
    for _ in range(n):
        first, _, vel = cars.rollout_check(m, states, actions, fov, B, edge, 0.001)
    t = (time.perf_counter() - t) / n
    print("R=%5d roll-outs x 200 steps x %d beams: %8.1f us per call, %7.2f us per roll-out, %6.0f Mrays/s; crashed %d"
          % (R, B, t * 1e6, t * 1e6 / R, R * 200 * B / t / 1e6, int((first >= 0).sum())), flush=True)
