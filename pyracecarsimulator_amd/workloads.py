"""Deterministic instances of BASELINE.json's configs.

Everything is generated from seeds with ``numpy.random.default_rng`` so the GPU box
rebuilds identical inputs; nothing here reads /root/reference.

``maps/map.pgm`` (configs 1-2) is missing from the reference mount (large blob stripped,
.MISSING_LARGE_BLOBS:4); its YAML (maps/map.yaml:1-7: 0.05 m/px, origin -51.224998)
implies a 2049 x 2049 grid, so the substitute is ``make_maze(2049, seed=1)`` with that
metadata.  Every report that uses it says so (``workload`` string).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import maps

#: reference constants (params.yaml:28-39)
SCAN_FOV = 4.71
SCAN_MAX_RANGE_M = 15.0
RESOLUTION = 0.05
MAX_RANGE_PX = int(SCAN_MAX_RANGE_M / RESOLUTION)   # racecar_simulator_v2.py:196 -> 300


@dataclass
class Workload:
    name: str
    gmap: maps.GridMap
    n_poses: int
    num_rays: int
    fov: float
    max_range_px: int
    method: str
    pose_seed: int
    theta_disc: int = 0
    noise_std: float = 0.0
    noise_seed: int = 0
    note: str = ""
    pose_kind: str = "uniform"    # "uniform": seeded free-cell poses; "rollout": MCTS roll-out poses

    @property
    def pose_note(self) -> str:
        return ("seeded MCTS roll-out poses (rl_car_rollout, scripts/mcts.py:214-231 schedule)"
                if self.pose_kind == "rollout" else "seeded free-space poses")

    def describe(self) -> str:
        return "%s: %s %dx%d, %d poses x %d beams, method %s%s" % (
            self.name, self.gmap.name, self.gmap.rows, self.gmap.cols, self.n_poses,
            self.num_rays, self.method, (" (" + self.note + ")") if self.note else "")


def cfg1() -> Workload:
    g = maps.make_maze(2049, cell=40, wall=3, p=0.45, seed=1,
                       origin=(-51.224998, -51.224998, 0.0))
    return Workload("cfg1", g, 1, 1081, SCAN_FOV, MAX_RANGE_PX, "RM", 2,
                    note="maps/map.pgm missing -> seeded 2049^2 maze with maps/map.yaml metadata")


def cfg2(n_poses: int = 4096) -> Workload:
    w = cfg1()
    return Workload("cfg2", w.gmap, n_poses, 1081, SCAN_FOV, MAX_RANGE_PX, "RMGPU", 2,
                    note=w.note)


def cfg3(n_poses: int = 65536) -> Workload:
    g = maps.make_maze(2000, cell=40, wall=3, p=0.45, seed=3)
    return Workload("cfg3", g, n_poses, 1081, SCAN_FOV, MAX_RANGE_PX, "GLT", 4,
                    theta_disc=1442, note="GiantLUT theta_disc=1442 ~ 2pi*1081/4.71")


def cfg4(n_poses: int = 1 << 20) -> Workload:
    g = maps.load_colombia()
    return Workload("cfg4", g, n_poses, 1081, SCAN_FOV, MAX_RANGE_PX, "RMGPU", 7,
                    note="maps/colombia, MCTS roll-out poses: %d roll-outs x %d steps, random (speed, "
                         "steer) every %d steps, dt %g" % (-(-n_poses // ROLLOUT_STEPS), ROLLOUT_STEPS,
                                                          ROLLOUT_ACTION_EVERY, ROLLOUT_DT),
                    pose_kind="rollout")


def cfg5(n_poses: int = 262144) -> Workload:
    g = maps.make_maze(4096, cell=64, wall=3, p=0.45, seed=5)
    return Workload("cfg5", g, n_poses, 720, SCAN_FOV * 720.0 / 1080.0, MAX_RANGE_PX, "RMGPU", 6,
                    noise_std=0.01, noise_seed=6,
                    note="policy window lidar[180:900] (scripts/policy.py:32) + Gaussian noise")


#: MCTS.rollout (scripts/mcts.py:202-231): max_iterations 200 (params.yaml:126), a new random action
#: every 10 steps, simulator step dt 0.01 (scripts/racecar_simulator_v2.py:84 via params.yaml:49)
ROLLOUT_STEPS = 200
ROLLOUT_ACTION_EVERY = 10
ROLLOUT_DT = 0.01


def rollout_inputs(w: "Workload", n_rollouts: int, seed: int, dt=None):
    """Seeded start states (R, 11) and action schedules (R, 20, 2) of ``n_rollouts`` MCTS roll-outs:
    starts on free cells with >= 2 px clearance, heading uniform, initial speed U[0, max_speed/2];
    actions as MCTS.rollout draws them: speed U[0, max_speed], steer U[-max_steer_ang, max_steer_ang]
    (scripts/mcts.py:216-221; limits params.yaml:9-10)."""
    from .racecar import DEFAULT_CAR
    starts = maps.sample_free_poses(w.gmap, n_rollouts, seed, 2.0, dt)
    rng = np.random.default_rng(seed + 7919)
    states = np.zeros((n_rollouts, 11), dtype=np.float64)
    states[:, 0:3] = starts
    states[:, 3] = rng.uniform(0.0, DEFAULT_CAR["max_speed"] / 2.0, n_rollouts)
    n_act = -(-ROLLOUT_STEPS // ROLLOUT_ACTION_EVERY)
    actions = np.empty((n_rollouts, n_act, 2), dtype=np.float64)
    actions[:, :, 0] = rng.uniform(0.0, DEFAULT_CAR["max_speed"], (n_rollouts, n_act))
    actions[:, :, 1] = rng.uniform(-DEFAULT_CAR["max_steer_ang"], DEFAULT_CAR["max_steer_ang"],
                                   (n_rollouts, n_act))
    return states, actions


def rollout_poses(w: "Workload", n_poses: int, seed: int, dt=None, device: int = 0, lo: int = 0,
                  hi=None) -> np.ndarray:
    """configs[3]'s pose batch as SURVEY.md §8(d) defines it: ceil(n/200) roll-outs x 200 steps
    integrated by the roll-out generator of this library (``rl_car_rollout``: Car::control +
    Car::updatePosition on the GPU, pinned to the reference's compiled Car by
    tests/golden/car_rollouts_ref.npz), truncated to ``n_poses``.  Needs the MI355X.

    ``lo``/``hi`` select poses [lo, hi) of that batch: start states and actions are drawn for the
    WHOLE batch (so the batch is the same whoever asks), but only the roll-outs that cover the block
    are integrated, on ``device`` — a rank of an N-GPU job generates its own shard on its own GPU."""
    from .racecar import CarBatch
    hi = n_poses if hi is None else hi
    R = -(-n_poses // ROLLOUT_STEPS)
    states, actions = rollout_inputs(w, R, seed, dt)
    r_lo, r_hi = lo // ROLLOUT_STEPS, -(-hi // ROLLOUT_STEPS)
    cars = CarBatch(device=device)
    poses, _, _ = cars.rollout(states[r_lo:r_hi], actions[r_lo:r_hi], ROLLOUT_STEPS, ROLLOUT_ACTION_EVERY,
                               ROLLOUT_DT)
    cars.close()
    off = lo - r_lo * ROLLOUT_STEPS
    return np.ascontiguousarray(poses.reshape(-1, 3)[off:off + (hi - lo)])


CONFIGS = {"cfg1": cfg1, "cfg2": cfg2, "cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}


def make_poses(w: Workload, dt=None, n_poses=None, seed=None, lo: int = 0, hi=None, device: int = 0) -> np.ndarray:
    """Seeded free-space poses for a workload.  ``dt`` (cells) restricts the draw to cells
    with >= 2 px clearance; without it any free cell qualifies.
    ``lo``/``hi``: only poses [lo, hi) of the batch of ``n_poses`` (a rank's block)."""
    n = w.n_poses if n_poses is None else n_poses
    seed = w.pose_seed if seed is None else seed
    hi = n if hi is None else hi
    if w.pose_kind == "rollout":
        return rollout_poses(w, n, seed, dt, device=device, lo=lo, hi=hi)
    return np.ascontiguousarray(maps.sample_free_poses(w.gmap, n, seed, 2.0, dt)[lo:hi])


#: BASELINE.json configs whose pose count is a GLOBAL batch to be sharded over the GPUs ("1M poses ...
#: sharded over 8", "256k poses ... 8xMI355X"): strong scaling.  The others fix the poses per GPU (weak).
GLOBAL_BATCH = ("cfg4", "cfg5")


def batch_layout(w: Workload, world: int, poses_per_gpu: int = 0):
    """(n_local, n_global, scaling) of a ``world``-GPU run: an explicit per-GPU pose count and the
    per-GPU configs scale weakly (n_global = world * n_local); configs 4 and 5 shard their global batch
    (n_local = n_global // world — every rank the same count, as the all-gather needs)."""
    if poses_per_gpu:
        return poses_per_gpu, poses_per_gpu * world, "weak"
    if w.name in GLOBAL_BATCH and world > 1:
        n_local = w.n_poses // world
        return n_local, n_local * world, "strong"
    return w.n_poses, w.n_poses * world, "weak"


def make_global_poses(w: Workload, world: int = 1, dt=None) -> np.ndarray:
    """The seeded global batch of ``world * n_poses`` poses; rank r scans block ``shard_range(r)``."""
    return make_poses(w, dt=dt, n_poses=w.n_poses * world)


def rank_poses(w: Workload, n_global: int, rank: int, world: int, dt=None, seed=None, device: int = 0):
    """Rank ``rank``'s contiguous block of the seeded global batch of ``n_global`` poses — the same
    poses ``make_poses(w, n_poses=n_global)[lo:hi]`` returns, generated without the other ranks'
    roll-outs and on this rank's own device."""
    lo, hi = shard_range(n_global, rank, world)
    return make_poses(w, dt=dt, n_poses=n_global, seed=seed, lo=lo, hi=hi, device=device)


def shard_range(n: int, rank: int, world: int):
    """Contiguous pose block of ``rank``: [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
