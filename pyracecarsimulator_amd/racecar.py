"""Batched vehicle roll-outs for the scan path.

The reference's ``racecar`` package is a C++ ``Car`` behind a Cython ``PyCar``
(/root/reference/racecar/src/racecar.cpp, racecar/pywrapper/racecar.pyx:75-114).  Only the two
pieces of it that sit directly on either side of the batched scan are provided here, batched and
on the GPU:

* before the scan — the roll-out pose generator: ``MCTS.rollout`` calls ``control`` +
  ``updatePosition(0.01)`` 200 times per roll-out (scripts/mcts.py:214-231).  ``CarBatch.rollout``
  integrates any number of roll-outs at once, one GPU lane each, in float64;
* after the scan — the crash test ``isCrashed`` over ``setCarEdgeDistances``' table
  (racecar.cpp:239-292, 305-328; scripts/racecar_simulator_v2.py:47-50, 146-167):
  ``edge_distances`` (host table, same quirks) + the fused device test in ``range_libc``.

``CarBatch.rollout_check`` chains roll-outs -> poses -> scan -> per-roll-out crash index on the
device (poses and ranges never cross PCIe).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import f32p, f64p

#: constructor order of Car (racecar/include/racecar.hpp:32-36) with the reference's values
#: (params.yaml:3-21,47)
CAR_PARAM_ORDER = ("wb", "fc", "h_cg", "l_f", "l_r", "cs_f", "cs_r", "mass", "I_z", "ttc_thresh",
                   "width", "length", "max_steer_vel", "max_steer_ang", "max_speed", "max_accel",
                   "max_decel")
DEFAULT_CAR = dict(wb=0.3302, fc=1.0, h_cg=0.08255, l_f=0.15875, l_r=0.17145, cs_f=2.3, cs_r=2.3,
                   mass=3.17, I_z=0.0398378, ttc_thresh=0.001, width=0.2032, length=0.4064,
                   max_steer_vel=5.0, max_steer_ang=0.4189, max_speed=7.0, max_accel=3.0,
                   max_decel=20.0)


def edge_distances(num_rays, min_ang, scan_ang_inc, scan_dist_to_base, width, wheelbase):
    """Car::setCarEdgeDistances (racecar.cpp:239-292) as float64[num_rays]: distance from the lidar
    to the car's outline along each beam — native (``rl_car_edge_distances``, host C++ of
    libscan_amd.so; needs no GPU).  Kept as the reference computes it: the angle is incremented BEFORE
    use (table shifted by one beam, :256), ``PI = 3.145`` (racecar.hpp:117), and a beam at exactly
    0 rad gets ``side / sin(-0.0001)`` (a large negative number, :277-283)."""
    out = np.empty(int(num_rays), dtype=np.float64)
    _lib.check(_lib.lib().rl_car_edge_distances(int(num_rays), float(min_ang), float(scan_ang_inc),
                                                float(scan_dist_to_base), float(width), float(wheelbase),
                                                out.ctypes.data_as(f64p)))
    return out


def is_crashed(rays, num_rays, poses, edge, crash_thresh):
    """Car::isCrashed (racecar.cpp:305-328) over host ranges (``rl_car_is_crashed``): index of the
    first crashed scan, else ``-(poses+1)``."""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    edge = np.ascontiguousarray(edge, dtype=np.float64)
    if rays.size < num_rays * poses or edge.size < num_rays:
        raise ValueError("is_crashed: rays needs poses*num_rays values, edge num_rays")
    first = C.c_int(0)
    _lib.check(_lib.lib().rl_car_is_crashed(rays.ctypes.data_as(f32p), int(num_rays), int(poses),
                                            edge.ctypes.data_as(f64p), float(crash_thresh), C.byref(first)))
    return int(first.value)


class CarBatch:
    """Many ``Car`` objects stepped together on one MI355X, or on several (``device=[...]``) (no CPU path)."""

    def __init__(self, params=None, device=0):
        p = dict(DEFAULT_CAR)
        p.update(params or {})
        self.params = p
        arr = np.array([p[k] for k in CAR_PARAM_ORDER], dtype=np.float64)
        self._h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            # several devices (pair with a method of PyOMap(device=[...]) on the same list): roll-outs are cut
            # into contiguous blocks, one per device, inside this one process
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            _lib.check(_lib.lib().rl_car_create_multi(devs, len(device), arr.ctypes.data_as(f64p), C.byref(self._h)))
        else:
            _lib.check(_lib.lib().rl_car_create(int(device), arr.ctypes.data_as(f64p), C.byref(self._h)))

    @staticmethod
    def _prep(states, actions, n_steps, action_every):
        states = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 11)
        n_act = (n_steps + action_every - 1) // action_every
        actions = np.ascontiguousarray(actions, dtype=np.float64).reshape(states.shape[0], n_act, 2)
        return states, actions

    def rollout(self, states, actions, n_steps=200, action_every=10, dt=0.01):
        """states float64 (R, 11) in getState layout; actions float64 (R, ceil(n_steps/every), 2) as
        (speed, steer).  Returns (poses float32 (R, n_steps, 3), final states (R, 11), velocities
        (R, n_steps)) — the arrays MCTS.rollout builds (all_sim_states, rewards)."""
        states, actions = self._prep(states, actions, n_steps, action_every)
        R = states.shape[0]
        poses = np.empty((R, n_steps, 3), dtype=np.float32)
        out = np.empty((R, 11), dtype=np.float64)
        vel = np.empty((R, n_steps), dtype=np.float64)
        _lib.check(_lib.lib().rl_car_rollout(
            self._h, states.ctypes.data_as(f64p), actions.ctypes.data_as(f64p), R, int(n_steps),
            int(action_every), float(dt), poses.ctypes.data_as(f32p), out.ctypes.data_as(f64p),
            vel.ctypes.data_as(f64p)))
        return poses, out, vel

    def rollout_check(self, method, states, actions, fov, num_rays, edge, crash_thresh, n_steps=200,
                      action_every=10, dt=0.01):
        """MCTS.rollout + checkCollisionMany for R roll-outs in one call; returns (first crashed
        pose per roll-out int32 (R,), final states (R, 11), velocities (R, n_steps))."""
        states, actions = self._prep(states, actions, n_steps, action_every)
        R = states.shape[0]
        edge = np.ascontiguousarray(edge, dtype=np.float64)
        first = np.zeros(R, dtype=np.int32)
        out = np.empty((R, 11), dtype=np.float64)
        vel = np.empty((R, n_steps), dtype=np.float64)
        _lib.check(_lib.lib().rl_car_rollout_check(
            self._h, method._h, states.ctypes.data_as(f64p), actions.ctypes.data_as(f64p), R,
            int(n_steps), int(action_every), float(dt), float(fov), int(num_rays),
            edge.ctypes.data_as(f64p), float(crash_thresh), first.ctypes.data_as(C.POINTER(C.c_int)),
            out.ctypes.data_as(f64p), vel.ctypes.data_as(f64p)))
        return first, out, vel

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().rl_car_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
