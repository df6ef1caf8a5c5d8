"""Follow-the-Gap steering on the GPU — drop-in for the reference's ``followgap`` module.

The reference wraps a header-only C++ class (followgap/followgap.hpp) in Cython
(followgap/followgap.pyx:23-31) and calls it on ONE scan at a time: ``PyFollowGap(10, 15.0,
max_steer_ang, 0.004)`` at scripts/mcts.py:97-99 and scripts/two_player/simple_driver.py:31,
``fg.eval(lidar, len(lidar))`` at scripts/mcts.py:267 and simple_driver.py:51.  ``PyFollowGap``
here keeps that constructor and ``eval`` and adds the batched forms (one wave per scan,
``csrc/consumer_kernels.h``), host or device resident, so a batch of scans that was just produced
on the GPU never has to leave it to be turned into steering angles.

Results are bit-identical to the reference's compiled header (tests/golden/followgap_ref.npz).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import f32p


class PyFollowGap:
    """``PyFollowGap(window_size, max_distance, max_angle, angle_inc)`` (followgap.pyx:23-25).
    ``window_size`` is stored and unused, as in the reference (FollowGap::eval never reads it)."""

    def __init__(self, ws, md, ma, angle_inc, device=0):
        self.window_size, self.max_distance = int(ws), float(md)
        self.max_angle, self.angle_inc = float(ma), float(angle_inc)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().rl_followgap_create(int(device), self.window_size, self.max_distance,
                                                  self.max_angle, self.angle_inc, C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().rl_followgap_destroy(h)
            except Exception:
                pass

    @staticmethod
    def _f32c(a, what):
        if not isinstance(a, np.ndarray) or a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
            # the Cython signature is np.ndarray[float, ndim=1, mode="c"] (followgap.pyx:30)
            raise ValueError("%s must be a C-contiguous float32 numpy array" % what)
        return a

    def eval(self, lidar, size):
        """Steering angle for one scan: ``fg.eval(lidar, len(lidar))`` (followgap.pyx:30-31)."""
        lidar = self._f32c(lidar, "lidar")
        if lidar.ndim != 1:
            raise ValueError("lidar must be one-dimensional")
        size = int(size)
        if size > lidar.shape[0]:
            raise ValueError("size exceeds the scan length")
        out = np.empty(1, dtype=np.float32)
        _lib.check(_lib.lib().rl_followgap_eval(self._h, lidar.ctypes.data_as(f32p), 1, size,
                                                out.ctypes.data_as(f32p)))
        return float(out[0])

    def eval_many(self, scans, size=None):
        """Steering angles float32 (n,) for scans float32 (n, size) — or a flat (n*size,) array with
        ``size`` given, the layout ``scanMany`` returns."""
        scans = self._f32c(scans, "scans")
        if scans.ndim == 2:
            n, size = scans.shape
        else:
            if not size or scans.size % int(size):
                raise ValueError("flat scans need a size that divides their length")
            size = int(size)
            n = scans.size // size
        out = np.empty(n, dtype=np.float32)
        _lib.check(_lib.lib().rl_followgap_eval(self._h, scans.ctypes.data_as(f32p), int(n), int(size),
                                                out.ctypes.data_as(f32p)))
        return out

    def eval_many_device(self, d_scans_ptr, n_scans, size, d_angles_ptr, stream=0):
        """Device-resident form: ``d_scans_ptr`` -> float32[n_scans*size] (e.g. the output of
        ``calc_range_fan_device``), ``d_angles_ptr`` -> float32[n_scans]; asynchronous on ``stream``."""
        _lib.check(_lib.lib().rl_followgap_eval_device(self._h, C.c_void_p(int(d_scans_ptr)), int(n_scans),
                                                       int(size), C.c_void_p(int(d_angles_ptr)),
                                                       C.c_void_p(int(stream))))
