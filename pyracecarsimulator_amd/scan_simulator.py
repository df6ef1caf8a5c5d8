"""``ScanSimulator2D`` — the reference's lidar scan façade on the MI355X library.

Same constructor, attributes and methods as /root/reference/scripts/scan_simulator.py
(``__init__`` :13-42, ``setMap`` :44-60, ``setRaytracingMethod`` :62-79, ``updateMap``
:81-86, ``scan`` :88-111, ``scanMany`` :113-135), Python 3, with the range_libc objects
replaced by ``pyracecarsimulator_amd.range_libc``.

Reference behaviours kept on purpose (SURVEY.md §7 "quirks"):
* ``scan``/``scanMany`` return the cached buffer itself (alias) — ``copy=True`` opts out;
* ``scanMany`` always scans ``batch_size`` poses, whatever ``len(poses)`` is (:119);
* noise stays off unless ``enable_noise`` is set (it is commented out at :109-111).
The one deliberate change: the sparse ``input_vector_many`` (12 B per *ray*, one live row
per pose) is kept only as the reference-visible attribute; what crosses PCIe is the dense
``(batch, 3)`` pose block.
"""
from __future__ import annotations

import math
import sys

import numpy as np

from . import _lib, range_libc


class ScanSimulator2D:

    def __init__(self, num_rays, fov, scan_std, batch_size=100):
        self.batch_size = batch_size
        self.num_rays = num_rays
        self.fov = fov
        self.scan_std = scan_std
        self.twopi = math.pi * 2

        # cached vectors (scan_simulator.py:32-40)
        # (the two result vectors live in pinned host memory of the library: the kernels write the
        #  ranges straight into them, no staging copy on the way back)
        self.output_vector = _lib.pinned_zeros(self.num_rays, np.float32)
        self.noise = np.zeros(self.num_rays, dtype=np.float32)
        self.input_vector = np.zeros((self.num_rays, 3), dtype=np.float32)
        self._addr_of = (None, None, 0, 0)       # (input_vector, output_vector, their addresses)
        self.output_vector_many = _lib.pinned_zeros(batch_size * self.num_rays, np.float32)
        self.input_vector_many = np.zeros((batch_size * self.num_rays, 3), dtype=np.float32)
        self._poses_many = np.zeros((batch_size, 3), dtype=np.float32)
        self._many_addr = (None, None, 0, 0)

        self.hasMap = False
        self.scan_method = None
        self.enable_noise = False
        self.noise_seed = 0

    def setMap(self, ros_map, max_range_px, resolution, origin):
        """ros_map: a ``range_libc.PyOMap`` (scripts/ros_interface.py:210,223)."""
        self.omap = ros_map
        self.origin_x = origin[0]
        self.origin_y = origin[1]
        self.origin_c = math.cos(origin[2])
        self.origin_s = math.sin(origin[2])
        self.mrx = max_range_px
        self.res = resolution
        self.hasMap = True

    def setRaytracingMethod(self, method="RM"):
        if not self.hasMap:
            print("for set RaytracingMethod use setMap first")
            return
        if method == "RM":
            self.scan_method = range_libc.PyRayMarching(self.omap, self.mrx)
        elif method == "RMGPU":
            self.scan_method = range_libc.PyRayMarchingGPU(self.omap, self.mrx)
        else:
            print("Only ray marching is supported")
            sys.exit()
        self._apply_noise()

    def setNoise(self, enable, seed=0):
        """Turn on the Gaussian range noise the reference leaves commented out
        (scan_simulator.py:109-111); applied on the device, keyed by (seed, ray id)."""
        self.enable_noise = bool(enable)
        self.noise_seed = int(seed)
        self._apply_noise()

    def _apply_noise(self):
        if self.scan_method is not None:
            self.scan_method.set_noise(self.scan_std if self.enable_noise else 0.0,
                                       self.noise_seed, 0)

    def updateMap(self, ros_map):
        """scan_simulator.py:81-86 is a stub that only stores the map; here a NumPy grid of
        the same shape is pushed to the device and the distance transform rebuilt."""
        self.ros_map = ros_map
        if self.hasMap and not isinstance(ros_map, range_libc.PyOMap):
            self.omap.update(ros_map)

    def scan(self, x, y, theta, copy=False):
        if not self.hasMap:
            print("Doing a scan without a defined map")
        self.input_vector[0, 0] = x
        self.input_vector[0, 1] = y
        self.input_vector[0, 2] = theta
        fast = getattr(self.scan_method, "_fan_rows_ptr", None)
        if fast is not None:
            # same call (scan_simulator.py:103-106) on the cached vectors' addresses
            if self._addr_of[0] is not self.input_vector or self._addr_of[1] is not self.output_vector:
                self._addr_of = (self.input_vector, self.output_vector,
                                 self.input_vector.ctypes.data, self.output_vector.ctypes.data)
            fast(self._addr_of[2], self._addr_of[3], self.num_rays, self.fov, self.num_rays)
        else:                                    # a foreign range_libc object: the public form
            self.scan_method.calc_range_many(self.input_vector, self.output_vector, self.fov,
                                             self.num_rays)
        return self.output_vector.copy() if copy else self.output_vector

    def scanMany(self, poses, copy=False):
        n, b = self.num_rays, self.batch_size
        p = self._poses_many
        if isinstance(poses, np.ndarray) and poses.ndim == 2 and poses.shape[0] >= b \
                and poses.shape[1] >= 3:
            p[:, :] = poses[:b, :3]              # same rows as the loop below, one copy
        else:
            for i in range(b):                   # scan_simulator.py:119-127
                p[i, 0] = poses[i][0]
                p[i, 1] = poses[i][1]
                p[i, 2] = poses[i][2]
        self.input_vector_many[::n, :] = p       # reference-visible sparse layout
        m = self.scan_method
        if getattr(m, "_fan_dense_raw", None) is not None:
            # the dense form of the same call on the cached vectors' addresses (no per-call ctypes objects or
            # re-validation: ~2.5 us of the ~43 a 200-pose roll-out takes end to end)
            if self._many_addr[0] is not p or self._many_addr[1] is not self.output_vector_many:
                self._many_addr = (p, self.output_vector_many, p.ctypes.data, self.output_vector_many.ctypes.data)
            rc = m._fan_dense_raw(m._h, self._many_addr[2], b, self.fov, n, self._many_addr[3], None, None)
            if rc:
                _lib.check(rc)
        else:                                    # a foreign range_libc object: the public form
            m.calc_range_fan(p, self.output_vector_many, self.fov, n)
        return self.output_vector_many.copy() if copy else self.output_vector_many
