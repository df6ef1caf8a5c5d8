"""``range_libc``-compatible surface for the scan path, backed by libscan_amd.so.

Mirrors the objects the reference builds and calls (same names, argument meaning
and in-place/None-returning behaviour):

* ``PyOMap(map_msg)``                      scripts/ros_interface.py:210, scripts/mcts_driver.py:278
* ``PyRayMarching(omap, max_range_px)``    scripts/scan_simulator.py:72-73
* ``PyRayMarchingGPU(omap, max_range_px)`` scripts/scan_simulator.py:74-76
* ``PyCDDTCast(omap, max_range_px, theta_disc)``   scripts/two_player/scan.py:46
* ``obj.calc_range_many(ins, outs, fov, num_rays)`` scripts/scan_simulator.py:103-106,130-133
* ``obj.calc_range_many(ins, outs)``                scripts/two_player/scan.py:69-70

plus range_libc's remaining casters (``PyBresenhamsLine``, ``PyGiantLUTCast``).
Array contract: ``ins`` float32 C-contiguous (n,3), ``outs`` float32
C-contiguous (n,); anything else raises ``ValueError`` like the Cython
``np.ndarray[float, ndim=2, mode="c"]`` signature does.

Usage as a drop-in: ``from pyracecarsimulator_amd import range_libc`` (or put
``sys.modules["range_libc"] = pyracecarsimulator_amd.range_libc`` before importing
the reference's scan_simulator).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import f32p, f64p, i32p, u16p, u8p

__all__ = ["PyOMap", "PyRayMarching", "PyRayMarchingGPU", "PyCDDTCast", "PyBresenhamsLine",
           "PyGiantLUTCast", "USE_CUDA", "SHOULD_USE_CUDA"]

#: range_libc exports these flags; the AMD build always has its device path.
USE_CUDA = True
SHOULD_USE_CUDA = True


def _yaw_from_quaternion(q) -> float:
    """yaw of tf.transformations.euler_from_quaternion (scripts/ros_interface.py:212-216)."""
    x, y, z, w = (float(q.x), float(q.y), float(q.z), float(q.w))
    return math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))


class PyOMap:
    """Occupancy grid + world transform, resident on one MI355X.

    Accepts what the reference passes — a ``nav_msgs/OccupancyGrid``-like object with
    ``.info.{width,height,resolution,origin}`` and row-major ``.data`` (occupied where
    ``data > 10``) — or, since ROS messages do not exist on the GPU
    box, a NumPy ``(H, W)`` array (nonzero = occupied) with ``resolution`` and
    ``origin=(x, y, yaw)``, or a ``maps.GridMap``.

    ``device`` may be a LIST of device indices (``rl_map_create_multi``): the map then lives on every
    one of them, and the range methods created on it cut each host-pointer batch (``calc_range_many``,
    ``calc_range_fan``, ``check_collision_*``) into contiguous pose blocks, one per device, inside the
    one calling process — the reference's ``scanMany`` / ``checkCollisionMany`` callers
    (scripts/scan_simulator.py:113-135, scripts/mcts.py:237) get all the GPUs of the node unchanged.
    """

    def __init__(self, arg1, arg2=None, resolution=None, origin=None, device=0):
        occ, res, org = self._ingest(arg1, arg2, resolution, origin)
        self.occ = np.ascontiguousarray(occ, dtype=np.uint8)
        if self.occ.ndim != 2:
            raise ValueError("PyOMap: occupancy must be a 2-D array")
        self.height, self.width = (int(v) for v in self.occ.shape)
        self.resolution = float(res)
        self.origin = tuple(float(v) for v in org)
        self._h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            self.devices = [int(d) for d in device]
            if not self.devices:
                raise ValueError("PyOMap: empty device list")
            self.device = self.devices[0]
            arr = (C.c_int * len(self.devices))(*self.devices)
            _lib.check(_lib.lib().rl_map_create_multi(
                self.occ.ctypes.data_as(u8p), self.height, self.width, self.resolution,
                self.origin[0], self.origin[1], self.origin[2], arr, len(self.devices), C.byref(self._h)))
            return
        self.device = int(device)
        self.devices = [self.device]
        _lib.check(_lib.lib().rl_map_create(
            self.occ.ctypes.data_as(u8p), self.height, self.width, self.resolution,
            self.origin[0], self.origin[1], self.origin[2], self.device, C.byref(self._h)))

    @staticmethod
    def _ingest(arg1, arg2, resolution, origin):
        if hasattr(arg1, "info") and hasattr(arg1, "data"):          # OccupancyGrid duck type
            info = arg1.info
            w, h = int(info.width), int(info.height)
            data = np.asarray(arg1.data).reshape(h, w)
            pos, ori = info.origin.position, info.origin.orientation
            return (data > 10), float(info.resolution), (pos.x, pos.y, _yaw_from_quaternion(ori))
        if hasattr(arg1, "occ") and hasattr(arg1, "resolution"):     # maps.GridMap
            return arg1.occ != 0, arg1.resolution, arg1.origin
        arr = np.asarray(arg1)
        if arr.ndim == 2:
            if resolution is None and arg2 is not None and not isinstance(arg2, (tuple, list)):
                resolution = arg2
            return arr != 0, (1.0 if resolution is None else resolution), \
                ((0.0, 0.0, 0.0) if origin is None else origin)
        raise TypeError("PyOMap: expected an OccupancyGrid-like message, a GridMap or a 2-D array")

    # -- range_libc.PyOMap helpers ------------------------------------------
    def isOccupied(self, x, y):
        return bool(self.occ[int(x), int(y)])

    def update(self, occ):
        """Replace the grid (same shape) and rebuild the device tables — the per-scan
        rebuild of scripts/two_player/rcs_two_player.py:110-121."""
        occ = np.ascontiguousarray(np.asarray(occ) != 0, dtype=np.uint8)
        if occ.shape != self.occ.shape:
            raise ValueError("PyOMap.update: shape mismatch")
        self.occ = occ
        self._base_occ = None
        _lib.check(_lib.lib().rl_map_update(self._h, occ.ctypes.data_as(u8p)))

    def stamp_cells(self, flat_idx, value=255):
        """The two-player tick without re-uploading the grid (rl_map_stamp_cells): the occupancy becomes the BASE map
        (as constructed / last ``update``d) with the cells ``flat_idx`` (row * width + col; out-of-range ones are
        skipped like the reference's guard, scripts/two_player/rcs_two_player.py:113) occupied, tables rebuilt on the
        device.  Each stamp replaces the previous one; an empty list restores the base map."""
        idx = np.ascontiguousarray(flat_idx, dtype=np.int32).ravel()
        if getattr(self, "_base_occ", None) is None:
            self._base_occ = self.occ.copy()
        _lib.check(_lib.lib().rl_map_stamp_cells(self._h, idx.ctypes.data_as(_lib.i32p), int(idx.size), int(value) & 0xff))
        occ = self._base_occ.copy()
        ok = idx[(idx >= 0) & (idx < occ.size)]
        occ.reshape(-1)[ok] = 1 if (int(value) & 0xff) else 0
        self.occ = occ

    def distance_transform(self):
        """float32 (H, W) exact EDT in cells as built on the device (test hook)."""
        out = np.empty((self.height, self.width), dtype=np.float32)
        _lib.check(_lib.lib().rl_map_get_dt(self._h, out.ctypes.data_as(f32p)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().rl_map_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check_ins_outs(ins, outs):
    if not isinstance(ins, np.ndarray) or not isinstance(outs, np.ndarray):
        raise TypeError("calc_range_many: ins/outs must be numpy arrays")
    if ins.dtype != np.float32 or outs.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float'")
    if ins.ndim != 2 or outs.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 2 and 1)")
    if not ins.flags.c_contiguous or not outs.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    if ins.shape[1] != 3:
        raise ValueError("ins must have shape (n, 3)")
    if outs.shape[0] < ins.shape[0]:
        raise ValueError("outs is shorter than ins")


def _check_ranges_out(ranges, n):
    """An optional ranges buffer the C side will fill with n float32 values."""
    if ranges is None:
        return
    if not isinstance(ranges, np.ndarray) or ranges.dtype != np.float32 or not ranges.flags.c_contiguous \
            or ranges.size < n:
        raise ValueError("ranges must be a C-contiguous float32 array with n_poses*num_rays elements")


class _RangeMethod:
    KIND = None

    def __init__(self, omap, max_range_px, theta_disc=0):
        if not isinstance(omap, PyOMap):
            raise TypeError("expected a PyOMap")
        self.omap = omap                         # keep the map alive
        self.max_range_px = float(max_range_px)
        self.theta_disc = int(theta_disc)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().rl_method_create(omap._h, self.KIND, self.max_range_px,
                                               self.theta_disc, C.byref(self._h)))
        self._fan_raw = _lib.raw("rl_calc_range_many_fan")
        self._fan_dense_raw = _lib.raw("rl_calc_range_fan")

    # -- the reference's entry point ----------------------------------------
    def calc_range_many(self, ins, outs, fov=None, num_rays=None):
        """In-place, returns None.  2 args: one (x, y, theta) row per ray
        (scripts/two_player/scan.py:69-70).  4 args: the fork's fan form — pose p in row
        ``p*num_rays``, result ``outs[p*num_rays + j]`` (scripts/scan_simulator.py:103-106,
        130-133)."""
        _check_ins_outs(ins, outs)
        L = _lib.lib()
        if fov is None and num_rays is None:
            _lib.check(L.rl_calc_range_many(self._h, ins.ctypes.data_as(f32p),
                                            outs.ctypes.data_as(f32p), ins.shape[0]))
        elif fov is None or num_rays is None:
            raise TypeError("calc_range_many takes (ins, outs) or (ins, outs, fov, num_rays)")
        else:
            _lib.check(L.rl_calc_range_many_fan(self._h, ins.ctypes.data_as(f32p),
                                                outs.ctypes.data_as(f32p), ins.shape[0],
                                                float(fov), int(num_rays)))
        return None

    def _fan_rows_ptr(self, ins_addr, outs_addr, n_rows, fov, num_rays):
        """4-argument calc_range_many on buffers whose addresses the caller keeps (ScanSimulator2D's
        cached vectors): same C entry point, no per-call ctypes pointer objects, no re-validation."""
        rc = self._fan_raw(self._h, ins_addr, outs_addr, n_rows, fov, num_rays)
        if rc:
            _lib.check(rc)

    def calc_range(self, x, y, heading):
        """Scalar query in world coordinates (upstream's RayMarchingGPU does not support
        this and returns -1; here it is one 1-ray launch)."""
        ins = np.array([[x, y, heading]], dtype=np.float32)
        outs = np.zeros(1, dtype=np.float32)
        self.calc_range_many(ins, outs)
        return float(outs[0])

    # -- dense fast paths -------------------------------------
    def calc_range_fan(self, poses, outs, fov, num_rays, hit_cells=None, steps=None):
        """poses float32 (P,3) -> outs float32 (P*num_rays,), optional diagnostics."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        n = poses.shape[0] * int(num_rays)
        if outs.dtype != np.float32 or not outs.flags.c_contiguous or outs.size < n:
            raise ValueError("outs must be C-contiguous float32 with P*num_rays elements")
        if hit_cells is not None and (hit_cells.dtype != np.int32 or hit_cells.size < 2 * n
                                      or not hit_cells.flags.c_contiguous):
            raise ValueError("hit_cells must be C-contiguous int32 (P*num_rays, 2)")
        if steps is not None and (steps.dtype != np.uint16 or steps.size < n
                                  or not steps.flags.c_contiguous):
            raise ValueError("steps must be C-contiguous uint16 (P*num_rays,)")
        _lib.check(_lib.lib().rl_calc_range_fan(
            self._h, poses.ctypes.data_as(f32p), poses.shape[0], float(fov), int(num_rays),
            outs.ctypes.data_as(f32p),
            hit_cells.ctypes.data_as(i32p) if hit_cells is not None else None,
            steps.ctypes.data_as(u16p) if steps is not None else None))
        return None

    def calc_range_fan_device(self, d_poses_ptr, n_poses, fov, num_rays, d_outs_ptr,
                              d_hits_ptr=0, d_steps_ptr=0, stream=0):
        """Asynchronous launch on device pointers (ints), e.g. torch ``tensor.data_ptr()``."""
        _lib.check(_lib.lib().rl_calc_range_fan_device(
            self._h, C.c_void_p(d_poses_ptr), int(n_poses), float(fov), int(num_rays),
            C.c_void_p(d_outs_ptr), C.c_void_p(d_hits_ptr or None),
            C.c_void_p(d_steps_ptr or None), C.c_void_p(stream or None)))

    def calc_range_many_device(self, d_ins_ptr, d_outs_ptr, n, stream=0):
        _lib.check(_lib.lib().rl_calc_range_many_device(
            self._h, C.c_void_p(d_ins_ptr), C.c_void_p(d_outs_ptr), int(n),
            C.c_void_p(stream or None)))

    def check_collision_many(self, poses, fov, num_rays, edge_distances, crash_thresh,
                             ranges=None):
        """Scan + Car::isCrashed fused (racecar/src/racecar.cpp:305-328): index of the first
        crashed pose, else ``-(n_poses+1)``."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        edge = np.ascontiguousarray(edge_distances, dtype=np.float64)
        if edge.size < num_rays:
            raise ValueError("edge_distances needs num_rays entries")
        _check_ranges_out(ranges, poses.shape[0] * int(num_rays))
        first = C.c_int(0)
        _lib.check(_lib.lib().rl_check_collision_many(
            self._h, poses.ctypes.data_as(f32p), poses.shape[0], float(fov), int(num_rays),
            edge.ctypes.data_as(f64p), float(crash_thresh), C.byref(first),
            ranges.ctypes.data_as(f32p) if ranges is not None else None))
        return int(first.value)

    def check_collision_groups(self, poses, group, fov, num_rays, edge_distances, crash_thresh,
                               ranges=None):
        """isCrashed per roll-out of a batch: ``poses`` holds consecutive roll-outs of ``group``
        poses; returns int32[n_groups] (first crashed pose inside each roll-out, else -(group+1))."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        if poses.shape[0] % group:
            raise ValueError("number of poses is not a multiple of the group size")
        n_groups = poses.shape[0] // group
        edge = np.ascontiguousarray(edge_distances, dtype=np.float64)
        if edge.size < num_rays:
            raise ValueError("edge_distances needs num_rays entries")
        _check_ranges_out(ranges, poses.shape[0] * int(num_rays))
        first = np.zeros(n_groups, dtype=np.int32)
        _lib.check(_lib.lib().rl_check_collision_groups(
            self._h, poses.ctypes.data_as(f32p), n_groups, int(group), float(fov), int(num_rays),
            edge.ctypes.data_as(f64p), float(crash_thresh),
            first.ctypes.data_as(C.POINTER(C.c_int)),
            ranges.ctypes.data_as(f32p) if ranges is not None else None))
        return first

    def check_collision_groups_device(self, d_poses_ptr, n_groups, group, fov, num_rays, d_edge_ptr,
                                      crash_thresh, d_first_ptr, d_ranges_ptr=0, stream=0):
        """Asynchronous grouped scan + crash test on device pointers (ints)."""
        _lib.check(_lib.lib().rl_check_collision_groups_device(
            self._h, C.c_void_p(d_poses_ptr), int(n_groups), int(group), float(fov), int(num_rays),
            C.c_void_p(d_edge_ptr), float(crash_thresh), C.c_void_p(d_first_ptr),
            C.c_void_p(d_ranges_ptr or None), C.c_void_p(stream or None)))

    def calc_range_fan_multi_device(self, poses, d_outs_ptr, fov, num_rays, consumer=0, chunks=0):
        """Multi-device handle, results in DEVICE memory of replica ``consumer``'s GPU (rl_calc_range_fan_multi_device):
        every device marches its pose block into its own HBM and sends it to the consumer chunk by chunk
        (hipMemcpyPeerAsync: xGMI between peers) under the next chunk's march.  ``poses``: host array (n, 3);
        ``d_outs_ptr``: device address (int) of n * num_rays float32 on the consumer's device.  Synchronous."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        _lib.check(_lib.lib().rl_calc_range_fan_multi_device(
            self._h, poses.ctypes.data_as(f32p), poses.shape[0], float(fov), int(num_rays), int(consumer),
            C.c_void_p(d_outs_ptr), int(chunks)))

    def check_collision_groups_multi_device(self, poses, group, fov, num_rays, edge_distances, crash_thresh, d_first_ptr,
                                            consumer=0):
        """... the fused crash indices (int32 per roll-out of ``group`` poses) of every device's block, gathered in the
        consumer device's memory at ``d_first_ptr`` (rl_check_collision_groups_multi_device)."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        if poses.shape[0] % group:
            raise ValueError("number of poses is not a multiple of the group size")
        edge = np.ascontiguousarray(edge_distances, dtype=np.float64)
        if edge.size < num_rays:
            raise ValueError("edge_distances needs num_rays entries")
        _lib.check(_lib.lib().rl_check_collision_groups_multi_device(
            self._h, poses.ctypes.data_as(f32p), poses.shape[0] // int(group), int(group), float(fov), int(num_rays),
            edge.ctypes.data_as(f64p), float(crash_thresh), int(consumer), C.c_void_p(d_first_ptr)))

    @property
    def n_devices(self):
        """Devices behind this handle (> 1 for a method of a multi-device ``PyOMap``)."""
        return int(_lib.lib().rl_method_n_devices(self._h))

    def replica(self, i):
        """The per-device method ``i`` of a multi-device handle as a borrowed object for the ``*_device``
        calls (device pointers belong to one device); the handle itself for a single-device method."""
        p = _lib.lib().rl_method_replica(self._h, int(i))
        if not p:
            raise IndexError("replica %d out of range (%d devices)" % (i, self.n_devices))
        if p == self._h.value:
            return self
        r = object.__new__(type(self))
        r.__dict__.update(omap=self.omap, max_range_px=self.max_range_px, theta_disc=self.theta_disc,
                          _h=C.c_void_p(p), _fan_raw=self._fan_raw, _fan_dense_raw=self._fan_dense_raw, _parent=self)   # (_parent: borrowed handle)
        return r

    def set_noise(self, std, seed=0, ray_offset=0):
        _lib.check(_lib.lib().rl_set_noise(self._h, float(std), int(seed), int(ray_offset)))

    def set_option(self, name, value):
        _lib.check(_lib.lib().rl_method_set_option(self._h, name.encode(), int(value)))

    def get_info(self, name):
        v = C.c_int64(0)
        _lib.check(_lib.lib().rl_method_get_info(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def plan_fan(self, n_poses, num_rays, aux=False, crash=False):
        """What a fan call of this shape would launch with the handle's current options
        (kernel with template arguments, grid, LDS, binning pass): rl_method_plan_fan."""
        pl = _lib.LaunchPlan()
        _lib.check(_lib.lib().rl_method_plan_fan(self._h, int(n_poses), int(num_rays), int(bool(aux)),
                                                 int(bool(crash)), C.byref(pl)))
        return pl.as_dict()

    def last_plan(self):
        """The plan the last fan launch of this handle executed (rl_method_last_plan)."""
        pl = _lib.LaunchPlan()
        _lib.check(_lib.lib().rl_method_last_plan(self._h, C.byref(pl)))
        return pl.as_dict()

    def debug_stamps(self):
        """(n_waves, 4) uint64 diagnostics of the last stream-kernel launch (option debug_stamps)."""
        n = self.get_info("last_grid") * 4 * 4
        buf = np.zeros(max(n, 4), dtype=np.uint64)
        got = _lib.lib().rl_debug_read_stamps(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        if got < 0:
            _lib.check(got)
        return buf[:got].reshape(-1, 4)

    def last_kernel_ms(self):
        """Device time of the last launch; needs ``set_option("timing", 1)`` before that launch."""
        ms = C.c_float(0)
        _lib.check(_lib.lib().rl_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "_parent", None) is not None:       # a borrowed replica: the parent owns the handle
            return
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().rl_method_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PyRayMarching(_RangeMethod):
    """range_libc.PyRayMarching(omap, mrx) — step coefficient 0.999 (CPU RayMarching)."""
    KIND = _lib.RL_RM


class PyRayMarchingGPU(_RangeMethod):
    """range_libc.PyRayMarchingGPU(omap, mrx) — step coefficient 1.0 (kernels.cu)."""
    KIND = _lib.RL_RM_GPU


class PyBresenhamsLine(_RangeMethod):
    """range_libc.PyBresenhamsLine(omap, mrx)."""
    KIND = _lib.RL_BRESENHAM


class PyCDDTCast(_RangeMethod):
    """range_libc.PyCDDTCast(omap, mrx, theta_disc) (scripts/two_player/scan.py:46)."""
    KIND = _lib.RL_CDDT

    def __init__(self, omap, max_range_px, theta_disc):
        super().__init__(omap, max_range_px, theta_disc)

    def prune(self, max_range=None):   # upstream API; pruning is a no-op for exactness
        return None


class PyGiantLUTCast(_RangeMethod):
    """range_libc.PyGiantLUTCast(omap, mrx, theta_disc): uint16 table [row][col][theta_bin] built
    on the device at first use (rows*cols*theta_disc*2 bytes of HBM)."""
    KIND = _lib.RL_GIANT_LUT

    def __init__(self, omap, max_range_px, theta_disc):
        super().__init__(omap, max_range_px, theta_disc)

    def table(self, row0=0, row1=None):
        """(row1-row0, cols, theta_disc) uint16 slab of the device table (test hook)."""
        row1 = self.omap.height if row1 is None else row1
        out = np.empty((row1 - row0, self.omap.width, self.theta_disc), dtype=np.uint16)
        _lib.check(_lib.lib().rl_method_read_lut(self._h, row0, row1, out.ctypes.data_as(u16p)))
        return out
