"""ctypes binding of oracle/librangelib_oracle.so — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see rangelib_oracle.h): range_libc is absent from the reference
mount and the reference has no golden vectors for this path; this oracle is a
restatement of range_libc's published algorithm anchored on the reference's call
sites (scripts/scan_simulator.py:72-76,103-106,130-133).

The library is built by ``make -C oracle`` (also done by __graft_entry__.build()).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u16p = C.POINTER(C.c_uint16)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)


class _OrcMap(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("occ", _u8p), ("res", C.c_float),
                ("ox", C.c_float), ("oy", C.c_float), ("wa_cos", C.c_float),
                ("wa_sin", C.c_float), ("wa", C.c_float), ("inv_res", C.c_float)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "librangelib_oracle.so")
    src = os.path.join(_HERE, "rangelib_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "librangelib_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


_NATIVE = None


def native_lib():
    """The same restatement built ``-O3 -march=native`` FOR THE HOST IT RUNS ON (SURVEY.md section 8d asks for
    that build of the timed CPU baseline; the checker stays the portable ``-O2`` build).  Compiled on first
    use into oracle/_native/ under a name that carries a hash of this CPU's feature flags, so a file built on
    another machine is never loaded.  Returns None when it cannot be built or loaded here.  Only the two
    timed casters are bound (orc_rm_fan, orc_bl_fan)."""
    global _NATIVE
    if _NATIVE is not None:
        return _NATIVE or None
    import hashlib
    flags = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = line
                break
    except OSError:
        pass
    tag = hashlib.sha1(flags.encode()).hexdigest()[:12]
    out_dir = os.path.join(_HERE, "_native")
    so = os.path.join(out_dir, "librangelib_oracle_native_%s.so" % tag)
    src = os.path.join(_HERE, "rangelib_oracle.c")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(out_dir, exist_ok=True)
            subprocess.check_call(["gcc"] + NATIVE_FLAGS.split() + ["-shared", "-o", so, src, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(so)
        mp = C.POINTER(_OrcMap)
        L.orc_rm_fan.argtypes = [mp, _f32p, C.c_float, C.c_float, _f32p, C.c_int, C.c_float,
                                 C.c_int, _f32p, _i32p, _u16p, C.c_int]
        L.orc_bl_fan.argtypes = [mp, C.c_float, _f32p, C.c_int, C.c_float, C.c_int, _f32p, _i32p,
                                 _u16p, C.c_int]
        _NATIVE = L
    except (OSError, subprocess.CalledProcessError):
        _NATIVE = False
    return _NATIVE or None


#: flags of native_lib(): the arithmetic switches of the checker build are kept (no contraction, no fast-math), so
#: the two builds return the same bits — bench.py checks that on its sample before it quotes the faster one
NATIVE_FLAGS = "-O3 -march=native -fPIC -std=c11 -ffp-contract=off -fno-fast-math -fopenmp"


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_map_init.argtypes = [C.POINTER(_OrcMap), _u8p, C.c_int, C.c_int, C.c_float,
                                   C.c_float, C.c_float, C.c_float]
        L.orc_sincosf.argtypes = [C.c_float, _f32p, _f32p]
        L.orc_edt.argtypes = [_u8p, C.c_int, C.c_int, _f32p]
        L.orc_edt_sq.argtypes = [_u8p, C.c_int, C.c_int, _u32p]
        mp = C.POINTER(_OrcMap)
        L.orc_rm_fan.argtypes = [mp, _f32p, C.c_float, C.c_float, _f32p, C.c_int, C.c_float,
                                 C.c_int, _f32p, _i32p, _u16p, C.c_int]
        L.orc_rm_rays.argtypes = [mp, _f32p, C.c_float, C.c_float, _f32p, C.c_int, _f32p, _i32p,
                                  _u16p, C.c_int]
        L.orc_rm_rays_libm.argtypes = [mp, _f32p, C.c_float, C.c_float, _f32p, C.c_int, _f32p, _i32p, _u16p]
        L.orc_rm_fan_libm.argtypes = [mp, _f32p, C.c_float, C.c_float, _f32p, C.c_int, C.c_float, C.c_int,
                                      _f32p, _i32p, _u16p]
        L.orc_bl_rays_libm.argtypes = [mp, C.c_float, _f32p, C.c_int, _f32p, _i32p, _u16p]
        L.orc_bl_fan_libm.argtypes = [mp, C.c_float, _f32p, C.c_int, C.c_float, C.c_int, _f32p, _i32p, _u16p]
        L.orc_libm_sincosf.argtypes = [_f32p, C.c_long, _f32p, _f32p]
        L.orc_lit_sincosf.argtypes = [_f32p, C.c_long, _f32p, _f32p]
        L.orc_libm_restatement_check.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_long), C.c_int]
        L.orc_libm_restatement_check.restype = C.c_long
        L.orc_bl_fan.argtypes = [mp, C.c_float, _f32p, C.c_int, C.c_float, C.c_int, _f32p, _i32p,
                                 _u16p, C.c_int]
        L.orc_bl_rays.argtypes = [mp, C.c_float, _f32p, C.c_int, _f32p, _i32p, _u16p, C.c_int]
        L.orc_lut_build.argtypes = [mp, _f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                    _u16p, C.c_int]
        L.orc_lut_fan.argtypes = [mp, _u16p, C.c_int, C.c_float, _f32p, C.c_int, C.c_float,
                                  C.c_int, _f32p, C.c_int]
        L.orc_lut_rays.argtypes = [mp, _u16p, C.c_int, C.c_float, _f32p, C.c_int, _f32p, C.c_int]
        L.orc_lut_pose_cells.argtypes = [mp, _f32p, C.c_int, _i32p, _i32p]
        L.orc_lut_fan_rows.argtypes = [mp, _u16p, C.c_int, C.c_float, _f32p, C.c_int, C.c_float,
                                       C.c_int, _f32p]
        L.orc_cddt_build.argtypes = [mp, C.c_int]
        L.orc_cddt_build.restype = C.c_void_p
        L.orc_cddt_free.argtypes = [C.c_void_p]
        L.orc_cddt_fan.argtypes = [mp, C.c_void_p, C.c_float, _f32p, C.c_int, C.c_float, C.c_int,
                                   _f32p, C.c_int]
        L.orc_cddt_rays.argtypes = [mp, C.c_void_p, C.c_float, _f32p, C.c_int, _f32p, C.c_int]
        L.orc_cddt_build_libm.argtypes = [mp, C.c_int]
        L.orc_cddt_build_libm.restype = C.c_void_p
        L.orc_cddt_rays_libm.argtypes = [mp, C.c_void_p, C.c_float, _f32p, C.c_int, _f32p]
        L.orc_lut_build_libm.argtypes = [mp, _f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _u16p, C.c_int]
        L.orc_lut_fan_rows_libm.argtypes = [mp, _u16p, C.c_int, C.c_float, _f32p, C.c_int, C.c_float, C.c_int, _f32p]
        L.orc_edge_distances.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_double, _f64p]
        L.orc_is_crashed.argtypes = [_f32p, C.c_int, C.c_int, _f64p, C.c_double]
        L.orc_is_crashed.restype = C.c_int
        L.orc_followgap_eval.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_followgap_eval.restype = C.c_float
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def sincosf(x):
    """Vector wrapper around orc_sincosf -> (sin, cos) float32 arrays."""
    x = np.atleast_1d(np.asarray(x, dtype=np.float32))
    s = np.empty_like(x)
    c = np.empty_like(x)
    L = lib()
    sv, cv = C.c_float(), C.c_float()
    for i, v in enumerate(x):
        L.orc_sincosf(C.c_float(float(v)), C.byref(sv), C.byref(cv))
        s[i], c[i] = sv.value, cv.value
    return s, c


def edt(occ):
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    out = np.empty(occ.shape, dtype=np.float32)
    lib().orc_edt(_p(occ, _u8p), occ.shape[0], occ.shape[1], _p(out, _f32p))
    return out


def edt_sq(occ):
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    out = np.empty(occ.shape, dtype=np.uint32)
    lib().orc_edt_sq(_p(occ, _u8p), occ.shape[0], occ.shape[1], _p(out, _u32p))
    return out


class OracleMap:
    """OMap + DistanceTransform + every CPU caster of the oracle on one grid."""

    def __init__(self, occ, resolution, origin, max_range_px):
        self.occ = np.ascontiguousarray(occ, dtype=np.uint8)
        self.rows, self.cols = self.occ.shape
        self.max_range_px = float(max_range_px)
        self.resolution = float(resolution)
        self._m = _OrcMap()
        lib().orc_map_init(C.byref(self._m), _p(self.occ, _u8p), self.rows, self.cols,
                           float(resolution), float(origin[0]), float(origin[1]), float(origin[2]))
        self._dt = None
        self._cddt = {}

    @classmethod
    def from_gridmap(cls, g, max_range_px):
        return cls(g.occ, g.resolution, g.origin, max_range_px)

    @property
    def dt(self):
        if self._dt is None:
            self._dt = edt(self.occ)
        return self._dt

    def __del__(self):
        for h in getattr(self, "_cddt", {}).values():
            lib().orc_cddt_free(h)

    # -- RayMarching -------------------------------------------------------
    def rm_fan(self, poses, fov, num_rays, step_coeff=0.999, nthreads=1, want_hits=True,
               want_steps=True, native=False):
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        n = poses.shape[0] * num_rays
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32) if want_hits else None
        steps = np.empty(n, dtype=np.uint16) if want_steps else None
        (native_lib() if native else lib()).orc_rm_fan(C.byref(self._m), _p(self.dt, _f32p), self.max_range_px, step_coeff,
                         _p(poses, _f32p), poses.shape[0], fov, num_rays, _p(ranges, _f32p),
                         _p(hits, _i32p), _p(steps, _u16p), nthreads)
        return ranges, hits, steps

    def rm_rays(self, ins, step_coeff=0.999, nthreads=1):
        ins = np.ascontiguousarray(ins, dtype=np.float32).reshape(-1, 3)
        n = ins.shape[0]
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32)
        steps = np.empty(n, dtype=np.uint16)
        lib().orc_rm_rays(C.byref(self._m), _p(self.dt, _f32p), self.max_range_px, step_coeff,
                          _p(ins, _f32p), n, _p(ranges, _f32p), _p(hits, _i32p), _p(steps, _u16p),
                          nthreads)
        return ranges, hits, steps

    def rm_rays_libm(self, ins, step_coeff=0.999, full=False):
        """Upstream-literal form (libm trig, unfused, calc_range(y, x, theta')).  ``full``: also the
        hit cells (col, row) and sample counts."""
        ins = np.ascontiguousarray(ins, dtype=np.float32).reshape(-1, 3)
        n = ins.shape[0]
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32) if full else None
        steps = np.empty(n, dtype=np.uint16) if full else None
        lib().orc_rm_rays_libm(C.byref(self._m), _p(self.dt, _f32p), self.max_range_px, step_coeff,
                               _p(ins, _f32p), n, _p(ranges, _f32p), _p(hits, _i32p), _p(steps, _u16p))
        return (ranges, hits, steps) if full else ranges

    def rm_fan_libm(self, poses, fov, num_rays, step_coeff=0.999):
        """The 4-arg fan stated literally: one libm cast per beam at theta + (-fov/2 + j*fov/B)."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        n = poses.shape[0] * num_rays
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32)
        steps = np.empty(n, dtype=np.uint16)
        lib().orc_rm_fan_libm(C.byref(self._m), _p(self.dt, _f32p), self.max_range_px, step_coeff,
                              _p(poses, _f32p), poses.shape[0], fov, num_rays, _p(ranges, _f32p),
                              _p(hits, _i32p), _p(steps, _u16p))
        return ranges, hits, steps

    # -- BresenhamsLine ----------------------------------------------------
    def bl_rays_libm(self, ins):
        """Upstream-literal BresenhamsLine (libm trig, un-fused end point and hit distance) -> ranges, hit cells (col, row), steps."""
        ins = np.ascontiguousarray(ins, dtype=np.float32).reshape(-1, 3)
        n = ins.shape[0]
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32)
        steps = np.empty(n, dtype=np.uint16)
        lib().orc_bl_rays_libm(C.byref(self._m), self.max_range_px, _p(ins, _f32p), n, _p(ranges, _f32p), _p(hits, _i32p),
                               _p(steps, _u16p))
        return ranges, hits, steps

    def bl_fan_libm(self, poses, fov, num_rays):
        """The 4-argument fan of the upstream-literal BresenhamsLine: one cast per beam at theta + (-fov/2 + j*fov/B)."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        n = poses.shape[0] * num_rays
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32)
        steps = np.empty(n, dtype=np.uint16)
        lib().orc_bl_fan_libm(C.byref(self._m), self.max_range_px, _p(poses, _f32p), poses.shape[0], fov, num_rays,
                              _p(ranges, _f32p), _p(hits, _i32p), _p(steps, _u16p))
        return ranges, hits, steps

    def bl_fan(self, poses, fov, num_rays, nthreads=1, native=False):
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        n = poses.shape[0] * num_rays
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32)
        steps = np.empty(n, dtype=np.uint16)
        (native_lib() if native else lib()).orc_bl_fan(C.byref(self._m), self.max_range_px, _p(poses, _f32p), poses.shape[0],
                         fov, num_rays, _p(ranges, _f32p), _p(hits, _i32p), _p(steps, _u16p),
                         nthreads)
        return ranges, hits, steps

    def bl_rays(self, ins, nthreads=1):
        ins = np.ascontiguousarray(ins, dtype=np.float32).reshape(-1, 3)
        n = ins.shape[0]
        ranges = np.empty(n, dtype=np.float32)
        hits = np.empty((n, 2), dtype=np.int32)
        steps = np.empty(n, dtype=np.uint16)
        lib().orc_bl_rays(C.byref(self._m), self.max_range_px, _p(ins, _f32p), n,
                          _p(ranges, _f32p), _p(hits, _i32p), _p(steps, _u16p), nthreads)
        return ranges, hits, steps

    # -- GiantLUT ------------------------------------------------------------
    def lut_build(self, theta_disc, r0=0, r1=None, step_coeff=0.999, nthreads=0):
        r1 = self.rows if r1 is None else r1
        lut = np.empty((r1 - r0, self.cols, theta_disc), dtype=np.uint16)
        lib().orc_lut_build(C.byref(self._m), _p(self.dt, _f32p), self.max_range_px, step_coeff,
                            theta_disc, r0, r1, _p(lut, _u16p),
                            nthreads or lib().orc_max_threads())
        return lut

    def lut_fan(self, lut, poses, fov, num_rays, nthreads=1):
        """``lut`` must cover every row (r0=0, r1=rows)."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        ranges = np.empty(poses.shape[0] * num_rays, dtype=np.float32)
        lib().orc_lut_fan(C.byref(self._m), _p(lut, _u16p), lut.shape[2], self.max_range_px,
                          _p(poses, _f32p), poses.shape[0], fov, num_rays, _p(ranges, _f32p),
                          nthreads)
        return ranges

    def lut_pose_cells(self, poses):
        """(row, col) of the table cell each pose reads (-1, -1 outside the map)."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        r = np.empty(poses.shape[0], np.int32)
        c = np.empty(poses.shape[0], np.int32)
        lib().orc_lut_pose_cells(C.byref(self._m), _p(poses, _f32p), poses.shape[0], _p(r, _i32p),
                                 _p(c, _i32p))
        return r, c

    def lut_fan_rows(self, pose_rows, poses, fov, num_rays):
        """Fan query with one theta row per pose (``pose_rows`` uint16 (P, theta_disc)): for
        tables too large for the host."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        pose_rows = np.ascontiguousarray(pose_rows, dtype=np.uint16)
        assert pose_rows.shape[0] == poses.shape[0]
        ranges = np.empty(poses.shape[0] * num_rays, dtype=np.float32)
        lib().orc_lut_fan_rows(C.byref(self._m), _p(pose_rows, _u16p), pose_rows.shape[1],
                               self.max_range_px, _p(poses, _f32p), poses.shape[0], fov, num_rays,
                               _p(ranges, _f32p))
        return ranges

    def lut_build_libm(self, theta_disc, r0=0, r1=None, step_coeff=0.999, nthreads=0):
        """Upstream-literal GiantLUT rows [r0, r1): libm cosf / sinf per bin, un-fused march from the cell corner."""
        r1 = self.rows if r1 is None else r1
        lut = np.empty((r1 - r0, self.cols, theta_disc), dtype=np.uint16)
        lib().orc_lut_build_libm(C.byref(self._m), _p(self.dt, _f32p), self.max_range_px, step_coeff, theta_disc, r0, r1,
                                 _p(lut, _u16p), nthreads or lib().orc_max_threads())
        return lut

    def lut_fan_rows_libm(self, pose_rows, poses, fov, num_rays):
        """Fan query on literal rows: un-fused world->grid, per-ray float32 angles, fmod + roundf bin rule."""
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        pose_rows = np.ascontiguousarray(pose_rows, dtype=np.uint16)
        assert pose_rows.shape[0] == poses.shape[0]
        ranges = np.empty(poses.shape[0] * num_rays, dtype=np.float32)
        lib().orc_lut_fan_rows_libm(C.byref(self._m), _p(pose_rows, _u16p), pose_rows.shape[1], self.max_range_px,
                                    _p(poses, _f32p), poses.shape[0], fov, num_rays, _p(ranges, _f32p))
        return ranges

    def lut_rays(self, lut, ins, nthreads=1):
        ins = np.ascontiguousarray(ins, dtype=np.float32).reshape(-1, 3)
        ranges = np.empty(ins.shape[0], dtype=np.float32)
        lib().orc_lut_rays(C.byref(self._m), _p(lut, _u16p), lut.shape[2], self.max_range_px,
                           _p(ins, _f32p), ins.shape[0], _p(ranges, _f32p), nthreads)
        return ranges

    # -- CDDT ------------------------------------------------------------------
    def _cddt_handle(self, theta_disc):
        if theta_disc not in self._cddt:
            self._cddt[theta_disc] = lib().orc_cddt_build(C.byref(self._m), theta_disc)
        return self._cddt[theta_disc]

    def cddt_fan(self, theta_disc, poses, fov, num_rays, nthreads=1):
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        ranges = np.empty(poses.shape[0] * num_rays, dtype=np.float32)
        lib().orc_cddt_fan(C.byref(self._m), self._cddt_handle(theta_disc), self.max_range_px,
                           _p(poses, _f32p), poses.shape[0], fov, num_rays, _p(ranges, _f32p),
                           nthreads)
        return ranges

    def cddt_rays_libm(self, theta_disc, ins):
        """The 2-argument per-ray query on the upstream-literal table (libm trig per bin, un-fused projection,
        fmod + roundf bin rule) — what scripts/two_player/scan.py:57-70 feeds: per-ray float32 thetas."""
        key = ("libm", theta_disc)
        if key not in self._cddt:
            self._cddt[key] = lib().orc_cddt_build_libm(C.byref(self._m), theta_disc)
        ins = np.ascontiguousarray(ins, dtype=np.float32).reshape(-1, 3)
        ranges = np.empty(ins.shape[0], dtype=np.float32)
        lib().orc_cddt_rays_libm(C.byref(self._m), self._cddt[key], self.max_range_px, _p(ins, _f32p), ins.shape[0],
                                 _p(ranges, _f32p))
        return ranges

    def cddt_rays(self, theta_disc, ins, nthreads=1):
        ins = np.ascontiguousarray(ins, dtype=np.float32).reshape(-1, 3)
        ranges = np.empty(ins.shape[0], dtype=np.float32)
        lib().orc_cddt_rays(C.byref(self._m), self._cddt_handle(theta_disc), self.max_range_px,
                            _p(ins, _f32p), ins.shape[0], _p(ranges, _f32p), nthreads)
        return ranges


def libm_sincosf(x):
    """(sinf, cosf) of a float32 array by THIS host's libm."""
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().orc_libm_sincosf(_p(x, _f32p), x.size, _p(s, _f32p), _p(c, _f32p))
    return s, c


def lit_sincosf(x):
    """(sinf, cosf) by the statement of glibc's algorithm that the product's audit mode runs on the device."""
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().orc_lit_sincosf(_p(x, _f32p), x.size, _p(s, _f32p), _p(c, _f32p))
    return s, c


def libm_restatement_mismatches(first=0, step=1, nthreads=0):
    """Float bit patterns first, first + step, ... (both signs): (sinf, cosf) inputs on which that statement and this
    host's libm differ.  step 1 walks every finite float."""
    bad_cos = C.c_long(0)
    bad_sin = lib().orc_libm_restatement_check(first, step, C.byref(bad_cos), nthreads or max_threads())
    return int(bad_sin), int(bad_cos.value)


def edge_distances(num_rays, min_ang, inc, scan_dist_to_base, width, wheelbase):
    out = np.empty(num_rays, dtype=np.float64)
    lib().orc_edge_distances(num_rays, min_ang, inc, scan_dist_to_base, width, wheelbase,
                             _p(out, _f64p))
    return out


def is_crashed(rays, num_rays, poses, edge, crash_thresh):
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    edge = np.ascontiguousarray(edge, dtype=np.float64)
    return int(lib().orc_is_crashed(_p(rays, _f32p), num_rays, poses, _p(edge, _f64p),
                                    crash_thresh))


def followgap_eval(lidar, max_distance, max_angle, angle_inc, size=None):
    """FollowGap::eval restatement (followgap/followgap.hpp:104-129) for one scan."""
    lidar = np.ascontiguousarray(lidar, dtype=np.float32)
    n = len(lidar) if size is None else int(size)
    return float(lib().orc_followgap_eval(_p(lidar, _f32p), n, max_distance, max_angle, angle_inc))


def max_threads():
    return int(lib().orc_max_threads())
