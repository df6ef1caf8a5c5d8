"""ctypes loader for libscan_amd.so (C ABI: include/scanlib.h).

The library is the product: hand-written HIP kernels for gfx950 behind a C ABI.
There is no Python/NumPy fallback — if the shared object is missing or no HIP
device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
#: SCANLIB_SO: an alternative build of the library (A/B runs of two builds on one box, tools/ab_so.sh)
_SO = os.environ.get("SCANLIB_SO") or os.path.join(_HERE, "libscan_amd.so")
_LIB = None

RL_OK = 0
RL_BRESENHAM, RL_RM, RL_RM_GPU, RL_CDDT, RL_GIANT_LUT = 0, 1, 2, 3, 4

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
u16p = C.POINTER(C.c_uint16)
u8p = C.POINTER(C.c_uint8)



class PlanOpts(C.Structure):
    """rl_plan_opts (include/scanlib.h): the options that shape a launch."""
    _fields_ = [(n, C.c_int) for n in (
        "variant", "grid_mult", "wg_threads", "low_water", "sort_poses", "xcd_bands", "slots", "tiled",
        "inline_prep", "inline_max", "inline_map_kb", "stripe_max", "order_inline", "bin_multi_min",
        "bin_generic", "run_log2", "cddt_bins", "cddt_sort", "lut_debug", "debug_stamps", "slice_log2",
        "cddt_theta_min", "cddt_search", "code_map", "code_min_rays", "tail_pct", "tail_wg_pct", "code_entries")]


class LaunchPlan(C.Structure):
    """rl_launch_plan (include/scanlib.h)."""
    _fields_ = [(n, C.c_int) for n in (
        "kernel", "grid", "block", "lds_bytes", "binning", "record_source", "slots", "bands", "run_log2",
        "k_max", "tiled", "aux", "crash", "nl", "ch", "slices", "slice_poses", "code", "code_entries", "gen1")] + [("name", C.c_char * 192)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "name"}
        d["name"] = self.name.decode()
        d["kernel"] = KERNEL_IDS.get(self.kernel, str(self.kernel))
        d["binning"] = BINNINGS.get(self.binning, str(self.binning))
        return d


KERNEL_IDS = {0: "none", 1: "rm_chunk", 2: "rm_stream", 3: "occ_lds", 4: "bl_stream", 5: "bl_lds", 6: "lut_lds",
              7: "lut_fan", 8: "cddt_bins", 9: "cddt_rays", 10: "cddt_theta", 11: "rm_literal", 12: "rm_stream_literal"}
BINNINGS = {0: "none", 1: "small_keys", 2: "small_records", 3: "grid_sort", 4: "grid_unsorted", 5: "generic"}

#: every symbol include/scanlib.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rl_version": (C.c_char_p, []),
    "rl_last_error": (C.c_char_p, []),
    "rl_device_count": (C.c_int, []),
    "rl_map_create": (C.c_int, [u8p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.POINTER(C.c_void_p)]),
    "rl_map_create_multi": (C.c_int, [u8p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "rl_map_n_devices": (C.c_int, [C.c_void_p]),
    "rl_map_replica": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rl_map_update": (C.c_int, [C.c_void_p, u8p]),
    "rl_map_stamp_cells": (C.c_int, [C.c_void_p, i32p, C.c_int, C.c_uint8]),
    "rl_map_destroy": (None, [C.c_void_p]),
    "rl_map_rows": (C.c_int, [C.c_void_p]),
    "rl_map_cols": (C.c_int, [C.c_void_p]),
    "rl_map_device": (C.c_int, [C.c_void_p]),
    "rl_map_get_dt": (C.c_int, [C.c_void_p, f32p]),
    "rl_map_get_occ": (C.c_int, [C.c_void_p, u8p]),
    "rl_method_create": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int,
                                   C.POINTER(C.c_void_p)]),
    "rl_method_destroy": (None, [C.c_void_p]),
    "rl_method_kind": (C.c_int, [C.c_void_p]),
    "rl_method_n_devices": (C.c_int, [C.c_void_p]),
    "rl_method_replica": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rl_calc_range_many": (C.c_int, [C.c_void_p, f32p, f32p, C.c_int]),
    "rl_calc_range_many_fan": (C.c_int, [C.c_void_p, f32p, f32p, C.c_int, C.c_float, C.c_int]),
    "rl_calc_range_fan": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_float, C.c_int, f32p, i32p,
                                    u16p]),
    "rl_calc_range_fan_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rl_calc_range_many_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p]),
    "rl_set_noise": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64, C.c_uint64]),
    "rl_check_collision_many": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_float, C.c_int, f64p,
                                          C.c_double, C.POINTER(C.c_int), f32p]),
    "rl_check_collision_groups": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_int, C.c_float, C.c_int, f64p,
                                            C.c_double, C.POINTER(C.c_int), f32p]),
    "rl_check_collision_groups_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                                   C.c_int, C.c_void_p, C.c_double, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]),
    "rl_calc_range_fan_multi_device": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "rl_check_collision_groups_multi_device": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_int, C.c_float, C.c_int, f64p,
                                                         C.c_double, C.c_int, C.c_void_p]),
    "rl_followgap_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                      C.POINTER(C.c_void_p)]),
    "rl_followgap_destroy": (None, [C.c_void_p]),
    "rl_followgap_eval": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_int, f32p]),
    "rl_followgap_eval_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p]),
    "rl_car_create": (C.c_int, [C.c_int, f64p, C.POINTER(C.c_void_p)]),
    "rl_car_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, f64p, C.POINTER(C.c_void_p)]),
    "rl_car_destroy": (None, [C.c_void_p]),
    "rl_car_rollout": (C.c_int, [C.c_void_p, f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_double, f32p,
                                 f64p, f64p]),
    "rl_car_rollout_check": (C.c_int, [C.c_void_p, C.c_void_p, f64p, f64p, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.c_float, C.c_int, f64p, C.c_double,
                                       C.POINTER(C.c_int), f64p, f64p]),
    "rl_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "rl_host_free": (C.c_int, [C.c_void_p]),
    "rl_car_edge_distances": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                        f64p]),
    "rl_car_is_crashed": (C.c_int, [f32p, C.c_int, C.c_int, f64p, C.c_double, C.POINTER(C.c_int)]),
    "rl_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rl_method_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "rl_method_get_info": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "rl_method_read_lut": (C.c_int, [C.c_void_p, C.c_int, C.c_int, u16p]),
    "rl_debug_read_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "rl_probe_gather_rate": (C.c_int, [C.c_int, C.c_int, f64p, f64p, C.POINTER(C.c_int)]),
    "rl_plan_default_opts": (C.c_int, [C.POINTER(PlanOpts)]),
    "rl_plan_fan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(PlanOpts), C.c_int,
                              C.c_int, C.c_int, C.c_int, C.POINTER(LaunchPlan)]),
    "rl_method_plan_fan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(LaunchPlan)]),
    "rl_method_last_plan": (C.c_int, [C.c_void_p, C.POINTER(LaunchPlan)]),
    "rl_launch_contexts": (C.c_int, []),
    "rl_probe_hbm": (C.c_int, [C.c_int, C.c_size_t, f64p]),
    "rl_probe_hbm_nt": (C.c_int, [C.c_int, C.c_size_t, f64p]),
    "rl_probe_literal_sincosf": (C.c_int, [C.c_int, f32p, C.c_size_t, f32p, f32p]),
    "rl_ranges_to_u16_device": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
    "rl_ranges_from_u16_device": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
}


def plan_fan(kind, rows, cols, n_poses, num_rays, max_range_px=300.0, theta_disc=0, n_cu=256, aux=False,
             crash=False, **opts):
    """The launch plan of a fan call (rl_plan_fan: pure host arithmetic, no device needed) as a dict.
    ``opts`` override fields of the default rl_plan_opts."""
    o = PlanOpts()
    check(lib().rl_plan_default_opts(C.byref(o)))
    for k, v in opts.items():
        if not hasattr(o, k):
            raise KeyError("unknown plan option %r" % k)
        setattr(o, k, int(v))
    pl = LaunchPlan()
    check(lib().rl_plan_fan(int(kind), int(n_cu), int(rows), int(cols), float(max_range_px), int(theta_disc),
                            C.byref(o), int(n_poses), int(num_rays), int(bool(aux)), int(bool(crash)), C.byref(pl)))
    return pl.as_dict()


class ScanLibError(RuntimeError):
    """A libscan_amd.so call returned a negative rl_status."""

    def __init__(self, code, msg):
        super().__init__("libscan_amd: %s (rl_status %d)" % (msg, code))
        self.code = code


def build(force: bool = False) -> str:
    """Compile libscan_amd.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))
            if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "scanlib.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO)
                                             for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-j4", "-C", os.path.join(_HERE, "csrc"), "../libscan_amd.so"]
                              + (["-B"] if force else []))
    return _SO


def _share_torch_hip_runtime():
    """PyTorch wheels carry their own copy of the HIP/HSA runtime (torch/lib/libamdhip64.so, same
    SONAME as /opt/rocm's).  A process must run ONE runtime: if libscan_amd.so pulls in the system
    copy first and torch is imported later, torch's HSA runtime finds the device already taken
    ("No HIP GPUs are available").  So when torch is installed but not imported yet, its runtime is
    loaded first and libscan_amd.so binds to it — the same arrangement as importing torch before
    this package.  Without torch (or with SCANLIB_SYSTEM_HIP=1) the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("SCANLIB_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libscan_amd.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise ImportError(
                "libscan_amd.so is missing: run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or make -C pyracecarsimulator_amd/csrc). There is no CPU fallback.")
        _share_torch_hip_runtime()
        L = C.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("SCANLIB_SO"):          # an older build in an A/B run: newer entry points absent
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


_RAW = {}


def raw(name):
    """The same entry point with every pointer argument typed ``void*``: callers that keep the
    addresses of long-lived buffers (ints) skip the per-call ``ndarray.ctypes.data_as`` objects —
    a few microseconds on the 25-us scan() path."""
    fn = _RAW.get(name)
    if fn is None:
        res, args = SYMBOLS[name]
        args = [C.c_void_p if (isinstance(a, type) and issubclass(a, C._Pointer)) else a for a in args]
        fn = C.CFUNCTYPE(res, *args)((name, lib()))
        _RAW[name] = fn
    return fn


def check(code: int) -> None:
    if code != RL_OK:
        raise ScanLibError(code, lib().rl_last_error().decode("utf-8", "replace"))


class _PinnedOwner:
    """Keeps one rl_host_alloc block alive for the NumPy array built on it."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr and _LIB is not None:
                _LIB.rl_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_zeros(shape, dtype):
    """A zeroed NumPy array in pinned host memory of the library (rl_host_alloc): host-pointer scans
    write their ranges straight into it.  Falls back to ordinary memory when no device is usable (the
    array is then just an array; scans would fail earlier anyway)."""
    import numpy as np
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    p = C.c_void_p()
    try:
        rc = lib().rl_host_alloc(max(n, 1) * dt.itemsize, C.byref(p))
    except Exception:
        rc = -1
    if rc != RL_OK or not p.value:
        return np.zeros(shape, dtype=dt)
    nbytes = max(n, 1) * dt.itemsize

    class _Block(C.c_char * nbytes):              # (a Python subclass: instances can carry the owner)
        pass

    buf = _Block.from_address(p.value)
    buf._owner = _PinnedOwner(p.value)            # freed when the last array on the block is gone
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
