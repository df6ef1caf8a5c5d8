#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ and pyracecarsimulator_amd/data/.

Run in the build container only (reads /root/reference; the GPU box never does):

* pyracecarsimulator_amd/data/colombia_map.npz   the one map the reference mount holds (maps/colombia/map.pgm +
                          map.yaml values), re-encoded — a data file, not source.
* golden/protocol.json    GOLD-C: call protocol of the reference's own ScanSimulator2D
                          (scripts/scan_simulator.py run through lib2to3 on a /tmp copy, with a
                          recording stub in place of range_libc): shapes, dtypes, live rows,
                          fov/num_rays arguments, buffer aliasing.
* golden/car_ref.npz      GOLD-D: outputs of the reference's compiled Car (oracle/_ref):
                          setCarEdgeDistances-driven isCrashed codes on seeded scans.
* golden/followgap_ref.npz  GOLD-E: steering angles of the reference's compiled FollowGap::eval
                          (followgap/followgap.hpp via oracle/_ref/libfollowgap_ref.so) on real
                          scans (rm_colombia ranges), random scans and edge cases.
* golden/rm_*.npz         GOLD-A/B: ranges, hit cells, step counts of the C oracle on seeded
                          poses (cross-checked here against the independent NumPy statement
                          before being written).
* golden/table_libm_forms.npz  CDDT / GiantLUT queries of the same poses through the upstream-literal TABLE statements
* golden/rm_libm_forms.npz  the same poses (first 16 per map) through the upstream-literal libm
                          form of the oracle: what the canonical form's deviation is gated against.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

from pyracecarsimulator_amd import maps  # noqa: E402
from oracle import oracle as O, np_statement as N  # noqa: E402

CAR = dict(wb=0.3302, fc=0.523, h_cg=0.074, l_f=0.15875, l_r=0.17145, cs_f=4.718, cs_r=5.4562,
           mass=3.47, I_z=0.04712, ttc_thresh=0.001, width=0.2032, length=0.5,
           max_steer_vel=3.2, max_steer_ang=0.4189, max_speed=7.0, max_accel=7.51,
           max_decel=8.26)


def colombia():
    img = maps.read_pgm(os.path.join(REF, "maps/colombia/map.pgm"))
    meta = maps.read_map_yaml(os.path.join(REF, "maps/colombia/map.yaml"))
    np.savez_compressed(os.path.join(ROOT, "pyracecarsimulator_amd/data/colombia_map.npz"),
                        image=img, resolution=meta["resolution"], origin=np.array(meta["origin"]),
                        negate=meta["negate"], occupied_thresh=meta["occupied_thresh"],
                        free_thresh=meta["free_thresh"])


def protocol():
    tmp = tempfile.mkdtemp(prefix="scanproto")
    shutil.copy(os.path.join(REF, "scripts/scan_simulator.py"), tmp)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", tmp],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    calls = []

    class _Rec:
        def __init__(self, *a):
            self.args = a

        def calc_range_many(self, ins, outs, *rest):
            calls.append({"ins_shape": list(ins.shape), "ins_dtype": str(ins.dtype),
                          "ins_c_contiguous": bool(ins.flags.c_contiguous),
                          "outs_shape": list(outs.shape), "outs_dtype": str(outs.dtype),
                          "nonzero_rows": np.nonzero(ins.any(axis=1))[0].tolist(),
                          "extra_args": [float(rest[0]), int(rest[1])] if rest else []})
            outs[:] = np.arange(outs.size, dtype=np.float32)

    stub = types.ModuleType("range_libc")
    stub.PyRayMarching = type("PyRayMarching", (_Rec,), {})
    stub.PyRayMarchingGPU = type("PyRayMarchingGPU", (_Rec,), {})
    sys.modules["range_libc"] = stub
    sys.path.insert(0, tmp)
    import scan_simulator as ref
    sim = ref.ScanSimulator2D(1081, 4.71, 0.01, batch_size=4)
    sim.setMap("omap", 300, 0.05, (0.0, 0.0, 0.0))
    sim.setRaytracingMethod("RMGPU")
    method_cls = type(sim.scan_method).__name__
    method_args = [str(a) for a in sim.scan_method.args]
    out1 = sim.scan(1.0, 2.0, 0.5)
    alias1 = out1 is sim.output_vector
    poses = np.array([[1, 2, 0.1], [3, 4, 0.2], [5, 6, 0.3], [7, 8, 0.4], [9, 9, 9]], np.float32)
    out2 = sim.scanMany(poses)
    alias2 = out2 is sim.output_vector_many
    proto = {"source": "scripts/scan_simulator.py (lib2to3 copy, recording range_libc stub)",
             "ctor": {"num_rays": 1081, "fov": 4.71, "scan_std": 0.01, "batch_size": 4},
             "attributes": sorted(k for k in vars(sim) if not k.startswith("_")),
             "method_class_for_RMGPU": method_cls, "method_ctor_args": method_args,
             "scan_call": calls[0], "scan_returns_cached_buffer": alias1,
             "scanMany_call": calls[1], "scanMany_returns_cached_buffer": alias2,
             "scanMany_pose_rows_used": 4, "scanMany_poses_given": 5}
    sys.path.remove(tmp)
    del sys.modules["range_libc"], sys.modules["scan_simulator"]
    with open(os.path.join(GOLD, "protocol.json"), "w") as f:
        json.dump(proto, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


def car_ref():
    so = os.path.join(ROOT, "oracle/_ref/libracecar_ref.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"],
                          stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.ref_car_create.restype = C.c_void_p
    L.ref_car_create.argtypes = [C.POINTER(C.c_double)]
    L.ref_car_set_edge_distances.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.ref_car_is_crashed.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
    L.ref_car_is_crashed.restype = C.c_int
    order = ["wb", "fc", "h_cg", "l_f", "l_r", "cs_f", "cs_r", "mass", "I_z", "ttc_thresh",
             "width", "length", "max_steer_vel", "max_steer_ang", "max_speed", "max_accel",
             "max_decel"]
    args = (C.c_double * 17)(*[CAR[k] for k in order])
    car = L.ref_car_create(args)
    cases = []
    rng = np.random.default_rng(42)
    for num_rays, fov, poses in [(1081, 4.71, 1), (1081, 4.71, 6), (1080, 4.71, 5), (720, 3.14, 4),
                                 (64, 6.2, 3)]:
        L.ref_car_set_edge_distances(car, num_rays, -fov / 2.0, fov / num_rays, 0.275)
        edge = O.edge_distances(num_rays, -fov / 2.0, fov / num_rays, 0.275, CAR["width"], CAR["wb"])
        for trial in range(6):
            rays = rng.uniform(0.3, 15.0, size=poses * num_rays).astype(np.float32)
            if trial % 3 == 1:      # plant a crash in a random pose/beam
                p, j = rng.integers(0, poses), rng.integers(0, num_rays)
                rays[p * num_rays + j] = np.float32(max(edge[j], 0.0) * 0.5)
            if trial % 3 == 2:      # borderline: exactly at edge + thresh
                p, j = rng.integers(0, poses), rng.integers(0, num_rays)
                rays[p * num_rays + j] = np.float32(edge[j] + CAR["ttc_thresh"])
            code = L.ref_car_is_crashed(car, rays.ctypes.data_as(C.POINTER(C.c_float)), num_rays,
                                        poses)
            cases.append((num_rays, fov, poses, rays, code))
    np.savez_compressed(
        os.path.join(GOLD, "car_ref.npz"),
        car=np.array([CAR[k] for k in order]), car_keys=np.array(order),
        num_rays=np.array([c[0] for c in cases]), fov=np.array([c[1] for c in cases]),
        poses=np.array([c[2] for c in cases]), codes=np.array([c[4] for c in cases]),
        **{"rays_%d" % i: c[3] for i, c in enumerate(cases)})


def car_rollouts():
    """GOLD-D2: roll-outs integrated by the reference's compiled Car (control + updatePosition,
    racecar.cpp:53-98,294-303) with the action schedule of scripts/mcts.py:214-231."""
    from pyracecarsimulator_amd import racecar as RC
    L = C.CDLL(os.path.join(ROOT, "oracle/_ref/libracecar_ref.so"))
    L.ref_car_create.restype = C.c_void_p
    L.ref_car_create.argtypes = [C.POINTER(C.c_double)]
    L.ref_car_control.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.ref_car_update_position.argtypes = [C.c_void_p, C.c_double]
    L.ref_car_get_state.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.ref_car_set_state.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    params = np.array([RC.DEFAULT_CAR[k] for k in RC.CAR_PARAM_ORDER])
    car = L.ref_car_create((C.c_double * 17)(*params))
    rng = np.random.default_rng(123)
    R, n_steps, every = 48, 200, 10
    states = np.zeros((R, 11))
    states[:, 0] = rng.uniform(-5, 5, R)
    states[:, 1] = rng.uniform(-5, 5, R)
    states[:, 2] = rng.uniform(-3.1, 3.1, R)
    states[:, 3] = rng.uniform(0, 6, R) * (rng.random(R) < 0.8)
    states[:, 4] = rng.uniform(-0.4, 0.4, R)
    states[R // 2:, 5] = rng.uniform(-1, 1, R - R // 2)
    states[R // 2:, 6] = rng.uniform(-0.1, 0.1, R - R // 2)
    states[R // 2:, 7] = 1.0
    actions = np.stack([rng.uniform(0, 7, (R, 20)), rng.uniform(-0.4189, 0.4189, (R, 20))], -1)
    poses = np.zeros((R, n_steps, 3), np.float32)
    vel = np.zeros((R, n_steps))
    out = np.zeros((R, 11))
    for r in range(R):
        st = (C.c_double * 11)(*states[r])
        L.ref_car_set_state(car, st)
        for i in range(n_steps):
            L.ref_car_control(car, actions[r, i // every, 0], actions[r, i // every, 1])
            L.ref_car_update_position(car, 0.01)
            L.ref_car_get_state(car, st)
            poses[r, i] = (st[0], st[1], st[2])
            vel[r, i] = st[3]
        out[r] = list(st)
    np.savez_compressed(os.path.join(GOLD, "car_rollouts_ref.npz"), params=params, states=states,
                        actions=actions, poses=poses, velocities=vel, final=out, n_steps=n_steps,
                        action_every=every, dt=0.01)


def followgap_ref():
    """GOLD-E.  Needs rm_colombia.npz (made by rm_golden) for the real scans."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "oracle/_ref/libfollowgap_ref.so"))
    L.ref_followgap_eval.restype = C.c_float
    L.ref_followgap_eval.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    rng = np.random.default_rng(77)
    scans = []
    z = np.load(os.path.join(GOLD, "rm_colombia.npz"))
    B = int(z["num_rays"])
    real = z["ranges_gpu"].reshape(-1, B)
    scans += [real[i].copy() for i in range(len(real))]                    # 1081-beam scans
    scans += [real[i, 180:900].copy() for i in range(0, len(real), 3)]     # the policy's 720-beam window
    for size in (10, 11, 17, 63, 64, 65, 128, 720, 1081, 2000):
        for kind in range(6):
            if kind == 0:
                v = rng.uniform(0.0, 20.0, size)
            elif kind == 1:
                v = rng.uniform(0.0, 1.7, size)                            # no beam above the 1.75 m gap level
            elif kind == 2:
                v = np.where(rng.random(size) < 0.3, 0.0, rng.uniform(0.5, 16.0, size))   # zeros
            elif kind == 3:
                v = np.full(size, 3.0)                                     # all equal (argmin tie -> 0)
            elif kind == 4:
                v = rng.choice([1.75, 1.7500001, 1.7499999, 15.0, 15.000001, 30.0], size)  # thresholds
            else:
                v = rng.normal(3.0, 3.0, size)                             # negative ranges (noise)
            scans.append(v.astype(np.float32))
    params = (10, 15.0, 0.4189, 0.004)                                     # scripts/mcts.py:97-99
    keep, angles = [], []
    for v in scans:
        # the reference reads one past the array when the chosen gap is the single last beam: skip
        w = v.copy()
        w[: len(w) - 10] = np.minimum(w[: len(w) - 10], np.float32(params[1]))
        mp = 0
        for i in range(len(w)):
            if w[i] != 0 and w[i] < w[mp]:
                mp = i
        w[mp] = 0
        for i in range(-5, 5):
            if 0 < mp + i < len(w) - 1:
                w[mp + i] = 0
        f = np.concatenate([[0], (w > 1.75).astype(np.int8), [0]])
        d = np.diff(f)
        starts, ends = np.where(d == 1)[0], np.where(d == -1)[0]
        if len(starts):
            k = int(np.argmax(ends - starts))
            best = (2 * int(starts[k]) + int(ends[k] - starts[k]) + 1) // 2
            if best >= len(v):
                continue
        buf = np.ascontiguousarray(v, dtype=np.float32)
        a = L.ref_followgap_eval(buf.ctypes.data_as(C.POINTER(C.c_float)), len(buf), *params)
        keep.append(buf)
        angles.append(a)
        assert np.float32(O.followgap_eval(buf, *params[1:])).tobytes() == np.float32(a).tobytes() or \
            (np.isnan(a) and np.isnan(O.followgap_eval(buf, *params[1:]))), (len(buf), a)
    offs = np.cumsum([0] + [len(k) for k in keep])
    np.savez_compressed(os.path.join(GOLD, "followgap_ref.npz"), scans=np.concatenate(keep),
                        offsets=offs, angles=np.array(angles, np.float32),
                        params=np.array(params, np.float64))
    print("followgap_ref", len(keep), "scans",
          os.path.getsize(os.path.join(GOLD, "followgap_ref.npz")) // 1024, "KiB")


def rm_golden(name, g, n_poses, seed, mrx=300, fov=4.71, num_rays=1081):
    om = O.OracleMap.from_gridmap(g, mrx)
    assert np.array_equal(om.dt, N.edt(g.occ)), "EDT: C oracle vs scipy statement"
    poses = maps.sample_free_poses(g, n_poses, seed, dt=om.dt)
    out = {"poses": poses, "fov": fov, "num_rays": num_rays, "max_range_px": mrx,
           "occ_packed": np.packbits(g.occ, axis=1), "shape": np.array(g.occ.shape),
           "resolution": g.resolution, "origin": np.array(g.origin)}
    for tag, sc in (("cpu", 0.999), ("gpu", 1.0)):
        r, h, s = om.rm_fan(poses, fov, num_rays, step_coeff=sc)
        r2, h2, s2 = N.rm_fan(g.occ, g.resolution, g.origin, mrx, poses, fov, num_rays, sc)
        assert np.array_equal(r, r2) and np.array_equal(h, h2) and np.array_equal(s, s2), \
            "C oracle vs NumPy statement disagree on " + name
        out["ranges_" + tag] = r
        out["hits_" + tag] = h.astype(np.int16)
        out["steps_" + tag] = s
    rb, hb, sb = om.bl_fan(poses, fov, num_rays)
    out["ranges_bl"] = rb
    out["hits_bl"] = hb.astype(np.int16)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "poses", n_poses, "mean steps", out["steps_cpu"].mean(),
          os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024, "KiB")


def libm_forms(n_poses=16):
    """GOLD-A/B in the UPSTREAM-LITERAL form (orc_rm_fan_libm: one libm cosf/sinf cast per beam at
    theta + (-fov/2 + j*fov/B) rounded to float32, every product and sum its own rounding, calc_range(y, x,
    theta') — the closest available statement of range_libc's own arithmetic, which is absent from the
    reference mount) for the first ``n_poses`` poses of every rm_*.npz: hit cells, ranges, sample counts for
    both step coefficients.  The GPU tests gate the canonical form's deviation from these numbers
    (tests/test_gpu_parity.py::test_device_fan_vs_upstream_literal_libm_form) instead of leaving it in prose.
    Generated with this container's glibc; the arrays are committed data."""
    out = {}
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        z = dict(np.load(os.path.join(GOLD, name + ".npz")))
        rows, cols = (int(v) for v in z["shape"])
        occ = np.ascontiguousarray(np.unpackbits(z["occ_packed"], axis=1)[:, :cols].astype(np.uint8))
        g = maps.GridMap(occ, float(z["resolution"]), tuple(float(v) for v in z["origin"]), name)
        om = O.OracleMap.from_gridmap(g, int(z["max_range_px"]))
        poses = z["poses"][:n_poses]
        for tag, sc in (("cpu", 0.999), ("gpu", 1.0)):
            r, h, s_ = om.rm_fan_libm(poses, float(z["fov"]), int(z["num_rays"]), step_coeff=sc)
            out["%s_ranges_%s" % (name, tag)] = r
            out["%s_hits_%s" % (name, tag)] = h.astype(np.int16)
            out["%s_steps_%s" % (name, tag)] = s_
        out[name + "_n_poses"] = np.int32(len(poses))
    np.savez_compressed(os.path.join(GOLD, "rm_libm_forms.npz"), **out)
    print("rm_libm_forms", os.path.getsize(os.path.join(GOLD, "rm_libm_forms.npz")) // 1024, "KiB")


def table_libm_forms(n_poses=8):
    """The TABLE methods in the upstream-literal form (orc_cddt_build_libm / orc_cddt_rays_libm,
    orc_lut_build_libm / orc_lut_fan_rows_libm: libm cosf / sinf per bin of a double-precision bin angle, un-fused
    projection and world->grid, fmod + roundf bin rule) for the first ``n_poses`` poses of every rm_*.npz:
    * CDDT theta_disc 112 (the reference's, scripts/two_player/rcs_two_player.py:121) and 360, queried the way the
      reference's only CDDT user does — the 2-argument per-ray form with host-built float32 thetas
      (scripts/two_player/scan.py:57-70);
    * GiantLUT theta_disc 180: the literal theta rows of the sampled poses' cells + the literal fan query on them.
    The GPU tests gate the device tables' distance to these numbers
    (tests/test_gpu_parity.py::test_device_tables_vs_upstream_literal_libm_forms).  Generated with this container's
    glibc; the arrays are committed data."""
    out = {}
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        z = dict(np.load(os.path.join(GOLD, name + ".npz")))
        rows, cols = (int(v) for v in z["shape"])
        occ = np.ascontiguousarray(np.unpackbits(z["occ_packed"], axis=1)[:, :cols].astype(np.uint8))
        g = maps.GridMap(occ, float(z["resolution"]), tuple(float(v) for v in z["origin"]), name)
        mrx = int(z["max_range_px"])
        om = O.OracleMap.from_gridmap(g, mrx)
        poses = np.ascontiguousarray(z["poses"][:n_poses])
        fov, B = float(z["fov"]), int(z["num_rays"])
        ins = table_ray_rows(poses, fov, B)
        for td in (112, 360):
            out["%s_cddt%d" % (name, td)] = om.cddt_rays_libm(td, ins)
        td = 180
        rr, cc = om.lut_pose_cells(poses)
        pose_rows = np.zeros((len(poses), td), np.uint16)
        for i, (r_, c_) in enumerate(zip(rr, cc)):
            if r_ >= 0:
                pose_rows[i] = om.lut_build_libm(td, int(r_), int(r_) + 1, nthreads=1)[0, int(c_)]
        out[name + "_lut180_rows"] = pose_rows
        out[name + "_lut180_fan"] = om.lut_fan_rows_libm(pose_rows, poses, fov, B)
        out[name + "_n_poses"] = np.int32(len(poses))
    np.savez_compressed(os.path.join(GOLD, "table_libm_forms.npz"), **out)
    print("table_libm_forms", os.path.getsize(os.path.join(GOLD, "table_libm_forms.npz")) // 1024, "KiB")


def bl_libm_forms(n_poses=8):
    """BresenhamsLine in the upstream-literal form (orc_bl_rays_libm / orc_bl_fan_libm: libm cosf / sinf of
    theta' = -theta + rotation constant, calc_range(y, x, theta'), un-fused end point and hit distance) for the first
    ``n_poses`` poses of every rm_*.npz: the 2-argument per-ray form on host-built float32 thetas and the 4-argument
    fan form; ranges, hit cells, step counts.  The GPU test gates the device's canonical walk against them
    (tests/test_gpu_parity.py::test_device_bresenham_vs_upstream_literal_libm_form).  Generated with this container's
    glibc; the arrays are committed data."""
    out = {}
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        z = dict(np.load(os.path.join(GOLD, name + ".npz")))
        rows, cols = (int(v) for v in z["shape"])
        occ = np.ascontiguousarray(np.unpackbits(z["occ_packed"], axis=1)[:, :cols].astype(np.uint8))
        g = maps.GridMap(occ, float(z["resolution"]), tuple(float(v) for v in z["origin"]), name)
        om = O.OracleMap.from_gridmap(g, int(z["max_range_px"]))
        poses = np.ascontiguousarray(z["poses"][:n_poses])
        fov, B = float(z["fov"]), int(z["num_rays"])
        r, h, s = om.bl_rays_libm(table_ray_rows(poses, fov, B))
        out[name + "_rays_ranges"], out[name + "_rays_hits"], out[name + "_rays_steps"] = r, h.astype(np.int16), s
        r, h, s = om.bl_fan_libm(poses, fov, B)
        out[name + "_fan_ranges"], out[name + "_fan_hits"], out[name + "_fan_steps"] = r, h.astype(np.int16), s
        out[name + "_n_poses"] = np.int32(len(poses))
    np.savez_compressed(os.path.join(GOLD, "bl_libm_forms.npz"), **out)
    print("bl_libm_forms", os.path.getsize(os.path.join(GOLD, "bl_libm_forms.npz")) // 1024, "KiB")


def table_ray_rows(poses, fov, B):
    """(x, y, theta) rows of the 2-argument form for the fans of ``poses``: theta = heading + the float32 np.arange
    angle row scripts/two_player/scan.py:57-62 builds on the host."""
    ang = (np.float32(-0.5) * np.float32(fov) + np.arange(B, dtype=np.float32) * (np.float32(fov) / np.float32(B))).astype(np.float32)
    ins = np.zeros((len(poses) * B, 3), np.float32)
    for p in range(len(poses)):
        ins[p * B:(p + 1) * B, :2] = poses[p, :2]
        ins[p * B:(p + 1) * B, 2] = poses[p, 2] + ang
    return ins


def main():
    os.makedirs(GOLD, exist_ok=True)
    colombia()
    protocol()
    car_ref()
    car_rollouts()
    rm_golden("rm_colombia", maps.load_colombia(), 64, 101)        # GOLD-A: 64 poses
    rm_golden("rm_maze256", maps.make_maze(256, cell=32, wall=2, p=0.45, seed=7), 64, 102)   # GOLD-B
    g = maps.make_maze(192, cell=24, wall=2, p=0.5, seed=9, resolution=0.1,
                       origin=(-3.0, 2.5, 0.6))           # rotated origin (yaw != 0)
    rm_golden("rm_maze192_yaw", g, 16, 103, mrx=120, fov=6.0, num_rays=360)
    libm_forms()
    table_libm_forms()
    bl_libm_forms()
    followgap_ref()


if __name__ == "__main__":
    main()
