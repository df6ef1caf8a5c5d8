#!/usr/bin/env python3
"""Randomised parity fuzz on the GPU: random map shapes / origins / yaw / ranges / fans / batch
sizes, every method against the CPU oracle, bit-exact.  Prints the first mismatch with its seed."""
import os, sys, time, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyracecarsimulator_amd import maps, range_libc
from oracle import oracle as O

def run(seconds, seed):
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    n_cases = 0
    t0 = t_mark = time.time()
    while time.time() < t_end:
        n_cases += one_case(int(rng.integers(0, 2**31)))
        if time.time() - t_mark >= 300.0:          # a progress line every five minutes (a run cut short still says how far it got)
            t_mark = time.time()
            print("... %d cases after %.0f s, all bit-identical so far" % (n_cases, t_mark - t0), flush=True)
    return n_cases


def followgap_case(r):
    """FollowGap::eval for a batch of random scans (consumer_kernels.h: followgap_bits_kernel up to 1280 beams, followgap_kernel
    beyond) against the CPU restatement, bit for bit: runs of random lengths around the 1.75 threshold, zeros, NaNs, the minimum
    anywhere, random clamp / steering limits."""
    from pyracecarsimulator_amd.followgap import PyFollowGap
    size = int(r.choice([10, 11, 37, 64, 65, 127, 128, 360, 720, 1081, 1280, 1281, int(r.integers(10, 1400)), int(r.integers(10, 3000))]))
    md = float(r.choice([15.0, 1.0, 1.75, 3.0, 100.0]))
    ma = float(r.choice([0.4189, 1.0e6, 0.01]))
    inc = float(r.choice([0.004, 0.0058, 0.1]))
    n = int(r.choice([1, 3, 40, 130]))
    scans = np.empty((n, size), np.float32)
    for k in range(n):
        mean = float(r.choice([1.1, 2.0, 5.0, 20.0, 100.0, 1000.0]))
        pos, hi = 0, bool(r.integers(0, 2))
        while pos < size:
            ln = int(r.geometric(1.0 / mean))
            val = r.choice([1.7500001, 2.0, 9.0, 14.0, 40.0]) if hi else r.choice([1.75, 1.7499999, 1.0, 0.0, 0.5])
            scans[k, pos:pos + ln] = val * (1.0 if r.random() < 0.5 else r.uniform(1.0, 1.0001))
            pos, hi = pos + ln, not hi
        kind = r.integers(0, 6)
        if kind == 0:
            scans[k] = r.uniform(0.0, 20.0, size)
        elif kind == 1:
            scans[k, r.integers(0, size)] = np.nan
        elif kind == 2:
            scans[k, r.integers(0, size, 3)] = -1.0
        if r.random() < 0.7:
            scans[k, int(r.choice([0, 1, 5, size - 1, size - 2, size - 6, int(r.integers(0, size))]))] = 0.25
    fg = PyFollowGap(10, md, ma, inc)
    got = fg.eval_many(scans)
    want = np.array([O.followgap_eval(scans[i], md, ma, inc) for i in range(n)], np.float32)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "FollowGap size %d md %g ma %g inc %g: scans %s" % (size, md, ma, inc, np.where(~same)[0][:5])


def one_case(seed):
    if True:
        if os.environ.get("FUZZ_TRACE"):
            print("case", seed, flush=True)
        r = np.random.default_rng(seed)
        try:
            followgap_case(np.random.default_rng(seed ^ 0x5F0))
        except AssertionError as e:
            raise AssertionError("MISMATCH seed=%d: %s" % (seed, e))
        rows, cols = int(r.integers(1, 400)), int(r.integers(1, 400))
        kind = r.integers(0, 4)
        if kind == 0:
            occ = (r.random((rows, cols)) < r.choice([0.0, 0.002, 0.02, 0.2, 0.9])).astype(np.uint8)
        elif kind == 1:
            occ = np.zeros((rows, cols), np.uint8); occ[0, :] = occ[-1, :] = 1; occ[:, 0] = occ[:, -1] = 1
        elif kind == 2:
            n = max(rows, cols, 8)
            occ = maps.make_maze(n, cell=int(r.integers(4, 40)), wall=int(r.integers(1, 4)), p=0.5, seed=seed).occ[:rows, :cols].copy()
        else:
            occ = np.ones((rows, cols), np.uint8); occ[rows // 4: 3 * rows // 4 + 1, cols // 4: 3 * cols // 4 + 1] = 0
        res = float(r.choice([0.05, 0.1, 1.0, 0.013]))
        origin = (float(r.uniform(-50, 50)), float(r.uniform(-50, 50)), float(r.choice([0.0, 0.0, r.uniform(-3.2, 3.2)])))
        mrx = float(r.choice([1, 7, 30, 120, 300, 512]))
        B = int(r.choice([1, 2, 63, 64, 65, 100, 360, 720, 1081, 2000]))
        fov = float(r.choice([4.71, 6.283, 0.5, -2.0, 0.0]))
        P = int(r.choice([1, 2, 7, 64, 65, 300, 600, 2500, 9000]))
        if P == 9000:
            B = int(r.choice([1, 3, 64, 100]))            # grid-wide binning path, kept cheap
        g = maps.GridMap(occ, res, origin, "fuzz")
        # poses: inside, near edges, outside
        gx = r.uniform(-3, cols + 3, P); gy = r.uniform(-3, rows + 3, P); th = r.uniform(-7, 7, P)
        if r.random() < 0.25:
            # coordinates a hair below an integer on an identity grid: float accumulation (x0 + 1 + 1 ...) then
            # rounds up across powers of two and walks gain a cell (the round-2 K2 window bug lived there)
            res, origin = 1.0, (0.0, 0.0, 0.0)
            snap = r.random(P) < 0.5
            eps = 2.0 ** -r.integers(14, 21, P)
            gx = np.where(snap, np.floor(r.uniform(1, cols + 1, P)) - eps, gx)
            gy = np.where(snap, np.floor(r.uniform(1, rows + 1, P)) - eps, gy)
            g = maps.GridMap(occ, res, origin, "fuzz")
        c, s = np.cos(origin[2]), np.sin(origin[2])
        poses = np.stack([origin[0] + (c * gx - s * gy) * res, origin[1] + (s * gx + c * gy) * res, th + origin[2]], 1).astype(np.float32)
        try:
            om = O.OracleMap.from_gridmap(g, mrx)
            omap = range_libc.PyOMap(g)
            assert np.array_equal(omap.distance_transform(), om.dt), "EDT"
            n = P * B
            for name, cls, args, ofun in (
                    ("RM", range_libc.PyRayMarching, (), lambda: om.rm_fan(poses, fov, B, 0.999)),
                    ("RMGPU", range_libc.PyRayMarchingGPU, (), lambda: om.rm_fan(poses, fov, B, 1.0)),
                    ("BL", range_libc.PyBresenhamsLine, (), lambda: om.bl_fan(poses, fov, B))):
                for variant in (1, 0):
                    m = cls(omap, mrx, *args); m.set_option("variant", variant)
                    # random schedule: every path the host policy can take, whatever the map size
                    sched = {}
                    if variant == 1 and r.random() < 0.7:
                        sched = {"inline_map_kb": int(r.choice([0, 2048])), "inline_max": int(r.choice([0, 512])),
                                 "stripe_max": int(r.choice([0, 2560, 8192])), "order_inline": int(r.integers(0, 2)),
                                 "tiled": int(r.integers(0, 2)), "low_water": int(r.choice([0, 5, 12, 24, 40])),
                                 "run_log2": int(r.choice([-1, 0, 2, 5])), "xcd_bands": int(r.choice([1, 3, 8])),
                                 "grid_mult": int(r.choice([1, 8])), "wg_threads": int(r.choice([256, 512, 1024])),
                                 "bin_multi_min": int(r.choice([64, 8192])), "bin_ppw": int(r.choice([256, 512, 2048])), "tile_stripe": int(r.choice([-1, 0, 1, 3, 100])), "pinned_max_rays": int(r.choice([0, 262144])),
                                 "slots": int(r.choice([1, 2, 3])), "spec_drain": int(r.choice([0, 8, 64])),
                                 "spec_stretch": int(r.choice([1, 4, 16])), "drain_cap": int(r.choice([1, 24, 64])),
                                 "drain_stretch": int(r.choice([1, 8])), "handoff": int(r.integers(0, 2)), "group_drain": int(r.choice([0, 2, 16])),
                                 "handoff_cap": int(r.choice([8, 16, 32, 64])), "handoff_wg": int(r.choice([64, 128, 256])),
                                 "code_map": int(r.choice([0, 2])), "code_min_rays": 0}
                        if os.environ.get("FUZZ_NO_GROUP"):
                            sched["group_drain"] = 0
                        if os.environ.get("FUZZ_NO_HANDOFF"):
                            sched["handoff"] = 0
                        for k_, v_ in sched.items():
                            m.set_option(k_, v_)
                        if os.environ.get("FUZZ_TRACE"):
                            print("  ", name, variant, sched, flush=True)
                    out = np.empty(n, np.float32); hits = np.empty((n, 2), np.int32); st = np.empty(n, np.uint16)
                    m.calc_range_fan(poses, out, fov, B, hit_cells=hits, steps=st)
                    r0, h0, s0 = ofun()
                    assert np.array_equal(out, r0), "%s v%d ranges %s" % (name, variant, sched)
                    if name.startswith("RM") and variant == 1:
                        # the ranges-only launch is its own instantiation (several rays per lane, drain compaction /
                        # hand-off to rm_leftover_kernel): diagnostics force one ray per lane
                        out2 = np.full(n, -3.0, np.float32)
                        m.calc_range_fan(poses, out2, fov, B)
                        assert np.array_equal(out2, r0), "%s v%d ranges-only %s" % (name, variant, sched)
                    assert np.array_equal(hits, h0), "%s v%d hits %s" % (name, variant, sched)
                    assert np.array_equal(st, s0), "%s v%d steps %s" % (name, variant, sched)
                    if sched:                      # ranges-only launch takes the non-diagnostic kernels
                        out2 = np.empty(n, np.float32); m.calc_range_fan(poses, out2, fov, B)
                        assert np.array_equal(out2, r0), "%s v%d ranges-only %s" % (name, variant, sched)
                    m.close()
            # the audit mode (variant 3): upstream-literal arithmetic against the oracle's libm forms — the fan form
            # with hit cells and sample counts, and the 2-argument per-ray form
            if n <= 400000 or r.random() < 0.2:
                for cls, sc in ((range_libc.PyRayMarching, 0.999), (range_libc.PyRayMarchingGPU, 1.0)):
                    m = cls(omap, mrx); m.set_option("variant", 3)
                    m.set_option("code_map", int(r.choice([0, 2]))); m.set_option("code_min_rays", 0); m.set_option("slots", int(r.choice([0, 2])))
                    o3 = np.empty(n, np.float32); m.calc_range_fan(poses, o3, fov, B)      # ranges only: the stream form (code map or float32)
                    assert np.array_equal(o3, om.rm_fan_libm(poses, fov, B, step_coeff=sc)[0]), "literal stream form %g %s" % (sc, m.last_plan()["name"])
                    out = np.empty(n, np.float32); hits = np.empty((n, 2), np.int32); st = np.empty(n, np.uint16)
                    m.calc_range_fan(poses, out, fov, B, hit_cells=hits, steps=st)
                    r0, h0, s0 = om.rm_fan_libm(poses, fov, B, step_coeff=sc)
                    assert np.array_equal(out, r0), "audit mode ranges %g" % sc
                    assert np.array_equal(hits, h0) and np.array_equal(st, s0), "audit mode hits / steps %g" % sc
                    ins = poses[: min(P, 3000)].copy()
                    o2 = np.empty(len(ins), np.float32); m.calc_range_many(ins, o2)
                    assert np.array_equal(o2, om.rm_rays_libm(ins, step_coeff=sc)), "audit mode rays %g" % sc
                    m.close()
            if rows * cols <= 20000:
                td = int(r.choice([2, 16, 112, 113, 360]))
                m = range_libc.PyCDDTCast(omap, mrx, td)
                m.set_option("cddt_bins", int(r.integers(0, 2))); m.set_option("cddt_lds_sort", int(r.choice([128, 16384]))); m.set_option("cddt_sort", int(r.integers(0, 2)))
                m.set_option("cddt_theta_min", int(r.choice([0, 1, 32768])))      # pose-major | theta-major | by size
                m.set_option("cddt_search", int(r.integers(0, 2)))
                out = np.empty(n, np.float32); m.calc_range_fan(poses, out, fov, B)
                assert np.array_equal(out, om.cddt_fan(td, poses, fov, B)), "CDDT td=%d" % td
                m.close()
            if rows * cols <= 6000:
                td = int(r.choice([2, 30, 180, 181, 514, 720, 1024, 1442]))   # 16-B row loads per lane: 1, 2, 3
                m = range_libc.PyGiantLUTCast(omap, mrx, td)
                lut = om.lut_build(td, nthreads=8)
                assert np.array_equal(m.table(), lut), "LUT table td=%d" % td
                out = np.empty(n, np.float32); m.calc_range_fan(poses, out, fov, B)
                assert np.array_equal(out, om.lut_fan(lut, poses, fov, B)), "LUT fan td=%d" % td
                m.close()
            # fused / generic crash tests, whole batch and grouped
            edge = r.uniform(0.05, 0.6, B)
            thr = 0.001
            rr = om.rm_fan(poses, fov, B, 1.0)[0]
            m = range_libc.PyRayMarchingGPU(omap, mrx); m.set_option("slots", int(r.choice([1, 2]))); m.set_option("code_map", int(r.choice([0, 2]))); m.set_option("code_min_rays", 0)
            m.set_option("handoff", 0 if os.environ.get("FUZZ_NO_HANDOFF") else int(r.integers(0, 2))); m.set_option("handoff_cap", int(r.choice([8, 64])))
            if os.environ.get("FUZZ_NO_GROUP"):
                m.set_option("group_drain", 0)
            assert m.check_collision_many(poses, fov, B, edge, thr) == O.is_crashed(rr, B, P, edge, thr), "crash many"
            grp = next(k for k in (7, 5, 4, 3, 2, 1) if P % k == 0)
            want = [O.is_crashed(rr[k * grp * B:(k + 1) * grp * B], B, grp, edge, thr) for k in range(P // grp)]
            assert m.check_collision_groups(poses, grp, fov, B, edge, thr).tolist() == want, "crash groups"
            m.close()
            # map update: stamp a block, tables must follow
            occ2 = occ.copy(); occ2[rows // 3: rows // 3 + 3, cols // 3: cols // 3 + 3] ^= 1
            omap.update(occ2)
            om2 = O.OracleMap(occ2, res, origin, mrx)
            m = range_libc.PyRayMarching(omap, mrx)          # (default arithmetic of "RM": upstream-literal)
            out = np.empty(n, np.float32); m.calc_range_fan(poses, out, fov, B)
            assert np.array_equal(out, om2.rm_fan_libm(poses, fov, B, step_coeff=0.999)[0]), "after map update"
            m.close(); omap.update(occ)
            if r.random() < 0.25:
                # one handle over several devices (the box has one GPU: device 0 two or three times): the batch cut into
                # pose blocks, ranges with noise-free parity against the oracle, crash indices global
                multi = range_libc.PyOMap(g, device=[0] * int(r.integers(2, 4)))
                mm = range_libc.PyRayMarchingGPU(multi, mrx); mm.set_option("multi_min_poses", int(r.choice([1, 16, 64])))
                out = np.empty(n, np.float32); mm.calc_range_fan(poses, out, fov, B)
                assert np.array_equal(out, rr), "multi-device fan"
                assert mm.check_collision_many(poses, fov, B, edge, thr) == O.is_crashed(rr, B, P, edge, thr), "multi-device crash many"
                assert mm.check_collision_groups(poses, grp, fov, B, edge, thr).tolist() == want, "multi-device crash groups"
                mm.close(); multi.close()
            ins = poses[r.integers(0, P, 500)].copy(); ins[:, 2] = r.uniform(-9, 9, 500).astype(np.float32)
            outs = np.empty(500, np.float32)
            m = range_libc.PyRayMarching(omap, mrx); m.calc_range_many(ins, outs)
            assert np.array_equal(outs, om.rm_rays_libm(ins, step_coeff=0.999)), "RM rays (literal)"
            m.set_option("variant", 1); m.calc_range_many(ins, outs)
            assert np.array_equal(outs, om.rm_rays(ins)[0]), "RM rays"
            m.close(); omap.close()
        except AssertionError as e:
            raise AssertionError("MISMATCH seed=%d: %s | map %dx%d kind %d res %g origin %s mrx %g B %d fov %g P %d"
                                 % (seed, e, rows, cols, kind, res, origin, mrx, B, fov, P))
    return 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print("fuzz ok: %d random cases, all methods bit-identical to the oracle" % run(a.seconds, a.seed))
