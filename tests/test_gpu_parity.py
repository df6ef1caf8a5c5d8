"""GPU parity tests (run with -m gpu on the MI355X box).  Every test calls the HIP kernels
through the C ABI (libscan_amd.so) and compares with the CPU oracle on the same seeded inputs,
with the committed golden vectors, or — at BASELINE.json's full sizes — through
size-independent properties.  Bar: bit-exact ranges / hit cells / step counts for ray marching."""
import math
import os

import numpy as np
import pytest

from conftest import GOLD, load_golden
from pyracecarsimulator_amd import ScanSimulator2D, _lib, maps, range_libc, workloads

pytestmark = pytest.mark.gpu

SCAN_FOV_720 = 4.71 * 720.0 / 1080.0


@pytest.fixture(scope="module", autouse=True)
def _gpu(need_gpu):
    yield


def _fan(method, poses, fov, B):
    n = len(poses) * B
    out = np.empty(n, np.float32)
    hits = np.empty((n, 2), np.int32)
    steps = np.empty(n, np.uint16)
    method.calc_range_fan(poses, out, fov, B, hit_cells=hits, steps=steps)
    return out, hits, steps


# ---------------------------------------------------------------- K0: EDT
@pytest.mark.parametrize("case", ["colombia", "maze", "random", "single", "empty", "wide"])
def test_device_edt_bit_equal_to_oracle(oracle_mod, case):
    rng = np.random.default_rng(1)
    occ = {"colombia": lambda: maps.load_colombia().occ,
           "maze": lambda: maps.make_maze(300, cell=30, wall=2, seed=2).occ,
           "random": lambda: (rng.random((211, 173)) < 0.01).astype(np.uint8),
           "single": lambda: np.pad(np.ones((1, 1), np.uint8), ((40, 9), (3, 77))),
           "empty": lambda: np.zeros((33, 65), np.uint8),
           "wide": lambda: (rng.random((5, 3000)) < 0.002).astype(np.uint8)}[case]()
    omap = range_libc.PyOMap(occ, 0.05)
    assert np.array_equal(omap.distance_transform(), oracle_mod.edt(occ))


# ---------------------------------------------------------------- schedules of K1 are result-neutral
@pytest.mark.parametrize("opts", [
    {"variant": 0},                                             # chunk kernel (K1)
    {"variant": 1},                                             # stream kernel (K1b), defaults
    {"variant": 1, "low_water": 0, "wg_threads": 256},          # no refill until all lanes done
    {"variant": 1, "low_water": 63, "wg_threads": 512},         # refill after every step
    {"variant": 1, "sort_poses": 0, "xcd_bands": 1},            # unsorted, one band
    {"variant": 1, "xcd_bands": 3, "grid_mult": 1},             # odd band count, small grid
    {"variant": 1, "grid_mult": 16, "wg_threads": 256},
    {"variant": 1, "inline_prep": 0},                           # single-workgroup binning launch
    {"variant": 1, "inline_prep": 0, "bin_generic": 1},         # ... its generic (any size) form
    {"variant": 1, "inline_prep": 0, "bin_multi_min": 64},      # grid-wide binning kernels
    {"variant": 1, "inline_prep": 1, "inline_max": 100000, "xcd_bands": 5},   # workgroups derive their own pose records
    {"variant": 1, "slice_log2": 14},                           # batch cut into pose slices (>= 2^30 rays in production)
    {"variant": 1, "bin_multi_min": 64},                        # grid-wide binning kernels
    {"variant": 1, "bin_multi_min": 1 << 30},                   # single-workgroup binning
    {"variant": 1, "inline_map_kb": 0},                         # big-map policy: binning launch from 512 poses up
    {"variant": 1, "inline_map_kb": 0, "run_log2": 3},          # ... with runs of 8 blocks per workgroup turn
    {"variant": 1, "inline_prep": 0, "run_log2": 5, "grid_mult": 1},
    {"variant": 1, "inline_map_kb": 0, "inline_max": 0},        # stripe bands compacted inside the march kernel
    {"variant": 1, "inline_map_kb": 0, "inline_max": 0, "xcd_bands": 5, "grid_mult": 2},
    {"variant": 1, "inline_map_kb": 0, "stripe_max": 0},        # keys-only binning launch + records derived in the march
    {"variant": 1, "inline_map_kb": 0, "stripe_max": 0, "xcd_bands": 3, "grid_mult": 1},
    {"variant": 1, "inline_map_kb": 0, "stripe_max": 0, "order_inline": 0},   # classic: binning launch writes the records
    {"variant": 1, "slots": 2},                                 # two rays per lane (ranges-only launches)
    {"variant": 1, "slots": 3},                                 # three rays per lane
    {"variant": 1, "slots": 2, "inline_map_kb": 0, "stripe_max": 0},          # ... behind the keys-only binning launch
    {"variant": 1, "slots": 2, "inline_prep": 0, "low_water": 0},             # ... records from the binning launch
    {"variant": 1, "slots": 2, "inline_prep": 0, "bin_multi_min": 64, "xcd_bands": 3, "grid_mult": 2},
    {"variant": 1, "tiled": 0},                                 # row-major padded EDT (default: 4x8-cell tiles)
    {"variant": 1, "tiled": 0, "inline_prep": 0, "xcd_bands": 1},
    {"variant": 1, "nt_store": 0},                              # plain range stores (default: non-temporal)
    {"variant": 1, "nt_store": 0, "slots": 2},
    {"variant": 1, "slots": 2, "group_drain": 16},              # the last 32 / 16 rays of a wave on 2 / 4 lanes per ray (default: off)
    {"variant": 1, "slots": 3, "group_drain": 4, "drain_cap": 24},
    {"variant": 1, "slots": 2, "group_drain": 16, "drain_cap": 8, "drain_stretch": 1},   # straight into 4 lanes per ray
    {"variant": 1, "slots": 2, "handoff": 1},                   # dry waves hand their last rays to rm_leftover_kernel
    {"variant": 1, "slots": 2, "handoff": 1, "handoff_cap": 8, "handoff_wg": 64, "inline_map_kb": 0, "stripe_max": 0},
    {"variant": 1, "slots": 2, "handoff": 1, "handoff_cap": 64, "inline_prep": 0, "grid_mult": 2},
    {"variant": 1, "slots": 3, "handoff": 1, "handoff_cap": 32, "handoff_wg": 128},
    {"variant": 1, "slots": 2, "code_map": 2, "code_min_rays": 0},                  # the step map as u16 palette codes, the palette in LDS (round 6)
    {"variant": 1, "slots": 2, "code_map": 2, "code_min_rays": 0, "inline_map_kb": 0, "stripe_max": 0},
    {"variant": 1, "slots": 2, "code_map": 2, "code_min_rays": 0, "inline_prep": 0, "low_water": 0},
    {"variant": 1, "slots": 2, "code_map": 2, "code_min_rays": 0, "inline_prep": 0, "bin_multi_min": 64, "xcd_bands": 3, "grid_mult": 2},
    {"variant": 1, "slots": 2, "code_map": 2, "code_min_rays": 0, "group_drain": 16, "drain_cap": 8, "drain_stretch": 1},
    {"variant": 1, "slots": 2, "code_map": 2, "code_min_rays": 0, "drain_cap": 3, "spec_stretch": 1},
    {"variant": 1, "slots": 2, "grid_mult": 1, "tail_pct": 25},                  # two generations of workgroups (round 6 A/B: off by default)
    {"variant": 1, "slots": 2, "grid_mult": 1, "tail_pct": 40, "tail_wg_pct": 200, "inline_prep": 0, "code_min_rays": 0},
    {"variant": 1, "slots": 1, "grid_mult": 1, "tail_pct": 15, "inline_map_kb": 0, "stripe_max": 0},
    {"variant": 1, "slots": 2, "tile_stripe": 0, "inline_map_kb": 0, "stripe_max": 0},          # binning tiles row-major (default: stripes of tile rows, column by column)
    {"variant": 1, "slots": 2, "tile_stripe": 3, "inline_map_kb": 0, "stripe_max": 0, "xcd_bands": 3},
    {"variant": 1, "slots": 2, "tile_stripe": 1, "inline_prep": 0},
    {"variant": 1, "slots": 2, "tile_stripe": 2, "inline_prep": 0, "bin_generic": 1},
    {"variant": 1, "slots": 2, "tile_stripe": 5, "inline_prep": 0, "bin_multi_min": 64, "code_map": 2, "code_min_rays": 0},
    {"variant": 1, "tile_stripe": 4096, "inline_prep": 0, "bin_multi_min": 64},                 # (more rows than the map has tiles: row-major)
])
def test_every_kernel_schedule_is_bit_identical(oracle_mod, opts):
    g = maps.make_maze(400, cell=40, wall=3, p=0.45, seed=21, origin=(-7.0, 3.0, -0.4))
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    poses = maps.sample_free_poses(g, 333, 4, dt=om.dt)
    poses[17] = [np.nan, 0, 0]
    poses[200] = [1e6, 1e6, 1.0]
    poses[201] = [g.origin[0] - 0.3 * g.resolution, g.origin[1] + 1.0, 0.2]   # -1 < gx < 0
    for cls, sc in ((range_libc.PyRayMarchingGPU, 1.0), (range_libc.PyRayMarching, 0.999)):
        m = cls(omap, 300)
        for k, v in opts.items():
            m.set_option(k, v)
            assert m.get_info(k) == v
        r, h, s = _fan(m, poses, 4.71, 1081)
        r0, h0, s0 = om.rm_fan(poses, 4.71, 1081, step_coeff=sc, nthreads=4)
        assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s, s0)
        # ranges-only launch (no diagnostics) takes the non-AUX template
        r1 = np.empty_like(r)
        m.calc_range_fan(poses, r1, 4.71, 1081)
        assert np.array_equal(r1, r0)
        # small batches take the unsorted single-band path
        r, h, s = _fan(m, poses[:5], 4.71, 70)
        r0, h0, s0 = om.rm_fan(poses[:5], 4.71, 70, step_coeff=sc)
        assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s, s0)


# ---------------------------------------------------------------- K1 vs golden vectors
@pytest.mark.parametrize("name", ["rm_colombia", "rm_maze256", "rm_maze192_yaw"])
def test_rm_fan_reproduces_golden_vectors(name):
    g, z = load_golden(name)
    omap = range_libc.PyOMap(g)
    fov, B, mrx = float(z["fov"]), int(z["num_rays"]), int(z["max_range_px"])
    L = np.load(os.path.join(GOLD, "rm_libm_forms.npz"))
    npz = int(L[name + "_n_poses"])
    for tag, cls in (("cpu", range_libc.PyRayMarching), ("gpu", range_libc.PyRayMarchingGPU)):
        m = cls(omap, mrx)
        if tag == "cpu":
            # "RM" names range_libc's CPU RayMarching (scripts/scan_simulator.py:72-73): by default it computes the
            # upstream-literal statement — the committed libm-form vectors, bit for bit: ranges, hit cells, sample counts
            assert m.get_info("variant") == 3
            r, h, s = _fan(m, z["poses"][:npz], fov, B)
            assert np.array_equal(r, L["%s_ranges_cpu" % name]) and np.array_equal(h, L["%s_hits_cpu" % name].astype(np.int32))
            assert np.array_equal(s, L["%s_steps_cpu" % name])
            rr = np.empty(npz * B, np.float32)
            m.calc_range_fan(z["poses"][:npz], rr, fov, B)             # ranges only: the stream form
            assert m.last_plan()["kernel"] == "rm_stream_literal" and np.array_equal(rr, L["%s_ranges_cpu" % name])
            m.set_option("variant", 1)                                 # the canonical arithmetic stays selectable
        else:
            assert m.get_info("variant") == 1
        r, h, s = _fan(m, z["poses"], fov, B)
        assert np.array_equal(r, z["ranges_" + tag]), tag
        assert np.array_equal(h, z["hits_" + tag].astype(np.int32)), tag
        assert np.array_equal(s, z["steps_" + tag]), tag


def _libm_gate(a, ha, b, hb, res, what):
    """The gate north_star's wording allows the canonical form against upstream's literal arithmetic: hit cells
    identical on all but <= 1e-4 of the rays (grazing rays that catch or pass a corner cell), every range within
    1.5 cells, every ray whose hit cell agrees within 1e-3 cell."""
    ha, hb = np.asarray(ha, np.int32).reshape(-1, 2), np.asarray(hb, np.int32).reshape(-1, 2)
    mism = (ha != hb).any(axis=1)
    err = np.abs(a - b) / res
    assert mism.mean() <= 1e-4 or mism.sum() <= 2, (what, int(mism.sum()), len(a))
    assert err.max() <= 1.5, (what, float(err.max()))
    assert err[~mism].max() <= 1e-3, (what, float(err[~mism].max()))
    return int(mism.sum()), float(err.max())


def test_device_fan_vs_upstream_literal_libm_form(oracle_mod):
    """Parity gap, as narrow as the mount allows (range_libc itself is absent: PARITY UNPINNED).  The device
    kernels reproduce the oracle's CANONICAL float32 form bit for bit; upstream computes every beam's direction
    with libm cosf/sinf of theta + alpha_j.  This test gates the distance between the two on (1) the committed
    upstream-literal vectors of GOLD-A/B (tests/golden/rm_libm_forms.npz, made by make_fixtures.py::libm_forms)
    and (2) a 64-pose subsample of cfg2 cast live by the oracle's libm form: hit-cell mismatches <= 1e-4 of the
    rays, every range within 1.5 cells (one cell on all but the rays whose hit cell moved)."""
    L = np.load(os.path.join(GOLD, "rm_libm_forms.npz"))
    seen = []
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        omap = range_libc.PyOMap(g)
        fov, B, mrx = float(z["fov"]), int(z["num_rays"]), int(z["max_range_px"])
        npz = int(L[name + "_n_poses"])
        for tag, cls in (("cpu", range_libc.PyRayMarching), ("gpu", range_libc.PyRayMarchingGPU)):
            mc = cls(omap, mrx)
            mc.set_option("variant", 1)                 # the CANONICAL form of both classes against the literal vectors
            r, h, s_ = _fan(mc, z["poses"][:npz], fov, B)
            seen.append((name, tag) + _libm_gate(r, h, L["%s_ranges_%s" % (name, tag)], L["%s_hits_%s" % (name, tag)],
                                                 g.resolution, name + "/" + tag))
            assert (s_ != L["%s_steps_%s" % (name, tag)]).mean() < 2e-3
    w = workloads.cfg2()
    g, B, mrx = w.gmap, w.num_rays, w.max_range_px
    omap = range_libc.PyOMap(g)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    poses = workloads.make_poses(w, dt=om.dt)[np.linspace(0, w.n_poses - 1, 64).astype(np.int64)]
    for tag, cls, sc in (("cpu", range_libc.PyRayMarching, 0.999), ("gpu", range_libc.PyRayMarchingGPU, 1.0)):
        r, h, _ = _fan(cls(omap, mrx), poses, w.fov, B)
        rb, hb, _ = om.rm_fan_libm(poses, w.fov, B, step_coeff=sc)
        seen.append(("cfg2", tag) + _libm_gate(r, h, rb, hb, g.resolution, "cfg2/" + tag))
    print("canonical (device) vs upstream-literal libm form — (map, coeff, rays with another hit cell, max |d| in cells):", seen)


def _ray_rows(poses, fov, B):
    """(x, y, theta) rows of the 2-argument form: heading + the float32 np.arange angle row of scripts/two_player/scan.py:57-62"""
    ang = (np.float32(-0.5) * np.float32(fov) + np.arange(B, dtype=np.float32) * (np.float32(fov) / np.float32(B))).astype(np.float32)
    ins = np.zeros((len(poses) * B, 3), np.float32)
    for q in range(len(poses)):
        ins[q * B:(q + 1) * B, :2] = poses[q, :2]
        ins[q * B:(q + 1) * B, 2] = poses[q, 2] + ang
    return ins


def test_device_tables_vs_upstream_literal_libm_forms():
    """The TABLE methods against upstream's literal arithmetic (range_libc absent: PARITY UNPINNED; the closest
    available statement is oracle/rangelib_oracle.c's *_libm table forms — libm cosf / sinf per bin of a
    double-precision bin angle, un-fused projection and world->grid, fmod + roundf bin rule).  The device builds the
    CANONICAL tables bit for bit (test_cddt_* / test_giant_lut_*); this gate pins their distance to the literal form on
    the committed vectors (tests/golden/table_libm_forms.npz, make_fixtures.py::table_libm_forms), queried the way
    the reference's only CDDT user queries (2-argument per-ray form, scripts/two_player/scan.py:57-70):
    CDDT theta_disc 112 (scripts/two_player/rcs_two_player.py:121) and 360: every ray within 1e-3 cell, no ray in
    another bucket; GiantLUT theta_disc 180: every range within one table code of the literal fan query."""
    L = np.load(os.path.join(GOLD, "table_libm_forms.npz"))
    seen = []
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        omap = range_libc.PyOMap(g)
        fov, B, mrx = float(z["fov"]), int(z["num_rays"]), int(z["max_range_px"])
        n = int(L[name + "_n_poses"])
        poses = np.ascontiguousarray(z["poses"][:n])
        ins = _ray_rows(poses, fov, B)
        for td in (112, 360):
            m = range_libc.PyCDDTCast(omap, mrx, td)
            got = np.empty(len(ins), np.float32)
            m.calc_range_many(ins, got)
            e = np.abs(got - L["%s_cddt%d" % (name, td)]) / g.resolution
            seen.append((name, "cddt%d" % td, float((e == 0).mean()), float(e.max())))
            assert e.max() <= 1e-3, (name, td, float(e.max()), int((e > 1e-3).sum()))
            m.close()
        m = range_libc.PyGiantLUTCast(omap, mrx, 180)
        fan = np.empty(n * B, np.float32)
        m.calc_range_fan(poses, fan, fov, B)
        code = mrx / 65535.0
        e = np.abs(fan - L[name + "_lut180_fan"]) / g.resolution
        seen.append((name, "lut180", float((e == 0).mean()), float(e.max())))
        assert (e > 1.01 * code).mean() <= 1e-3 and np.median(e) == 0.0, (name, float(e.max()))
        m.close()
    print("device tables vs upstream-literal libm table forms — (map, method, share of rays bit-equal, max |d| in cells):", seen)


def test_device_bresenham_vs_upstream_literal_libm_form():
    """BresenhamsLine against upstream's literal arithmetic (range_libc absent: PARITY UNPINNED; the closest available
    statement is oracle/rangelib_oracle.c's orc_bl_*_libm — libm cosf / sinf of theta' = -theta + rotation constant,
    un-fused end point and hit distance): the device walks the CANONICAL statement bit for bit (test_bresenham_*); this
    gate pins its distance to the literal form on the committed vectors (tests/golden/bl_libm_forms.npz,
    make_fixtures.py::bl_libm_forms), both kernels (stream and LDS window): in the fan form the same hit cell and step
    count on every ray and every range within 1e-3 cell; in the per-ray form at most 1e-4 of the rays (one ray of a
    small set) off by more than 1e-3 cell, none by more than 1.5 cells."""
    L = np.load(os.path.join(GOLD, "bl_libm_forms.npz"))
    seen = []
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        omap = range_libc.PyOMap(g)
        fov, B, mrx = float(z["fov"]), int(z["num_rays"]), int(z["max_range_px"])
        n = int(L[name + "_n_poses"])
        poses = np.ascontiguousarray(z["poses"][:n])
        for variant in (1, 0):
            m = range_libc.PyBresenhamsLine(omap, mrx)
            m.set_option("variant", variant)
            r, h, s = _fan(m, poses, fov, B)
            e = np.abs(r - L[name + "_fan_ranges"]) / g.resolution
            assert np.array_equal(h, L[name + "_fan_hits"].astype(np.int32)), (name, variant, "hit cells")
            assert np.array_equal(s, L[name + "_fan_steps"]), (name, variant, "step counts")
            assert e.max() <= 1e-3, (name, variant, float(e.max()))
            seen.append((name, "fan v%d" % variant, float((e == 0).mean()), float(e.max())))
            m.close()
        m = range_libc.PyBresenhamsLine(omap, mrx)
        got = np.empty(n * B, np.float32)
        m.calc_range_many(_ray_rows(poses, fov, B), got)
        # (per-ray form: the canonical direction is det_sincosf(theta), the literal one libm cosf / sinf of
        #  -theta + rotation constant — a walk that grazes a corner can take the neighbouring cell: 0-2 of 69 184 rays on
        #  the CPU statements; the same gate as the ray-marching methods' _libm_gate)
        e = np.abs(got - L[name + "_rays_ranges"]) / g.resolution
        off = e > 1e-3
        assert int(off.sum()) <= max(1, int(1e-4 * e.size)) and e.max() <= 1.5, (name, "2-argument form", int(off.sum()), float(e.max()))
        seen.append((name, "rays", float((e == 0).mean()), float(e.max())))
        m.close()
    print("device Bresenham vs the upstream-literal libm form — (map, form, share of rays bit-equal, max |d| in cells):", seen)


def _exact_rm_gate(err_cells, theta_disc, mean_range_cells, origin_cells, what):
    """K2 / K3 ranges "<= 1 cell vs oracle RM".  One cell is what the METHODS' conventions allow
    on most rays, not on all of them, so the gate is the distribution (recorded on bench.py's line as vs_exact_rm):
      * floor, whatever theta_disc: exact ray marching reports the distance to the hit cell's INTEGER corner, the table
        methods the distance to the wall's edge-cell geometry (CDDT) / a cast from the pose cell's corner (GiantLUT,
        + origin_cells = sqrt(2)/2 on average) — both within a cell of the wall, median difference ~0.6 cell, p90 ~1.5;
      * angular term: the ray's angle is rounded to the nearest of theta_disc bins, half a bin at most:
        E = (pi / theta_disc) x mean range; at unit incidence slope it shifts the hit by E;
      * tails: a ray that grazes a corner in one statement and passes it in the other differs by up to the range
        window — p99 and max are recorded and bounded by the window only.
    Gates: median <= 0.75 + 0.1 E, p90 <= 1.75 + 0.6 E (+ origin term), >= 65 % of the rays within one cell."""
    E = math.pi / theta_disc * mean_range_cells
    med, p90, p99, mx = (float(np.median(err_cells)), float(np.percentile(err_cells, 90)),
                         float(np.percentile(err_cells, 99)), float(err_cells.max()))
    within = float((err_cells <= 1.0).mean())
    print("%s vs exact ray marching (cells): median %.3f p90 %.3f p99 %.2f max %.1f, within one cell %.4f; E = %.3f"
          % (what, med, p90, p99, mx, within, E))
    assert med <= 0.75 + 0.1 * E + 0.25 * origin_cells, (what, med)
    assert p90 <= 1.75 + 0.6 * E + origin_cells, (what, p90)
    assert within >= 0.65, (what, within)
    return med, p90, p99, mx, within


def test_audit_mode_trig_on_the_device_equals_this_hosts_libm(oracle_mod):
    """The audit mode's sinf / cosf on the device (csrc/literal_kernels.h: glibc's algorithm in double precision) against
    THIS host's libm, bit for bit: angles a scan can produce, every binade up to the largest float, denormals, zeros,
    infinities and NaN — and against the oracle's own statement of that algorithm (which tests/test_oracle.py walks
    over the float range against libm on the CPU box)."""
    import ctypes
    rng = np.random.default_rng(11)
    x = np.concatenate([
        rng.uniform(-13.0, 13.0, 400000).astype(np.float32),
        (rng.uniform(-1.0, 1.0, 200000) * np.exp2(rng.uniform(-140.0, 127.9, 200000))).astype(np.float32),
        rng.integers(0, 1 << 32, 400000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 120.0, -120.0, 0.78539816, 2.4414062e-4, 1e-45, 3.4028235e38],
                 np.float32)])
    s, c = np.empty_like(x), np.empty_like(x)
    f32p = ctypes.POINTER(ctypes.c_float)
    _lib.check(_lib.lib().rl_probe_literal_sincosf(0, x.ctypes.data_as(f32p), x.size, s.ctypes.data_as(f32p),
                                                   c.ctypes.data_as(f32p)))
    for name, (ws, wc) in (("libm", oracle_mod.libm_sincosf(x)), ("statement", oracle_mod.lit_sincosf(x))):
        for got, want, fn in ((s, ws, "sinf"), (c, wc, "cosf")):
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(got), nan), (name, fn)
            bad = (got.view(np.uint32) != want.view(np.uint32)) & ~nan
            assert not bad.any(), (name, fn, int(bad.sum()), x[bad][:5], got[bad][:5], want[bad][:5])


def test_audit_mode_reproduces_the_upstream_literal_form_bit_for_bit(oracle_mod):
    """variant 3 — range_libc's CPU arithmetic stated literally (libm trig per ray, un-fused products and sums,
    calc_range(y, x, theta')) — against the oracle's libm forms: ranges, hit cells and sample counts of the fan form
    on the three golden maps (both step coefficients) and a cfg2 subsample, the committed upstream-literal vectors
    (tests/golden/rm_libm_forms.npz), the 2-argument per-ray form, and the poses that cannot be cast.  Where the
    canonical default differs from upstream's literal arithmetic on <= 1e-4 of the rays, this mode does not differ."""
    L = np.load(os.path.join(GOLD, "rm_libm_forms.npz"))
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        omap = range_libc.PyOMap(g)
        om = oracle_mod.OracleMap.from_gridmap(g, int(z["max_range_px"]))
        fov, B, mrx = float(z["fov"]), int(z["num_rays"]), int(z["max_range_px"])
        poses = np.array(z["poses"], np.float32)
        npz = int(L[name + "_n_poses"])
        for tag, cls, sc in (("cpu", range_libc.PyRayMarching, 0.999), ("gpu", range_libc.PyRayMarchingGPU, 1.0)):
            m = cls(omap, mrx)
            m.set_option("variant", 3)
            r, h, s_ = _fan(m, poses, fov, B)
            assert m.last_plan()["kernel"] == "rm_literal"
            rb, hb, sb = om.rm_fan_libm(poses, fov, B, step_coeff=sc)
            assert np.array_equal(r, rb) and np.array_equal(h, hb) and np.array_equal(s_, sb), (name, tag)
            n = npz * B
            assert np.array_equal(r[:n], L["%s_ranges_%s" % (name, tag)]), (name, tag, "committed vectors")
            assert np.array_equal(h[:n], L["%s_hits_%s" % (name, tag)].astype(np.int32).reshape(-1, 2))
            assert np.array_equal(s_[:n], L["%s_steps_%s" % (name, tag)])
            # ranges-only launch, and the upstream 2-argument form: one (x, y, theta) row per ray
            r1 = np.empty_like(r)
            m.calc_range_fan(poses, r1, fov, B)
            assert np.array_equal(r1, rb)
            # (ranges only: the production form — the stream kernel's schedule with the literal arithmetic)
            assert m.last_plan()["kernel"] == "rm_stream_literal", m.last_plan()
            ins = np.zeros((4000, 3), np.float32)
            rng = np.random.default_rng(5)
            pick = rng.integers(0, len(poses), len(ins))
            ins[:, :2] = poses[pick, :2]
            ins[:, 2] = rng.uniform(-12.0, 12.0, len(ins)).astype(np.float32)
            ins[7] = [np.nan, 0.0, 0.0]
            ins[8] = [1e6, -1e6, 1.0]
            ins[9, 2] = 3e7                                  # (beyond the fast range reduction of libm's sinf)
            ins[10, 2] = np.inf
            o2 = np.empty(len(ins), np.float32)
            m.calc_range_many(ins, o2)
            assert np.array_equal(o2, om.rm_rays_libm(ins, step_coeff=sc)), (name, tag, "2-argument form")
    w = workloads.cfg2()
    g, B, mrx = w.gmap, w.num_rays, w.max_range_px
    omap = range_libc.PyOMap(g)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    poses = workloads.make_poses(w, dt=om.dt)[np.linspace(0, w.n_poses - 1, 64).astype(np.int64)]
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    m.set_option("variant", 3)
    r, h, s_ = _fan(m, poses, w.fov, B)
    rb, hb, sb = om.rm_fan_libm(poses, w.fov, B, step_coeff=1.0)
    assert np.array_equal(r, rb) and np.array_equal(h, hb) and np.array_equal(s_, sb)
    # the fused crash test in the literal arithmetic: Car::isCrashed (oracle._ref-pinned restatement) over the LITERAL ranges
    edge = oracle_mod.edge_distances(B, -w.fov / 2, w.fov / B, 0.275, 0.2032, 0.3302)
    assert m.check_collision_many(poses, w.fov, B, edge, 0.001) == oracle_mod.is_crashed(rb, B, len(poses), edge, 0.001)
    assert m.last_plan()["kernel"] == "rm_stream_literal" and m.last_plan()["crash"] == 1


def test_code_map_is_engaged_and_bit_identical(oracle_mod):
    """Round 6: the step map as 16-bit palette codes + the palette of exact float32 steps in LDS (rm_fan_stream_kernel<...,
    CODE = 2>).  Same sample sequence as the float32 step map by construction — so the ranges, the fused crash index and the
    upstream-literal form are bit-identical to the oracle —, and the launches really take it (plan: code 2, the palette size
    the handle reports), on a maze (hundreds of distinct steps), colombia and an empty map (a palette of stop codes and steps
    past max_range only); a map whose palette does not fit falls back to the float32 map."""
    from pyracecarsimulator_amd import racecar as RC
    B, fov = 1081, 4.71
    edge = RC.edge_distances(B, -fov / 2.0, fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    cases = [("maze400", maps.make_maze(400, cell=40, wall=3, p=0.45, seed=21, origin=(-7.0, 3.0, -0.4)), 300),
             ("colombia", maps.load_colombia(), 300),
             ("maze700_r120", maps.make_maze(700, cell=64, wall=2, p=0.5, seed=4), 120)]
    for name, g, mrx in cases:
        om = oracle_mod.OracleMap.from_gridmap(g, mrx)
        omap = range_libc.PyOMap(g)
        poses = maps.sample_free_poses(g, 300, 9, dt=om.dt)
        poses[11] = [np.nan, 0, 0]
        poses[12] = [1e6, 1e6, 1.0]
        for cls, sc in ((range_libc.PyRayMarchingGPU, 1.0), (range_libc.PyRayMarching, 0.999)):
            m = cls(omap, mrx)
            m.set_option("variant", 1)
            m.set_option("slots", 2)
            assert m.get_info("code_map") == 2 and m.get_info("code_min_rays") == 1 << 22         # (the defaults)
            m.set_option("code_min_rays", 0)
            want = om.rm_fan(poses, fov, B, step_coeff=sc, nthreads=4, want_hits=False, want_steps=False)[0]
            for opts in ({}, {"inline_prep": 0}, {"inline_map_kb": 0, "stripe_max": 0, "inline_prep": 1}):
                for k, v in opts.items():
                    m.set_option(k, v)
                got = np.full(len(poses) * B, -1.0, np.float32)
                m.calc_range_fan(poses, got, fov, B)
                pl = m.last_plan()
                assert pl["code"] == 2 and pl["code_entries"] == m.get_info("code_entries") >= 2, (name, opts, pl)
                assert pl["name"].endswith(", 2, false, 2>"), pl["name"]
                assert np.array_equal(got, want), (name, cls.__name__, opts, int((got != want).sum()))
                # fused crash test on the code map
                grp = 100
                first = m.check_collision_groups(poses, grp, fov, B, edge, 0.001)
                assert m.last_plan()["code"] == 2 and m.last_plan()["crash"] == 1
                ref = [oracle_mod.is_crashed(want[q * grp * B:(q + 1) * grp * B], B, grp, edge, 0.001) for q in range(len(poses) // grp)]
                assert first.tolist() == ref, (name, cls.__name__, opts)
            # upstream-literal arithmetic on the code map
            m.set_option("variant", 3)
            got = np.full(len(poses) * B, -1.0, np.float32)
            m.calc_range_fan(poses, got, fov, B)
            assert m.last_plan()["kernel"] == "rm_stream_literal" and m.last_plan()["code"] == 2, m.last_plan()
            assert np.array_equal(got, om.rm_fan_libm(poses, fov, B, step_coeff=sc)[0]), (name, cls.__name__, "literal")
            # diagnostics (hit cells / sample counts) stay on the float32 map
            m.set_option("variant", 1)
            r, h, s_ = _fan(m, poses[:40], fov, B)
            r0, h0, s0 = om.rm_fan(poses[:40], fov, B, step_coeff=sc)
            assert m.last_plan()["code"] == 0 and np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s_, s0)
            m.close()
        # a map update rebuilds the palette and the code map
        occ2 = np.array(g.occ, copy=True)
        occ2[g.rows // 3:g.rows // 3 + 9, g.cols // 4:g.cols // 4 + 30] = 1
        m = range_libc.PyRayMarchingGPU(omap, mrx)
        m.set_option("slots", 2)
        m.set_option("code_min_rays", 0)
        got = np.empty(len(poses) * B, np.float32)
        m.calc_range_fan(poses, got, fov, B)
        omap.update(occ2)
        om2 = oracle_mod.OracleMap(occ2, g.resolution, g.origin, mrx)
        m.calc_range_fan(poses, got, fov, B)
        assert m.last_plan()["code"] == 2
        assert np.array_equal(got, om2.rm_fan(poses, fov, B, step_coeff=1.0, nthreads=4, want_hits=False, want_steps=False)[0]), name
        m.close()
        omap.close()


def test_code_map_falls_back_when_the_palette_does_not_fit(oracle_mod):
    """A map with more distinct steps than the LDS palette holds (an open 900^2 room with a few obstacles and a 700-cell range:
    every d^2 = a^2 + b^2 up to 700^2 occurs — tens of thousands of them against plan::CODE_MAX_ENTRIES = 4096): the handle
    reports no code map, the plan stays on the float32 step map, the results are the oracle's.  And a range so long that
    the palette's d^2 histogram itself would not fit (max_range 2000 cells) does the same without trying."""
    occ = np.zeros((900, 900), np.uint8)
    occ[0, :] = occ[-1, :] = occ[:, 0] = occ[:, -1] = 1
    occ[450, 450] = occ[100, 700] = occ[777, 123] = 1
    g = maps.GridMap(occ, 0.05, (-3.0, -2.0, 0.3), "open900")
    B, fov = 720, 6.0
    for mrx in (700, 2000):
        om = oracle_mod.OracleMap.from_gridmap(g, mrx)
        omap = range_libc.PyOMap(g)
        poses = maps.sample_free_poses(g, 96, 5, dt=om.dt)
        m = range_libc.PyRayMarchingGPU(omap, mrx)
        m.set_option("slots", 2)
        m.set_option("code_min_rays", 0)
        got = np.empty(len(poses) * B, np.float32)
        m.calc_range_fan(poses, got, fov, B)
        assert m.get_info("code_map") == 2 and m.get_info("code_entries") == 0, (mrx, m.get_info("code_entries"))
        pl = m.last_plan()
        assert pl["code"] == 0 and pl["name"].endswith(", 2, false, 0>"), pl
        assert np.array_equal(got, om.rm_fan(poses, fov, B, step_coeff=1.0, nthreads=4, want_hits=False, want_steps=False)[0]), mrx
        m.close()
        omap.close()


def test_upstream_literal_mode_in_production_shape(oracle_mod):
    """variant 3 as a PRODUCTION mode: the upstream-literal arithmetic — per-ray theta_p +
    (-fov/2 + j * inc) in float32, glibc sinf / cosf at claim time, un-fused position and hit range, calc_range(y, x,
    theta') — on the stream kernel's schedule (rm_fan_stream_kernel<.., LIT>): two rays per lane, pipelined, noise, the
    fused crash test per roll-out (what scripts/racecar_simulator_v2.py:146-167 consumes), batches beyond one INLINE
    launch in pose slices.  Everything bit-identical to the checker's orc_rm_fan_libm, isCrashed over ITS ranges."""
    torch = pytest.importorskip("torch")
    from pyracecarsimulator_amd import racecar as RC
    w = workloads.cfg2()
    g, B, mrx = w.gmap, w.num_rays, w.max_range_px
    omap = range_libc.PyOMap(g)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    all_poses = workloads.make_poses(w, dt=om.dt)
    edge = RC.edge_distances(B, -w.fov / 2.0, w.fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    for cls, sc in ((range_libc.PyRayMarchingGPU, 1.0), (range_libc.PyRayMarching, 0.999)):
        m = cls(omap, mrx)
        m.set_option("variant", 3)
        for n, opts in ((200, {}), (600, {"slots": 2}), (600, {"slots": 2, "group_drain": 8}), (1000, {"slots": 1}),
                        (1000, {"slots": 2, "inline_map_kb": 0, "stripe_max": 0}), (1000, {"slots": 2, "grid_mult": 3})):
            for k, v in opts.items():
                m.set_option(k, v)
            poses = np.ascontiguousarray(all_poses[:n])
            poses[3] = [np.nan, 0.0, 0.0]
            poses[5] = [1e6, -1e6, 1.0]
            poses[7, 2] = np.inf
            want = om.rm_fan_libm(poses, w.fov, B, step_coeff=sc)[0]
            got = np.full(n * B, -1.0, np.float32)
            m.calc_range_fan(poses, got, w.fov, B)
            assert m.last_plan()["kernel"] == "rm_stream_literal", (n, opts, m.last_plan())
            assert np.array_equal(got, want), (cls.__name__, n, opts, int((got != want).sum()))
            # fused crash test, roll-outs of 100 poses: indices of Car::isCrashed over the literal ranges
            grp = 100
            first = m.check_collision_groups(poses, grp, w.fov, B, edge, 0.001)
            ref = [oracle_mod.is_crashed(want[q * grp * B:(q + 1) * grp * B], B, grp, edge, 0.001) for q in range(n // grp)]
            assert first.tolist() == ref, (cls.__name__, n, opts)
        m.close()
    # a batch beyond one INLINE launch: pose slices of 4096, noise keyed by the global ray id, crash marks shifted per slice
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    m.set_option("variant", 3)
    n, Bs = 9000, 96
    poses = np.ascontiguousarray(workloads.make_poses(w, dt=om.dt, n_poses=n, seed=77))
    want = om.rm_fan_libm(poses, 2.0, Bs, step_coeff=1.0)[0]
    got = np.empty(n * Bs, np.float32)
    m.calc_range_fan(poses, got, 2.0, Bs)
    pl = m.plan_fan(n, Bs)                      # (last_plan() is the last SLICE's launch)
    assert pl["kernel"] == "rm_stream_literal" and pl["slices"] == 3 and pl["slice_poses"] == 4096, pl
    assert m.last_plan()["kernel"] == "rm_stream_literal"
    assert np.array_equal(got, want)
    edge96 = np.full(Bs, 0.3)
    first = m.check_collision_groups(poses, 90, 2.0, Bs, edge96, 0.001)
    assert first.tolist() == [oracle_mod.is_crashed(want[q * 90 * Bs:(q + 1) * 90 * Bs], Bs, 90, edge96, 0.001) for q in range(100)]
    # noise: the sliced literal launch and the one-lane-per-ray kernel (diagnostics force it) draw the same noise
    m.set_noise(0.01, 5, 1000)
    a, b, hb = np.empty(n * Bs, np.float32), np.empty(n * Bs, np.float32), np.empty((n * Bs, 2), np.int32)
    m.calc_range_fan(poses, a, 2.0, Bs)
    m.calc_range_fan(poses, b, 2.0, Bs, hit_cells=hb)
    assert m.last_plan()["kernel"] == "rm_literal" and np.array_equal(a, b)
    assert 0.009 < float((a - want).std()) < 0.011
    # device-resident, four launches in flight on four streams (the shape bench.py --opt variant=3 times)
    m.set_noise(0.0, 0, 0)
    m.set_option("slots", 2)
    m.set_option("grid_mult", 3)
    from pyracecarsimulator_amd.pipeline import concurrent_streams
    streams = concurrent_streams(4)
    n = 4096
    batches = [np.ascontiguousarray(workloads.make_poses(w, dt=om.dt, n_poses=n, seed=900 + k)) for k in range(len(streams))]
    d_p = [torch.from_numpy(b).cuda() for b in batches]
    d_o = [torch.empty(n * B, dtype=torch.float32, device="cuda") for _ in batches]
    for rep in range(3):
        for k, st in enumerate(streams):
            m.calc_range_fan_device(d_p[k].data_ptr(), n, w.fov, B, d_o[k].data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    sub = np.linspace(0, n - 1, 24).astype(np.int64)
    for k in range(len(streams)):
        want = om.rm_fan_libm(batches[k][sub], w.fov, B, step_coeff=1.0)[0]
        assert np.array_equal(d_o[k].view(n, B)[torch.from_numpy(sub).cuda()].reshape(-1).cpu().numpy(), want), k


def test_pyomap_from_occupancy_grid_message_scans_like_the_oracle(oracle_mod):
    """Row a6 end to end: PyOMap(map_msg) with a quaternion origin (yaw != 0) and map_server data
    binarised as /root/reference/scripts/ros_interface.py:80-86 does -> ScanSimulator2D.scan."""
    from conftest import occupancy_grid_msg
    g = maps.make_maze(200, cell=25, wall=2, p=0.5, seed=12, resolution=0.05, origin=(-3.0, 1.5, 0.7))
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    poses = maps.sample_free_poses(g, 40, 3, dt=om.dt)
    want = om.rm_fan_libm(poses, 4.71, 1081, step_coeff=0.999)[0]      # "RM": the upstream-literal arithmetic
    for binarise in (True, False):
        omap = range_libc.PyOMap(occupancy_grid_msg(g, binarise))
        assert (omap.height, omap.width) == (g.rows, g.cols)
        assert np.array_equal(omap.distance_transform(), om.dt)
        sim = ScanSimulator2D(1081, 4.71, 0.01, batch_size=40)
        sim.setMap(omap, 300, g.resolution, omap.origin)
        sim.setRaytracingMethod("RM")
        assert np.array_equal(sim.scanMany(poses), want)
        assert np.array_equal(sim.scan(*poses[7]), want[7 * 1081:8 * 1081])


# ---------------------------------------------------------------- K1 vs oracle, seeded sweeps
@pytest.mark.parametrize("B,fov", [(1, 4.71), (63, 1.0), (64, 6.283), (65, 4.71), (1081, 4.71),
                                   (1080, 4.71), (720, 3.14), (2500, 6.0)])
def test_rm_fan_vs_oracle_beam_counts(oracle_mod, B, fov):
    g = maps.make_maze(320, cell=32, wall=3, p=0.5, seed=B, resolution=0.05,
                       origin=(-3.0, 4.0, 0.1 * (B % 7)))
    mrx = 150
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    poses = maps.sample_free_poses(g, 37, B + 1, dt=om.dt)
    for sc, cls in ((0.999, range_libc.PyRayMarching), (1.0, range_libc.PyRayMarchingGPU)):
        m = cls(omap, mrx)
        if cls is range_libc.PyRayMarching:
            # range_libc's CPU RayMarching: the upstream-literal arithmetic by default — ranges, hit cells, sample counts
            # (diagnostics: one lane per ray) and the ranges-only launch (stream form from 64 beams up), bit for bit
            rl, hl, sl = _fan(m, poses, fov, B)
            r0, h0, s0 = om.rm_fan_libm(poses, fov, B, step_coeff=sc)
            assert np.array_equal(rl, r0) and np.array_equal(hl, h0) and np.array_equal(sl, s0)
            rr = np.empty(len(poses) * B, np.float32)
            m.calc_range_fan(poses, rr, fov, B)
            assert np.array_equal(rr, r0) and m.last_plan()["kernel"] == ("rm_stream_literal" if B >= 64 else "rm_literal")
            m.set_option("variant", 1)
        r, h, s = _fan(m, poses, fov, B)
        r0, h0, s0 = oracle_mod.OracleMap.rm_fan(om, poses, fov, B, step_coeff=sc)
        assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s, s0)


def test_one_handle_called_with_many_fans_keeps_its_direction_tables_straight(oracle_mod):
    """The beam-direction table is cached per (fov, num_rays) on the handle, four fans deep: seven
    fans alternating on one handle go through hits, fills and evictions, ranges / hit cells / sample
    counts of every call equal the oracle's."""
    g = maps.make_maze(256, cell=32, wall=3, p=0.5, seed=77, origin=(1.0, -2.0, 0.4))
    mrx = 120
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    poses = maps.sample_free_poses(g, 70, 5, dt=om.dt)
    fans = [(1081, 4.71), (720, 4.71), (1081, 3.0), (64, 6.283), (271, 4.71), (1081, -4.71), (100, 0.5)]
    want = {fan: om.rm_fan(poses, fan[1], fan[0], step_coeff=1.0) for fan in fans}
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    for fan in fans + fans[::-1] + fans[::2] + fans:
        r, h, s = _fan(m, poses, fan[1], fan[0])
        r0, h0, s0 = want[fan]
        assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s, s0), fan


def test_edge_cases(oracle_mod):
    g = maps.make_room(128, wall=2)
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarching(omap, 300)
    m.set_option("variant", 1)                  # (the canonical kernels; the literal form's edge cases: test_upstream_literal_*)
    poses = np.array([
        [3.0, 3.0, 0.0], [3.0, 3.0, 1e-30], [3.0, 3.0, -1e-38], [3.0, 3.0, 1e-42],   # tiny / denormal
        [0.01, 0.01, 0.7],                      # inside the wall
        [-0.3 * 0.05, 1.0, 0.0],                # -1 < gx < 0: in-map by truncation
        [-5.0, -5.0, 0.3], [100.0, 3.0, 3.0],   # outside the map
        [np.nan, 1.0, 0.0], [1.0, np.inf, 0.0], [1.0, 1.0, np.nan], [1e30, -1e30, 5.0],
        [3.0, 3.0, 1e4], [3.0, 3.0, -777.0],    # large headings
    ], np.float32)
    r, h, s = _fan(m, poses, 4.71, 129)
    r0, h0, s0 = om.rm_fan(poses, 4.71, 129)
    assert np.array_equal(r, r0, equal_nan=True) and np.array_equal(h, h0) and np.array_equal(s, s0)
    # largest supported fan (LDS table), and one beam too many
    big = np.empty(2 * 7680, np.float32)
    m.calc_range_fan(poses[:2], big, 6.0, 7680)
    assert np.array_equal(big, om.rm_fan(poses[:2], 6.0, 7680)[0])
    with pytest.raises(Exception):
        m.calc_range_fan(poses[:2], np.empty(2 * 7681, np.float32), 6.0, 7681)
    # empty batch: nothing happens, nothing raises
    out = np.empty(0, np.float32)
    m.calc_range_fan(np.zeros((0, 3), np.float32), out, 4.71, 129)
    m.calc_range_many(np.zeros((0, 3), np.float32), out)
    # map without any obstacle: every ray is a miss at max range
    omap2 = range_libc.PyOMap(np.zeros((50, 60), np.uint8), 0.1)
    r, h, s = _fan(range_libc.PyRayMarchingGPU(omap2, 40), np.array([[3.0, 2.5, 0.2]], np.float32), 6.0, 100)
    assert np.all(r == np.float32(40) * np.float32(0.1)) and np.all(h == -1)


def test_two_arg_rays_api_and_four_arg_sparse_api(oracle_mod):
    g, z = load_golden("rm_maze192_yaw")
    mrx, B, fov = int(z["max_range_px"]), int(z["num_rays"]), float(z["fov"])
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarching(omap, mrx)
    # upstream 2-arg form: one (x,y,theta) row per ray (scripts/two_player/scan.py:57-70)
    rng = np.random.default_rng(3)
    ins = z["poses"][rng.integers(0, len(z["poses"]), 5000)].copy()
    ins[:, 2] = rng.uniform(-7, 7, len(ins)).astype(np.float32)
    outs = np.zeros(len(ins), np.float32)
    assert m.calc_range_many(ins, outs) is None
    r0 = om.rm_rays_libm(ins, step_coeff=0.999)                   # PyRayMarching: the upstream-literal arithmetic
    assert np.array_equal(outs, r0)
    assert m.calc_range(*ins[7]) == float(r0[7])
    m.set_option("variant", 1)
    m.calc_range_many(ins, outs)
    assert np.array_equal(outs, om.rm_rays(ins)[0])               # ... the canonical one on request
    m.set_option("variant", 3)
    # the fork's 4-arg form exactly as ScanSimulator2D calls it: sparse ins, pose p at row p*B
    P = 5
    sparse = np.zeros((P * B, 3), np.float32)
    sparse[::B] = z["poses"][:P]
    outs = np.full(P * B, -1, np.float32)
    assert m.calc_range_many(sparse, outs, fov, B) is None
    assert np.array_equal(outs, om.rm_fan_libm(z["poses"][:P], fov, B, step_coeff=0.999)[0])
    m.set_option("variant", 1)
    m.calc_range_many(sparse, outs, fov, B)
    assert np.array_equal(outs, z["ranges_cpu"][:P * B])
    with pytest.raises(ValueError):
        m.calc_range_many(sparse.astype(np.float64), outs, fov, B)
    with pytest.raises(TypeError):
        m.calc_range_many(sparse, outs, fov)


def test_scan_simulator_end_to_end(oracle_mod):
    g = maps.load_colombia()
    mrx = int(15.0 / g.resolution)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    poses = maps.sample_free_poses(g, 10, 77, dt=om.dt)
    for method, sc in (("RM", 0.999), ("RMGPU", 1.0)):
        sim = ScanSimulator2D(1081, 4.71, 0.01, batch_size=8)
        sim.setMap(omap, mrx, g.resolution, g.origin)
        sim.setRaytracingMethod(method)
        want = (om.rm_fan_libm if method == "RM" else om.rm_fan)(poses[:8], 4.71, 1081, step_coeff=sc)[0]
        many = sim.scanMany(poses)                  # 10 given, batch_size 8 scanned
        assert many is sim.output_vector_many and np.array_equal(many, want)
        one = sim.scan(float(poses[2, 0]), float(poses[2, 1]), float(poses[2, 2]))
        assert one is sim.output_vector and np.array_equal(one, want[2 * 1081:3 * 1081])
        first = one.copy()
        sim.scan(*[float(v) for v in poses[3]])     # alias: the earlier result is overwritten
        assert not np.array_equal(one, first)


def test_noise_statistics_and_shard_invariance(oracle_mod):
    g, z = load_golden("rm_maze256")
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, 300)
    poses, B = z["poses"], 1081
    clean = np.empty(len(poses) * B, np.float32)
    m.calc_range_fan(poses, clean, 4.71, B)
    assert np.array_equal(clean, z["ranges_gpu"])
    m.set_noise(0.01, seed=6)
    noisy = np.empty_like(clean)
    m.calc_range_fan(poses, noisy, 4.71, B)
    d = (noisy - clean).astype(np.float64)
    assert abs(d.mean()) < 3e-4 and abs(d.std() - 0.01) < 3e-4
    k = (d / 0.01)
    assert abs((k ** 3).mean()) < 0.05 and abs((k ** 4).mean() - 3.0) < 0.15      # gaussian moments
    again = np.empty_like(clean)
    m.calc_range_fan(poses, again, 4.71, B)
    assert np.array_equal(noisy, again)                                           # counter-based
    m.set_option("slice_log2", 13)            # launch cut into pose slices: same global ray ids
    m.calc_range_fan(poses, again, 4.71, B)
    assert np.array_equal(noisy, again)
    for mode in (1, 2):                       # the event pair spans every slice, not the last one
        m.set_option("timing", mode)
        m.set_option("slice_log2", 30)
        m.calc_range_fan(poses, again, 4.71, B)
        whole = m.last_kernel_ms()
        m.set_option("slice_log2", 13)
        m.calc_range_fan(poses, again, 4.71, B)
        assert m.last_kernel_ms() > 1.5 * whole > 0.0
    m.set_option("timing", 0)
    m.set_option("slice_log2", 30)
    # sharding: second half scanned alone with ray_offset reproduces the unsharded noise
    half = len(poses) // 2
    m.set_noise(0.01, seed=6, ray_offset=half * B)
    part = np.empty((len(poses) - half) * B, np.float32)
    m.calc_range_fan(poses[half:], part, 4.71, B)
    assert np.array_equal(part, noisy[half * B:])
    m.set_noise(0.0)
    m.calc_range_fan(poses, again, 4.71, B)
    assert np.array_equal(again, clean)


def test_fused_crash_test_matches_is_crashed(oracle_mod):
    g = maps.load_colombia()
    mrx, B, fov = 300, 1081, 4.71
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    edge = oracle_mod.edge_distances(B, -fov / 2, fov / B, 0.275, 0.2032, 0.3302)
    rng = np.random.default_rng(8)
    seen = set()
    for trial in range(12):
        clear = 2.0 if trial % 2 == 0 else 8.0
        rr, cc = np.nonzero(om.dt >= clear)
        k = rng.integers(0, rr.size, 40)
        poses = np.stack([cc[k] * 0.05 + g.origin[0] + 0.02, rr[k] * 0.05 + g.origin[1] + 0.02,
                          rng.uniform(-3, 3, 40)], 1).astype(np.float32)
        ranges = np.empty(40 * B, np.float32)
        code = m.check_collision_many(poses, fov, B, edge, 0.001, ranges=ranges)
        want_r, _, _ = om.rm_fan(poses, fov, B, step_coeff=1.0)
        assert np.array_equal(ranges, want_r)
        assert code == oracle_mod.is_crashed(want_r, B, 40, edge, 0.001)
        assert m.check_collision_many(poses, fov, B, edge, 0.001) == code     # ranges not requested
        seen.add(code >= 0)
    assert seen == {True, False}
    assert m.check_collision_many(np.zeros((0, 3), np.float32), fov, B, edge, 0.001) == -1


def test_map_update_rebuilds_tables(oracle_mod):
    g = maps.make_maze(200, cell=25, wall=2, seed=5)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarching(omap, 120)
    m.set_option("variant", 1)
    poses = maps.sample_free_poses(g, 6, 1)
    occ2 = g.occ.copy()
    occ2[90:110, 90:110] = 1                         # stamp an obstacle (two-player car outline)
    omap.update(occ2)
    om2 = oracle_mod.OracleMap(occ2, g.resolution, g.origin, 120)
    assert np.array_equal(omap.distance_transform(), om2.dt)
    r, h, s = _fan(m, poses, 4.71, 360)
    r0, h0, s0 = om2.rm_fan(poses, 4.71, 360)
    assert np.array_equal(r, r0) and np.array_equal(h, h0)


def test_device_resident_api_with_torch_streams(oracle_mod):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    g, z = load_golden("rm_colombia")
    omap = range_libc.PyOMap(g, device=0)
    m = range_libc.PyRayMarchingGPU(omap, 300)
    with pytest.raises(Exception):
        m.last_kernel_ms()                       # timing events are opt-in
    m.set_option("timing", 1)
    B = 1081
    d_poses = torch.from_numpy(z["poses"]).cuda()
    d_out = torch.zeros(len(z["poses"]) * B, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        m.calc_range_fan_device(d_poses.data_ptr(), len(z["poses"]), 4.71, B, d_out.data_ptr(),
                                stream=side.cuda_stream)
    side.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), z["ranges_gpu"])
    assert m.last_kernel_ms() > 0.0


# ---------------------------------------------------------------- full BASELINE size: properties
def test_cfg1_one_pose_through_scan_simulator(oracle_mod):
    """configs[0]: the map of maps/map.yaml (2049^2; the .pgm is missing from the mount, so the seeded maze
    stands in), ONE pose, 1081 beams, "RM" (range_libc's CPU RayMarching, step coefficient 0.999)
    through ScanSimulator2D.scan — the reference's own plumbing case (scripts/scan_simulator.py:88-111) —
    bit for bit against the oracle, for a handful of poses one at a time."""
    w = workloads.cfg1()
    assert (w.n_poses, w.num_rays, w.method) == (1, 1081, "RM")
    g = w.gmap
    omap = range_libc.PyOMap(g)
    om = oracle_mod.OracleMap.from_gridmap(g, w.max_range_px)
    # the device EDT at FULL size against the oracle's own EDT (Felzenszwalb in C, ~1 s at 4096^2): nothing of the
    # device is fed to the checker
    assert np.array_equal(omap.distance_transform(), om.dt), "device EDT differs from the oracle's at full size"
    sim = ScanSimulator2D(w.num_rays, w.fov, 0.01, batch_size=1)
    sim.setMap(omap, w.max_range_px, g.resolution, g.origin)
    sim.setRaytracingMethod(w.method)
    poses = np.concatenate([workloads.make_poses(w, dt=om.dt), maps.sample_free_poses(g, 7, 99, dt=om.dt)])
    for p in poses:
        got = sim.scan(float(p[0]), float(p[1]), float(p[2]))
        assert got is sim.output_vector
        # "RM" = range_libc's CPU RayMarching = the upstream-literal statement (per-ray float32 angle, glibc sinf / cosf,
        # un-fused march: the checker's rm_fan_libm), bit for bit
        want = om.rm_fan_libm(p[None, :], w.fov, w.num_rays, step_coeff=0.999)[0]
        assert np.array_equal(got, want)
    assert sim.scan_method.get_info("variant") == 3 and sim.scan_method.last_plan()["kernel"] == "rm_stream_literal"
    sim.scan_method.set_option("variant", 1)                      # the canonical arithmetic on request
    for p in poses[:3]:
        assert np.array_equal(sim.scan(float(p[0]), float(p[1]), float(p[2])),
                              om.rm_fan(p[None, :], w.fov, w.num_rays, step_coeff=0.999)[0])


def test_cfg2_full_size_properties(oracle_mod):
    """configs[1]: 2049^2 map, 4096 poses x 1081 beams (4.4 M rays) — checked through properties
    that do not need the oracle on every ray, plus an oracle spot check on a pose subsample."""
    w = workloads.cfg2()
    g, B, mrx = w.gmap, w.num_rays, w.max_range_px
    omap = range_libc.PyOMap(g)
    dt = omap.distance_transform()
    assert np.array_equal(dt == 0, g.occ != 0)
    poses = workloads.make_poses(w, dt=dt)
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    r, h, s = _fan(m, poses, w.fov, B)
    hit = h[:, 0] >= 0
    # (a) every reported hit cell is an occupied cell inside the map
    assert np.all(g.occ[h[hit, 1], h[hit, 0]] == 1)
    # (b) range == |origin - hit cell corner| * res in float32, recomputed independently
    inv = np.float32(1.0 / float(np.float32(g.resolution)))
    gx = np.repeat(((poses[:, 0] - np.float32(g.origin[0])) * inv).astype(np.float32), B)
    gy = np.repeat(((poses[:, 1] - np.float32(g.origin[1])) * inv).astype(np.float32), B)
    xd = (h[hit, 0].astype(np.float32) - gx[hit]).astype(np.float64)
    yd = (h[hit, 1].astype(np.float32) - gy[hit]).astype(np.float64)
    want = (np.sqrt((xd * xd + yd * yd).astype(np.float32)).astype(np.float32) * np.float32(g.resolution))
    assert np.abs(r[hit] - want).max() <= 2e-6
    # (c) misses report exactly max range; all ranges bounded; step counts sane
    assert np.all(r[~hit] == np.float32(mrx) * np.float32(g.resolution))
    assert r.min() >= 0 and r.max() <= (mrx + 1.5) * g.resolution and s.min() >= 1
    # (d) determinism + batch-split invariance
    r2 = np.empty_like(r)
    m.calc_range_fan(poses, r2, w.fov, B)
    assert np.array_equal(r, r2)
    cut = 1357
    ra, rb = np.empty(cut * B, np.float32), np.empty((len(poses) - cut) * B, np.float32)
    m.calc_range_fan(poses[:cut], ra, w.fov, B)
    m.calc_range_fan(poses[cut:], rb, w.fov, B)
    assert np.array_equal(np.concatenate([ra, rb]), r)
    # (e) oracle spot check: every 64th pose, bit-exact
    sub = np.arange(0, len(poses), 64)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    r0, h0, s0 = om.rm_fan(poses[sub], w.fov, B, step_coeff=1.0, nthreads=4)
    pick = (sub[:, None] * B + np.arange(B)[None, :]).ravel()
    assert np.array_equal(r[pick], r0) and np.array_equal(h[pick], h0) and np.array_equal(s[pick], s0)


# ---------------------------------------------------------------- K2: BresenhamsLine (LDS tile)
@pytest.mark.parametrize("variant", [0, 1])        # 0: LDS-window kernel (K2), 1: stream kernel (K2b)
@pytest.mark.parametrize("name", ["rm_colombia", "rm_maze256", "rm_maze192_yaw"])
def test_bresenham_fan_reproduces_golden_vectors(oracle_mod, name, variant):
    g, z = load_golden(name)
    omap = range_libc.PyOMap(g)
    fov, B, mrx = float(z["fov"]), int(z["num_rays"]), int(z["max_range_px"])
    m = range_libc.PyBresenhamsLine(omap, mrx)
    m.set_option("variant", variant)
    r, h, s = _fan(m, z["poses"], fov, B)
    assert np.array_equal(r, z["ranges_bl"]) and np.array_equal(h, z["hits_bl"].astype(np.int32))
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    r0, h0, s0 = om.bl_fan(z["poses"], fov, B)
    assert np.array_equal(s, s0)
    r1 = np.empty_like(r)
    m.calc_range_fan(z["poses"], r1, fov, B)                  # non-AUX template
    assert np.array_equal(r1, r0)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("mrx", [40, 300, 700])               # 700: window > LDS -> global bit map
def test_bresenham_vs_oracle_edge_cases_and_rays(oracle_mod, mrx, variant):
    g = maps.make_maze(300, cell=30, wall=2, p=0.5, seed=mrx, origin=(2.0, -1.0, 0.35))
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyBresenhamsLine(omap, mrx)
    m.set_option("variant", variant)
    poses = maps.sample_free_poses(g, 40, 3)
    poses[3] = [np.nan, 0, 0]
    poses[4] = [1e20, 1.0, 0.5]
    poses[5] = [g.origin[0] - 0.01, g.origin[1] - 0.01, 0.8]          # just outside the map
    poses[6] = [g.origin[0] + 0.001, g.origin[1] + 0.001, 0.8]        # inside the border wall
    poses[7, 2] = 1e-30
    r, h, s = _fan(m, poses, 4.71, 257)
    r0, h0, s0 = om.bl_fan(poses, 4.71, 257)
    assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s, s0)
    rng = np.random.default_rng(2)
    ins = poses[rng.integers(0, 40, 3000)].copy()
    ins[:, 2] = rng.uniform(-8, 8, 3000).astype(np.float32)
    outs = np.empty(3000, np.float32)
    m.calc_range_many(ins, outs)
    r0, _, _ = om.bl_rays(ins)
    assert np.array_equal(outs, r0)


def test_occ_fan_lds_north_star_shape_within_one_cell_of_ray_marching(oracle_mod):
    """occupancy window + fan in LDS, unit-step march, wave ballot picks the
    first hit.  Not sphere tracing, so not bit-identical: acceptance is "within one cell of the oracle's
    RayMarching", met on all but corner-grazing rays; hits land on occupied cells, misses agree."""
    for name in ("rm_colombia", "rm_maze256"):
        g, z = load_golden(name)
        om = oracle_mod.OracleMap.from_gridmap(g, 300)
        omap = range_libc.PyOMap(g)
        m = range_libc.PyRayMarchingGPU(omap, 300)
        m.set_option("variant", 2)
        poses = z["poses"]
        r, h, s = _fan(m, poses, 4.71, 1081)
        r0, h0, _ = om.rm_fan(poses, 4.71, 1081, step_coeff=1.0)
        err = np.abs(r - r0) / g.resolution
        within = float((err <= 1.0001).mean())
        print("%s: occ_fan_lds within one cell of exact ray marching on %.4f of the rays, identical on %.4f"
              % (name, within, (err == 0).mean()))
        # colombia's one-cell-thick walls: sphere tracing steps by the distance between CELL INDICES, so
        # it can pass a thin wall at a shallow angle where unit steps hit it — the dense march is the
        # stricter of the two there
        assert within > (0.95 if name == "rm_colombia" else 0.995), (name, within)
        assert np.median(err) == 0.0
        hit = h[:, 0] >= 0
        assert g.occ[h[hit, 1], h[hit, 0]].all()                    # every reported hit cell is occupied
        assert ((h0[:, 0] >= 0) == hit).mean() > 0.98               # hit / miss decisions agree
        # edge cases: outside the map, NaN
        bad = np.array([[-50.0, 0.0, 0.0], [np.nan, 0.0, 0.0]], np.float32)
        rb = np.empty(2 * 1081, np.float32)
        m.calc_range_fan(bad, rb, 4.71, 1081)
        assert np.array_equal(rb, np.full(2 * 1081, np.float32(300.0) * np.float32(g.resolution)))


@pytest.mark.parametrize("variant", [0, 1])
def test_bresenham_coordinate_rounds_up_across_a_power_of_two(oracle_mod, variant):
    """The walk's float coordinate x0 + 1 + 1 + ... can gain a cell: 127.99999237 + 1 lies exactly half
    way between two float32 values above 128 and rounds to 129.0.  When the skipped integer (128) is the
    walk's end cell, the end test never fires, the walk runs to its step budget and stands budget + 1
    cells from its start cell — an occupied cell exactly there must be hit.  (Found by the 30-minute fuzz
    of round 2: K2's LDS window had no margin for that cell and read stale LDS one row outside.)"""
    occ = np.zeros((160, 96), np.uint8)
    occ[132, :] = 1                                    # 11 rows above the start row 121
    occ[:100, 68] = 1                                  # 11 columns right of the start column 57
    g = maps.GridMap(occ, 1.0, (0.0, 0.0, 0.0), "binade")
    mrx = 7                                            # budget: 10 steps
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    y0 = np.float32(122.0) - np.float32(2.0 ** -17)    # 121.99999237: reaches 127.99999237 after 6 steps
    x0 = np.float32(58.0) - np.float32(2.0 ** -18)     # 57.99999619: reaches 63.99999619 after 6 steps
    poses = np.array([[50.5, y0, 0.9], [20.25, y0, 0.9], [x0, 20.5, 0.67], [x0, 60.25, 0.67],
                      [50.5, np.float32(122.25), 0.9]], np.float32)
    r0, h0, s0 = om.bl_fan(poses, 0.15, 64)
    assert (h0[:128, 1] == 132).all() and (s0[:128] == 10).all()          # the statement: 11 rows in 10 steps
    assert (h0[128:256, 0] == 68).all() and (h0[256:] == -1).all()
    omap = range_libc.PyOMap(g)
    m = range_libc.PyBresenhamsLine(omap, mrx)
    m.set_option("variant", variant)
    for B, fov in ((64, 0.15), (1, 0.0), (100, 6.283)):
        r, h, s = _fan(m, poses, fov, B)
        r1, h1, s1 = om.bl_fan(poses, fov, B)
        assert np.array_equal(r, r1) and np.array_equal(h, h1) and np.array_equal(s, s1), (variant, B)


# ---------------------------------------------------------------- K3: GiantLUT
def test_giant_lut_table_and_queries_bit_equal_to_oracle(oracle_mod):
    g = maps.make_maze(96, cell=16, wall=2, p=0.5, seed=4, origin=(-1.0, 0.5, 0.2))
    mrx, td = 80, 180
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyGiantLUTCast(omap, mrx, td)
    lut0 = om.lut_build(td, nthreads=4)
    assert np.array_equal(m.table(), lut0)
    assert np.array_equal(m.table(10, 13), lut0[10:13])
    poses = maps.sample_free_poses(g, 50, 1)
    poses[0] = [-50.0, 0.0, 0.0]                       # outside: max range
    poses[1] = [np.nan, 0.0, 0.0]
    out = np.empty(50 * 271, np.float32)
    m.calc_range_fan(poses, out, 4.71, 271)
    assert np.array_equal(out, om.lut_fan(lut0, poses, 4.71, 271))
    rng = np.random.default_rng(5)
    ins = poses[rng.integers(0, 50, 4000)].copy()
    ins[:, 2] = rng.uniform(-20, 20, 4000).astype(np.float32)
    outs = np.empty(4000, np.float32)
    m.calc_range_many(ins, outs)
    assert np.array_equal(outs, om.lut_rays(lut0, ins))
    # LUT answers track exact ray marching (cell-corner origin + angular rounding)
    rm, _, _ = om.rm_rays(ins[2:])
    err = np.abs(outs[2:] - np.minimum(rm, mrx * g.resolution)) / g.resolution
    assert np.median(err) < 1.5
    with pytest.raises(Exception):
        m.calc_range_fan(poses, out, 4.71, 271, hit_cells=np.empty((50 * 271, 2), np.int32))


# theta_disc decides how many 16-B loads per lane fetch a pose's theta row into LDS (NL = 1..3; wider
# rows take the per-beam kernel): every width, with the trimmed fetch (fov < 2pi), the whole-row
# fetch (fov >= 2pi, negative fov), headings that wrap the bin range, 12- and 17-chunk fans
@pytest.mark.parametrize("td", [510, 514, 720, 1024, 1026, 1442, 1536, 1538, 1443])
def test_giant_lut_fan_every_row_width_bit_equal_to_oracle(oracle_mod, td):
    g = maps.make_maze(56, cell=14, wall=2, p=0.5, seed=td, origin=(2.0, -1.5, -0.3))
    mrx = 60
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyGiantLUTCast(omap, mrx, td)
    lut0 = om.lut_build(td, nthreads=oracle_mod.max_threads())
    assert np.array_equal(m.table(), lut0)
    rng = np.random.default_rng(td)
    poses = maps.sample_free_poses(g, 96, td)
    poses[:, 2] = rng.uniform(-7.0, 7.0, len(poses)).astype(np.float32)
    # headings that put the first beam's bin right at the wrap of the row
    for k, b in enumerate((0.0, 0.49, 0.51, td - 1.0, td - 0.5, td / 2.0)):
        poses[8 + k, 2] = np.float32(b * 2 * math.pi / td + 4.71 / 2 - g.origin[2])
    poses[0] = [-50.0, 0.0, 0.0]
    poses[1] = [np.nan, 0.0, 0.0]
    poses[2, 2] = 1e4
    poses[3, 2] = -3e7                                      # |bin index| > 2^23: integer path
    poses[4, 2] = 2.5e4                                     # between the one-wrap form's 2^22 bound and 2^23 (td 1442)
    for B, fov in ((1081, 4.71), (271, 4.71), (1081, 2 * math.pi), (1088, 7.0), (720, -2.0), (64, 0.0),
                   (1081, 6.25), (700, 6.2831855)):
        out = np.empty(len(poses) * B, np.float32)
        m.calc_range_fan(poses, out, fov, B)
        want = om.lut_fan(lut0, poses, fov, B)
        assert np.array_equal(out, want), (td, B, fov, int((out != want).sum()))
        m.set_option("lut_debug", 16)                       # plain instead of non-temporal row loads: same bits
        out[:] = -1.0
        m.calc_range_fan(poses, out, fov, B)
        m.set_option("lut_debug", 0)
        assert np.array_equal(out, want), (td, B, fov, "plain loads")
    # the last cell of the table: the row's 16-B tail loads stay inside the allocation
    corner = np.array([[g.origin[0], g.origin[1], 0.3]], np.float32)
    c, s_ = math.cos(g.origin[2]), math.sin(g.origin[2])
    gx, gy = g.cols - 0.5, g.rows - 0.5
    corner[0, 0] += (c * gx - s_ * gy) * g.resolution
    corner[0, 1] += (s_ * gx + c * gy) * g.resolution
    out = np.empty(1081, np.float32)
    m.calc_range_fan(corner, out, 4.71, 1081)
    assert np.array_equal(out, om.lut_fan(lut0, corner, 4.71, 1081))


# ---------------------------------------------------------------- K3b: CDDT
@pytest.mark.parametrize("td", [112, 720, 113])        # 112: scripts/two_player/rcs_two_player.py:121
def test_cddt_queries_bit_equal_to_oracle(oracle_mod, td):
    g, z = load_golden("rm_maze256")
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyCDDTCast(omap, 300, td)
    poses = z["poses"][:12]
    out = np.empty(12 * 1081, np.float32)
    m.calc_range_fan(poses, out, 4.71, 1081)
    assert np.array_equal(out, om.cddt_fan(td, poses, 4.71, 1081))
    rng = np.random.default_rng(6)
    ins = z["poses"][rng.integers(0, len(z["poses"]), 5000)].copy()
    ins[:, 2] = rng.uniform(-10, 10, 5000).astype(np.float32)
    outs = np.empty(5000, np.float32)
    m.calc_range_many(ins, outs)                       # the reference's 2-arg CDDT call
    assert np.array_equal(outs, om.cddt_rays(td, ins))
    # rebuilt after a map change (two-player: car outline stamped every scan)
    occ2 = g.occ.copy()
    occ2[100:120, 100:140] = 1
    omap.update(occ2)
    om2 = oracle_mod.OracleMap(occ2, g.resolution, g.origin, 300)
    m.calc_range_many(ins, outs)
    assert np.array_equal(outs, om2.cddt_rays(td, ins))


def test_cddt_fan_searches_only_the_bins_the_fan_touches(oracle_mod):
    """The per-bin fan kernel searches the circular run of theta bins between the last and the first beam
    (one bin of margin): narrow and wide fans, fans past a full turn, one beam, headings whose bin index wraps
    many times or leaves the float-exact range — bit-equal to the per-ray statement of the oracle."""
    g, z = load_golden("rm_maze256")
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    rng = np.random.default_rng(11)
    poses = z["poses"][:40].copy()
    poses[:, 2] = rng.uniform(-math.pi, math.pi, 40).astype(np.float32)
    poses[5, 2] = 1.0e4
    poses[6, 2] = -3.0e5
    poses[7, 2] = 2.0e7                                   # |bin index| beyond 2^23 at theta_disc 720: integer path
    poses[8, 2] = 0.0
    poses[9, 2] = math.pi
    for td in (4, 16, 108, 113, 720):                    # (113: odd, table bin 0 has no half-turn partner)
        m = range_libc.PyCDDTCast(omap, 300, td)
        for fov, B in ((0.05, 33), (1.0, 257), (4.71, 1081), (6.2, 720), (6.2831855, 1081), (7.5, 1081), (4.71, 1),
                       (3.0, 2)):
            out = np.empty(len(poses) * B, np.float32)
            m.calc_range_fan(poses, out, fov, B)
            assert m.last_plan()["kernel"] == ("cddt_bins" if td <= B else "cddt_rays")
            want = om.cddt_fan(td, poses, fov, B)
            assert np.array_equal(out, want), (td, fov, B)
            if td <= B:                                   # the theta-major pair of kernels on the same fans
                m.set_option("cddt_theta_min", 1)
                # round 5's search kernel (look-ups prepared once per pose), round 4's, and search + fan fused per 64-pose tile
                for search in (1, 0, 2):
                    m.set_option("cddt_search", search)
                    out[:] = -1.0
                    m.calc_range_fan(poses, out, fov, B)
                    assert m.last_plan()["kernel"] == "cddt_theta"
                    # (the fused form needs its tile's (theta_disc | 1) + 1 rows of 64 floats in 64 KiB of LDS)
                    fits = ((td | 1) + 1) * 256 <= 65536
                    assert m.last_plan()["name"] == ("scan::cddt_theta_search_kernel", "scan::cddt_theta_search2_kernel",
                                                     "scan::cddt_theta_fused_kernel" if fits else "scan::cddt_theta_search2_kernel")[search]
                    assert np.array_equal(out, want), ("theta-major", search, td, fov, B)
                m.set_option("cddt_search", 1)
                m.set_option("cddt_theta_min", 32768)


@pytest.mark.parametrize("lds_sort", [16384, 128])      # 128: buckets above it take the global rank sort
def test_cddt_long_walls_fill_large_buckets(oracle_mod, lds_sort):
    """Straight walls parallel to a bin's direction put thousands of values into ONE bucket: the
    workgroup-per-bucket sorts (LDS bitonic, global rank sort) and the LDS bucket histograms of the
    projection, against the oracle; per-bin fan kernel and per-ray kernel."""
    occ = np.zeros((60, 2600), np.uint8)
    occ[10, 5:2590] = 1                                   # 2585 edge cells in one bucket of bin 0
    occ[30:33, 100:2000] = 1                              # 3 thick
    occ[5:55, 1300] = 1                                   # a vertical wall (bin theta_disc/4)
    for i in range(50):                                   # a diagonal
        occ[5 + i, 2100 + i] = 1
    g = maps.GridMap(occ, 0.05, (-3.0, 1.0, 0.0), "walls")
    mrx = 400
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    poses = maps.sample_free_poses(g, 40, 3)
    for td in (16, 112):
        m = range_libc.PyCDDTCast(omap, mrx, td)
        m.set_option("cddt_lds_sort", lds_sort)
        want = om.cddt_fan(td, poses, 4.71, 1081)
        for bins in (1, 0):
            m.set_option("cddt_bins", bins)
            out = np.empty(len(poses) * 1081, np.float32)
            m.calc_range_fan(poses, out, 4.71, 1081)
            assert np.array_equal(out, want), (td, bins)
        m.set_option("cddt_bins", 1)
        m.set_option("cddt_theta_min", 1)                 # theta-major (buckets beyond 1024 values: several separator lines)
        for search in (1, 0, 2):
            m.set_option("cddt_search", search)
            out = np.empty(len(poses) * 1081, np.float32)
            m.calc_range_fan(poses, out, 4.71, 1081)
            assert m.last_plan()["kernel"] == "cddt_theta" and np.array_equal(out, want), (td, search)
        m.set_option("cddt_search", 1)
        m.set_option("cddt_theta_min", 32768)
        # a rebuild (map update) goes through the same enqueue-only path again
        occ2 = occ.copy()
        occ2[40, 50:2500] = 1
        omap.update(occ2)
        om2 = oracle_mod.OracleMap(occ2, g.resolution, g.origin, mrx)
        out = np.empty(len(poses) * 1081, np.float32)
        m.calc_range_fan(poses, out, 4.71, 1081)
        assert np.array_equal(out, om2.cddt_fan(td, poses, 4.71, 1081)), td
        omap.update(occ)


def test_cfg3_giant_lut_full_size(oracle_mod):
    """configs[2]: 2000^2 maze, theta_disc 1442 (11.5 GB table), 65536 poses x 1081 beams."""
    w = workloads.cfg3()
    g, B, mrx, td = w.gmap, w.num_rays, w.max_range_px, w.theta_disc
    omap = range_libc.PyOMap(g)
    m = range_libc.PyGiantLUTCast(omap, mrx, td)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    # the device EDT at FULL size against the oracle's own EDT (Felzenszwalb in C, ~1 s at 4096^2): nothing of the
    # device is fed to the checker
    assert np.array_equal(omap.distance_transform(), om.dt), "device EDT differs from the oracle's at full size"
    # table slab vs the oracle (2 rows = 5.8 M entries), bit-exact
    slab = om.lut_build(td, 1000, 1002, nthreads=oracle_mod.max_threads())
    assert np.array_equal(m.table(1000, 1002), slab)
    poses = workloads.make_poses(w, dt=om.dt)
    out = np.empty(len(poses) * B, np.float32)
    m.calc_range_fan(poses, out, w.fov, B)
    assert out.min() >= 0 and out.max() <= mrx * g.resolution + 1e-5
    again = np.empty_like(out)
    m.calc_range_fan(poses, again, w.fov, B)
    assert np.array_equal(out, again)
    # the production kernel (theta_disc 1442 -> lut_fan_lds_kernel<3,17>, trimmed row fetch) bit for
    # bit against the oracle's fan query on a pose subsample: the oracle reads the DEVICE table rows
    # of the sampled poses' cells (the table itself is pinned by the slab above; 11.5 GB do not fit
    # the host), so every bin a beam reads must be the bin the oracle reads
    sub = np.concatenate([np.arange(0, len(poses), 683), [len(poses) - 1]])
    rr, cc = om.lut_pose_cells(poses[sub])
    assert (rr >= 0).all()
    row_cache = {}
    pose_rows = np.empty((len(sub), td), np.uint16)
    for i, (r_, c_) in enumerate(zip(rr, cc)):
        if int(r_) not in row_cache:
            row_cache[int(r_)] = m.table(int(r_), int(r_) + 1)[0]
        pose_rows[i] = row_cache[int(r_)][int(c_)]
    pick = (sub[:, None] * B + np.arange(B)[None, :]).ravel()
    want = om.lut_fan_rows(pose_rows, poses[sub], w.fov, B)
    assert np.array_equal(out[pick], want), int((out[pick] != want).sum())
    # ... also with the whole-row fetch (fov >= 2pi) and a negative fov, same poses
    for fov2, B2 in ((2 * math.pi, 1081), (-4.71, 1081), (SCAN_FOV_720, 720)):
        o2 = np.empty(len(sub) * B2, np.float32)
        m.calc_range_fan(poses[sub], o2, fov2, B2)
        assert np.array_equal(o2, om.lut_fan_rows(pose_rows, poses[sub], fov2, B2)), (fov2, B2)
    # against exact ray marching on a pose subsample: within 2 cells for nearly all beams
    sub = np.arange(0, len(poses), 1024)
    rm, _, _ = om.rm_fan(poses[sub], w.fov, B, nthreads=oracle_mod.max_threads())
    pick = (sub[:, None] * B + np.arange(B)[None, :]).ravel()
    err = np.abs(out[pick] - np.minimum(rm, mrx * g.resolution)) / g.resolution
    assert np.median(err) < 1.0 and (err < 2.0).mean() > 0.9
    # ... as the distribution SURVEY section 8(c)'s one-cell tolerance means (thresholds derived in _exact_rm_gate)
    _exact_rm_gate(err, td, float(np.minimum(rm, mrx * g.resolution).mean() / g.resolution), 0.71, "GiantLUT theta_disc %d, cfg3" % td)


# ---------------------------------------------------------------- configs 4 and 5: one GPU's shard
def _free_poses(g, dt, n, seed):
    return maps.sample_free_poses(g, n, seed, 2.0, dt)


def test_cfg4_colombia_rollout_shard_properties(oracle_mod):
    """configs[3]: maps/colombia, 2^20 MCTS roll-out poses (5243 roll-outs x 200 steps through
    rl_car_rollout with the scripts/mcts.py:214-231 action schedule) sharded 8 ways ->
    this is ONE rank's block (131072 poses x 1081 beams = 141.7 M rays), checked against the oracle on
    a subsample, through properties, and through the sharding contract (halves == whole)."""
    w = workloads.cfg4()
    assert w.pose_kind == "rollout"
    g, B, mrx = w.gmap, w.num_rays, w.max_range_px
    omap = range_libc.PyOMap(g)
    dt = omap.distance_transform()
    all_poses = workloads.make_poses(w, dt=dt)
    assert all_poses.shape == (1 << 20, 3) and np.isfinite(all_poses).all()
    # roll-out structure: consecutive poses of one roll-out are one 0.01 s step apart (<= 7 m/s)
    ro = all_poses[:200 * 5242].reshape(5242, 200, 3)
    hop = np.hypot(np.diff(ro[:, :, 0], axis=1), np.diff(ro[:, :, 1], axis=1))
    assert hop.max() <= 7.0 * 0.01 * 1.05 and hop.mean() > 0.005
    # ... and the generator is pinned: the same inputs through the reference-pinned host API
    st, ac = workloads.rollout_inputs(w, 5243, w.pose_seed, dt)
    from pyracecarsimulator_amd import racecar as RC
    again = RC.CarBatch().rollout(st[:64], ac[:64], 200, 10, 0.01)[0].reshape(-1, 3)
    assert np.array_equal(again, all_poses[:64 * 200])
    lo, hi = workloads.shard_range(1 << 20, 3, 8)              # rank 3 of 8
    assert hi - lo == 131072
    poses = np.ascontiguousarray(all_poses[lo:hi])
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    out = np.empty(len(poses) * B, np.float32)
    steps = np.empty(len(poses) * B, np.uint16)
    m.calc_range_fan(poses, out, w.fov, B, steps=steps)
    assert out.min() >= 0.0 and out.max() <= (mrx + 1.5) * g.resolution
    # oracle on every 512th pose, bit-exact (ranges and sample counts)
    sub = np.arange(0, len(poses), 512)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    r0, _, s0 = om.rm_fan(poses[sub], w.fov, B, step_coeff=1.0, nthreads=8)
    pick = (sub[:, None] * B + np.arange(B)[None, :]).ravel()
    assert np.array_equal(out[pick], r0) and np.array_equal(steps[pick], s0)
    # how the clustered roll-out poses differ from uniform free-cell poses (reported, not asserted on)
    uni = _free_poses(g, dt, 4096, 1003)
    su = np.empty(len(uni) * B, np.uint16)
    m.calc_range_fan(uni, np.empty(len(uni) * B, np.float32), w.fov, B, steps=su)
    print("cfg4 samples/ray: roll-out poses %.3f (p99 %d), uniform free-cell poses %.3f; poses outside the "
          "map or inside walls: %.2f %%" % (steps.mean(), np.percentile(steps[::97], 99), su.mean(),
                                            100.0 * (steps.reshape(-1, B).max(axis=1) <= 1).mean()))
    # two half-blocks reproduce the block (what all-gather of shards relies on)
    half = len(poses) // 2
    a, b = np.empty(half * B, np.float32), np.empty((len(poses) - half) * B, np.float32)
    m.calc_range_fan(poses[:half], a, w.fov, B)
    m.calc_range_fan(poses[half:], b, w.fov, B)
    assert np.array_equal(out[:half * B], a) and np.array_equal(out[half * B:], b)
    # fused crash test: whole block == isCrashed over the ranges, and per 200-pose roll-out
    edge = oracle_mod.edge_distances(B, -w.fov / 2, w.fov / B, 0.275, 0.2032, 0.3302)
    code = m.check_collision_many(poses, w.fov, B, edge, 0.001)
    assert code == oracle_mod.is_crashed(out, B, len(poses), edge, 0.001)
    n_ro = 640
    first = m.check_collision_groups(poses[:n_ro * 200], 200, w.fov, B, edge, 0.001)
    exp = [oracle_mod.is_crashed(out[k * 200 * B:(k + 1) * 200 * B], B, 200, edge, 0.001) for k in range(n_ro)]
    assert first.tolist() == exp
    assert sum(e >= 0 for e in exp) > 0 and sum(e < 0 for e in exp) > 0     # both outcomes occur


def test_cfg5_noise_shard_reproduces_unsharded(oracle_mod):
    """configs[4]: 4096^2 maze, 262144 poses x 720 beams + Gaussian noise, pose batch sharded 8 ways.
    One rank's TRUE shard (32768 poses) against the oracle on a subsample; the FULL batch through
    size-independent properties: a rank that scans its block with ray_offset = first global ray id
    reproduces the unsharded noisy scan bit for bit, noise has the configured sigma, noise-free ranges
    are idempotent."""
    w = workloads.cfg5()
    g, B, mrx = w.gmap, w.num_rays, w.max_range_px
    omap = range_libc.PyOMap(g)
    dt = omap.distance_transform()
    n = w.n_poses
    assert n == 262144
    poses = workloads.make_poses(w, dt=dt)
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    clean = np.empty(n * B, np.float32)
    m.calc_range_fan(poses, clean, w.fov, B)
    assert clean.min() >= 0.0 and clean.max() <= (mrx + 1.5) * g.resolution
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    # the 4096^2 device EDT against the oracle's own (32-bit d^2, the column pass's tiling at 4096 rows)
    assert np.array_equal(dt, om.dt), "device EDT differs from the oracle's at 4096^2"
    # rank 5's shard, every 128th pose of it, against the oracle
    lo5, hi5 = workloads.shard_range(n, 5, 8)
    assert hi5 - lo5 == 32768
    sub = np.arange(lo5, hi5, 128)
    r0, _, _ = om.rm_fan(poses[sub], w.fov, B, step_coeff=1.0, nthreads=8)
    pick = (sub[:, None] * B + np.arange(B)[None, :]).ravel()
    assert np.array_equal(clean[pick], r0)
    shard = np.empty((hi5 - lo5) * B, np.float32)
    m.calc_range_fan(poses[lo5:hi5], shard, w.fov, B)
    assert np.array_equal(shard, clean[lo5 * B:hi5 * B])
    # noise on the full batch
    m.set_noise(w.noise_std, w.noise_seed, 0)
    whole = np.empty(n * B, np.float32)
    m.calc_range_fan(poses, whole, w.fov, B)
    d = (whole[::7] - clean[::7]).astype(np.float64)
    assert abs(d.mean()) < 1e-4 and abs(d.std() - w.noise_std) < 1e-4
    # every rank's block with its own ray_offset reproduces its part of the unsharded scan
    for r in range(8):
        lo, hi = workloads.shard_range(n, r, 8)
        m.set_noise(w.noise_std, w.noise_seed, lo * B)
        part = np.empty((hi - lo) * B, np.float32)
        m.calc_range_fan(poses[lo:hi], part, w.fov, B)
        assert np.array_equal(part, whole[lo * B:hi * B]), r


# ---------------------------------------------------------------- "next" rows: roll-outs + crash
def test_rollout_generator_matches_reference_car():
    """GOLD-D2: 48 roll-outs x 200 steps integrated by the reference's compiled Car (oracle/_ref)."""
    import os
    from conftest import GOLD
    from pyracecarsimulator_amd import racecar as RC
    z = np.load(os.path.join(GOLD, "car_rollouts_ref.npz"))
    cars = RC.CarBatch(dict(zip(RC.CAR_PARAM_ORDER, z["params"])))
    poses, final, vel = cars.rollout(z["states"], z["actions"], int(z["n_steps"]),
                                     int(z["action_every"]), float(z["dt"]))
    # float64 ODE, libm vs OCML trig: <= 1e-9 relative, not bitwise
    assert np.allclose(final, z["final"], rtol=1e-9, atol=1e-9)
    assert np.allclose(vel, z["velocities"], rtol=1e-9, atol=1e-9)
    assert np.abs(poses.astype(np.float64) - z["poses"].astype(np.float64)).max() < 2e-6
    assert (poses == z["poses"]).mean() > 0.99
    # the survey's probe: 200 x control(2.0, 0.1) from rest
    p, f, _ = cars.rollout(np.zeros((1, 11)), np.tile([2.0, 0.1], (1, 20, 1)))
    assert np.allclose(f[0, :4], [2.080581685, 0.773390818, 0.665095230, 1.675438380], atol=2e-9)
    assert f[0, 10] == 200 and f[0, 7] == 1.0


def test_rollout_check_chain_equals_staged_oracle(oracle_mod):
    """roll-outs -> poses -> scan -> per-roll-out crash index in ONE call == the same stages done
    separately with the oracle scanning the generated poses."""
    from pyracecarsimulator_amd import racecar as RC
    g = maps.load_colombia()
    mrx, B, fov = 300, 1081, 4.71
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    rng = np.random.default_rng(9)
    R, n_steps = 24, 200
    start = maps.sample_free_poses(g, R, 5, 6.0, om.dt)
    states = np.zeros((R, 11))
    states[:, :3] = start
    states[:, 3] = rng.uniform(0, 3, R)
    actions = np.stack([rng.uniform(0, 7, (R, 20)), rng.uniform(-0.4189, 0.4189, (R, 20))], -1)
    cars = RC.CarBatch()
    edge = RC.edge_distances(B, -fov / 2, fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    first, final, vel = cars.rollout_check(m, states, actions, fov, B, edge, 0.001)
    poses, final2, vel2 = cars.rollout(states, actions)
    assert np.array_equal(final, final2) and np.array_equal(vel, vel2)
    want_r, _, _ = om.rm_fan(poses.reshape(-1, 3), fov, B, step_coeff=1.0, nthreads=8)
    want = [oracle_mod.is_crashed(want_r[r * n_steps * B:(r + 1) * n_steps * B], B, n_steps, edge, 0.001)
            for r in range(R)]
    assert first.tolist() == want
    assert any(w >= 0 for w in want) and any(w < 0 for w in want)
    # the grouped test on its own, for a method without a fused path (CDDT) and for Bresenham
    for cls, args, ofun in ((range_libc.PyCDDTCast, (112,), lambda p: om.cddt_fan(112, p, fov, B)),
                            (range_libc.PyBresenhamsLine, (), lambda p: om.bl_fan(p, fov, B)[0])):
        mm = cls(omap, mrx, *args)
        sub = poses[:6].reshape(-1, 3)
        got = mm.check_collision_groups(sub, n_steps, fov, B, edge, 0.001)
        rr = ofun(sub)
        want = [oracle_mod.is_crashed(rr[r * n_steps * B:(r + 1) * n_steps * B], B, n_steps, edge, 0.001)
                for r in range(6)]
        assert got.tolist() == want
        assert mm.check_collision_many(sub[:n_steps], fov, B, edge, 0.001) == want[0]


def test_grouped_crash_device_api_fused_and_generic(oracle_mod):
    torch = pytest.importorskip("torch")
    from pyracecarsimulator_amd import racecar as RC
    g = maps.load_colombia()
    mrx, B, fov, group = 300, 1081, 4.71, 50
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    poses = np.concatenate([maps.sample_free_poses(g, 200, 12, 9.0, om.dt),      # far from walls
                            maps.sample_free_poses(g, 200, 13, 2.0, om.dt)])     # some too close
    edge = RC.edge_distances(B, -fov / 2, fov / B, 0.275, 0.2032, 0.3302)
    want_r, _, _ = om.rm_fan(poses, fov, B, step_coeff=1.0, nthreads=8)
    want = [oracle_mod.is_crashed(want_r[k * group * B:(k + 1) * group * B], B, group, edge, 0.001)
            for k in range(8)]
    d_poses = torch.from_numpy(poses).cuda()
    d_edge = torch.from_numpy(edge).cuda()
    d_first = torch.zeros(8, dtype=torch.int32, device="cuda")
    d_ranges = torch.zeros(400 * B, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    m = range_libc.PyRayMarchingGPU(omap, mrx)
    # default schedule; two rays per lane draining in place; two rays per lane handing the last rays of a dry wave
    # (and their crash test) to rm_leftover_kernel
    for opts in ({}, {"slots": 2}, {"slots": 2, "handoff": 1, "handoff_cap": 8}, {"slots": 2, "handoff": 1, "handoff_cap": 64}):
        for k, v in opts.items():
            m.set_option(k, v)
        for ranges_ptr in (d_ranges.data_ptr(), 0):              # with and without storing the ranges
            d_first.zero_()
            d_ranges.zero_()
            m.check_collision_groups_device(d_poses.data_ptr(), 8, group, fov, B, d_edge.data_ptr(), 0.001,
                                            d_first.data_ptr(), ranges_ptr, stream=st)
            torch.cuda.synchronize()
            assert d_first.cpu().tolist() == want, opts
            if ranges_ptr:
                assert np.array_equal(d_ranges.cpu().numpy(), want_r), opts
    m.set_option("slots", 0)
    m.set_option("handoff", 0)
    m.check_collision_groups_device(d_poses.data_ptr(), 8, group, fov, B, d_edge.data_ptr(), 0.001,
                                    d_first.data_ptr(), d_ranges.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert any(w >= 0 for w in want) and any(w < 0 for w in want)
    assert m.check_collision_groups(poses, group, fov, B, edge, 0.001).tolist() == want


@pytest.mark.parametrize("variant", [0, 1])
def test_rays_leaving_an_open_map_count_no_border_sample(oracle_mod, variant):
    """Map without border walls (found by tests/gpu_fuzz.py, seed 35492827): beams that leave the
    map must report the oracle's sample count — the border read of the padded EDT is not a sample."""
    g = maps.GridMap(maps.make_maze(212, cell=20, wall=2, p=0.5, seed=3).occ[:, :155].copy()[5:-5, 5:],
                     0.05, (31.0, 17.0, 0.0), "open")
    om = oracle_mod.OracleMap.from_gridmap(g, 120)
    omap = range_libc.PyOMap(g)
    poses = maps.sample_free_poses(g, 64, 1)
    for cls, sc in ((range_libc.PyRayMarching, 0.999), (range_libc.PyRayMarchingGPU, 1.0)):
        m = cls(omap, 120)
        m.set_option("variant", variant)
        for B, fov in ((2, -2.0), (1081, 4.71)):
            r, h, s = _fan(m, poses, fov, B)
            r0, h0, s0 = om.rm_fan(poses, fov, B, step_coeff=sc)
            assert (h0[:, 0] < 0).any()                          # some beams do leave the map
            assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s, s0)


def test_randomised_parity_fuzz_short():
    """A few seconds of tests/gpu_fuzz.py (random map shapes, origins, yaw, ranges, fans, batch sizes,
    every method and kernel variant, crash tests, map updates) — all bit-identical to the oracle.
    Longer runs: ``python tests/gpu_fuzz.py --seconds 300 --seed N`` (4000+ cases clean in round 1)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gpu_fuzz", os.path.join(ROOT, "tests", "gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(8.0, 2026) > 10


def _ref_config():
    from pyracecarsimulator_amd import racecar as RC
    cfg = dict(RC.DEFAULT_CAR)
    cfg.update(scan_dist_to_base=0.275, batch_size=40, scan_beams=1080, scan_fov=4.71, scan_std=0.01,
               scan_max_range=15.0, free_thresh=0.8)          # params.yaml:28-39,44-47
    return cfg


def test_racecar_simulator_facade_drives_like_the_reference(oracle_mod):
    """RacecarSimulator (scripts/racecar_simulator_v2.py) end to end: drive/updatePose against the
    reference's compiled Car (oracle/_ref, when present) and runScan/checkCollision[Many] against the
    oracle scan + isCrashed."""
    import ctypes as C
    import os
    from conftest import ROOT
    from pyracecarsimulator_amd import RacecarSimulator, racecar as RC
    g = maps.load_colombia()
    cfg = _ref_config()
    sim = RacecarSimulator(cfg)
    omap = range_libc.PyOMap(g)
    sim.setMap(omap, g.resolution, g.origin)
    sim.setRaytracingMethod("RMGPU")
    mrx = int(15.0 / g.resolution)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    start = maps.sample_free_poses(g, 1, 21, 10.0, om.dt)[0]
    st = np.zeros(11)
    st[:3] = start
    sim.setState(st)
    ref = None
    so = os.path.join(ROOT, "oracle/_ref/libracecar_ref.so")
    if os.path.exists(so):
        L = C.CDLL(so)
        L.ref_car_create.restype = C.c_void_p
        L.ref_car_create.argtypes = [C.POINTER(C.c_double)]
        L.ref_car_control.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ref_car_update_position.argtypes = [C.c_void_p, C.c_double]
        L.ref_car_get_state.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.ref_car_set_state.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        ref = L.ref_car_create((C.c_double * 17)(*[cfg[k] for k in RC.CAR_PARAM_ORDER]))
        L.ref_car_set_state(ref, (C.c_double * 11)(*st))
    edge = oracle_mod.edge_distances(1080, -4.71 / 2, 4.71 / 1080, 0.275, cfg["width"], cfg["wb"])
    for i in range(30):
        sim.drive(2.0 + 0.1 * i, 0.3 * math.sin(i / 3.0))
        sim.updatePose()
        if ref is not None:
            L.ref_car_control(ref, 2.0 + 0.1 * i, 0.3 * math.sin(i / 3.0))
            L.ref_car_update_position(ref, 0.01)
            buf = (C.c_double * 11)()
            L.ref_car_get_state(ref, buf)
            assert np.allclose(sim.getState(), np.array(buf), rtol=1e-9, atol=1e-9)
        sim.runScan()
        pose = np.array([sim.getScanPose()], np.float32)
        want, _, _ = om.rm_fan(pose, 4.71, 1080, step_coeff=1.0)
        assert np.array_equal(sim.getScan(), want)
        assert sim.checkCollision() == oracle_mod.is_crashed(want, 1080, 1, edge, cfg["ttc_thresh"])
    assert sim.getTravelDistance() > 0 and sim.getMeanVelocity() > 0
    poses = maps.sample_free_poses(g, 45, 5, 2.0, om.dt)
    want, _, _ = om.rm_fan(poses[:40], 4.71, 1080, step_coeff=1.0)
    assert sim.checkCollisionMany(poses) == oracle_mod.is_crashed(want, 1080, 40, edge, cfg["ttc_thresh"])
    with pytest.raises(IndexError):
        sim.checkCollisionMany(poses[:10])
    # batched roll-outs through the façade
    states = np.tile(sim.getState(), (6, 1))
    acts = np.stack([np.full((6, 4), 3.0), np.linspace(-0.4, 0.4, 6)[:, None] * np.ones((6, 4))], -1)
    first, final, vel = sim.rolloutMany(states, acts, n_steps=40)
    assert first.shape == (6,) and final.shape == (6, 11) and vel.shape == (6, 40)
    sim.stop()
    assert not sim.getState().any()


# ---------------------------------------------------------------- FollowGap consumer
@pytest.mark.gpu
def test_followgap_kernel_reproduces_reference_build_vectors():
    """GOLD-E through the C ABI: one wave per scan, bit-identical to the reference's compiled header."""
    from pyracecarsimulator_amd.followgap import PyFollowGap
    z = np.load(os.path.join(GOLD, "followgap_ref.npz"))
    offs, prm = z["offsets"], z["params"]
    fg = PyFollowGap(int(prm[0]), float(prm[1]), float(prm[2]), float(prm[3]))
    for i in range(len(offs) - 1):
        v = np.ascontiguousarray(z["scans"][offs[i]:offs[i + 1]])
        got = np.float32(fg.eval(v, len(v)))
        assert got.tobytes() == np.float32(z["angles"][i]).tobytes(), (i, len(v), got, z["angles"][i])
    with pytest.raises(_lib.ScanLibError):
        fg.eval(np.ones(9, np.float32), 9)
    with pytest.raises(ValueError):
        fg.eval(np.ones(20, np.float64), 20)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [10, 11, 63, 64, 65, 128, 129, 720, 1081, 1088, 1089, 1217, 1280, 1281, 4097])
def test_followgap_batches_equal_the_oracle(oracle_mod, size):
    from pyracecarsimulator_amd.followgap import PyFollowGap
    rng = np.random.default_rng(size)
    n = 257
    scans = rng.uniform(0.0, 20.0, (n, size)).astype(np.float32)
    scans[rng.random((n, size)) < 0.15] = 0.0
    scans[::7] = np.minimum(scans[::7], 1.7)                     # no gap at all
    scans[1::7, : size // 2] = 1.0                               # one long gap on the right
    scans[2::7] = rng.normal(2.0, 2.0, (len(scans[2::7]), size)).astype(np.float32)   # negatives
    scans[3, :] = 3.0                                            # all equal
    scans[4, 0] = np.nan                                         # NaN seed: min_point stays 0
    scans[5, 3:9] = np.nan
    scans[6, :] = 0.0
    scans[8, -1], scans[8, :-1] = 5.0, 1.0                       # gap = the single last beam
    fg = PyFollowGap(10, 15.0, 0.4189, 0.004)
    got = fg.eval_many(scans)
    want = np.array([oracle_mod.followgap_eval(scans[i], 15.0, 0.4189, 0.004) for i in range(n)], np.float32)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), np.where(~same)[0][:10]
    flat = fg.eval_many(scans.reshape(-1), size)
    assert np.array_equal(flat.view(np.uint32), got.view(np.uint32))
    # structured scans for the one-bit-per-beam search (followgap_bits_kernel, size <= 1280): runs of every length (many per
    # lane chunk, runs over many chunks, runs that end at the last beam), ties between equal runs, the minimum — and with it
    # the safety bubble — at the ends of the scan, next to and inside the longest run, values on either side of 1.75 by one ulp
    n2 = 400
    s2 = np.empty((n2, size), np.float32)
    for k in range(n2):
        mean = [1.2, 2.0, 4.0, 9.0, 40.0, 300.0][k % 6]
        v, pos, hi = np.empty(size, np.float32), 0, bool(k & 1)
        while pos < size:
            ln = int(rng.geometric(1.0 / mean))
            v[pos:pos + ln] = rng.choice([1.7500001, 2.0, 9.0, 14.0]) if hi else rng.choice([1.75, 1.7499999, 1.0, 0.0])
            pos, hi = pos + ln, not hi
        if k % 5 == 1:                                           # two equal longest runs: the first wins
            ln = max(2, size // 5)
            v[:] = 1.0
            v[1:1 + ln] = 3.0
            v[size - 1 - ln:size - 1] = 3.0
        where = [0, 1, 4, 5, 6, size - 1, size - 2, size - 6, size // 2, int(rng.integers(size))][k % 10]
        if k % 3:
            v[where] = 0.25                                      # the unique minimum: the bubble covers where-5 ... where+4
        s2[k] = v
    # (a steering limit nothing reaches: every `best` beam gives its own angle)
    for blk in (s2, scans):
        got2 = PyFollowGap(10, 15.0, 1.0e6, 0.004).eval_many(blk)
        want2 = np.array([oracle_mod.followgap_eval(blk[i], 15.0, 1.0e6, 0.004) for i in range(len(blk))], np.float32)
        same2 = (got2.view(np.uint32) == want2.view(np.uint32)) | (np.isnan(got2) & np.isnan(want2))
        assert same2.all(), np.where(~same2)[0][:10]
    got2 = fg.eval_many(s2)
    want2 = np.array([oracle_mod.followgap_eval(s2[i], 15.0, 0.4189, 0.004) for i in range(n2)], np.float32)
    same2 = (got2.view(np.uint32) == want2.view(np.uint32)) | (np.isnan(got2) & np.isnan(want2))
    assert same2.all(), np.where(~same2)[0][:10]


@pytest.mark.gpu
def test_followgap_consumes_a_scanned_batch_on_the_device(oracle_mod):
    """scan -> steering without the ranges leaving HBM: calc_range_fan_device feeds
    rl_followgap_eval_device on the same stream."""
    import torch
    from pyracecarsimulator_amd.followgap import PyFollowGap
    g = maps.load_colombia()
    omap = range_libc.PyOMap(g)
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    poses = maps.sample_free_poses(g, 300, 11, dt=om.dt)
    m = range_libc.PyRayMarchingGPU(omap, 300)
    fg = PyFollowGap(10, 15.0, 0.4189, 0.004)
    d_poses = torch.from_numpy(poses).cuda()
    d_out = torch.empty(len(poses) * 1081, dtype=torch.float32, device="cuda")
    d_ang = torch.empty(len(poses), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    m.calc_range_fan_device(d_poses.data_ptr(), len(poses), 4.71, 1081, d_out.data_ptr(), stream=st)
    fg.eval_many_device(d_out.data_ptr(), len(poses), 1081, d_ang.data_ptr(), stream=st)
    torch.cuda.synchronize()
    r0, _, _ = om.rm_fan(poses, 4.71, 1081, step_coeff=1.0)
    r0 = r0.reshape(len(poses), 1081)
    want = np.array([oracle_mod.followgap_eval(r0[i], 15.0, 0.4189, 0.004) for i in range(len(poses))], np.float32)
    assert np.array_equal(d_ang.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_small_host_calls_zero_copy_equals_staged_path(oracle_mod):
    """scan()/scanMany()-sized host calls go through pinned, device-mapped memory (no staging
    copies); same bits as the staged path and as the oracle, for every method."""
    from pyracecarsimulator_amd import racecar as RC
    g = maps.load_colombia()
    omap = range_libc.PyOMap(g)
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    poses = maps.sample_free_poses(g, 50, 21, dt=om.dt)
    edge = RC.edge_distances(1081, -4.71 / 2, 4.71 / 1081, 0.275, 0.2032, 0.3302)
    for cls, args in ((range_libc.PyRayMarchingGPU, ()), (range_libc.PyRayMarching, ()),
                      (range_libc.PyBresenhamsLine, ()), (range_libc.PyCDDTCast, (108,))):
        m = cls(omap, 300, *args)
        outs = {}
        for mode in (65536, 0):
            m.set_option("pinned_max_rays", mode)
            for n in (1, 7, 50):
                r = np.full(n * 1081, -1.0, np.float32)
                m.calc_range_fan(poses[:n], r, 4.71, 1081)
                outs[(mode, n)] = r
                c = m.check_collision_many(poses[:n], 4.71, 1081, edge, 0.001)
                outs[(mode, n, "c")] = c
        for n in (1, 7, 50):
            assert np.array_equal(outs[(65536, n)], outs[(0, n)]), (cls.__name__, n)
            assert outs[(65536, n, "c")] == outs[(0, n, "c")]
        # the 2-argument per-ray form (scripts/two_player/scan.py:69-70), zero-copy vs staged
        ang = poses[0, 2] + np.linspace(-2.0, 2.0, 777, dtype=np.float32)
        ins = np.ascontiguousarray(np.stack([np.full(777, poses[0, 0]), np.full(777, poses[0, 1]), ang], 1),
                                   dtype=np.float32)
        two = {}
        for mode in (65536, 0):
            m.set_option("pinned_max_rays", mode)
            two[mode] = np.full(777, -1.0, np.float32)
            m.calc_range_many(ins, two[mode])
        assert np.array_equal(two[65536], two[0]) and (two[0] >= 0).all()
        if cls is range_libc.PyRayMarchingGPU:
            r0, _, _ = om.rm_fan(poses, 4.71, 1081, step_coeff=1.0)
            assert np.array_equal(outs[(65536, 50)], r0)


@pytest.mark.gpu
def test_pinned_result_vectors_are_written_directly(oracle_mod):
    """ScanSimulator2D keeps its cached result vectors in pinned blocks of the library (rl_host_alloc):
    the kernel writes the ranges straight into them.  Same results as the staged path, for calls below
    and above the zero-copy threshold, and the block outlives the simulator while an array refers to it."""
    import gc
    g = maps.load_colombia()
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    sim = ScanSimulator2D(1081, 4.71, 0.01, batch_size=300)
    assert sim.output_vector_many.base is not None and sim.output_vector.base is not None   # pinned blocks
    sim.setMap(omap, 300, g.resolution, g.origin)
    sim.setRaytracingMethod("RMGPU")
    poses = maps.sample_free_poses(g, 300, 12, dt=om.dt)
    want = om.rm_fan(poses, 4.71, 1081, step_coeff=1.0, nthreads=4)[0]
    out = sim.scanMany(poses)
    assert out is sim.output_vector_many and np.array_equal(out, want)        # 324 300 rays: above 262 144
    one = sim.scan(*poses[5])
    assert np.array_equal(one, want[5 * 1081:6 * 1081])
    # an ordinary array takes the staged path: same bits
    plain = np.empty_like(want)
    sim.scan_method.calc_range_fan(poses, plain, 4.71, 1081)
    assert np.array_equal(plain, want)
    # a larger pinned buffer used directly through the dense API, with the crash test riding along
    big = _lib.pinned_zeros(2000 * 1081, np.float32)
    p2 = maps.sample_free_poses(g, 2000, 13, dt=om.dt)
    w2 = om.rm_fan(p2, 4.71, 1081, step_coeff=1.0, nthreads=oracle_mod.max_threads())[0]
    sim.scan_method.calc_range_fan(p2, big, 4.71, 1081)
    assert np.array_equal(big, w2)
    edge = oracle_mod.edge_distances(1081, -4.71 / 2, 4.71 / 1081, 0.275, 0.2032, 0.3302)
    big[:] = 0
    code = sim.scan_method.check_collision_many(p2, 4.71, 1081, edge, 0.001, ranges=big)
    assert code == oracle_mod.is_crashed(w2, 1081, 2000, edge, 0.001) and np.array_equal(big, w2)
    keep = sim.output_vector_many
    del sim
    gc.collect()
    assert np.array_equal(keep, want)                      # the block lives as long as the array does


@pytest.mark.gpu
def test_big_host_calls_overlap_copy_and_march_in_pose_slices(oracle_mod):
    """A plain host-pointer scan of >= overlap_min_rays rays is cut into four pose slices, the device-to-host copy of
    one slice running on a second stream while the next slice marches: same bits as the unsliced call — pageable and
    pinned result buffers, pose counts that do not divide by four, and noise keyed by the GLOBAL ray id."""
    g = maps.load_colombia()
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    m = range_libc.PyRayMarchingGPU(omap, 300)
    assert m.get_info("overlap_min_rays") == 1 << 24
    for n in (4, 7, 301):
        poses = maps.sample_free_poses(g, n, 40 + n, dt=om.dt)
        want = om.rm_fan(poses, 4.71, 1081, step_coeff=1.0, nthreads=4)[0]
        for out in (np.zeros(n * 1081, np.float32), _lib.pinned_zeros(n * 1081, np.float32)):
            m.set_option("direct_max_rays", 0)             # a pinned block takes the DMA path too
            m.set_option("overlap_min_rays", 1)
            m.calc_range_fan(poses, out, 4.71, 1081)
            assert np.array_equal(out, want), n
            m.set_option("overlap_min_rays", 0)
            out[:] = 0
            m.calc_range_fan(poses, out, 4.71, 1081)
            assert np.array_equal(out, want), n
    poses = maps.sample_free_poses(g, 301, 77, dt=om.dt)
    m.set_noise(0.01, 6)
    a, b = np.zeros(301 * 1081, np.float32), np.zeros(301 * 1081, np.float32)
    m.set_option("overlap_min_rays", 1)
    m.calc_range_fan(poses, a, 4.71, 1081)
    m.set_option("overlap_min_rays", 0)
    m.calc_range_fan(poses, b, 4.71, 1081)
    assert np.array_equal(a, b) and a.std() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["one_row", "two_rows", "column", "uniform", "few"])
def test_stripe_bands_partition_clustered_poses(oracle_mod, layout):
    """Stripe mode ranks poses by (row bin, index) and cuts the ranks into equal bands; poses piled
    into one or two row bins put every cut inside a bin (ordered boundary counts)."""
    g = maps.make_maze(640, cell=40, wall=3, p=0.45, seed=5)
    om = oracle_mod.OracleMap.from_gridmap(g, 200)
    omap = range_libc.PyOMap(g)
    rng = np.random.default_rng(8)
    n = {"few": 70}.get(layout, 1500)
    free = maps.sample_free_poses(g, 4000, 3, dt=om.dt)
    gy = (free[:, 1] - g.origin[1]) / g.resolution
    gx = (free[:, 0] - g.origin[0]) / g.resolution
    if layout == "one_row":
        k = np.argsort(np.abs(gy - 321.0))[:n]
    elif layout == "two_rows":
        k = np.concatenate([np.argsort(np.abs(gy - 100.0))[: n // 3], np.argsort(np.abs(gy - 500.0))[: n - n // 3]])
    elif layout == "column":
        k = np.argsort(np.abs(gx - 200.0))[:n]
    else:
        k = rng.permutation(len(free))[:n]
    poses = free[k].copy()
    rng.shuffle(poses)
    poses[3] = [np.nan, 0.0, 0.0]
    poses[11] = [1e7, -1e7, 0.3]
    m = range_libc.PyRayMarchingGPU(omap, 200)
    for k_, v in {"inline_map_kb": 0, "inline_max": 0}.items():
        m.set_option(k_, v)
    B = 361
    r, h, s = _fan(m, poses, 4.0, B)
    r0, h0, s0 = om.rm_fan(poses, 4.0, B, step_coeff=1.0, nthreads=4)
    assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(s, s0)


@pytest.mark.gpu
def test_two_player_front_end_rebuilds_in_place(oracle_mod):
    """scripts/two_player/scan.py driven like rcs_two_player.py:110-124: build(map) + scan(pose)
    every tick with the other car stamped into the grid; same ranges as the oracle's CDDT on each
    tick's map, device objects reused."""
    from pyracecarsimulator_amd.two_player import ScanSimulator2D as TwoPlayerScan
    g = maps.load_colombia()
    om0 = oracle_mod.OracleMap.from_gridmap(g, 300)
    pose = maps.sample_free_poses(g, 1, 77, dt=om0.dt)[0]
    B, fov, td = 1080, 4.71, 112
    sim = TwoPlayerScan(B, fov, 0.01)
    first_method = None
    for tick in range(4):
        occ = g.occ.copy()
        occ[40 + 7 * tick: 48 + 7 * tick, 200:212] = 1                   # the other car's outline moves
        sim.build(maps.GridMap(occ, g.resolution, g.origin, name="tick"), 300, td)
        first_method = first_method or sim.scan_method
        assert sim.scan_method is first_method                           # no new device objects
        out = sim.scan(float(pose[0]), float(pose[1]), float(pose[2]))
        assert out is sim.output_vector
        om = oracle_mod.OracleMap(occ, g.resolution, g.origin, 300)
        want = om.cddt_rays(td, sim.input_vector)
        assert np.array_equal(out, want), tick
        assert np.array_equal(sim.omap.occ != 0, occ != 0) and np.array_equal(sim.omap.distance_transform(), om.dt), tick
    sim.build(maps.GridMap(g.occ, g.resolution, g.origin, name="base"), 300, td)          # the original map: the stamps' base
    # the caller has the outline cells at hand (rcs_two_player.py:110-116: x * map_width + y, indices past the grid skipped)
    cells = np.array([50 * g.cols + 100, 51 * g.cols + 100, 52 * g.cols + 101, g.rows * g.cols + 5, -3, 2**31 - 1], np.int64)
    sim.build_with_outline(cells)
    occ = g.occ.copy()
    occ.reshape(-1)[cells[:3]] = 1
    om = oracle_mod.OracleMap(occ, g.resolution, g.origin, 300)
    assert np.array_equal(sim.omap.distance_transform(), om.dt)
    out = sim.scan(float(pose[0]), float(pose[1]), float(pose[2]))
    assert np.array_equal(out, om.cddt_rays(td, sim.input_vector))
    sim.build_with_outline([])                                            # no outline: the original map again
    assert np.array_equal(sim.omap.distance_transform(), om0.dt) and np.array_equal(sim.omap.occ != 0, g.occ != 0)
    # a new grid through build() (rl_map_update) is the new base of later stamps
    occ = g.occ.copy()
    occ[100:104, 100:140] = 0
    occ[60:64, 210:214] = 1
    sim.build(maps.GridMap(occ, g.resolution, g.origin, name="edited"), 300, td)
    assert sim.scan_method is first_method
    om = oracle_mod.OracleMap(occ, g.resolution, g.origin, 300)
    assert np.array_equal(sim.omap.distance_transform(), om.dt)
    occ2 = occ.copy()
    occ2[70:73, 220:226] = 1
    sim.build_with_outline(np.flatnonzero((occ2 != 0).reshape(-1) & (occ == 0).reshape(-1)))
    om2 = oracle_mod.OracleMap(occ2, g.resolution, g.origin, 300)
    assert np.array_equal(sim.omap.distance_transform(), om2.dt)
    out = sim.scan(float(pose[0]), float(pose[1]), float(pose[2]))
    assert np.array_equal(out, om2.cddt_rays(td, sim.input_vector))
    # ray marching on a stamped map (step map + code map follow the map's epoch), and a multi-device map
    m = range_libc.PyRayMarchingGPU(sim.omap, 300)
    got = np.empty(B, np.float32)
    m.calc_range_fan(pose[None, :], got, fov, B)
    assert np.array_equal(got, om2.rm_fan(pose[None, :], fov, B, step_coeff=1.0)[0])
    m.close()
    multi = range_libc.PyOMap(g, device=[0, 0])
    multi.stamp_cells(cells)
    occ = g.occ.copy()
    occ.reshape(-1)[cells[:3]] = 1
    assert np.array_equal(multi.distance_transform(), oracle_mod.OracleMap(occ, g.resolution, g.origin, 300).dt)
    multi.close()


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 2])              # 2: two rays per lane, crash test fused into that form too
@pytest.mark.parametrize("n", [200, 513, 3000])
def test_fused_crash_marks_poses_then_reduces(oracle_mod, n, slots):
    """Whole-batch and grouped crash index for batches on both sides of the direct / per-pose-mark
    switch (512 poses), many poses crashing at once; equal to isCrashed on the oracle's ranges."""
    from pyracecarsimulator_amd import racecar as RC
    g = maps.load_colombia()
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    omap = range_libc.PyOMap(g)
    poses = maps.sample_free_poses(g, n, 31, dt=om.dt)
    B, fov = 360, 4.71
    edge = RC.edge_distances(B, -fov / 2, fov / B, 0.275, 0.2032, 0.3302) + 0.25      # wide car: many crashes
    r0, _, _ = om.rm_fan(poses, fov, B, step_coeff=1.0, nthreads=4)
    m = range_libc.PyRayMarchingGPU(omap, 300)
    m.set_option("slots", slots)
    want = oracle_mod.is_crashed(r0, B, n, edge, 0.001)
    assert m.check_collision_many(poses, fov, B, edge, 0.001) == want
    kept = np.empty(n * B, np.float32)                     # ... with the ranges kept: identical to the oracle
    assert m.check_collision_many(poses, fov, B, edge, 0.001, ranges=kept) == want and np.array_equal(kept, r0)
    far = np.full(B, -100.0)                                                          # nothing crashes
    assert m.check_collision_many(poses, fov, B, far, 0.001) == -(n + 1)
    grp = next(k for k in (40, 27, 25, 19, 8, 3, 1) if n % k == 0)
    got = m.check_collision_groups(poses, grp, fov, B, edge, 0.001)
    exp = [oracle_mod.is_crashed(r0[k * grp * B:(k + 1) * grp * B], B, grp, edge, 0.001) for k in range(n // grp)]
    assert got.tolist() == exp


# ---------------------------------------------------------------- K1b drain phase: value speculation on the step
@pytest.mark.parametrize("coeff_cls", ["RM", "RMGPU"])
def test_speculating_drain_loop_on_long_chains_bit_equal_to_oracle(oracle_mod, coeff_cls):
    """march_drain4 (one ray per lane, stream dry): rays grazing long straight walls (the step repeats: the
    loop consumes four samples per round trip), rays along a diagonal staircase wall (the step alternates:
    the first prediction fails, the loop must fall back to a plain stretch) and ordinary rays, with the loop
    off, on from 8 / 64 live lanes, stretches of 1 and 16 plain samples — ranges bit-equal to the oracle."""
    n = 640
    occ = np.zeros((n, n), np.uint8)
    occ[0, :] = occ[-1, :] = occ[:, 0] = occ[:, -1] = 1
    occ[100:103, 20:620] = 1                       # long horizontal walls
    occ[300:302, 40:600] = 1
    occ[120:600, 500:503] = 1                      # a long vertical wall
    for k in range(0, 260):                        # a diagonal staircase wall
        occ[330 + k, 60 + k] = 1
        occ[331 + k, 60 + k] = 1
    g = maps.GridMap(occ, 0.05, (-3.0, 1.0, 0.2), "walls")
    mrx = 300
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    omap = range_libc.PyOMap(g)
    assert np.array_equal(omap.distance_transform(), om.dt)
    rng = np.random.default_rng(5)
    P, B, fov = 700, 257, 0.35
    # grid poses a cell or two off the walls, heading almost along them (both directions), plus random ones
    gx = np.concatenate([rng.uniform(30, 600, 250), rng.uniform(496.0, 499.5, 150), 62 + rng.uniform(0, 250, 150),
                         rng.uniform(5, 630, 150)])
    gy = np.concatenate([np.where(rng.random(250) < 0.5, rng.uniform(103.5, 106.0, 250), rng.uniform(296.0, 299.5, 250)),
                         rng.uniform(130, 590, 150), np.zeros(150), rng.uniform(5, 630, 150)])
    gy[400:550] = 330 + (gx[400:550] - 60) + rng.uniform(3.0, 6.0, 150)        # just above the staircase
    th = np.concatenate([rng.choice([0.0, math.pi], 250) + rng.uniform(-0.03, 0.03, 250),
                         rng.choice([math.pi / 2, -math.pi / 2], 150) + rng.uniform(-0.03, 0.03, 150),
                         rng.choice([math.pi / 4, -3 * math.pi / 4], 150) + rng.uniform(-0.02, 0.02, 150),
                         rng.uniform(-math.pi, math.pi, 150)])
    c, s_ = math.cos(g.origin[2]), math.sin(g.origin[2])
    # every 9th pose outside the map, every 11th inside a wall: rays that are born finished / hit at once, claimed
    # next to the long chains (the compaction of the several-rays-per-lane drain must not lose their stores)
    gx[::9] = rng.uniform(-40.0, -2.0, len(gx[::9]))
    gy[::11] = 101.3
    gx[::11] = rng.uniform(30, 600, len(gx[::11]))
    poses = np.stack([g.origin[0] + (c * gx - s_ * gy) * g.resolution, g.origin[1] + (s_ * gx + c * gy) * g.resolution,
                      th + g.origin[2]], 1).astype(np.float32)
    coeff = 1.0 if coeff_cls == "RMGPU" else 0.999
    want, _, steps = om.rm_fan(poses, fov, B, step_coeff=coeff, nthreads=oracle_mod.max_threads())
    assert steps.max() >= 120 and (steps >= 60).sum() > 500          # the chains this test is about exist
    cls = range_libc.PyRayMarchingGPU if coeff_cls == "RMGPU" else range_libc.PyRayMarching
    m = cls(omap, mrx)
    m.set_option("variant", 1)             # (the drain loops of the canonical kernels, either step coefficient)
    m.set_option("slots", 1)
    out = np.empty(P * B, np.float32)
    for big_map_policy in (0, 1):
        m.set_option("inline_map_kb", 0 if big_map_policy else 2048)
        for sd, stretch, lw in ((0, 16, 12), (8, 16, 12), (8, 1, 12), (64, 16, 12), (64, 1, 0), (8, 4, 40), (3, 2, 12)):
            m.set_option("spec_drain", sd)
            m.set_option("spec_stretch", stretch)
            m.set_option("low_water", lw)
            out[:] = -1.0
            m.calc_range_fan(poses, out, fov, B)
            assert np.array_equal(out, want), (coeff_cls, big_map_policy, sd, stretch, lw, int((out != want).sum()))
    assert m.last_plan()["slots"] == 1 and m.get_info("spec_drain") == 3
    # several rays per lane: once at most drain_cap (<= 64) rays of a wave are live after the stream ran dry they
    # are compacted into slot A and finished by the same loops
    for slots in (2, 3):
        m.set_option("slots", slots)
        for big_map_policy in (0, 1):
            m.set_option("inline_map_kb", 0 if big_map_policy else 2048)
            for sd, cap, stretch, lw, gm in ((0, 64, 16, 12, 8), (8, 64, 8, 12, 8), (8, 24, 1, 12, 3), (8, 1, 16, 0, 1),
                                             (8, 48, 4, 30, 8), (8, 64, 1, 40, 3), (8, 7, 8, 12, 2)):
                for k_, v_ in (("spec_drain", sd), ("drain_cap", cap), ("drain_stretch", stretch), ("low_water", lw),
                               ("grid_mult", gm)):
                    m.set_option(k_, v_)
                out[:] = -1.0
                m.calc_range_fan(poses, out, fov, B)
                assert m.last_plan()["slots"] == slots
                assert np.array_equal(out, want), (coeff_cls, slots, big_map_policy, sd, cap, stretch, lw, gm,
                                                   int((out != want).sum()))
    m.set_option("drain_cap", 64)
    m.set_option("drain_stretch", 8)
    # ... and with the fused crash test riding along (two rays per lane keeps it)
    edge = oracle_mod.edge_distances(B, -fov / 2, fov / B, 0.275, 0.2032, 0.3302)
    m.set_option("slots", 2)
    m.set_option("spec_drain", 8)
    m.set_option("grid_mult", 8)
    m.set_option("low_water", 12)
    got = m.check_collision_groups(poses, 100, fov, B, edge, 0.001, ranges=out)
    exp = [oracle_mod.is_crashed(want[k * 100 * B:(k + 1) * 100 * B], B, 100, edge, 0.001) for k in range(P // 100)]
    assert got.tolist() == exp and np.array_equal(out, want)


def test_cfg3_cddt_full_size(oracle_mod):
    """configs[2] names "CDDT/LUT": the CDDT variant at full size — 2000^2 maze, 65536 poses x 1081 beams,
    theta_disc 108 (/root/reference/scripts/two_player/scan.py:46; 112 in rcs_two_player.py:121 is covered by
    the small-map tests) — bit-equal to the oracle's CDDT on every 1024th pose, deterministic, within range,
    and the same answer from the per-ray kernel and from the two-argument per-ray API."""
    w = workloads.cfg3()
    g, B, mrx = w.gmap, w.num_rays, w.max_range_px
    omap = range_libc.PyOMap(g)
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    # the device EDT at FULL size against the oracle's own EDT (Felzenszwalb in C, ~1 s at 4096^2): nothing of the
    # device is fed to the checker
    assert np.array_equal(omap.distance_transform(), om.dt), "device EDT differs from the oracle's at full size"
    m = range_libc.PyCDDTCast(omap, mrx, 108)
    poses = workloads.make_poses(w, dt=om.dt)
    assert len(poses) == 65536
    assert m.plan_fan(len(poses), B)["kernel"] == "cddt_theta"        # theta-major from 32 768 poses up
    out = np.empty(len(poses) * B, np.float32)
    m.calc_range_fan(poses, out, w.fov, B)
    assert m.last_plan()["kernel"] == "cddt_theta"
    assert out.min() >= 0.0 and out.max() <= mrx * g.resolution
    m.set_option("cddt_search", 0)                                    # round 4's search kernel on the same batch
    old = np.empty_like(out)
    m.calc_range_fan(poses, old, w.fov, B)
    assert m.last_plan()["name"] == "scan::cddt_theta_search_kernel" and np.array_equal(old, out)
    m.set_option("cddt_search", 2)                                    # search + fan fused per tile of 64 poses
    old[:] = -1.0
    m.calc_range_fan(poses, old, w.fov, B)
    assert m.last_plan()["name"] == "scan::cddt_theta_fused_kernel" and np.array_equal(old, out)
    del old
    m.set_option("cddt_search", 1)
    m.set_option("cddt_theta_min", 0)                                 # the pose-major kernel on the same batch
    pm = np.empty_like(out)
    m.calc_range_fan(poses, pm, w.fov, B)
    assert m.last_plan()["kernel"] == "cddt_bins" and np.array_equal(pm, out)
    del pm
    m.set_option("cddt_theta_min", 32768)
    again = np.empty_like(out)
    m.calc_range_fan(poses, again, w.fov, B)
    assert np.array_equal(out, again)
    sub = np.arange(0, len(poses), 1024)
    pick = (sub[:, None] * B + np.arange(B)[None, :]).ravel()
    want = om.cddt_fan(108, poses[sub], w.fov, B, nthreads=oracle_mod.max_threads())
    assert np.array_equal(out[pick], want), int((out[pick] != want).sum())
    # the per-ray kernel (one search per ray) on the subsample, and the upstream 2-argument form
    m.set_option("cddt_bins", 0)
    o2 = np.empty(len(sub) * B, np.float32)
    m.calc_range_fan(poses[sub], o2, w.fov, B)
    assert np.array_equal(o2, want)
    ins = np.zeros((B, 3), np.float32)
    ins[:, :2] = poses[sub[3], :2]
    ins[:, 2] = poses[sub[3], 2] + (np.float32(-0.5) * np.float32(w.fov) + np.arange(B, dtype=np.float32) * (np.float32(w.fov) / np.float32(B)))
    o3 = np.empty(B, np.float32)
    m.calc_range_many(ins, o3)
    assert np.array_equal(o3, om.cddt_rays(108, ins))
    # against exact ray marching: CDDT is approximate in angle only — the median error stays below two cells
    rm = om.rm_fan(poses[sub], w.fov, B, nthreads=oracle_mod.max_threads(), want_hits=False, want_steps=False)[0]
    err = np.abs(want - np.minimum(rm, mrx * g.resolution)) / g.resolution
    assert np.median(err) < 2.0
    # ... as the distribution SURVEY section 8(c)'s one-cell tolerance means (thresholds derived in _exact_rm_gate), at
    # 108, at the reference's own theta_disc 112 (scripts/two_player/rcs_two_player.py:121) and at the beam spacing
    rbar = float(np.minimum(rm, mrx * g.resolution).mean() / g.resolution)
    _exact_rm_gate(err, 108, rbar, 0.0, "CDDT theta_disc 108, cfg3")
    m.set_option("cddt_bins", 1)
    for td2 in (112, 1442):
        m2 = range_libc.PyCDDTCast(omap, mrx, td2)
        o4 = np.empty(len(sub) * B, np.float32)
        m2.calc_range_fan(poses[sub], o4, w.fov, B)
        assert np.array_equal(o4, om.cddt_fan(td2, poses[sub], w.fov, B, nthreads=oracle_mod.max_threads()))
        _exact_rm_gate(np.abs(o4 - np.minimum(rm, mrx * g.resolution)) / g.resolution, td2, rbar, 0.0,
                       "CDDT theta_disc %d, cfg3" % td2)
        m2.close()


def test_hbm_probe_reports_plausible_rates():
    import ctypes
    out = (ctypes.c_double * 5)()
    _lib.check(_lib.lib().rl_probe_hbm(0, 512 << 20, out))
    assert all(1000.0 < v < 8000.0 for v in out), list(out)
    out3 = (ctypes.c_double * 3)()
    _lib.check(_lib.lib().rl_probe_hbm_nt(0, 512 << 20, out3))
    assert all(1000.0 < v < 8000.0 for v in out3), list(out3)
