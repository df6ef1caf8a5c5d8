"""CPU tests of the oracle itself: known answers, the independent NumPy statement, and the
committed golden vectors.  PARITY UNPINNED upstream (range_libc absent) — these
pins are the build's own."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from conftest import GOLD, ROOT, load_golden
from oracle import np_statement as N
from oracle import oracle as O
from pyracecarsimulator_amd import maps


# ---------------------------------------------------------------- trig / EDT
def test_det_sincos_matches_numpy_statement_and_libm(oracle_mod):
    x = np.concatenate([np.random.default_rng(0).uniform(-50, 50, 4000),
                        [0.0, -0.0, 1e-30, -1e-30, math.pi / 2, math.pi, -math.pi, 2 * math.pi,
                         0.7853981633974483, 1e-8, 100.0, -1000.0]]).astype(np.float32)
    s, c = oracle_mod.sincosf(x)
    s2, c2 = N.sincosf(x)
    assert np.array_equal(s, s2) and np.array_equal(c, c2)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2.5e-7
    s0, c0 = oracle_mod.sincosf(np.float32(0.0))
    assert s0[0] == 0.0 and c0[0] == 1.0


def test_edt_closed_forms(oracle_mod):
    # KAT-5: single pixel and two pixels
    occ = np.zeros((17, 23), np.uint8)
    occ[5, 7] = 1
    d = oracle_mod.edt(occ)
    rr, cc = np.mgrid[0:17, 0:23]
    want = np.sqrt(((rr - 5) ** 2 + (cc - 7) ** 2).astype(np.float32))
    assert np.array_equal(d, want)
    occ[12, 20] = 1
    d = oracle_mod.edt(occ)
    want2 = np.minimum(want, np.sqrt(((rr - 12) ** 2 + (cc - 20) ** 2).astype(np.float32)))
    assert np.array_equal(d, want2)
    # empty map: Felzenszwalb INF -> 1e10
    assert np.all(oracle_mod.edt(np.zeros((4, 5), np.uint8)) == np.float32(1e10))


@pytest.mark.parametrize("seed,shape", [(1, (64, 64)), (2, (97, 131)), (3, (200, 50))])
def test_edt_vs_scipy(oracle_mod, seed, shape):
    rng = np.random.default_rng(seed)
    occ = (rng.random(shape) < 0.03).astype(np.uint8)
    assert np.array_equal(oracle_mod.edt(occ), N.edt(occ))
    d2 = oracle_mod.edt_sq(occ)
    assert np.array_equal(np.sqrt(d2.astype(np.float32)), oracle_mod.edt(occ))


def test_edt_colombia_vs_scipy(oracle_mod):
    g = maps.load_colombia()
    assert g.occ.shape == (350, 435)
    assert np.array_equal(oracle_mod.edt(g.occ), N.edt(g.occ))


# ---------------------------------------------------------------- known answers
def _room(oracle_mod, n=64, res=0.05, mrx=300):
    g = maps.make_room(n, wall=1, resolution=res)
    return g, oracle_mod.OracleMap.from_gridmap(g, mrx)


def test_kat1_axis_aligned_room(oracle_mod):
    g, om = _room(oracle_mod)
    # pose at a cell centre (col 20.5, row 30.5); 4 beams at 0, pi/2, pi, 3pi/2 via rays API
    xw, yw = 20.5 * 0.05, 30.5 * 0.05
    ins = np.array([[xw, yw, 0.0], [xw, yw, math.pi / 2], [xw, yw, math.pi],
                    [xw, yw, -math.pi / 2]], np.float32)
    r, h, s = om.rm_rays(ins, step_coeff=0.999)
    # +x: wall column 63: the march lands in cell (63,30); range = distance to its corner
    gx, gy = np.float32(xw) * np.float32(20.0), np.float32(yw) * np.float32(20.0)
    assert h[0].tolist() == [63, 30]
    assert r[0] == np.float32(np.sqrt(np.float32((63 - gx) ** 2 + (30 - gy) ** 2)) * np.float32(0.05))
    assert h[1].tolist()[1] == 63          # +y: top wall row
    assert h[2].tolist()[0] == 0           # -x: left wall col
    assert h[3].tolist()[1] == 0           # -y: bottom wall row
    assert np.all(s >= 2)


def test_kat2_pose_inside_occupied_cell(oracle_mod):
    g, om = _room(oracle_mod)
    ins = np.array([[0.3 * 0.05, 10.6 * 0.05, 0.4]], np.float32)   # inside the left wall
    r, h, s = om.rm_rays(ins)
    gx, gy = ins[0, 0] * np.float32(20.0), ins[0, 1] * np.float32(20.0)
    fx, fy = gx - np.float32(0), gy - np.float32(10)
    want = np.sqrt(np.float32(fx * fx + fy * fy)) * np.float32(0.05)
    assert h[0].tolist() == [0, 10] and s[0] == 1
    assert abs(float(r[0]) - float(want)) < 1e-7 and r[0] > 0      # RM: NOT zero
    rb, hb, sb = om.bl_rays(ins)
    assert rb[0] == 0.0 and hb[0].tolist() == [0, 10]             # Bresenham: zero


def test_kat3_ray_leaving_the_map_and_kat4_open_field(oracle_mod):
    occ = np.zeros((700, 700), np.uint8)
    occ[0, 0] = 1                                    # one far obstacle so the EDT is finite
    om = oracle_mod.OracleMap(occ, 0.05, (0.0, 0.0, 0.0), 300)
    # KAT-3: pose near the right edge looking out -> leaves the map -> max range
    r, h, s = om.rm_rays(np.array([[699.5 * 0.05, 350 * 0.05, 0.0]], np.float32))
    assert r[0] == np.float32(300.0) * np.float32(0.05) and h[0].tolist() == [-1, -1]
    # KAT-4: open field > 300 px in every direction -> t >= max_range
    r, h, s = om.rm_fan(np.array([[350 * 0.05, 350 * 0.05, 0.3]], np.float32), 6.28, 90)
    assert np.all(r == np.float32(15.0)) and np.all(h == -1)
    # outside the map entirely / NaN pose: deterministic miss
    r, h, s = om.rm_rays(np.array([[-5.0, -5.0, 0.0], [np.nan, 1.0, 0.0], [1e30, 0, 0]], np.float32))
    assert np.all(r == np.float32(15.0)) and np.all(s == 0)


def test_negative_coordinate_caveat_kept(oracle_mod):
    # trunc(-0.3) = 0: a pose at -1 < g < 0 is treated as in-map
    g, om = _room(oracle_mod)
    r, h, s = om.rm_rays(np.array([[-0.3 * 0.05, 5.5 * 0.05, 0.0]], np.float32))
    assert h[0].tolist() == [0, 5] and s[0] == 1


# ---------------------------------------------------------------- independent statement + golden
@pytest.mark.parametrize("seed", [11, 12])
def test_rm_c_oracle_equals_numpy_statement(oracle_mod, seed):
    g = maps.make_maze(160, cell=20, wall=2, p=0.5, seed=seed, resolution=0.1,
                       origin=(1.5, -2.0, 0.3 * (seed - 11)))
    om = oracle_mod.OracleMap.from_gridmap(g, 90)
    poses = maps.sample_free_poses(g, 12, seed, dt=om.dt)
    for sc in (0.999, 1.0):
        r, h, s = om.rm_fan(poses, 4.71, 271, step_coeff=sc)
        r2, h2, s2 = N.rm_fan(g.occ, g.resolution, g.origin, 90, poses, 4.71, 271, sc)
        assert np.array_equal(r, r2) and np.array_equal(h, h2) and np.array_equal(s, s2)


@pytest.mark.parametrize("name", ["rm_colombia", "rm_maze256", "rm_maze192_yaw"])
def test_oracle_reproduces_golden_vectors(oracle_mod, name):
    g, z = load_golden(name)
    om = oracle_mod.OracleMap.from_gridmap(g, int(z["max_range_px"]))
    fov, B = float(z["fov"]), int(z["num_rays"])
    for tag, sc in (("cpu", 0.999), ("gpu", 1.0)):
        r, h, s = om.rm_fan(z["poses"], fov, B, step_coeff=sc, nthreads=2)
        assert np.array_equal(r, z["ranges_" + tag])
        assert np.array_equal(h, z["hits_" + tag].astype(np.int32))
        assert np.array_equal(s, z["steps_" + tag])
    rb, hb, _ = om.bl_fan(z["poses"], fov, B)
    assert np.array_equal(rb, z["ranges_bl"]) and np.array_equal(hb, z["hits_bl"].astype(np.int32))


def test_fan_equals_per_ray_rows_up_to_trig_form(oracle_mod):
    """4-arg fan vs the 2-arg per-ray form (scripts/two_player/scan.py:57-66): same geometry, the
    direction comes from angle addition vs one sincos of the summed angle -> <= 1 cell apart on
    all but grazing rays."""
    g, z = load_golden("rm_maze256")
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    poses, B, fov = z["poses"][:6], 1081, 4.71
    r_fan, _, _ = om.rm_fan(poses, fov, B)
    th = (poses[:, 2:3] + (np.float32(-fov / 2) + np.arange(B, dtype=np.float32) *
                           np.float32(fov / B))[None, :]).astype(np.float32)
    ins = np.stack([np.repeat(poses[:, 0], B), np.repeat(poses[:, 1], B), th.ravel()], 1)
    r_rays, _, _ = om.rm_rays(ins)
    close = np.abs(r_fan - r_rays) <= g.resolution * 1.0001
    assert close.mean() > 0.995


def test_canonical_form_vs_upstream_literal_libm(oracle_mod):
    """north_star asks for bit-exact HIT CELLS against range_libc's CPU RayMarching.  range_libc is
    absent, so the one contact with its literal arithmetic is this: the canonical march
    (deterministic sincos, explicit fma, (col,row) order) against the upstream-literal statement
    (libm cosf/sinf of -theta+rot_const, calc_range(y,x,theta'), every product and sum a separate
    rounding).  Per-ray form (2-arg calc_range_many): hit cell AND sample count identical on every
    ray of every map; ranges differ only by the rounding of the origin arithmetic."""
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        om = oracle_mod.OracleMap.from_gridmap(g, int(z["max_range_px"]))
        rng = np.random.default_rng(5)
        base = z["poses"][rng.integers(0, len(z["poses"]), 4000)]
        ins = base.copy()
        ins[:, 2] = rng.uniform(-np.pi, np.pi, len(ins)).astype(np.float32)
        for sc in (0.999, 1.0):
            a, ha, sa = om.rm_rays(ins, step_coeff=sc)
            b, hb, sb = om.rm_rays_libm(ins, step_coeff=sc, full=True)
            mism = (ha != hb).any(axis=1)
            print("%s coeff %.3f per-ray form: hit-cell mismatches %d of %d, sample-count mismatches %d, "
                  "ranges bit-equal %.4f, max |d| %.2e cell" % (name, sc, mism.sum(), len(a), (sa != sb).sum(),
                                                               (a == b).mean(), np.abs(a - b).max() / g.resolution))
            assert not mism.any() and np.array_equal(sa, sb)
            assert np.abs(a - b).max() <= 1e-4 * g.resolution and (a == b).mean() > 0.7


# Measured on the golden inputs (hit cells that differ / rays; rays further than one cell apart):
#   rm_colombia    64 x 1081: coeff 0.999 1 / 69184 (0 beyond one cell), coeff 1.0 1 / 69184 (1: 1.37 cells)
#   rm_maze256     64 x 1081: coeff 0.999 0,                              coeff 1.0 2 / 69184 (0)
#   rm_maze192_yaw 16 x 360 : coeff 0.999 0,                              coeff 1.0 1 / 5760  (0)
# The literal form rounds theta + alpha_j to float32 before libm's trig, the canonical form rotates a
# per-beam (cos, sin) table by the pose heading: a ray grazing a corner can land on the neighbouring
# cell (or pass it).  The bounds below leave room for another libm build, not for a different march.
def test_canonical_fan_vs_upstream_literal_fan(oracle_mod):
    """The fork's 4-arg calc_range_many stated literally (one libm cast per beam at
    theta + (-fov/2 + j*fov/B), scripts/scan_simulator.py:103-106) against the canonical fan every
    kernel reproduces bit for bit: ranges within ONE cell on every ray (north_star's tolerance), hit
    cells equal on all but a counted handful of grazing rays."""
    worst = 0.0
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        om = oracle_mod.OracleMap.from_gridmap(g, int(z["max_range_px"]))
        poses, B, fov = z["poses"], int(z["num_rays"]), float(z["fov"])
        for sc in (0.999, 1.0):
            a, ha, sa = om.rm_fan(poses, fov, B, step_coeff=sc)
            b, hb, sb = om.rm_fan_libm(poses, fov, B, step_coeff=sc)
            mism = int((ha != hb).any(axis=1).sum())
            frac = mism / len(a)
            worst = max(worst, frac)
            print("%s coeff %.3f fan form: hit-cell mismatches %d of %d (%.2e), ranges bit-equal %.4f, "
                  "max |d| %.3f cell" % (name, sc, mism, len(a), frac, (a == b).mean(),
                                         np.abs(a - b).max() / g.resolution))
            beyond = int((np.abs(a - b) > g.resolution * 1.0001).sum())
            print("    rays further than one cell apart: %d" % beyond)
            assert mism <= 4 and frac <= 2e-4 if len(a) > 20000 else mism <= 2
            assert beyond <= 2 and beyond <= mism                         # only rays whose hit cell moved
            assert (sa != sb).mean() < 1e-3                               # sample counts: equal on > 99.9 %
    assert worst <= 2e-4


def test_committed_upstream_literal_vectors_match_this_hosts_libm(oracle_mod):
    """tests/golden/rm_libm_forms.npz (the upstream-literal libm form of GOLD-A/B, the vectors the GPU gate
    test_device_fan_vs_upstream_literal_libm_form compares the device with) against the same form recomputed on
    this host: hit cells and sample counts identical; ranges bit-equal unless this host's libm rounds a cosf /
    sinf differently in the last place (then: within 1e-3 cell).  And the canonical form stays inside the gate
    against the committed vectors."""
    import os
    from conftest import GOLD
    L = np.load(os.path.join(GOLD, "rm_libm_forms.npz"))
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        om = oracle_mod.OracleMap.from_gridmap(g, int(z["max_range_px"]))
        npz = int(L[name + "_n_poses"])
        poses, B, fov = z["poses"][:npz], int(z["num_rays"]), float(z["fov"])
        for tag, sc in (("cpu", 0.999), ("gpu", 1.0)):
            r, h, s_ = om.rm_fan_libm(poses, fov, B, step_coeff=sc)
            gr, gh = L["%s_ranges_%s" % (name, tag)], L["%s_hits_%s" % (name, tag)].astype(np.int32)
            moved = (h != gh).any(axis=1)
            assert moved.sum() <= 1, (name, tag, int(moved.sum()))
            assert np.abs(r - gr)[~moved].max() <= 1e-3 * g.resolution
            a, ha, _ = om.rm_fan(poses, fov, B, step_coeff=sc)
            mism = (ha != gh).any(axis=1)
            assert mism.sum() <= 2 and np.abs(a - gr).max() <= 1.5 * g.resolution


def test_committed_table_literal_vectors_match_this_hosts_libm(oracle_mod):
    """tests/golden/table_libm_forms.npz — CDDT (theta_disc 112, the reference's, and 360; the 2-argument per-ray form
    scripts/two_player/scan.py:57-70 uses) and GiantLUT (theta_disc 180) in the upstream-literal statement (libm cosf /
    sinf per bin, un-fused projection, fmod + roundf bin rule) — recomputed on this host: equal to the committed arrays
    (bit for bit unless this host's libm rounds a cosf / sinf differently: then within 1e-3 cell), and the CANONICAL
    statement (what the device builds bit for bit) stays inside the gate the GPU test applies to the device:
    every ray within 1e-3 cell of the literal form, GiantLUT codes identical."""
    import importlib.util
    import os
    from conftest import GOLD
    spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(GOLD, "make_fixtures.py"))
    L = np.load(os.path.join(GOLD, "table_libm_forms.npz"))
    # (table_ray_rows restated: importing make_fixtures would pull the reference-side generators in)
    def ray_rows(poses, fov, B):
        ang = (np.float32(-0.5) * np.float32(fov) + np.arange(B, dtype=np.float32) * (np.float32(fov) / np.float32(B))).astype(np.float32)
        ins = np.zeros((len(poses) * B, 3), np.float32)
        for q in range(len(poses)):
            ins[q * B:(q + 1) * B, :2] = poses[q, :2]
            ins[q * B:(q + 1) * B, 2] = poses[q, 2] + ang
        return ins
    assert spec is not None
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        mrx = int(z["max_range_px"])
        om = oracle_mod.OracleMap.from_gridmap(g, mrx)
        n = int(L[name + "_n_poses"])
        poses, fov, B = np.ascontiguousarray(z["poses"][:n]), float(z["fov"]), int(z["num_rays"])
        ins = ray_rows(poses, fov, B)
        for td in (112, 360):
            want = L["%s_cddt%d" % (name, td)]
            lit = om.cddt_rays_libm(td, ins)
            assert np.abs(lit - want).max() <= 1e-3 * g.resolution, (name, td)
            can = om.cddt_rays(td, ins)
            assert np.abs(can - want).max() <= 1e-3 * g.resolution, (name, td, float(np.abs(can - want).max()))
        td = 180
        rr, cc = om.lut_pose_cells(poses)
        lit_rows, can_rows = np.zeros((n, td), np.uint16), np.zeros((n, td), np.uint16)
        for i, (r_, c_) in enumerate(zip(rr, cc)):
            if r_ >= 0:
                lit_rows[i] = om.lut_build_libm(td, int(r_), int(r_) + 1, nthreads=1)[0, int(c_)]
                can_rows[i] = om.lut_build(td, int(r_), int(r_) + 1, nthreads=1)[0, int(c_)]
        assert np.abs(lit_rows.astype(np.int32) - L[name + "_lut180_rows"].astype(np.int32)).max() <= 1
        assert np.abs(can_rows.astype(np.int32) - L[name + "_lut180_rows"].astype(np.int32)).max() <= 1
        assert np.abs(om.lut_fan_rows_libm(lit_rows, poses, fov, B) - L[name + "_lut180_fan"]).max() <= 1e-3 * g.resolution
        # canonical fan query on canonical rows against the literal answer: the same bins except at exact ties
        can_fan = om.lut_fan_rows(can_rows, poses, fov, B)
        assert (np.abs(can_fan - L[name + "_lut180_fan"]) > 1e-3 * g.resolution).mean() <= 1e-3


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_other_casters_c_oracle_equals_numpy_statements(oracle_mod, seed):
    """Rows a12-a14: BresenhamsLine, GiantLUTCast queries and CDDTCast (table + queries) of the C
    oracle against independently structured NumPy statements (oracle/np_statement.py: all rays stepped
    together, edge map by array shifts, buckets as np.unique arrays searched with searchsorted) —
    ranges, hit cells and step counts bit for bit, on seeded mazes with rotated origins, poses outside
    the map and in the (-1, 0) truncation strip, several fans and theta discretisations."""
    from oracle import np_statement as N
    rng = np.random.default_rng(seed)
    g = maps.make_maze(64 + 9 * seed, cell=10 + 3 * seed, wall=1 + seed % 3, p=0.5, seed=seed,
                       resolution=[0.05, 0.1, 1.0, 0.013][seed],
                       origin=(-1.0 + seed, 0.7 - seed, [0.0, 0.4, -2.5, 3.1][seed]))
    mrx = [40, 60, 25, 120][seed]
    om = oracle_mod.OracleMap.from_gridmap(g, mrx)
    poses = maps.sample_free_poses(g, 14, seed)
    poses[0] = [-50.0, 0.0, 0.0]                                   # far outside
    c, s_ = np.cos(g.origin[2]), np.sin(g.origin[2])
    poses[1, 0] = g.origin[0] + (c * -0.3 - s_ * 5.5) * g.resolution      # -1 < gx < 0
    poses[1, 1] = g.origin[1] + (s_ * -0.3 + c * 5.5) * g.resolution
    poses[2, 2] = 1e4
    for B, fov in ((97, 4.71), (33, 6.283), (64, -2.0)):
        r, h, st = N.bl_fan(g.occ, g.resolution, g.origin, mrx, poses, fov, B)
        r0, h0, s0 = om.bl_fan(poses, fov, B)
        assert np.array_equal(r, r0) and np.array_equal(h, h0) and np.array_equal(st, s0), ("BL", B, fov)
        td = int(rng.choice([30, 90, 181]))
        lut = om.lut_build(td, nthreads=oracle_mod.max_threads())
        assert np.array_equal(N.lut_fan(lut, g.occ.shape, g.resolution, g.origin, mrx, poses, fov, B),
                              om.lut_fan(lut, poses, fov, B)), ("LUT", td, B, fov)
    for td in (16, 37, 112):
        tab = N.CddtTable(g.occ, td)
        for B, fov in ((97, 4.71), (40, 6.283)):
            assert np.array_equal(N.cddt_fan(tab, g.resolution, g.origin, mrx, poses, fov, B),
                                  om.cddt_fan(td, poses, fov, B)), ("CDDT", td, B, fov)


def test_bresenham_close_to_ray_marching(oracle_mod):
    g, z = load_golden("rm_maze256")
    diff = np.abs(z["ranges_bl"] - z["ranges_cpu"])
    assert (diff <= 2.0 * g.resolution).mean() > 0.97


# ---------------------------------------------------------------- LUT / CDDT restatements
def test_giant_lut_matches_ray_marching_from_cell_corner(oracle_mod):
    g = maps.make_maze(96, cell=16, wall=2, p=0.5, seed=4)
    om = oracle_mod.OracleMap.from_gridmap(g, 80)
    td = 180
    lut = om.lut_build(td, nthreads=4)
    assert lut.shape == (96, 96, td)
    # a LUT query equals (quantised) RM from the cell corner at the bin angle
    rng = np.random.default_rng(0)
    rr, cc = np.nonzero(g.occ == 0)
    k = rng.integers(0, rr.size, 500)
    b = rng.integers(0, td, 500)
    ins = np.stack([cc[k] * g.resolution + g.origin[0] + 1e-4, rr[k] * g.resolution + g.origin[1] + 1e-4,
                    b * (2 * np.pi / td)], 1).astype(np.float32)
    q = om.lut_rays(lut, ins)
    exact, _, _ = om.rm_rays(np.stack([ins[:, 0] - 1e-4, ins[:, 1] - 1e-4, ins[:, 2]], 1))
    assert np.abs(q - np.minimum(exact, 80 * g.resolution)).max() < 0.01 * g.resolution + 2e-3
    # fan form == per-ray form on the same headings
    poses = maps.sample_free_poses(g, 5, 1)
    fan = om.lut_fan(lut, poses, 4.71, 271)
    assert fan.shape == (5 * 271,) and np.all(fan >= 0) and np.all(fan <= 80 * g.resolution + 1e-6)


def test_cddt_tracks_ray_marching(oracle_mod):
    g, z = load_golden("rm_maze256")
    om = oracle_mod.OracleMap.from_gridmap(g, 300)
    poses = z["poses"][:8]
    r_rm, _, _ = om.rm_fan(poses, 4.71, 1081)
    r_cd = om.cddt_fan(720, poses, 4.71, 1081)
    err = np.abs(r_cd - r_rm) / g.resolution
    assert np.median(err) < 1.0 and (err < 3.0).mean() > 0.9


# ---------------------------------------------------------------- consumers: Car (racecar.cpp)
def test_edge_distances_and_is_crashed_vs_reference_build(oracle_mod):
    """GOLD-D: codes produced by the reference's own compiled Car::isCrashed
    (racecar/src/racecar.cpp:305-328 via oracle/_ref) on seeded scans."""
    z = np.load(os.path.join(GOLD, "car_ref.npz"))
    car = dict(zip([str(k) for k in z["car_keys"]], z["car"]))
    for i in range(len(z["codes"])):
        B, fov, P = int(z["num_rays"][i]), float(z["fov"][i]), int(z["poses"][i])
        edge = oracle_mod.edge_distances(B, -fov / 2.0, fov / B, 0.275, car["width"], car["wb"])
        code = oracle_mod.is_crashed(z["rays_%d" % i], B, P, edge, car["ttc_thresh"])
        assert code == int(z["codes"][i]), i
    # semantics pinned by the survey probe: no crash -> -(poses+1); crash in pose k -> k
    edge = oracle_mod.edge_distances(8, -1.0, 0.25, 0.275, 0.2032, 0.3302)
    far = np.full(24, 10.0, np.float32)
    assert oracle_mod.is_crashed(far, 8, 3, edge, 0.001) == -4
    assert oracle_mod.is_crashed(far[:8], 8, 1, edge, 0.001) == -2
    far[8 + 2] = 0.0   # (beam 3 sits at angle 0: the reference's -1016 m edge quirk, racecar.cpp:277-278)
    assert oracle_mod.is_crashed(far, 8, 3, edge, 0.001) == 1


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle/_ref/libracecar_ref.so")),
                    reason="oracle/_ref not built")
def test_oracle_edge_table_equals_live_reference_build(oracle_mod):
    L = C.CDLL(os.path.join(ROOT, "oracle/_ref/libracecar_ref.so"))
    L.ref_car_create.restype = C.c_void_p
    L.ref_car_create.argtypes = [C.POINTER(C.c_double)]
    L.ref_car_set_edge_distances.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.ref_car_is_crashed.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
    z = np.load(os.path.join(GOLD, "car_ref.npz"))
    car = L.ref_car_create((C.c_double * 17)(*z["car"]))
    B, fov = 1081, 4.71
    L.ref_car_set_edge_distances(car, B, -fov / 2, fov / B, 0.275)
    edge = oracle_mod.edge_distances(B, -fov / 2, fov / B, 0.275, float(z["car"][10]), float(z["car"][0]))
    # probe the private table through isCrashed: a ray just under / over edge+thresh per beam
    thresh = float(z["car"][9])
    for j in range(0, B, 37):
        rays = np.full(B, 100.0, np.float32)
        rays[j] = np.float32(edge[j] + thresh) - np.float32(1e-3)
        assert L.ref_car_is_crashed(car, rays.ctypes.data_as(C.POINTER(C.c_float)), B, 1) == 0
        rays[j] = np.float32(edge[j] + thresh) + np.float32(1e-3)
        assert L.ref_car_is_crashed(car, rays.ctypes.data_as(C.POINTER(C.c_float)), B, 1) == -2


def test_random_maps_c_oracle_vs_numpy_statement(oracle_mod):
    """Randomised cross-check of the two CPU statements (C restatement vs NumPy/SciPy): shapes,
    origins, yaw, ranges, fans — bit-identical EDT, ranges, hit cells and sample counts."""
    rng = np.random.default_rng(77)
    for case in range(25):
        rows, cols = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        dens = float(rng.choice([0.0, 0.01, 0.05, 0.3]))
        occ = (rng.random((rows, cols)) < dens).astype(np.uint8)
        res = float(rng.choice([0.05, 0.1, 1.0]))
        origin = (float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20)),
                  float(rng.choice([0.0, rng.uniform(-3, 3)])))
        mrx = float(rng.choice([1, 9, 60, 300]))
        B, fov = int(rng.choice([1, 7, 64, 181])), float(rng.choice([4.71, 6.283, -1.0, 0.0]))
        P = int(rng.integers(1, 12))
        assert np.array_equal(oracle_mod.edt(occ), N.edt(occ))
        gx, gy = rng.uniform(-2, cols + 2, P), rng.uniform(-2, rows + 2, P)
        c, s = np.cos(origin[2]), np.sin(origin[2])
        poses = np.stack([origin[0] + (c * gx - s * gy) * res, origin[1] + (s * gx + c * gy) * res,
                          rng.uniform(-7, 7, P)], 1).astype(np.float32)
        om = oracle_mod.OracleMap(occ, res, origin, mrx)
        for sc in (0.999, 1.0):
            r, h, st = om.rm_fan(poses, fov, B, step_coeff=sc)
            r2, h2, st2 = N.rm_fan(occ, res, origin, mrx, poses, fov, B, sc)
            assert np.array_equal(r, r2) and np.array_equal(h, h2) and np.array_equal(st, st2), case


# ---------------------------------------------------------------- FollowGap
def _followgap_cases():
    z = np.load(os.path.join(GOLD, "followgap_ref.npz"))
    offs = z["offsets"]
    return [z["scans"][offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], z["angles"], z["params"]


def test_followgap_restatement_equals_reference_build_vectors(oracle_mod):
    """GOLD-E: orc_followgap_eval vs the angles the reference's own followgap.hpp produced (compiled
    in place into oracle/_ref, tests/golden/make_fixtures.py::followgap_ref): bit-identical."""
    scans, angles, prm = _followgap_cases()
    assert len(scans) >= 90
    for v, a in zip(scans, angles):
        got = np.float32(oracle_mod.followgap_eval(v, float(prm[1]), float(prm[2]), float(prm[3])))
        assert got.tobytes() == np.float32(a).tobytes(), (len(v), got, a)
    assert np.isnan(oracle_mod.followgap_eval(np.ones(9, np.float32), 15.0, 0.4, 0.004))   # size < 10


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle/_ref/libfollowgap_ref.so")),
                    reason="oracle/_ref not built")
def test_followgap_restatement_equals_live_reference_build(oracle_mod):
    L = C.CDLL(os.path.join(ROOT, "oracle/_ref/libfollowgap_ref.so"))
    L.ref_followgap_eval.restype = C.c_float
    L.ref_followgap_eval.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    rng = np.random.default_rng(5)
    n = 0
    for trial in range(300):
        size = int(rng.integers(12, 1500))
        v = rng.uniform(0.0, rng.choice([2.0, 5.0, 20.0]), size).astype(np.float32)
        v[rng.random(size) < 0.1] = 0.0
        v[-1] = 0.5                       # keeps the chosen gap off the last beam (reference reads past the end there)
        md, ma, inc = 15.0, float(rng.choice([0.4189, 0.2])), float(rng.choice([0.004, 0.00436]))
        ref = L.ref_followgap_eval(v.ctypes.data_as(C.POINTER(C.c_float)), size, 10, md, ma, inc)
        got = oracle_mod.followgap_eval(v, md, ma, inc)
        assert np.float32(got).tobytes() == np.float32(ref).tobytes() or (np.isnan(got) and np.isnan(ref))
        n += 1
    assert n == 300


def test_audit_mode_trig_statement_equals_this_hosts_libm():
    """The product's audit mode (variant 3, csrc/literal_kernels.h) evaluates glibc's sinf / cosf algorithm on the device.
    The oracle holds the same statement in C; here it is walked against THIS host's libm over every 61st float bit
    pattern of both signs (70 million inputs, ~2 s; step 1 — every finite float, 0 mismatches on glibc 2.35 / x86-64
    with FMA — takes ~20 s on 8 cores: `python -c "from oracle import oracle as O; print(O.libm_restatement_mismatches())"`).
    A host whose libm is another implementation fails here: the audit mode then reproduces glibc's arithmetic, not
    that host's, and tests/test_gpu_parity.py::test_audit_mode_* say so too."""
    assert O.libm_restatement_mismatches(first=0, step=61) == (0, 0)
    assert O.libm_restatement_mismatches(first=0x3f000000, step=1 << 30) == (0, 0)      # (tiny call: argument handling)
    x = np.array([0.3, -2.0, 11.0, 119.9, 120.0, 1e7, -3e38, 0.0, 1e-40], np.float32)
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(O.libm_sincosf(x), O.lit_sincosf(x)))


def test_committed_bresenham_literal_vectors_match_this_hosts_libm(oracle_mod):
    """tests/golden/bl_libm_forms.npz — BresenhamsLine in the upstream-literal statement (libm cosf / sinf of
    theta' = -theta + rotation constant, calc_range(y, x, theta'), un-fused end point and hit distance), 2-argument and
    4-argument forms — recomputed on this host: equal to the committed arrays, and the CANONICAL statement (what the
    device walks bit for bit) stays inside the gate the GPU test applies to the device: the same hit cell and step count
    on every ray, ranges within 1e-3 cell."""
    import os
    from conftest import GOLD
    L = np.load(os.path.join(GOLD, "bl_libm_forms.npz"))
    for name in ("rm_colombia", "rm_maze256", "rm_maze192_yaw"):
        g, z = load_golden(name)
        om = oracle_mod.OracleMap.from_gridmap(g, int(z["max_range_px"]))
        n = int(L[name + "_n_poses"])
        poses, fov, B = np.ascontiguousarray(z["poses"][:n]), float(z["fov"]), int(z["num_rays"])
        r, h, s = om.bl_fan_libm(poses, fov, B)
        assert np.abs(r - L[name + "_fan_ranges"]).max() <= 1e-3 * g.resolution, name
        assert np.array_equal(h, L[name + "_fan_hits"].astype(np.int32)) and np.array_equal(s, L[name + "_fan_steps"]), name
        rc, hc, sc = om.bl_fan(poses, fov, B)
        assert np.array_equal(hc, h) and np.array_equal(sc, s), (name, "canonical walk differs from the literal one")
        assert np.abs(rc - r).max() <= 1e-3 * g.resolution, (name, float(np.abs(rc - r).max()))
        ang = (np.float32(-0.5) * np.float32(fov) + np.arange(B, dtype=np.float32) * (np.float32(fov) / np.float32(B))).astype(np.float32)
        ins = np.zeros((n * B, 3), np.float32)
        for q in range(n):
            ins[q * B:(q + 1) * B, :2] = poses[q, :2]
            ins[q * B:(q + 1) * B, 2] = poses[q, 2] + ang
        r2, h2, s2 = om.bl_rays_libm(ins)
        assert np.abs(r2 - L[name + "_rays_ranges"]).max() <= 1e-3 * g.resolution, name
        assert np.array_equal(h2, L[name + "_rays_hits"].astype(np.int32)) and np.array_equal(s2, L[name + "_rays_steps"]), name
