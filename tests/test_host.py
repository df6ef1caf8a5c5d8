"""CPU tests of the host side: map ingestion, workload generators, the ScanSimulator2D call
protocol (GOLD-C, captured from the reference's own scan_simulator.py) and the C-ABI surface."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLD, ROOT
from pyracecarsimulator_amd import _lib, maps, range_libc, workloads
from pyracecarsimulator_amd.scan_simulator import ScanSimulator2D


# ---------------------------------------------------------------- maps
def test_pgm_p2_and_p5_roundtrip(tmp_path):
    img = (np.arange(12 * 7).reshape(7, 12) * 3 % 256).astype(np.uint8)
    p5 = tmp_path / "a.pgm"
    p5.write_bytes(b"P5\n# c\n12 7\n255\n" + img.tobytes())
    p2 = tmp_path / "b.pgm"
    p2.write_text("P2\n# 8-bit pgm gray\n12 7\n255\n" +
                  "\n".join(" ".join("%3d" % v for v in row) for row in img) + "\n")
    assert np.array_equal(maps.read_pgm(str(p5)), img)
    assert np.array_equal(maps.read_pgm(str(p2)), img)
    with pytest.raises(ValueError):
        (tmp_path / "c.pgm").write_text("P3\n1 1\n255\n0 0 0\n")
        maps.read_pgm(str(tmp_path / "c.pgm"))


def test_map_server_thresholds_flip_and_reference_binarisation(tmp_path):
    img = np.array([[0, 254, 205], [100, 255, 30]], np.uint8)      # row 0 = TOP of the map
    data = maps.occupancy_from_image(img, 0, 0.65, 0.196)
    # (255-p)/255: 0->1.0 occ, 254->0.004 free, 205->0.196.. unknown, 100->0.61 unknown, 255 free, 30->0.88 occ
    assert data.tolist() == [[-1, 0, 100], [100, 0, -1]]           # flipped vertically
    occ = maps.binarise_reference(data)                            # unknown(-1) -> free
    assert occ.tolist() == [[0, 0, 1], [1, 0, 0]]
    (tmp_path / "m").mkdir()
    (tmp_path / "m" / "map.pgm").write_bytes(b"P5\n3 2\n255\n" + img.tobytes())
    (tmp_path / "m" / "map.yaml").write_text(
        "image: map.pgm\nresolution: 0.05\norigin: [-1.0, -2.0, 0.0]\nnegate: 0\n"
        "occupied_thresh: 0.65\nfree_thresh: 0.196\n")
    g = maps.load_map_server_map(str(tmp_path / "m" / "map.yaml"))
    assert g.occ.tolist() == occ.tolist() and g.resolution == 0.05 and g.origin == (-1.0, -2.0, 0.0)


def test_colombia_fixture():
    g = maps.load_colombia()
    assert (g.rows, g.cols) == (350, 435) and abs(g.occ.mean() - 0.7173) < 1e-3
    assert g.resolution == 0.05 and g.origin == (-5.70654, -2.020793, 0.0)


def test_generators_are_deterministic():
    a, b = maps.make_maze(200, seed=3), maps.make_maze(200, seed=3)
    assert np.array_equal(a.occ, b.occ) and not np.array_equal(a.occ, maps.make_maze(200, seed=4).occ)
    assert a.occ[0].all() and a.occ[:, -1].all()
    p1, p2 = maps.sample_free_poses(a, 50, 9), maps.sample_free_poses(a, 50, 9)
    assert np.array_equal(p1, p2) and p1.dtype == np.float32 and p1.shape == (50, 3)
    col = np.floor((p1[:, 0] - a.origin[0]) / a.resolution).astype(int)
    row = np.floor((p1[:, 1] - a.origin[1]) / a.resolution).astype(int)
    assert np.all(a.occ[row, col] == 0)


def test_workload_configs_match_baseline_json():
    w2 = workloads.cfg2()
    assert (w2.gmap.rows, w2.n_poses, w2.num_rays, w2.method) == (2049, 4096, 1081, "RMGPU")
    assert w2.max_range_px == 300 and w2.fov == 4.71
    assert workloads.cfg5().num_rays == 720 and workloads.cfg5().noise_std == 0.01
    assert workloads.cfg4().gmap.name == "colombia" and workloads.cfg4().n_poses == 1 << 20
    for n, world in [(10, 3), (4096, 8), (7, 8)]:
        spans = [workloads.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


# ---------------------------------------------------------------- ScanSimulator2D protocol (GOLD-C)
class _RecordingMethod:
    def __init__(self):
        self.calls = []

    def set_noise(self, *a):
        pass

    def calc_range_many(self, ins, outs, fov=None, num_rays=None):
        self.calls.append(("many", ins.copy(), outs.shape, outs.dtype, fov, num_rays))
        outs[:] = np.arange(outs.size, dtype=np.float32)

    def calc_range_fan(self, poses, outs, fov, num_rays, hit_cells=None, steps=None):
        self.calls.append(("fan", poses.copy(), outs.shape, outs.dtype, fov, num_rays))
        outs[:] = np.arange(outs.size, dtype=np.float32)


def test_scan_simulator_reproduces_reference_call_protocol():
    proto = json.load(open(os.path.join(GOLD, "protocol.json")))
    c = proto["ctor"]
    sim = ScanSimulator2D(c["num_rays"], c["fov"], c["scan_std"], batch_size=c["batch_size"])
    sim.setMap("omap", 300, 0.05, (0.0, 0.0, 0.0))
    sim.scan_method = rec = _RecordingMethod()
    # every public attribute of the reference object exists here too
    assert set(proto["attributes"]) <= set(vars(sim))
    out = sim.scan(1.0, 2.0, 0.5)
    kind, ins, oshape, odt, fov, nr = rec.calls[0]
    ref = proto["scan_call"]
    assert kind == "many" and list(ins.shape) == ref["ins_shape"] and str(ins.dtype) == ref["ins_dtype"]
    assert np.nonzero(ins.any(axis=1))[0].tolist() == ref["nonzero_rows"]
    assert list(oshape) == ref["outs_shape"] and [fov, nr] == ref["extra_args"]
    assert (out is sim.output_vector) == proto["scan_returns_cached_buffer"]
    poses = np.array([[1, 2, 0.1], [3, 4, 0.2], [5, 6, 0.3], [7, 8, 0.4], [9, 9, 9]], np.float32)
    out = sim.scanMany(poses)            # 5 poses given, batch_size 4 used (scan_simulator.py:119)
    kind, p, oshape, odt, fov, nr = rec.calls[1]
    refm = proto["scanMany_call"]
    assert p.shape == (proto["scanMany_pose_rows_used"], 3) and np.array_equal(p, poses[:4])
    assert list(oshape) == refm["outs_shape"] and [fov, nr] == refm["extra_args"]
    assert list(sim.input_vector_many.shape) == refm["ins_shape"]
    assert np.nonzero(sim.input_vector_many.any(axis=1))[0].tolist() == refm["nonzero_rows"]
    assert (out is sim.output_vector_many) == proto["scanMany_returns_cached_buffer"]
    assert sim.scanMany(poses, copy=True) is not sim.output_vector_many


def test_set_raytracing_method_error_behaviour(capsys):
    sim = ScanSimulator2D(16, 4.71, 0.01, batch_size=2)
    sim.setRaytracingMethod("RM")                      # no map: prints and returns (:67-69)
    assert "setMap first" in capsys.readouterr().out and sim.scan_method is None
    sim.setMap("omap", 300, 0.05, (0, 0, 0))
    with pytest.raises(SystemExit):                    # unknown method: print + sys.exit() (:77-79)
        sim.setRaytracingMethod("CDDT")


def test_calc_range_many_array_contract():
    chk = range_libc._check_ins_outs
    ins, outs = np.zeros((8, 3), np.float32), np.zeros(8, np.float32)
    chk(ins, outs)
    for bad_ins, bad_outs in [(ins.astype(np.float64), outs), (ins, outs.astype(np.float64)),
                              (np.zeros((8, 2), np.float32), outs), (ins[::2], outs),
                              (ins, np.zeros(4, np.float32)), (np.zeros(24, np.float32), outs)]:
        with pytest.raises(ValueError):
            chk(bad_ins, bad_outs)
    with pytest.raises(TypeError):
        chk([[0, 0, 0]], outs)


# ---------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "scanlib.h")).read()
    declared = set(re.findall(r"\b(rl_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.rl_version()


def test_no_cpu_fallback_fails_loudly_without_device():
    L = _lib.lib()
    if L.rl_device_count() > 0:
        pytest.skip("a HIP device is visible here")
    with pytest.raises(_lib.ScanLibError) as e:
        range_libc.PyOMap(np.zeros((8, 8), np.uint8), 0.05)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pyracecarsimulator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.replace("SURVEY", ""), os.path.join(dirpath, f)


# ---------------------------------------------------------------- racecar host table
def test_edge_distances_host_table_equals_reference_behaviour(oracle_mod):
    from pyracecarsimulator_amd import racecar as RC
    z = np.load(os.path.join(GOLD, "car_ref.npz"))
    car = dict(zip([str(k) for k in z["car_keys"]], z["car"]))
    for B, fov in [(1081, 4.71), (1080, 4.71), (720, 3.14), (64, 6.2), (8, 2.0)]:
        mine = RC.edge_distances(B, -fov / 2.0, fov / B, 0.275, car["width"], car["wb"])
        ref = oracle_mod.edge_distances(B, -fov / 2.0, fov / B, 0.275, car["width"], car["wb"])
        assert np.array_equal(mine, ref)
    # and through the reference binary's isCrashed codes (GOLD-D)
    for i in range(0, len(z["codes"]), 5):
        B, fov, P = int(z["num_rays"][i]), float(z["fov"][i]), int(z["poses"][i])
        edge = RC.edge_distances(B, -fov / 2.0, fov / B, 0.275, car["width"], car["wb"])
        assert oracle_mod.is_crashed(z["rays_%d" % i], B, P, edge, car["ttc_thresh"]) == int(z["codes"][i])
    assert tuple(RC.CAR_PARAM_ORDER) == tuple(str(k) for k in z["car_keys"])


def test_params_yaml_loader_maps_rosparam_names(tmp_path):
    from pyracecarsimulator_amd import config, racecar as RC
    y = tmp_path / "params.yaml"
    y.write_text("wheelbase: 0.5\nscan_beams: 720\nscan_fov: 3.14\nC_S_front: 4.7\nbatch_size: 64\n"
                 "scan_method: \"RM\"\nmass: 3.5\nsome_topic: \"/x\"\n")
    cfg = config.load_params(str(y), max_speed=5.0)
    assert cfg["wb"] == 0.5 and cfg["scan_beams"] == 720 and cfg["cs_f"] == 4.7 and cfg["batch_size"] == 64
    assert cfg["scan_method"] == "RM" and cfg["max_speed"] == 5.0 and cfg["mass"] == 3.5
    assert cfg["l_r"] == config.DEFAULTS["l_r"] and "some_topic" not in cfg
    assert all(k in cfg for k in RC.CAR_PARAM_ORDER)           # everything Car's constructor needs
    assert config.load_params()["scan_max_range"] == 15.0


def test_pyomap_ingests_an_occupancy_grid_message_like_the_reference():
    """Row a6: the reference's real call form is PyOMap(map_msg) (/root/reference/scripts/
    ros_interface.py:210, mcts_driver.py:278, two_player/scan.py:45) after rewriting map_msg.data to
    {0, 255} (ros_interface.py:80-86); origin yaw comes from the quaternion (ros_interface.py:212-220)."""
    from conftest import occupancy_grid_msg
    g = maps.make_maze(48, cell=12, wall=2, p=0.5, seed=2, resolution=0.05, origin=(-1.5, 2.25, 0.6))
    for binarise in (True, False):                 # {0,255} as the reference feeds it; raw {-1,0,100} too
        msg = occupancy_grid_msg(g, binarise)
        occ, res, org = range_libc.PyOMap._ingest(msg, None, None, None)
        assert occ.shape == (g.rows, g.cols) and np.array_equal(occ.astype(np.uint8), (g.occ != 0).astype(np.uint8))
        assert res == g.resolution and org[0] == g.origin[0] and org[1] == g.origin[1]
        assert abs(org[2] - g.origin[2]) < 1e-12
    # yaw outside (-pi/2, pi/2) and negative
    for yaw in (2.5, -3.0, -0.4, 0.0):
        g2 = maps.GridMap(g.occ, g.resolution, (0.0, 0.0, yaw), "yaw")
        assert abs(range_libc.PyOMap._ingest(occupancy_grid_msg(g2), None, None, None)[2][2] - yaw) < 1e-12


# ---------------------------------------------------------------------------------------------
# launch planning (rl_plan_fan: pure host arithmetic, include/scanlib.h "launch planning")
# ---------------------------------------------------------------------------------------------
def _plan(kind, rows, cols, n, B, **kw):
    return _lib.plan_fan(kind, rows, cols, n, B, **kw)


def test_launch_plan_of_every_baseline_config_and_of_the_reference_rollout_batch():
    """Which kernel every BASELINE.json configuration launches, decided without a device."""
    RM, RMGPU, CDDT, GLT, BL = _lib.RL_RM, _lib.RL_RM_GPU, _lib.RL_CDDT, _lib.RL_GIANT_LUT, _lib.RL_BRESENHAM
    # cfg1: one pose — no binning pass, the two workgroups derive their records in LDS
    # ("RM" is range_libc's CPU RayMarching: the upstream-literal arithmetic by default — variant 3 —, canonical on request)
    p = _plan(RM, 2049, 2049, 1, 1081)
    assert (p["kernel"], p["binning"], p["record_source"], p["grid"], p["bands"]) == ("rm_stream_literal", "none", 1, 2, 1)
    p = _plan(RM, 2049, 2049, 1, 1081, variant=1)
    assert (p["kernel"], p["binning"], p["record_source"], p["grid"], p["bands"]) == ("rm_stream", "none", 1, 2, 1)
    # cfg2 through the library's defaults (a lone launch): keys-only binning + INLINE march, whole machine
    p = _plan(RMGPU, 2049, 2049, 4096, 1081)
    assert p["name"] == "scan::rm_fan_stream_kernel<false, false, 1024, true, true, 2, false, 0>"
    assert (p["binning"], p["record_source"], p["grid"], p["block"], p["bands"]) == ("small_keys", 3, 512, 1024, 8)
    # ... on the u16 code map once the handle knows the map's palette (626 entries on this maze): the palette rides in LDS
    pc = _plan(RMGPU, 2049, 2049, 4096, 1081, code_entries=626)
    assert pc["name"] == "scan::rm_fan_stream_kernel<false, false, 1024, true, true, 2, false, 2>"
    assert (pc["code"], pc["code_entries"], pc["grid"], pc["binning"]) == (2, 626, 512, "small_keys")
    assert pc["lds_bytes"] >= p["lds_bytes"] + 626 * 4 - 32 * 4 and pc["lds_bytes"] <= 72 * 1024
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, code_entries=626, code_map=0)["code"] == 0
    # (small launches are latency-bound: the palette look-up in every dependent sample costs them more than the smaller
    #  footprint buys — the reference's 200-pose roll-out stays on the float32 map)
    assert _plan(RMGPU, 2049, 2049, 200, 1080, code_entries=626)["code"] == 0
    assert _plan(RMGPU, 2049, 2049, 200, 1080, code_entries=626, code_min_rays=0)["code"] == 2
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, code_entries=5000)["code"] == 0          # palette beyond the LDS table
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, code_entries=626, aux=True)["code"] == 0  # diagnostics: float32 map
    assert _plan(RM, 2049, 2049, 4096, 1081, code_entries=626)["name"] == \
        "scan::rm_fan_stream_kernel<false, false, 1024, true, true, 2, true, 2>"
    # cfg2 the way bench.py pipelines it: three rays per lane, 0.75 workgroups per CU
    p = _plan(RMGPU, 2049, 2049, 4096, 1081, slots=3, grid_mult=3)
    assert p["name"] == "scan::rm_fan_stream_kernel<false, false, 1024, true, true, 3, false, 0>"
    assert (p["grid"], p["slots"], p["binning"]) == (192, 3, "small_keys")
    assert p["lds_bytes"] <= 72 * 1024
    # diagnostics and the fused crash test keep one or two rays per lane
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, slots=3, aux=True)["slots"] == 1
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, slots=3, crash=True)["name"] == \
        "scan::rm_fan_stream_kernel<false, true, 1024, true, true, 2, false, 0>"
    # cfg3: GiantLUT rows of 1442 bins = 3 x 16-B loads per lane, 17 chunks of 64 beams; and the CDDT variant
    p = _plan(GLT, 2000, 2000, 65536, 1081, theta_disc=1442)
    assert (p["name"], p["grid"], p["block"], p["lds_bytes"]) == ("scan::lut_fan_lds_kernel<3, 17>", 4096, 256, 12288)
    # cfg3's 65 536 poses run theta-major (all poses against one table bin at a time, bins pinned to XCDs;
    # the fan kernel takes 32 poses x 109 floats of LDS per pass); below cddt_theta_min pose-major:
    p = _plan(CDDT, 2000, 2000, 65536, 1081, theta_disc=108)
    assert (p["kernel"], p["block"], p["grid"], p["bands"], p["ch"], p["nl"], p["lds_bytes"], p["name"]) == \
        ("cddt_theta", 256, 54 * 256, 8, 5, 109, 32 * 110 * 4, "scan::cddt_theta_search2_kernel")   # units of 256 poses
    p4 = _plan(CDDT, 2000, 2000, 65536, 1081, theta_disc=108, cddt_search=0)                      # round 4's kernel: 128
    assert (p4["grid"], p4["name"]) == (16384, "scan::cddt_theta_search_kernel")
    assert _plan(CDDT, 2000, 2000, 65536, 1081, theta_disc=720)["ch"] == 4      # 16 poses x 721 floats
    assert _plan(CDDT, 2000, 2000, 65536, 1081, theta_disc=108, cddt_theta_min=0)["kernel"] == "cddt_bins"
    p = _plan(CDDT, 2000, 2000, 3, 1081, theta_disc=4, cddt_theta_min=1)
    assert (p["kernel"], p["grid"], p["bands"]) == ("cddt_theta", 2, 1)          # fewer units than XCDs: one run
    assert _plan(CDDT, 2000, 2000, 1 << 20, 1081, theta_disc=1024)["kernel"] == "cddt_bins"    # R would pass 2 GiB
    p = _plan(CDDT, 2000, 2000, 32767, 1081, theta_disc=108)
    # (54 table bins -> 64 lanes per pose, 4 poses per pass of a 256-lane workgroup, 8 workgroups per CU)
    assert (p["kernel"], p["block"], p["grid"], p["lds_bytes"], p["nl"], p["ch"]) == ("cddt_bins", 256, 2048, 1728, 64, 4)
    p = _plan(CDDT, 2000, 2000, 3, 1081, theta_disc=720)
    assert (p["block"], p["grid"], p["nl"], p["ch"], p["lds_bytes"]) == (1024, 2, 384, 2, 5760)
    assert _plan(CDDT, 2000, 2000, 65536, 64, theta_disc=108)["kernel"] == "cddt_rays"   # fewer beams than bins
    # cfg4: the whole 2^20-pose batch goes through in two pose slices (32-bit ray offsets); a rank's
    # shard in one; colombia sits in every XCD's L2, so big batches take grid-wide binning + two rays per lane
    p = _plan(RMGPU, 350, 435, 1 << 20, 1081)
    assert p["slices"] == 2 and p["slice_poses"] * 1081 < 1 << 30
    p = _plan(RMGPU, 350, 435, 131072, 1081)
    assert (p["slices"], p["binning"], p["record_source"], p["slots"], p["grid"]) == (1, "grid_sort", 0, 2, 512)
    assert p["name"] == "scan::rm_fan_stream_kernel<false, false, 1024, false, true, 2, false, 0>"
    # ... while a 4096-pose batch on it needs no binning at all
    assert _plan(RMGPU, 350, 435, 4096, 1081)["binning"] == "none"
    # cfg5: 720 beams, 4096^2
    p = _plan(RMGPU, 4096, 4096, 262144, 720)
    assert (p["binning"], p["slots"], p["slices"]) == ("grid_sort", 2, 1)
    assert _plan(RMGPU, 4096, 4096, 32768, 720)["binning"] == "grid_sort"
    # the reference's own batch: 200 roll-out poses x 1080 beams (params.yaml:126,28) — one launch, no binning
    p = _plan(RMGPU, 350, 435, 200, 1080)
    assert (p["kernel"], p["binning"], p["record_source"], p["slices"]) == ("rm_stream", "none", 1, 1)
    assert p["grid"] * 16 >= 200 * 1080 // 64            # a wave per 64-ray block: nothing queues
    # Bresenham: stream schedule with records binned; 2-arg per-ray API = fan of one beam
    assert _plan(BL, 2049, 2049, 4096, 1081)["name"] == "scan::bl_fan_stream_kernel<false, 1024>"
    assert _plan(RMGPU, 2049, 2049, 100000, 1)["kernel"] == "rm_stream"


def test_launch_plan_thresholds_and_lds_budget():
    """The planner's case boundaries on a big map, and that no plan asks for more LDS than two
    1024-lane workgroups per CU can have."""
    RMGPU = _lib.RL_RM_GPU
    src = {n: _plan(RMGPU, 2049, 2049, n, 1081)["record_source"] for n in (64, 511, 512, 1536, 1537, 8191, 8192, 65536)}
    assert src == {64: 1, 511: 1, 512: 2, 1536: 2, 1537: 3, 8191: 3, 8192: 0, 65536: 0}
    bins = {n: _plan(RMGPU, 2049, 2049, n, 1081)["binning"] for n in (511, 1536, 1537, 8192)}
    assert bins == {511: "none", 1536: "none", 1537: "small_keys", 8192: "grid_sort"}
    for rows, cols in ((350, 435), (2049, 2049), (4096, 4096)):
        for n in (1, 63, 64, 200, 512, 1000, 2560, 4096, 8191, 8192, 40000, 65536, 1 << 20):
            for B in (64, 271, 720, 1081, 7680):
                for kw in ({}, {"slots": 3, "grid_mult": 3}, {"slots": 2, "grid_mult": 4}, {"crash": True}, {"aux": True}):
                    p = _plan(RMGPU, rows, cols, n, B, **kw)
                    # (the fused crash test cannot be cut into pose slices: 2^30 rays and more keep the chunk kernel)
                    assert p["kernel"] == ("rm_chunk" if kw.get("crash") and n * B >= 1 << 30 else "rm_stream")
                    assert p["grid"] >= 1
                    assert p["lds_bytes"] <= (72 if p["record_source"] else 150) * 1024, (rows, n, B, kw, p)
                    if p["record_source"] in (2, 3):
                        assert p["bands"] == 8
                    if p["slots"] > 1:
                        assert not p["aux"]
    # options reach the plan
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, variant=0)["name"] == "scan::rm_fan_kernel<false, false>"
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, inline_prep=0)["binning"] == "small_records"
    assert _plan(RMGPU, 2049, 2049, 4096, 1081, tiled=0, slots=2)["slots"] == 1
    with pytest.raises(_lib.ScanLibError):
        _plan(_lib.RL_GIANT_LUT, 100, 100, 10, 1081, theta_disc=100, crash=True)
    with pytest.raises(KeyError):
        _plan(RMGPU, 100, 100, 10, 1081, no_such_option=1)
    assert _lib.lib().rl_launch_contexts() >= 4


def test_launch_plan_survives_zeroed_and_out_of_range_options():
    """rl_plan_fan is public ABI: a C caller's `rl_plan_opts o = {0}` (and any hand-filled struct) must come
    back as a plan of the nearest valid options — not a division by zero (wg_threads / 64, xcd_bands), an
    undefined shift (slice_log2 >= 64) or a block size no kernel was instantiated for."""
    import ctypes as C
    L = _lib.lib()
    for kind in (_lib.RL_RM_GPU, _lib.RL_RM, _lib.RL_BRESENHAM, _lib.RL_CDDT, _lib.RL_GIANT_LUT):
        for n, B in ((200, 1081), (4096, 1081), (20000, 1081), (65536, 720), (1, 1)):
            o = _lib.PlanOpts()                                   # all zero
            pl = _lib.LaunchPlan()
            _lib.check(L.rl_plan_fan(kind, 256, 2049, 2049, 300.0, 108, C.byref(o), n, B, 0, 0, C.byref(pl)))
            assert pl.grid >= 1 and pl.block in (64, 256, 512, 1024), (kind, n, B, pl.as_dict())
    # the advisor's two reproducers and friends
    assert _plan(_lib.RL_RM_GPU, 2049, 2049, 20000, 1081, wg_threads=0)["block"] in (256, 512, 1024)
    assert _plan(_lib.RL_RM_GPU, 2049, 2049, 200, 1081, xcd_bands=0)["bands"] >= 1
    assert _plan(_lib.RL_RM_GPU, 2049, 2049, 4096, 1081, wg_threads=100)["block"] in (256, 1024)
    assert _plan(_lib.RL_RM_GPU, 2049, 2049, 4096, 1081, slice_log2=99, grid_mult=-5, slots=7, run_log2=40)["grid"] >= 1
    assert _plan(_lib.RL_RM_GPU, 2049, 2049, 1 << 20, 1081, slice_log2=-3)["slices"] >= 1
    assert _plan(_lib.RL_BRESENHAM, 2049, 2049, 4096, 1081, grid_mult=0, xcd_bands=-1)["grid"] >= 1
    # variant 3, the audit mode (upstream-literal arithmetic): one lane per ray, no binning pass, no fused crash test;
    # the table methods and Bresenham keep their own kernels
    # (round 5: in PRODUCTION shape — the stream kernel's schedule with the literal arithmetic — whenever the records can
    #  be derived in LDS and no diagnostics are asked for; the one-lane-per-ray kernel otherwise)
    pl = _plan(_lib.RL_RM, 2049, 2049, 4096, 1081, variant=3)
    assert (pl["kernel"], pl["block"], pl["binning"], pl["slots"], pl["name"]) == (
        "rm_stream_literal", 1024, "small_keys", 2, "scan::rm_fan_stream_kernel<false, false, 1024, true, true, 2, true, 0>")
    pl = _plan(_lib.RL_RM_GPU, 2049, 2049, 200, 1081, variant=3, crash=True)              # the reference's roll-out batch
    assert (pl["kernel"], pl["crash"], pl["name"]) == ("rm_stream_literal", 1, "scan::rm_fan_stream_kernel<false, true, 1024, true, true, 2, true, 0>")
    pl = _plan(_lib.RL_RM_GPU, 2049, 2049, 65536, 1081, variant=3)                        # beyond one INLINE launch: pose slices
    assert (pl["kernel"], pl["slices"], pl["slice_poses"]) == ("rm_stream_literal", 16, 4096)
    assert _plan(_lib.RL_RM_GPU, 435, 350, 100000, 1081, variant=3)["kernel"] == "rm_stream_literal"   # small map
    pl = _plan(_lib.RL_RM, 2049, 2049, 4096, 1081, variant=3, aux=True)                   # diagnostics: one lane per ray
    assert (pl["kernel"], pl["block"], pl["binning"], pl["name"]) == ("rm_literal", 256, "none", "scan::rm_literal_kernel<true, false>")
    assert _plan(_lib.RL_RM_GPU, 435, 350, 7, 1081, variant=3, aux=True)["name"] == "scan::rm_literal_kernel<true, false>"
    assert _plan(_lib.RL_RM_GPU, 2049, 2049, 4096, 32, variant=3)["kernel"] == "rm_literal"            # fans below 64 beams
    assert _plan(_lib.RL_RM_GPU, 2049, 2049, 4096, 1081, variant=9)["kernel"] == "rm_stream_literal"  # (clamped to 3)
    assert _plan(_lib.RL_BRESENHAM, 2049, 2049, 4096, 1081, variant=3)["kernel"] == "bl_stream"
    o = _lib.PlanOpts()
    _lib.check(L.rl_plan_default_opts(C.byref(o)))
    o.variant = 3
    pl2 = _lib.LaunchPlan()
    # the fused crash test in the literal arithmetic: served by the stream form; refused (RL_ERR_UNSUPPORTED) only where
    # that form cannot run — fans below 64 beams go to the one-lane-per-ray kernel, which has no crash test
    assert L.rl_plan_fan(_lib.RL_RM_GPU, 256, 2049, 2049, 300.0, 0, C.byref(o), 4096, 1081, 0, 1, C.byref(pl2)) == 0
    assert pl2.as_dict()["kernel"] == "rm_stream_literal" and pl2.crash == 1
    assert L.rl_plan_fan(_lib.RL_RM_GPU, 256, 2049, 2049, 300.0, 0, C.byref(o), 4096, 32, 0, 1, C.byref(pl2)) == -4       # RL_ERR_UNSUPPORTED


def test_launch_plan_falls_back_when_the_tiled_step_map_or_the_lds_does_not_fit():
    """A very elongated map (padded side > 2^20) would need a tiled pitch of K > 24, beyond the 24-bit multiply
    of the march's address: the plan marches on the row-major step map with one ray per lane instead; a fan
    whose beam tables exceed a workgroup's 160 KB of LDS is refused with RL_ERR_UNSUPPORTED, not launched."""
    RMGPU = _lib.RL_RM_GPU
    p = _plan(RMGPU, 300, 1100000, 4096, 1081, slots=2)
    assert p["tiled"] == 0 and p["slots"] == 1 and ", false, 1, false, 0>" in p["name"]
    assert _plan(RMGPU, 2049, 2049, 4096, 1081)["tiled"] == 1
    assert _plan(RMGPU, 16384, 16384, 4096, 1081)["tiled"] == 1          # K = 19: fits
    with pytest.raises(_lib.ScanLibError) as e:
        _plan(RMGPU, 2049, 2049, 64, 21000)
    assert e.value.code == -4
    assert _plan(RMGPU, 2049, 2049, 64, 7680, crash=True)["lds_bytes"] <= 160 * 1024


def test_multi_device_entry_points_exist_and_fail_loudly_without_a_device():
    """rl_map_create_multi / rl_car_create_multi: argument errors are RL_ERR_INVALID, and on a box without a
    GPU creation fails with RL_ERR_NO_DEVICE like rl_map_create (no CPU fallback)."""
    import ctypes as C
    L = _lib.lib()
    occ = np.zeros((8, 8), np.uint8)
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert L.rl_map_create_multi(occ.ctypes.data_as(_lib.u8p), 8, 8, 0.05, 0.0, 0.0, 0.0, devs, 0, C.byref(h)) == -1
    assert L.rl_map_create_multi(occ.ctypes.data_as(_lib.u8p), 8, 8, 0.05, 0.0, 0.0, 0.0, None, 2, C.byref(h)) == -1
    assert L.rl_map_n_devices(None) == 0 and L.rl_method_n_devices(None) == 0
    assert L.rl_map_replica(None, 0) is None and L.rl_method_replica(None, 0) is None
    if L.rl_device_count() == 0:
        rc = L.rl_map_create_multi(occ.ctypes.data_as(_lib.u8p), 8, 8, 0.05, 0.0, 0.0, 0.0, devs, 2, C.byref(h))
        assert rc == -2 and b"no HIP device" in L.rl_last_error()
        from pyracecarsimulator_amd import range_libc
        with pytest.raises(_lib.ScanLibError):
            range_libc.PyOMap(occ, 0.05, device=[0, 0])
        params = np.zeros(17)
        assert L.rl_car_create_multi(devs, 2, params.ctypes.data_as(_lib.f64p), C.byref(h)) == -2


def test_housekeeping_gates():
    """VERDICT r04 #9, kept as a gate: DESIGN.md stays below 40 KB (history lives in docs/history/), the library's host
    side stays in its four translation units, and no launch function of abi_fan.hip grows past 100 lines (one function
    per kernel family instead of one 220-line launch_fan)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.getsize(os.path.join(root, "DESIGN.md")) <= 40000
    csrc = os.path.join(root, "pyracecarsimulator_amd", "csrc")
    units = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    assert units == ["abi_car.hip", "abi_fan.hip", "abi_map.hip", "abi_multi.hip"], units
    src = open(os.path.join(csrc, "abi_fan.hip")).read().split("\n")
    starts = [i for i, l in enumerate(src) if re.match(r"(static )?int launch_[a-z_]+\(", l)]
    assert len(starts) >= 10, starts
    for i in starts:
        j = next(k for k in range(i, len(src)) if src[k] == "}")
        assert j - i + 1 <= 100, (src[i], j - i + 1)


def test_kernel_register_budgets_are_a_build_gate(tmp_path):
    """the march loops pin physical VGPRs in inline assembly and the stream kernels are held at 64 VGPRs /
    80 SGPRs for their eighth wave per SIMD — a compiler bump that spends one register more must fail the BUILD, not
    silently cost 25 % (profiles/r05/sgpr_cap_ab.txt).  csrc/check_resources.py reads hipcc's kernel-resource-usage remarks
    (the Makefile keeps them next to the objects and runs the check before it links): the current build passes, every
    budgeted family is present, and a remark that is one register over fails."""
    import subprocess, sys
    csrc = os.path.join(ROOT, "pyracecarsimulator_amd", "csrc")
    obj = os.path.join(csrc, ".obj")
    if not (os.path.isdir(obj) and any(f.endswith(".remarks") for f in os.listdir(obj))):
        _lib.build(force=True)                     # (a tree whose objects predate the remarks: rebuild once)
    chk = os.path.join(csrc, "check_resources.py")
    r = subprocess.run([sys.executable, chk, obj, "--print"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if "rm_fan_stream_kernel<" in ln]
    assert len(rows) >= 40 and all(" occupancy 8" in ln for ln in rows), rows[:3]
    assert any("true, true, 2, false, 2>" in ln for ln in rows)          # the code-map kernel is among them
    # the gate itself: the same remarks with one kernel pushed to 65 VGPRs / 7 waves
    text = open(os.path.join(obj, "abi_fan.remarks"), errors="replace").read()
    i = text.index("rm_fan_stream_kernel")
    j = text.index("VGPRs:", i)
    k = text.index("Occupancy [waves/SIMD]:", i)
    bad = text[:j] + "VGPRs: 65" + text[text.index("\n", j):k] + "Occupancy [waves/SIMD]: 7" + text[text.index("\n", k):]
    d = tmp_path / "obj"
    d.mkdir()
    (d / "abi_fan.remarks").write_text(bad)
    for f in ("abi_map.remarks", "abi_multi.remarks", "abi_car.remarks"):
        (d / f).write_text(open(os.path.join(obj, f), errors="replace").read())
    r = subprocess.run([sys.executable, chk, str(d)], capture_output=True, text=True)
    assert r.returncode == 1 and "vgpr = 65" in r.stderr and "occupancy = 7" in r.stderr, r.stderr[-1500:]
