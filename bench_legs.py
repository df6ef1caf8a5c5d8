"""bench.py's short side legs: the other single-GPU configurations of BASELINE.json on the same driver command.

``bench.py --gpus 1`` keeps cfg2 (4096 x 1081, RMGPU) as ``value``; this module adds ``other_configs`` to the same
JSON line — cfg3 with the GiantLUT and the CDDT variant (the bandwidth-bound alternatives north_star names; CDDT at the
reference's theta_disc 112, scripts/two_player/rcs_two_player.py:121), cfg2 in the reduced ``crash`` / ``steer`` modes
(the exchanges that are supposed to scale), one rank's shard of cfg5 (4096^2 map, 720 beams, noise) — each a short
VERIFIED timed loop: {mrays_s, ms_per_step, frac, frac_hbm, verified, ...}.  A leg is a few timed bursts bracketed by
device synchronisation, every step in flight on its own seeded pose batch, inputs resident in HBM.

Verification of a leg (``verified``): every slot buffer bit-equal to a serial launch of the same method; a pose
subsample of batch 0 bit-equal to the CPU oracle's statement of THAT method (GiantLUT: the oracle's fan query over the
device table's rows of the sampled cells, the table itself is pinned by tests/); reduced modes: the fused result equal to
the reduction of the serial launch's ranges.  For the table methods the error against EXACT ray marching on the same
subsample goes on the line as ``vs_exact_rm`` (cells: median / p90 / p99 / max), SURVEY.md section 8(c)'s tolerance.
The oracle is only the checker here (it is loaded by bench.py's cpu_baseline leg); without it ``verified`` rests on
the serial-launch comparison and says so.
"""
from __future__ import annotations

import time

import numpy as np

HBM_PEAK = 8.0e12


def error_stats_cells(got_m, want_m, resolution):
    """|got - want| in cells: median / p90 / p99 / max and the share within one cell."""
    e = np.abs(np.asarray(got_m, np.float64) - np.asarray(want_m, np.float64)) / float(resolution)
    return {"median": round(float(np.median(e)), 4), "p90": round(float(np.percentile(e, 90)), 4),
            "p99": round(float(np.percentile(e, 99)), 4), "max": round(float(e.max()), 3),
            "within_one_cell": round(float((e <= 1.0).mean()), 5), "rays": int(e.size)}


LEGS = (
    # name, workload, method, theta_disc, poses, pipeline, reduce, steps, bursts
    ("cfg3_glt", "cfg3", "GLT", 1442, 65536, 1, None, 8, 5),
    ("cfg3_cddt", "cfg3", "CDDT", 112, 65536, 4, None, 64, 5),
    ("cfg3_cddt_theta108", "cfg3", "CDDT", 108, 65536, 4, None, 64, 5),
    ("cfg2_crash", "cfg2", "RMGPU", 0, 4096, 4, "crash", 80, 7),
    ("cfg2_steer", "cfg2", "RMGPU", 0, 4096, 4, "steer", 80, 7),
    ("cfg5_shard", "cfg5", "RMGPU", 0, 32768, 4, None, 40, 5),
    # cfg4: colombia (the one map the reference ships), one rank's shard of the 2^20 roll-out poses, serial
    ("cfg4_shard", "cfg4", "RMGPU", 0, 131072, 1, None, 8, 5),
    # cfg4 as the reference's caller produces it (scripts/mcts.py:202-245): 4096 roll-outs x 200 control steps ->
    # 819 200 poses -> scan -> first crashed pose per roll-out, ONE host call (rl_car_rollout_check); run_rollout_leg
    ("cfg4_rollout_check", "cfg4", "RMGPU", 0, 4096, 1, "rollout", 5, 3),
)


def _alg_bytes(method, mean_steps, B, nbar):
    import math
    pose = 12.0 / B
    if method == "RMGPU":
        return mean_steps * 4.0 + 4.0 + pose
    if method == "GLT":
        return 2.0 + 4.0 + pose
    probes = max(1, math.ceil(math.log2(nbar + 1.0))) if nbar > 0 else 1
    return 4.0 * probes + 8.0 + 4.0 + pose


def run_leg(spec, torch, dev, device_index, O=None, pmc_lookup=None, seed_shift=0, cache=None, streams=None):
    from pyracecarsimulator_amd import range_libc, workloads, racecar as RC
    from pyracecarsimulator_amd.followgap import PyFollowGap
    from pyracecarsimulator_amd.pipeline import concurrent_streams

    name, wl, method, theta, n, P, reduce_, steps, bursts = spec
    t_leg = time.perf_counter()
    w = workloads.CONFIGS[wl]()
    B = w.num_rays
    omap = range_libc.PyOMap(w.gmap, device=device_index)
    if method == "GLT":
        meth = range_libc.PyGiantLUTCast(omap, w.max_range_px, theta)
    elif method == "CDDT":
        meth = range_libc.PyCDDTCast(omap, w.max_range_px, theta)
    else:
        meth = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
    is_rm = method == "RMGPU"
    dt = omap.distance_transform()
    # (the caller's set of concurrently running streams when it has one: every new set costs a probe of 16 candidates)
    if P > 1:
        streams = list(streams[:P]) if (streams and len(streams) >= P) else concurrent_streams(P)
    else:
        streams = [torch.cuda.current_stream()]
    P = len(streams)
    default_gm = meth.get_info("grid_mult")

    def schedule(pipelined):
        meth.set_option("grid_mult", 3 if (pipelined and P > 1) else default_gm)
        if is_rm:
            meth.set_option("slots", 2 if (pipelined and P > 1) else 0)

    # (the legs of one workload share their seeded pose batches and the checker's map: generating them is most of a
    #  leg's wall time)
    cache = {} if cache is None else cache
    key = (wl, n, seed_shift)
    have = cache.setdefault(key, {"batches": []})["batches"]
    while len(have) < P:
        have.append(workloads.make_poses(w, dt=dt, n_poses=n, seed=w.pose_seed + 7919 * len(have) + seed_shift))
    batches = have[:P]
    d_poses = [torch.from_numpy(b).to(dev) for b in batches]
    if w.noise_std > 0:
        meth.set_noise(w.noise_std, w.noise_seed, 0)
    group = next(g for g in range(min(200, n), 0, -1) if n % g == 0)
    n_groups = n // group
    edge = RC.edge_distances(B, -w.fov / 2.0, w.fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    d_edge = torch.from_numpy(edge).to(dev)
    THRESH = 0.001
    # the PRODUCT's own pipeline object runs the steps (ShardedScan: one prepared C call per step and slot — a Python
    # wrapper call per step costs the host more than a 26-us step leaves it)
    from pyracecarsimulator_amd.distributed import ShardedScan
    fg = None
    pose_ptrs = [t.data_ptr() for t in d_poses]
    if reduce_ == "crash":
        sc = ShardedScan(n, B, dev, n_chunks=1, gather=True, streams=streams, mode="crash", n_items=n_groups, every=8)
        sc.bind_crash(meth, pose_ptrs, w.fov, group, d_edge.data_ptr(), THRESH)
    elif reduce_ == "steer":
        fg = PyFollowGap(10, 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004, device=device_index)
        sc = ShardedScan(n, B, dev, n_chunks=1, gather=True, streams=streams, mode="steer", n_items=n, every=8)
        sc.bind_steer(meth, fg, pose_ptrs, w.fov)
    else:
        sc = ShardedScan(n, B, dev, n_chunks=1, gather=False, streams=streams, mode="ranges",
                         max_range_m=w.max_range_px * w.gmap.resolution)
        sc.bind(meth, pose_ptrs, w.fov)
    outs = [sl.local for sl in sc.slots]

    schedule(True)
    for _ in range(max(P, 3)):
        sc.step()
    sc.finish()
    torch.cuda.synchronize()
    plan = meth.last_plan()
    walls = []
    for _ in range(bursts):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _i in range(steps):
            sc.step()
        sc.finish()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / steps)
    ms = float(np.median(walls)) * 1e3
    rays = n * B
    red = None
    if reduce_:
        # the last reduced result of every slot (slot k always scans batch k)
        red = [sc.results(sl)[0, -1].clone() for sl in sc.slots]

    # ---- verification
    ver = {}
    ok = True
    schedule(False)
    if is_rm:
        meth.set_option("slots", 1)
    cur = torch.cuda.current_stream().cuda_stream
    d_ref = torch.empty(n * B, dtype=torch.float32, device=dev)
    mean_steps = 0.0
    for k in range(P):
        if reduce_ == "steer":
            meth.set_option("nt_store", 0)
        meth.calc_range_fan_device(d_poses[k].data_ptr(), n, w.fov, B, d_ref.data_ptr(), stream=cur)
        torch.cuda.synchronize()
        same = bool(torch.equal(outs[k], d_ref))
        ok &= same
        if reduce_ == "crash":
            hit = ((d_ref.view(n, B).double() - d_edge.view(1, B)) < THRESH).any(dim=1).view(n_groups, group)
            first = torch.where(hit.any(dim=1), hit.to(torch.int32).argmax(dim=1).to(torch.int32),
                                torch.full((n_groups,), -(group + 1), dtype=torch.int32, device=dev))
            ok &= bool(torch.equal(red[k], first))
        elif reduce_ == "steer":
            ang = torch.empty(n, dtype=torch.float32, device=dev)
            fg.eval_many_device(d_ref.data_ptr(), n, B, ang.data_ptr(), stream=cur)
            torch.cuda.synchronize()
            ok &= bool(torch.equal(red[k], ang))
    ver["slots_equal_serial_launch"] = ok
    if is_rm:
        d_steps = torch.empty(n * B, dtype=torch.int16, device=dev)
        meth.calc_range_fan_device(d_poses[0].data_ptr(), n, w.fov, B, d_ref.data_ptr(), d_steps_ptr=d_steps.data_ptr(), stream=cur)
        torch.cuda.synchronize()
        mean_steps = float(d_steps.to(torch.int32).bitwise_and(0xFFFF).float().mean().item())
        del d_steps
    out = {}
    if O is not None:
        om = cache[key].get("om")
        if om is None:
            om = cache[key]["om"] = O.OracleMap.from_gridmap(w.gmap, w.max_range_px)
        sub = np.unique(np.linspace(0, n - 1, 16 if method == "GLT" else 32).astype(np.int64))
        poses = np.ascontiguousarray(batches[0][sub])
        nthr = O.max_threads()
        got = np.empty(len(sub) * B, np.float32)
        if w.noise_std > 0:
            meth.set_noise(0.0, 0, 0)
        meth.calc_range_fan(poses, got, w.fov, B)
        if w.noise_std > 0:
            meth.set_noise(w.noise_std, w.noise_seed, 0)
        exact = om.rm_fan(poses, w.fov, B, step_coeff=1.0, nthreads=nthr, want_hits=False, want_steps=False)[0]
        if is_rm:
            want = exact
        elif method == "CDDT":
            want = om.cddt_fan(theta, poses, w.fov, B, nthreads=nthr)
        else:
            rr, cc = om.lut_pose_cells(poses)
            rows = np.empty((len(sub), theta), np.uint16)
            cache = {}
            for i, (r_, c_) in enumerate(zip(rr, cc)):
                if int(r_) not in cache:
                    cache[int(r_)] = meth.table(int(r_), int(r_) + 1)[0]
                rows[i] = cache[int(r_)][int(c_)]
            want = om.lut_fan_rows(rows, poses, w.fov, B)
        same = bool(np.array_equal(got, want))
        ver["oracle_subsample"] = same
        ver["oracle_sample"] = "%d poses x %d beams of batch 0 (noise off) against the oracle's %s statement" % (len(sub), B, method)
        ok &= same
        if not is_rm:
            out["vs_exact_rm"] = dict(error_stats_cells(got, exact, w.gmap.resolution),
                                      what="|range - exact ray marching (oracle rm_fan, coefficient 1.0)| in cells, same subsample")
        if reduce_ == "steer":
            wr = want.reshape(len(sub), B)
            ref = np.array([O.followgap_eval(wr[i], 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004) for i in range(len(sub))], np.float32)
            same = bool(np.array_equal(red[0].cpu().numpy()[sub], ref))
            ver["oracle_followgap"] = same
            ok &= same
    else:
        ver["oracle_subsample"] = "not run (the oracle is loaded by the cpu_baseline leg only)"
    nbar = 0.0
    if method == "CDDT":
        nbar = meth.get_info("cddt_values") / max(meth.get_info("cddt_nonempty_buckets"), 1)
    bpr = _alg_bytes(method, mean_steps, B, nbar)
    frac = bpr * rays / (ms * 1e-3) / HBM_PEAK
    pe = pmc_lookup(wl, method, n, plan, "steer" if reduce_ == "steer" else None) if pmc_lookup else None
    frac_hbm = round(pe["bytes"] / (ms * 1e-3) / HBM_PEAK, 5) if pe else None
    out.update({"mrays_s": round(rays / (ms * 1e-3) / 1e6, 1), "ms_per_step": round(ms, 4),
                # CDDT: the yardstick is the bytes the kernels of a step MOVE (committed PMC pass) / time / peak — SURVEY
                # section 8(d)'s per-ray bisection bytes are not what a kernel that answers ~10 beams per look-up moves
                "frac": frac_hbm if method == "CDDT" else round(frac, 5),
                "frac_hbm": frac_hbm,
                "verified": bool(ok), "verification": ver,
                "config": {"workload": "%s: %s %dx%d, %d poses x %d beams" % (wl, w.gmap.name, w.gmap.rows, w.gmap.cols, n, B),
                           "method": method + (" theta_disc %d" % theta if theta else "") +
                                     (" + Gaussian noise" if w.noise_std > 0 else ""),
                           "schedule": ("%d steps in flight" % P) if P > 1 else "serial",
                           "reduce": {"crash": "fused Car::isCrashed per %d-pose roll-out" % group,
                                      "steer": "Follow-the-Gap per scan on the slot's stream", None: "none"}[reduce_],
                           "kernel": plan["name"], "grid": plan["grid"]},
                "steps": steps, "bursts": bursts, "algorithmic_bytes_per_ray": round(bpr, 3),
                "traffic_source": pe["profile"] if pe else None,
                "leg_seconds": None})
    if method == "CDDT":
        out["frac_bisection"] = round(frac, 5)
        out["frac_is"] = ("HBM bytes the step's kernels move (committed rocprofv3 PMC pass of this launch shape) / ms_per_step / 8 TB/s; "
                          "frac_bisection = SURVEY 8(d)'s per-RAY bisection bytes / time / peak, > 1 because ~%d beams share one look-up"
                          % max(1, B // max(1, theta or 108)))
    if is_rm:
        out["mean_samples_per_ray"] = round(mean_steps, 3)
    sc.unbind()
    del sc
    if fg is not None:
        del fg
    meth.close()
    omap.close()
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
    return out


def run_rollout_leg(spec, torch, dev, device_index, O=None, pmc_lookup=None):
    """The MCTS roll-out chain in one call (rl_car_rollout_check: Car::control + updatePosition x 200 per roll-out ->
    poses -> scan -> Car::isCrashed per roll-out; scripts/mcts.py:202-245, racecar_simulator_v2.py:146-167), host
    pointers in, crash indices out: states / actions (host) -> int32 per roll-out (host).  The rate counts the rays the
    chain marches.  Verified: indices == the staged form (rl_car_rollout's poses through rl_check_collision_groups), a
    subsample of roll-outs == Car::isCrashed over the CPU oracle's scan of those poses."""
    from pyracecarsimulator_amd import range_libc, workloads, maps, racecar as RC
    name, wl, method, _theta, R, _P, _reduce, steps, bursts = spec
    t_leg = time.perf_counter()
    w = workloads.CONFIGS[wl]()
    B, n_steps = w.num_rays, 200
    omap = range_libc.PyOMap(w.gmap, device=device_index)
    meth = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
    cars = RC.CarBatch(device=device_index)
    dt = omap.distance_transform()
    rng = np.random.default_rng(w.pose_seed + 17)
    start = maps.sample_free_poses(w.gmap, R, w.pose_seed + 3, 4.0, dt)
    states = np.zeros((R, 11))
    states[:, :3] = start
    states[:, 3] = 1.0
    actions = np.stack([rng.uniform(0, 7, (R, 20)), rng.uniform(-0.4189, 0.4189, (R, 20))], -1)
    edge = RC.edge_distances(B, -w.fov / 2.0, w.fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    THRESH = 0.001
    for _ in range(2):
        first, _, _ = cars.rollout_check(meth, states, actions, w.fov, B, edge, THRESH, n_steps=n_steps)
    plan = meth.last_plan()
    walls = []
    for _ in range(bursts):
        t0 = time.perf_counter()
        for _i in range(steps):
            first, final, vel = cars.rollout_check(meth, states, actions, w.fov, B, edge, THRESH, n_steps=n_steps)
        walls.append((time.perf_counter() - t0) / steps)
    ms = float(np.median(walls)) * 1e3
    rays = R * n_steps * B
    # ---- verification
    ver, ok = {}, True
    poses, final2, vel2 = cars.rollout(states, actions, n_steps=n_steps)
    staged = meth.check_collision_groups(poses.reshape(-1, 3), n_steps, w.fov, B, edge, THRESH)
    same = bool(np.array_equal(first, staged) and np.array_equal(final, final2) and np.array_equal(vel, vel2))
    ver["chain_equals_staged_calls"] = same
    ok &= same
    sub = np.unique(np.linspace(0, R - 1, 6).astype(np.int64))
    sub_poses = np.ascontiguousarray(poses[sub].reshape(-1, 3))
    st = np.empty(len(sub_poses) * B, np.uint16)
    rr = np.empty(len(sub_poses) * B, np.float32)
    meth.calc_range_fan(sub_poses, rr, w.fov, B, steps=st)
    mean_steps = float(st.astype(np.float64).mean())
    if O is not None:
        om = O.OracleMap.from_gridmap(w.gmap, w.max_range_px)
        want_r = om.rm_fan(sub_poses, w.fov, B, step_coeff=1.0, nthreads=O.max_threads(), want_hits=False, want_steps=False)[0]
        want = [O.is_crashed(want_r[i * n_steps * B:(i + 1) * n_steps * B], B, n_steps, edge, THRESH) for i in range(len(sub))]
        same = bool(np.array_equal(rr, want_r)) and first[sub].tolist() == want
        ver["oracle_subsample"] = same
        ver["oracle_sample"] = "%d roll-outs x %d poses x %d beams: ranges and Car::isCrashed index against the oracle" % (len(sub), n_steps, B)
        ok &= same
    else:
        ver["oracle_subsample"] = "not run (the oracle is loaded by the cpu_baseline leg only)"
    bpr = _alg_bytes("RMGPU", mean_steps, B, 0.0) + 24.0 / B      # (+ the pose the roll-out kernel writes and the march reads)
    pe = pmc_lookup(wl, "RMGPU+rollout", R * n_steps, plan) if pmc_lookup else None
    out = {"mrays_s": round(rays / (ms * 1e-3) / 1e6, 1), "ms_per_step": round(ms, 4),
           "us_per_rollout": round(ms * 1e3 / R, 3),
           "frac": round(bpr * rays / (ms * 1e-3) / HBM_PEAK, 5),
           "frac_hbm": round(pe["bytes"] / (ms * 1e-3) / HBM_PEAK, 5) if pe else None,
           "verified": bool(ok), "verification": ver,
           "config": {"workload": "%s: %s %dx%d, %d roll-outs x %d control steps x %d beams (scripts/mcts.py:202-245)"
                                  % (wl, w.gmap.name, w.gmap.rows, w.gmap.cols, R, n_steps, B),
                      "method": "RMGPU", "schedule": "one synchronous host call per step (rl_car_rollout_check): states / actions in, "
                                                     "int32 crash index per roll-out out",
                      "reduce": "fused Car::isCrashed per %d-pose roll-out" % n_steps,
                      "kernel": plan["name"], "grid": plan["grid"], "crashed_rollouts": int((first >= 0).sum())},
           "steps": steps, "bursts": bursts, "algorithmic_bytes_per_ray": round(bpr, 3), "mean_samples_per_ray": round(mean_steps, 3),
           "traffic_source": pe["profile"] if pe else None, "leg_seconds": None}
    cars.close()
    meth.close()
    omap.close()
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
    return out


def other_configs(torch, dev, device_index, O=None, pmc_lookup=None, only=None, budget_s=40.0, streams=None):
    """Run the side legs; one that raises is reported as {"error": ...} (not a verification failure), one that runs
    and mismatches carries verified: false — bench.py gates its exit code on that."""
    res = {}
    cache = {}
    t0 = time.perf_counter()
    for spec in LEGS:
        if only and spec[0] not in only:
            continue
        if time.perf_counter() - t0 > budget_s:
            res[spec[0]] = {"skipped": "time budget of %.0f s for the side legs used up" % budget_s}
            continue
        try:
            if spec[6] == "rollout":
                res[spec[0]] = run_rollout_leg(spec, torch, dev, device_index, O=O, pmc_lookup=pmc_lookup)
            else:
                res[spec[0]] = run_leg(spec, torch, dev, device_index, O=O, pmc_lookup=pmc_lookup, cache=cache, streams=streams)
        except Exception as e:                      # noqa: BLE001 — a side leg must not take the headline down
            res[spec[0]] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return res
