#!/usr/bin/env python3
"""bench.py — million rays/s of the batched lidar scan on MI355X, with roofline and CPU baseline.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; prints ONE JSON line on rank 0.
For N>1 the driver may launch it under ``torch.distributed.run`` (one rank per GPU, RCCL); started
bare with ``--gpus N`` it spawns that launcher itself (as a child process, before anything in this
process touches the GPU) and exits with the children's return code.

A "step" is one pass of the hot path over one synthetic pose batch: the fan-expanding ray march of
``n_poses x num_rays`` rays (ScanSimulator2D.scanMany -> calc_range_many,
/root/reference/scripts/scan_simulator.py:113-135) with poses and ranges resident in HBM.
Consecutive steps are enqueued round robin on ``--pipeline`` streams (default: 4 for batches up to
32768 poses) so that step k+1 fills the CUs step k's last long rays leave idle; every step is complete (and, for N>1, every
all-gather) before the clock stops.  ``--pipeline 1`` is the strictly serial schedule.
For N>1 every rank scans its own ``n_poses`` block (weak scaling) and the ranges are all-gathered
over xGMI, chunk by chunk, overlapped with the marches of the following steps (BASELINE.json
north_star); the figure for the reduced exchange (fused crash test, all-gather of the int32 crash
indices) rides along as ``crash_mode``.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)

KERNEL_OF = {"RM": "rm_fan_stream_kernel", "RMGPU": "rm_fan_stream_kernel", "BL": "bl_fan_stream_kernel",
             "GLT": "lut_fan_lds_kernel", "CDDT": "cddt_fan_bins_kernel"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--workload", default="cfg2",
                    help="cfg2 (default: 2049^2 maze, 4096x1081, RMGPU) | cfg3 | cfg4 | cfg5")
    ap.add_argument("--poses", type=int, default=0, help="poses per GPU (0 = workload default)")
    ap.add_argument("--method", default="", help="override: RM | RMGPU | BL | CDDT | GLT")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="steps in flight: consecutive steps go round robin to this many concurrent "
                         "streams (1 = serial: step k+1 starts after step k's last ray; 0 = auto: 4, "
                         "except ray-marching batches above 32768 poses, which fill the machine on their "
                         "own and lose L2 locality when two of them are co-resident)")
    ap.add_argument("--grid-mult", type=int, default=0,
                    help="workgroups (x256 threads) per CU of one launch; 0 = 3 when pipelined (launches of "
                         "0.75 workgroups of 1024 per CU, two rays per lane: several launches co-resident on "
                         "every CU), the library default 8 when serial")
    ap.add_argument("--chunks", type=int, default=0,
                    help="all-gather chunks per step (N>1); 0 = auto: 1 when steps are pipelined (the gather of "
                         "step k overlaps the marches of the following steps; every extra collective costs "
                         "~29 us of host time), 4 on the serial schedule (gather of chunk k overlaps march k+1)")
    ap.add_argument("--gather", default="ranges", choices=["ranges", "crash", "none"],
                    help="N>1 exchange per step: 'ranges' (default) = all-gather of every range, 4 B/ray "
                         "(BASELINE.json north_star); 'crash' = fused per-roll-out crash test, all-gather "
                         "of the int32 crash indices (what MCTS.rollout consumes); 'none' = shards stay put")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="'crash' mode: steps per all-gather bucket")
    ap.add_argument("--no-crash-line", action="store_true",
                    help="N>1: skip the extra timed loop that measures the 'crash' exchange")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (tuning)")
    ap.add_argument("--opt", action="append", default=[], help="kernel option name=int (tuning)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--dist-single", action="store_true",
                    help="with --gpus 1: still create the process group (one rank) and all-gather the ranges "
                         "through it — the RCCL code path on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true",
                    help="dry run of the N>1 code path on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    return ap.parse_args()


def spawn_ranks(a):
    """``bench.py --gpus N`` started bare: run the N ranks as children (fresh processes under
    torch.distributed.run), relay their output, exit with their return code.  Nothing in THIS
    process has touched the GPU (no torch import yet), and nothing is exec'ed."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def make_method(range_libc, omap, w, method):
    if method == "RM":
        return range_libc.PyRayMarching(omap, w.max_range_px)
    if method == "RMGPU":
        return range_libc.PyRayMarchingGPU(omap, w.max_range_px)
    if method == "BL":
        return range_libc.PyBresenhamsLine(omap, w.max_range_px)
    if method == "CDDT":
        return range_libc.PyCDDTCast(omap, w.max_range_px, w.theta_disc or 108)
    if method == "GLT":
        return range_libc.PyGiantLUTCast(omap, w.max_range_px, w.theta_disc or 1442)
    raise SystemExit("unknown method %r" % method)


def algorithmic_bytes_per_ray(method, mean_steps, num_rays, w):
    """bytes a ray must move, per kernel family."""
    pose = 12.0 / num_rays
    if method in ("RM", "RMGPU"):
        return mean_steps * 4.0 + 4.0 + pose            # S̄ EDT samples (f32) + range out + pose
    if method == "BL":
        win = 2 * w.max_range_px + 1
        return (win * win / 8.0) / num_rays + 4.0 + pose  # bit-packed window per pose + out
    if method == "GLT":
        return 2.0 + 4.0 + pose                          # one u16 entry + range out
    if method == "CDDT":
        return 4.0 * max(1.0, mean_steps) + 8.0 + 4.0 + pose
    return 4.0 + pose


def cpu_baseline(w, gmap, poses_all, method, seconds):
    """Oracle (kind "port": range_libc's CPU classes are absent from the reference mount) timed
    on this host's cores over a bounded sample of the same poses."""
    from oracle import oracle as O
    om = O.OracleMap.from_gridmap(gmap, w.max_range_px)
    _ = om.dt
    nthr = O.max_threads()
    step = 1.0 if method == "RMGPU" else 0.999
    B = w.num_rays

    def run(poses, nt):
        t = time.perf_counter()
        if method == "BL":
            om.bl_fan(poses, w.fov, B, nthreads=nt)
        else:
            om.rm_fan(poses, w.fov, B, step_coeff=step, nthreads=nt, want_hits=False,
                      want_steps=False)
        return time.perf_counter() - t

    # 1 thread: faithful to range_libc's serial loop; bounded sample
    n1 = min(len(poses_all), 512)
    run(poses_all[:64], 1)
    t1 = run(poses_all[:n1], 1)
    rate1 = n1 * B / t1
    # all cores: repeat the whole batch until ~seconds elapsed (threads ramp up slowly in VMs)
    batch = poses_all[:min(len(poses_all), 16384)]
    reps, tot, t_all = 0, 0, 0.0
    run(batch, nthr)
    while t_all < seconds and reps < 200:
        t_all += run(batch, nthr)
        tot += len(batch) * B
        reps += 1
    rate = tot / t_all
    name = "BresenhamsLine" if method == "BL" else "RayMarching"
    return {"value": round(rate / 1e6, 3), "unit": "Mrays/s", "cores": nthr, "kind": "port",
            "sample": "%s oracle (oracle/rangelib_oracle.c, OpenMP over poses), %d x (%d poses x %d "
                      "beams) of the same workload in %.1f s" % (name, reps, len(batch), B, t_all),
            "single_thread_Mrays_s": round(rate1 / 1e6, 3),
            "single_thread_sample": "%d poses x %d beams" % (n1, B)}


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    import torch
    import torch.distributed as dist
    from pyracecarsimulator_amd import _lib, range_libc, workloads
    from pyracecarsimulator_amd.distributed import ShardedScan, broadcast_map
    from pyracecarsimulator_amd.pipeline import concurrent_streams

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if a.same_device:
        local_rank = 0
        if a.backend == "nccl":      # RCCL refuses two ranks on one device ("Duplicate GPU detected")
            print("bench.py: --same-device runs over gloo (RCCL needs one GPU per rank)", file=sys.stderr)
            a.backend = "gloo"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or a.dist_single:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_single and world == 1:
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            s_.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner with C stdio on stdout, which a pipe delivers at process exit —
        # AFTER the JSON line.  While the process group lives, C-level stdout goes to stderr; it comes
        # back (flushed) for the one line this program owes its caller.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    w = workloads.CONFIGS[a.workload]()
    if a.poses:
        w.n_poses = a.poses
    method = a.method or w.method
    B = w.num_rays

    # map: built on rank 0, broadcast over RCCL (north_star), tables built per GPU
    multi = world > 1 or a.dist_single            # a process group exists
    gmap = w.gmap
    if multi:
        gmap = broadcast_map(w.gmap if rank == 0 else None, 0, dev)
    omap = range_libc.PyOMap(gmap, device=local_rank)
    meth = make_method(range_libc, omap, w, method)
    if a.variant >= 0:
        meth.set_option("variant", a.variant)
    if w.noise_std > 0:
        meth.set_noise(w.noise_std, w.noise_seed, rank * w.n_poses * B)

    # poses: one seeded global batch of world*n_poses, rank r takes block r (weak scaling)
    dt = omap.distance_transform()
    poses_all = workloads.make_global_poses(w, world, dt=dt)
    lo, hi = workloads.shard_range(len(poses_all), rank, world)
    poses = np.ascontiguousarray(poses_all[lo:hi])
    d_poses = torch.from_numpy(poses).to(dev)
    n = len(poses)

    mode = "none" if (a.no_gather or (world == 1 and not a.dist_single)) else a.gather
    # streams that really run concurrently (HIP maps streams onto a few hardware queues)
    P = a.pipeline if a.pipeline > 0 else (1 if (method in ("RM", "RMGPU", "BL") and n > 32768) else 4)
    streams = concurrent_streams(P) if P > 1 else [torch.cuda.current_stream()]
    P = len(streams)
    default_gm = meth.get_info("grid_mult")
    gm = a.grid_mult or (3 if P > 1 else default_gm)
    meth.set_option("grid_mult", gm)
    if P > 1 and method in ("RM", "RMGPU"):
        meth.set_option("slots", 3 if n <= 8192 else 2)   # several rays per lane: what launches in flight want
        #                                           (three pay for small batches only, measured)
    for kv in a.opt:
        k, v = kv.split("=")
        meth.set_option(k, int(v))
    n_chunks = a.chunks or (1 if P > 1 else 4)
    scan = ShardedScan(n, B, dev, n_chunks=n_chunks, gather=(mode == "ranges"), streams=streams,
                       gather_single_rank=a.dist_single)

    # a step = meth.calc_range_fan_device(local poses -> the slot's buffer) per chunk, prepared once
    scan.bind(meth, d_poses.data_ptr(), w.fov)

    # 'crash': the reference's consumer of a scanned batch is Car::isCrashed per roll-out
    # (scripts/racecar_simulator_v2.py:146-167); group = roll-out length (params.yaml:126 uses 200)
    group = next(gsz for gsz in range(min(200, n), 0, -1) if n % gsz == 0)
    n_groups = n // group
    crash_gather = d_edge = None
    if multi and method in ("RM", "RMGPU"):
        from pyracecarsimulator_amd import racecar as RC
        from pyracecarsimulator_amd.distributed import BucketedIndexGather
        edge = RC.edge_distances(B, -w.fov / 2.0, w.fov / B, 0.275, RC.DEFAULT_CAR["width"],
                                 RC.DEFAULT_CAR["wb"])
        d_edge = torch.from_numpy(edge).to(dev)
        # buckets of M steps, double-buffered: the (latency-bound, M x ~100 B) all-gather of bucket b
        # overlaps the marches of bucket b+1 on RCCL's stream
        crash_gather = BucketedIndexGather(n_groups, a.gather_every, dev)
    elif mode == "crash":
        mode = "ranges" if (world > 1 or a.dist_single) else "none"
    cur_stream = torch.cuda.current_stream().cuda_stream

    # untimed diagnostics launch: mean samples per ray (feeds the algorithmic-bytes figure)
    mean_steps = 0.0
    p99_steps = max_steps = 0.0
    if method in ("RM", "RMGPU", "BL"):
        d_steps = torch.empty(n * B, dtype=torch.int16, device=dev)
        meth.calc_range_fan_device(d_poses.data_ptr(), n, w.fov, B, scan.local.data_ptr(),
                                   d_steps_ptr=d_steps.data_ptr(), stream=cur_stream)
        torch.cuda.synchronize()
        st = d_steps.to(torch.int32).bitwise_and(0xFFFF).float()
        mean_steps = float(st.mean().item())
        sub = st[:: max(1, st.numel() // (1 << 20))]            # <= ~1M rays for the quantile
        p99_steps = float(torch.quantile(sub, 0.99).item())
        max_steps = float(st.max().item())
        del d_steps, st, sub

    def crash_step():
        meth.check_collision_groups_device(d_poses.data_ptr(), n_groups, group, w.fov, B,
                                           d_edge.data_ptr(), 0.001,
                                           crash_gather.slot_view().data_ptr(),
                                           scan.slots[0].local.data_ptr(), stream=cur_stream)
        crash_gather.step_done()

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, drain_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        drain_fn(None)
        # HIP events around the K timed steps, none between them (an event per step would put two extra
        # barrier packets between consecutive launches): one before the first step is enqueued — every
        # stream is idle, the barrier has just synchronised the device — and one per stream behind its
        # last step; the region ends with the latest of those.  (torch creates the HIP event at the
        # first record(): done here, outside the timed region.)
        e0 = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
        for e in [e0] + ends:
            e.record()
        barrier()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step_fn()
        drain_fn(ends)
        barrier()
        el = time.perf_counter() - t0
        dev_ms = max(e0.elapsed_time(e) for e in ends) / steps
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, dev_ms

    def scan_drain(ends):
        scan.finish(ends)

    def crash_drain(ends):
        crash_gather.flush()
        if ends is not None:
            for e in ends:
                e.record()                        # (the crash loop runs on the current stream)

    if mode == "crash":
        elapsed, step_ms = timed(crash_step, crash_drain, a.steps, a.warmup)
    else:
        elapsed, step_ms = timed(scan.step, scan_drain, a.steps, a.warmup)

    if a.dist_single and world == 1 and mode == "ranges":
        torch.cuda.synchronize()
        if not torch.equal(scan.global_order(), scan.local):
            raise SystemExit("dist-single: gathered ranges differ from the local ranges")
    rays_per_step = n * B * world
    value = rays_per_step * a.steps / elapsed / 1e6
    bpr = algorithmic_bytes_per_ray(method, mean_steps, B, w)

    out = {
        "metric": "million rays/sec, 1081-beam scans" if B == 1081 else
                  "million rays/sec, %d-beam scans" % B,
        "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded maze + seeded free-space poses; maps/map.pgm is missing from "
                "the reference mount)" if "maze" in gmap.name else "reference map fixture + " + w.pose_note,
        "config": {"workload": w.describe(), "method": method, "poses_per_gpu": n,
                   "global_poses": n * world, "num_rays": B, "fov": w.fov,
                   "max_range_px": w.max_range_px, "map": "%dx%d" % (gmap.rows, gmap.cols),
                   "parallelism": "pose-batch dp%d" % world,
                   "pipeline": "%d steps in flight on %d concurrent streams, grid_mult %d%s" % (
                       P, P, gm, ", %d rays per lane" % (3 if n <= 8192 else 2) if method in ("RM", "RMGPU") else "")
                               if P > 1 else "serial (one stream), grid_mult %d" % gm,
                   "gather": {"none": "none",
                              "ranges": "all-gather ranges (4 B/ray), %d chunks per step, overlapped with "
                                        "the following steps' marches" % len(scan.chunks),
                              "crash": "fused crash test per %d-pose roll-out, all-gather of int32 "
                                       "crash indices in buckets of %d steps" % (group, max(1, a.gather_every))}[mode]},
        "step_ms_avg": round(step_ms, 4),
        "mean_samples_per_ray": round(mean_steps, 3), "p99_samples_per_ray": round(p99_steps, 1),
        "max_samples_per_ray": round(max_steps, 1),
    }
    if multi:
        out["rccl_world"] = world
        out["gather_bytes_per_step"] = {"ranges": 4 * n * B * world, "crash": 4 * n_groups * world,
                                        "none": 0}[mode]
        if crash_gather is not None and mode != "crash" and not a.no_crash_line:
            # the reduced exchange on the same poses, serial schedule (its own timed loop)
            k2 = max(10, a.steps // 4)
            meth.set_option("grid_mult", default_gm)
            meth.set_option("slots", 0)
            el2, _ = timed(crash_step, crash_drain, k2, min(a.warmup, 5))
            out["crash_mode"] = {"value": round(rays_per_step * k2 / el2 / 1e6, 2), "unit": "Mrays/s",
                                 "ms_per_step": round(el2 / k2 * 1e3, 4), "steps": k2,
                                 "gather_bytes_per_step": 4 * n_groups * world,
                                 "what": "fused crash test per %d-pose roll-out, all-gather of the int32 crash "
                                         "indices in buckets of %d steps" % (group, max(1, a.gather_every))}
    if world == 1:
        # the dominant kernel.  `achieved` prices the ALGORITHMIC bytes of one launch against the time one
        # launch takes out of the timed region (HIP events around the K steps / K): with P launches in
        # flight that is the machine time a launch costs, not its begin-to-end span (a kernel trace shows
        # each launch ~P x longer, P of them overlapping).  `serial` is the same kernel alone on an idle
        # machine (library events around the march kernel on extra steps AFTER the timed region — a pair
        # of event records per launch costs ~12 us, so it stays out of `value`): the duration a kernel
        # trace of `--pipeline 1` reports.
        eff_ms = step_ms
        achieved = bpr * n * B / (eff_ms * 1e-3) / 1e9
        meth.set_option("grid_mult", default_gm)
        if method in ("RM", "RMGPU"):
            meth.set_option("slots", 0)
        meth.set_option("timing", 2)
        ks = []
        solo_out = scan.slots[0].local
        for _ in range(min(a.steps, 30)):
            meth.calc_range_fan_device(d_poses.data_ptr(), n, w.fov, B, solo_out.data_ptr(), stream=cur_stream)
            ks.append(meth.last_kernel_ms())
        meth.set_option("timing", 0)
        meth.set_option("grid_mult", gm)
        if P > 1 and method in ("RM", "RMGPU"):
            meth.set_option("slots", 3 if n <= 8192 else 2)
        k_ms = float(np.mean(ks))
        serial_ach = bpr * n * B / (k_ms * 1e-3) / 1e9
        out["kernel_ms_avg"] = round(eff_ms, 4)
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "traffic": _pmc_traffic(a.workload, method),
                           "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes, committed; "
                                             "not measured in this run)",
                           "bytes_per_ray": round(bpr, 3), "kernel": KERNEL_OF[method],
                           "launch_ms": round(eff_ms, 4), "launches_in_flight": P,
                           "serial": {"kernel_ms": round(k_ms, 4), "achieved": round(serial_ach, 2),
                                      "frac": round(serial_ach / HBM_PEAK_GBS, 5),
                                      "what": "the march kernel alone on an idle machine (grid_mult %d)" % default_gm}}
        if method in ("RM", "RMGPU") and mean_steps > 0:
            # the kernel's real limiters, next to the contractual HBM object.  (1) the CU's scattered-gather
            # rate, probed in this run; (2) VALU issue: wave-level VALU instructions per launch (rocprofv3
            # SQ_INSTS_VALU of the several-rays-per-lane kernel, committed) x 4 clocks each on 4 SIMDs per CU
            lanes, clk, ncu = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int(0)
            _lib.check(_lib.lib().rl_probe_gather_rate(local_rank, 46, ctypes.byref(lanes), ctypes.byref(clk),
                                                        ctypes.byref(ncu)))
            peak = lanes.value * ncu.value * clk.value
            # gathered samples: the statement's count less the t = 0 sample of every ray, which the kernel
            # reads once per pose with the pose record (pose_first_step)
            samples = max(mean_steps - 1.0, 0.0) * n * B
            out["roofline_gather"] = {"achieved_samples_per_s": round(samples / (eff_ms * 1e-3), 1),
                                      "peak": round(peak, 1), "frac": round(samples / (eff_ms * 1e-3) / peak, 5),
                                      "serial_frac": round(samples / (k_ms * 1e-3) / peak, 5),
                                      "probe": "%.2f active lanes/clk/CU x %d CUs x %.2f GHz (rl_probe_gather_rate, "
                                               "46 random lanes, this run)" % (lanes.value, ncu.value, clk.value / 1e9)}
            valu = _pmc_traffic(a.workload, method + "/valu_insts")
            if valu and P > 1 and n == workloads.CONFIGS[a.workload]().n_poses:
                floor_ms = valu * 4.0 / (4 * ncu.value * clk.value) * 1e3
                out["roofline_valu"] = {"wave_valu_insts_per_launch": valu, "floor_ms": round(floor_ms, 5),
                                        "frac": round(floor_ms / eff_ms, 5),
                                        "source": "profiles/pmc_traffic.json (SQ_INSTS_VALU, profiles/r02/"
                                                  "r2_pmc_cfg2_slots3_final; not measured in this run)"}
        if rank == 0 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, gmap, poses_all, method, a.cpu_seconds)
    if world > 1 or a.dist_single:
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


def _pmc_traffic(workload, method):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*.json), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("%s/%s" % (workload, method))
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
