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
32768 poses) so that step k+1 fills the CUs step k's last long rays leave idle — and every step in
flight scans its OWN seeded pose batch (P distinct batches, as consecutive MCTS roll-out batches
are); every step is complete (and, for N>1, every exchange) before the clock stops.
``--pipeline 1`` is the strictly serial schedule.

Timing: W warm-up steps, then ``--bursts`` (25) bursts of EXACTLY K steps, each bracketed by a barrier +
device synchronisation on both sides, maximum over ranks; ``value`` is the MEDIAN burst (min / max ride
along), so one line is not one sample of a 0.7-ms region.

Verification (untimed, after the bursts): every slot's range buffer must be bit-equal to a serial,
one-ray-per-lane launch of the same poses, and (in the cpu_baseline leg) a 64-pose subsample bit-equal
to the CPU oracle; the line carries ``"verified": true`` and the process exits non-zero otherwise.

N>1: cfg2 / cfg3 fix the poses per GPU (weak scaling); cfg4 / cfg5 shard BASELINE.json's GLOBAL batch
(2^20 / 262144 poses / N: strong scaling); ``--poses`` always means poses per GPU.  Exchange per step
(``--gather``): ``ranges`` = all-gather of every range over xGMI (BASELINE.json north_star, default),
``ranges_u16`` = the same on 16-bit ranges (lossy, labelled), ``root`` = gather to rank 0 only,
``crash`` = fused per-roll-out crash test + all-gather of the int32 crash indices (what
scripts/mcts.py:237-245 consumes), ``steer`` = Follow-the-Gap per scan + all-gather of the float32
steering angles (scripts/mcts.py:262-267), ``none``.  The reduced modes run on the same pipelined slot
streams as the plain scan (one bucket of results per slot) and also work at N = 1.  Every N>1 line
carries, next to ``value``: ``march_only``, ``crash_mode``, ``steer_mode`` (each its own timed loop)
and ``scaling_model`` (mode -> modelled 8-GPU speed-up from the xGMI ingress bytes).
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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--bursts", type=int, default=25,
                    help="timed bursts of --steps steps; value = the median burst (fewer when one burst "
                         "takes more than ~0.2 s)")
    ap.add_argument("--workload", default="cfg2",
                    help="cfg2 (default: 2049^2 maze, 4096x1081, RMGPU) | cfg3 | cfg4 | cfg5")
    ap.add_argument("--poses", type=int, default=0, help="poses per GPU (0 = workload default)")
    ap.add_argument("--method", default="", help="override: RM | RMGPU | BL | CDDT | GLT")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="steps in flight: consecutive steps go round robin to this many concurrent "
                         "streams (1 = serial: step k+1 starts after step k's last ray; 0 = auto: 4, "
                         "except ray-marching batches above 32768 poses, which fill the machine on their "
                         "own and lose L2 locality when two of them are co-resident, and GiantLUT, whose lone "
                         "launch runs at the HBM rate)")
    ap.add_argument("--grid-mult", type=int, default=0,
                    help="workgroups (x256 threads) per CU of one launch; 0 = 3 when pipelined (launches of "
                         "0.75 workgroups of 1024 per CU, two rays per lane: several launches co-resident on "
                         "every CU), the library default 8 when serial")
    ap.add_argument("--chunks", type=int, default=0,
                    help="exchange chunks per step (N>1); 0 = auto: 1 when steps are pipelined (the gather of "
                         "step k overlaps the marches of the following steps; every extra collective costs "
                         "~29 us of host time), 4 on the serial schedule (gather of chunk k overlaps march k+1)")
    ap.add_argument("--gather", default="ranges", choices=["ranges", "ranges_u16", "root", "crash", "steer", "none"],
                    help="N>1 exchange per step: 'ranges' (default) = all-gather of every range, 4 B/ray "
                         "(BASELINE.json north_star); 'ranges_u16' = the same on 16-bit fixed-point ranges "
                         "(2 B/ray, LOSSY: <= 0.11 mm at 15 m); 'root' = gather to rank 0 only (the reference's "
                         "consumer is one MCTS process); 'crash' = fused per-roll-out crash test, all-gather "
                         "of the int32 crash indices (what MCTS.rollout consumes); 'steer' = Follow-the-Gap per "
                         "scan, all-gather of the float32 steering angles (MCTS's expansion policy); 'none' = "
                         "shards stay put.  crash / steer also run at N = 1 (nothing to exchange)")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="'crash' / 'steer' modes: steps (of one slot) per all-gather bucket")
    ap.add_argument("--no-crash-line", action="store_true",
                    help="N>1: skip the extra timed loops that measure the 'crash' and 'steer' exchanges")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed output verification (tuning sweeps)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the same-batch figure, the end-to-end host latencies and the gather-rate probe")
    ap.add_argument("--theta-disc", type=int, default=0,
                    help="CDDT / GiantLUT: theta_disc of the table (default: the workload's; CDDT 108 — the reference's "
                         "two-player CDDT uses 112, which the driver line's other_configs leg times)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short verified side legs (cfg3 GiantLUT / CDDT, cfg2 crash / steer, a cfg5 shard) that "
                         "the default 1-GPU cfg2 run appends to its line as `other_configs`")
    ap.add_argument("--only-configs", default="", help="comma-separated subset of the side legs (tuning)")
    ap.add_argument("--selftest-corrupt", action="store_true",
                    help="flip one range of the last slot before the verification (tests that `verified` gates)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
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
        return range_libc.PyCDDTCast(omap, w.max_range_px, theta_disc_of(w, method))
    if method == "GLT":
        return range_libc.PyGiantLUTCast(omap, w.max_range_px, theta_disc_of(w, method))
    raise SystemExit("unknown method %r" % method)


_THETA_OVERRIDE = 0          # --theta-disc N (CDDT / GiantLUT): the table's angular resolution instead of the workload's


def theta_disc_of(w, method):
    if _THETA_OVERRIDE and method in ("CDDT", "GLT"):
        return _THETA_OVERRIDE
    # CDDT: 108 ~ the two-player game's 112 (scripts/two_player/rcs_two_player.py:121) on an even bin count
    # per quadrant; GiantLUT: bin spacing = beam spacing
    if method == "CDDT":
        return w.theta_disc if (w.theta_disc and w.method == "CDDT") else 108
    return w.theta_disc or 1442


def algorithmic_bytes_per_ray(method, mean_steps, num_rays, w, cddt_nbar=0.0):
    """bytes a ray must move, per kernel family -> (bytes, how they were counted)."""
    import math
    pose = 12.0 / num_rays
    if method in ("RM", "RMGPU"):
        return (mean_steps * 4.0 + 4.0 + pose,           # S̄ EDT samples (f32) + range out + pose
                "S x 4 + 4 + 12/B with S = %.3f mean samples per ray (this run's diagnostics launch)" % mean_steps)
    if method == "BL":
        win = 2 * w.max_range_px + 1
        return ((win * win / 8.0) / num_rays + 4.0 + pose,  # bit-packed window per pose + out
                "(2R+1)^2/8 bit-packed window per pose / B + 4 + 12/B")
    if method == "GLT":
        return 2.0 + 4.0 + pose, "one u16 table entry + 4 + 12/B"      # one u16 entry + range out
    if method == "CDDT":
        probes = max(1, math.ceil(math.log2(cddt_nbar + 1.0))) if cddt_nbar > 0 else 1
        return (4.0 * probes + 8.0 + 4.0 + pose,
                "4 x ceil(log2(n_bucket + 1)) + 8 + 4 + 12/B with n_bucket = %.1f stored values per non-empty bucket of "
                "this table: %d probes of a per-RAY bisection — the kernels answer one look-up per (pose, table bin), "
                "~%d beams share it, so measured traffic is far below this figure" % (
                    cddt_nbar, probes, max(1, num_rays // max(1, (w.theta_disc or 108)))))
    return 4.0 + pose, "4 + 12/B"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_flags():
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CFLAGS"):
                return "gcc " + line.split(":=", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, gmap, poses_all, method, seconds, check=None):
    """Oracle (kind "port": range_libc's CPU classes are absent from the reference mount) timed
    on this host's cores over a bounded sample of the same poses — RayMarching AND BresenhamsLine,
    the two CPU classes BASELINE.json's north_star names.  Two builds of the same C restatement are timed:
    ``-O3 -march=native`` compiled on this host (quoted as ``value`` when it builds and
    returns the checker's bits on the sample) and the portable ``-O2`` checker build (``o2_build``).
    ``check(om, O)`` (optional) runs the oracle-side output verification with the oracle map this leg has
    loaded anyway."""
    from oracle import oracle as O
    om = O.OracleMap.from_gridmap(gmap, w.max_range_px)
    _ = om.dt
    nthr = O.max_threads()
    step = 1.0 if method == "RMGPU" else 0.999
    B = w.num_rays

    def cast(kind, poses, nt, native):
        if kind == "BL":
            return om.bl_fan(poses, w.fov, B, nthreads=nt, native=native)[0]
        return om.rm_fan(poses, w.fov, B, step_coeff=step, nthreads=nt, want_hits=False, want_steps=False,
                         native=native)[0]

    def run(kind, poses, nt, native):
        t = time.perf_counter()
        cast(kind, poses, nt, native)
        return time.perf_counter() - t

    have_native = O.native_lib() is not None
    if have_native:
        # the optimised build must return the checker build's bits (same source, same arithmetic switches)
        have_native = all(np.array_equal(cast(k, poses_all[:48], nthr, True), cast(k, poses_all[:48], nthr, False))
                          for k in ("RM", "BL"))

    def measure(kind, secs, n1, native):
        # 1 thread: faithful to range_libc's serial loop; bounded sample
        run(kind, poses_all[:32], 1, native)
        n1 = min(len(poses_all), n1)
        t1 = run(kind, poses_all[:n1], 1, native)
        # all cores: repeat a batch until ~secs elapsed (threads ramp up slowly in VMs)
        batch = poses_all[:min(len(poses_all), 16384 if kind != "BL" else 2048)]
        reps, tot, t_all = 0, 0, 0.0
        run(kind, batch, nthr, native)
        while t_all < secs and reps < 200:
            t_all += run(kind, batch, nthr, native)
            tot += len(batch) * B
            reps += 1
        return (tot / t_all / 1e6, n1 * B / t1 / 1e6,
                "%d x (%d poses x %d beams) in %.1f s; 1 thread: %d poses" % (reps, len(batch), B, t_all, n1))

    primary = "BL" if method == "BL" else "RM"
    name = {"RM": "RayMarching", "BL": "BresenhamsLine"}
    n1 = {"RM": 512, "BL": 64}
    flags_o2 = oracle_flags()
    share = 0.6 if have_native else 1.0
    rate, rate1, sample = measure(primary, seconds * share, n1[primary], have_native)
    out = {"value": round(rate, 3), "unit": "Mrays/s", "cores": nthr, "kind": "port",
           "sample": "%s oracle (oracle/rangelib_oracle.c, OpenMP over poses), %s of the same workload"
                     % (name[primary], sample),
           "single_thread_Mrays_s": round(rate1, 3), "cpu_model": cpu_model(),
           "flags": ("gcc " + O.NATIVE_FLAGS + " (built on this host)") if have_native else flags_o2}
    if have_native:
        r_o2, r1_o2, s_o2 = measure(primary, seconds * 0.4, n1[primary], False)
        out["o2_build"] = {"value": round(r_o2, 3), "single_thread_Mrays_s": round(r1_o2, 3), "flags": flags_o2,
                           "sample": s_o2, "what": "the portable checker build of the same source, timed beside it "
                                                   "(bit-identical results on the sample)"}
    else:
        out["native_build"] = "not available on this host (gcc -O3 -march=native failed or returned different bits)"
    other = "RM" if primary == "BL" else "BL"
    r2, r21, s2 = measure(other, max(2.0, seconds / 3.0), n1[other], have_native)
    out["bresenham" if other == "BL" else "raymarching"] = {
        "value": round(r2, 3), "unit": "Mrays/s", "cores": nthr, "single_thread": round(r21, 3),
        "sample": "%s oracle, %s" % (name[other], s2)}
    if check is not None:
        out["_verification"] = check(om, O)
    return out


def pmc_entry(workload, method, n, plan, mode=None):
    """The committed rocprofv3 PMC pass of EXACTLY this launch shape (profiles/pmc_traffic.json), or None:
    workload, method, poses per launch, kernel (all template arguments, as the trace prints the symbol), grid and the
    exchange mode that changes the kernel's stores ("steer": plain range stores for FollowGap) must all match."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            entries = json.load(f).get("entries", [])
    except (OSError, ValueError):
        return None
    for e in entries:
        if (e.get("workload") == workload and e.get("method") == method and e.get("poses") == n and
                e.get("kernel") == plan["name"] and e.get("grid") == plan["grid"] and e.get("mode") == mode):
            return e
    return None


def main():
    a = parse_args()
    global _THETA_OVERRIDE
    _THETA_OVERRIDE = max(0, a.theta_disc)
    for kv in list(a.opt):                       # (`--opt variant=3` is `--variant 3`: the checker's statement follows it)
        if kv.startswith("variant="):
            a.variant = int(kv.split("=")[1])
            a.opt.remove(kv)
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
    from pyracecarsimulator_amd.distributed import (ShardedScan, broadcast_map, XGMI_LINKS, XGMI_LINK_GBS,
                                                    exchange_bytes as _exchange_bytes, scaling_model)
    from pyracecarsimulator_amd.pipeline import concurrent_streams

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if a.same_device:
        local_rank = 0
        if a.backend == "nccl":      # RCCL refuses two ranks on one device ("Duplicate GPU detected")
            print("bench.py: --same-device runs over gloo (RCCL needs one GPU per rank)", file=sys.stderr)
            a.backend = "gloo"
    elif world > 1 and torch.cuda.device_count() < world:
        # (device_count() does not initialise the GPU.)  One rank per GPU or nothing: a rank that silently shares a
        # device would report an N-GPU number measured on fewer GPUs
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible — one GPU per rank is required "
                         "(--same-device --backend gloo is the labelled dry run of the N>1 code path on one GPU)"
                         % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or a.dist_single            # a process group exists
    saved_stdout = None
    if multi:
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
    n, n_global, scaling = workloads.batch_layout(w, world, a.poses)
    w.n_poses = n
    method = a.method or w.method
    B = w.num_rays
    lo, _hi = workloads.shard_range(n_global, rank, world)

    # map: built on rank 0, broadcast over RCCL (north_star), tables built per GPU
    gmap = w.gmap
    if multi:
        gmap = broadcast_map(w.gmap if rank == 0 else None, 0, dev)
    omap = range_libc.PyOMap(gmap, device=local_rank)
    meth = make_method(range_libc, omap, w, method)
    if a.variant >= 0:
        meth.set_option("variant", a.variant)
    # the arithmetic in effect: "RM" (range_libc's CPU RayMarching) defaults to the upstream-literal form, variant 3
    variant_eff = meth.get_info("variant") if method in ("RM", "RMGPU") else a.variant
    if w.noise_std > 0:
        meth.set_noise(w.noise_std, w.noise_seed, lo * B)    # keyed by the GLOBAL ray id: shard-invariant
    max_range_m = w.max_range_px * gmap.resolution

    mode = "none" if a.no_gather else a.gather
    if not multi and mode in ("ranges", "ranges_u16", "root"):
        mode = "none"                             # (one GPU: nothing to exchange; crash / steer still reduce)
    is_rm = method in ("RM", "RMGPU")
    # streams that really run concurrently (HIP maps streams onto a few hardware queues); a handle keeps
    # rl_launch_contexts() per-stream scratch sets — more streams than that would silently serialise
    n_ctx = int(_lib.lib().rl_launch_contexts())
    # (GiantLUT: a lone launch streams its rows at the HBM rate since they are read with non-temporal loads; launches
    #  sharing the machine only get in each other's way: 838 serial, 771 / 726 Grays/s with 2 / 4 in flight)
    P = a.pipeline if a.pipeline > 0 else (1 if ((method in ("RM", "RMGPU", "BL") and n > 32768) or method == "GLT") else 4)
    if P > n_ctx:
        print("bench.py: --pipeline %d clamped to the library's %d launch contexts" % (P, n_ctx), file=sys.stderr)
        P = n_ctx
    streams = concurrent_streams(P) if P > 1 else [torch.cuda.current_stream()]
    P = len(streams)
    default_gm = meth.get_info("grid_mult")
    gm = a.grid_mult or (3 if P > 1 else default_gm)
    # (two rays per lane: since the waves compact their last rays — DESIGN.md section 4 — two beat three at
    # every batch but 8192 poses, where three lead by 2 %: profiles/r03/ab_slots.txt)
    pipe_slots = 2 if (P > 1 and is_rm) else 0

    def apply_schedule(pipelined: bool):
        meth.set_option("grid_mult", gm if pipelined else default_gm)
        if is_rm:
            meth.set_option("slots", pipe_slots if pipelined else 0)
        if pipelined:
            for kv in a.opt:
                k, v = kv.split("=")
                meth.set_option(k, int(v))

    apply_schedule(True)

    # poses: P seeded batches — every step in flight scans its own (slot k always batch k); rank r takes
    # block r of each global batch, generated on its own GPU
    dt = omap.distance_transform()
    batches = [workloads.rank_poses(w, n_global, rank, world, dt=dt, seed=w.pose_seed + 7919 * k, device=local_rank)
               for k in range(P)]
    d_poses = [torch.from_numpy(b).to(dev) for b in batches]
    del dt
    pose_ptrs = [t.data_ptr() for t in d_poses]

    n_chunks = a.chunks or (1 if P > 1 else 4)
    # the reduced exchanges: what the reference's consumers of a scanned batch read.  'crash': Car::isCrashed
    # per roll-out (scripts/racecar_simulator_v2.py:146-167; roll-out length params.yaml:126 = 200);
    # 'steer': FollowGap per scan (scripts/mcts.py:97-99,262-267)
    group = next(gsz for gsz in range(min(200, n), 0, -1) if n % gsz == 0)
    n_groups = n // group
    from pyracecarsimulator_amd import racecar as RC
    from pyracecarsimulator_amd.followgap import PyFollowGap
    edge = RC.edge_distances(B, -w.fov / 2.0, w.fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    d_edge = torch.from_numpy(edge).to(dev)
    CRASH_THRESH = 0.001                          # params.yaml:47
    fgap = PyFollowGap(10, 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004, device=local_rank) if B >= 10 else None
    if mode == "steer" and fgap is None:
        raise SystemExit("--gather steer needs at least 10 beams per scan")

    nt_default = meth.get_info("nt_store")          # (after --opt: the A/B of the non-temporal range stores)

    def make_scan(md):
        """A ShardedScan of this run's shape in exchange mode ``md``, bound to the method and the P batches."""
        meth.set_option("nt_store", nt_default)     # (bind_steer clears it for its own mode)
        if md in ("crash", "steer"):
            sc = ShardedScan(n, B, dev, n_chunks=1, gather=True, streams=streams, gather_single_rank=a.dist_single,
                             mode=md, n_items=n_groups if md == "crash" else n, every=a.gather_every)
            if md == "crash":
                sc.bind_crash(meth, pose_ptrs, w.fov, group, d_edge.data_ptr(), CRASH_THRESH)
            else:
                sc.bind_steer(meth, fgap, pose_ptrs, w.fov)
            return sc
        sc = ShardedScan(n, B, dev, n_chunks=n_chunks, gather=md in ("ranges", "ranges_u16", "root"),
                         streams=streams, gather_single_rank=a.dist_single,
                         mode=md if md in ("ranges", "ranges_u16", "root") else "ranges", root=0,
                         max_range_m=max_range_m)
        sc.bind(meth, pose_ptrs, w.fov, noise=(w.noise_std, w.noise_seed, lo * B) if w.noise_std > 0 else None)
        return sc

    scan = make_scan(mode)
    plan = meth.plan_fan(n, B, crash=(mode == "crash" and is_rm))   # what a step launches (kernel, grid)
    cur_stream = torch.cuda.current_stream().cuda_stream

    # untimed diagnostics launch: mean samples per ray (feeds the algorithmic-bytes figure)
    mean_steps = p99_steps = max_steps = 0.0
    if method in ("RM", "RMGPU", "BL"):
        d_steps = torch.empty(n * B, dtype=torch.int16, device=dev)
        meth.calc_range_fan_device(d_poses[0].data_ptr(), n, w.fov, B, scan.slots[0].local.data_ptr(),
                                   d_steps_ptr=d_steps.data_ptr(), stream=cur_stream)
        torch.cuda.synchronize()
        st = d_steps.to(torch.int32).bitwise_and(0xFFFF).float()
        mean_steps = float(st.mean().item())
        sub = st[:: max(1, st.numel() // (1 << 20))]            # <= ~1M rays for the quantile
        p99_steps = float(torch.quantile(sub, 0.99).item())
        max_steps = float(st.max().item())
        del d_steps, st, sub

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def burst(step_fn, drain_fn, steps, with_events=False):
        """EXACTLY ``steps`` steps between two barrier + synchronise brackets: (wall seconds, maximum
        over ranks; device ms per step from HIP events on the slot streams, or None)."""
        # The bursts `value` comes from carry NO HIP event: an event record is a barrier packet on its stream plus
        # ~2 us of host time, five of them (one in front, one per slot stream behind its last step) cost a 20-step
        # burst ~25 us of its ~600 (tools/r04/burst_overhead.py: an EMPTY bracket with the five records takes
        # 31 us, without them 6) — timing instrumentation, not the hot path.  The device-time figure
        # (`roofline.launch_ms`, `frac_device`) comes from separate bursts WITH the events (with_events=True): one
        # before the first step is enqueued — every stream is idle, the barrier has just synchronised the
        # device — and one per stream behind its last step; the region ends with the latest of those.  (torch
        # creates the HIP event at the first record(): done outside the timed region.)
        e0 = ends = None
        if with_events:
            e0 = torch.cuda.Event(enable_timing=True)
            ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
            for e in [e0] + ends:
                e.record()
        barrier()
        t0 = time.perf_counter()
        if with_events:
            e0.record()
        for _ in range(steps):
            step_fn()
        drain_fn(ends)
        barrier()
        el = time.perf_counter() - t0
        dev_ms = (max(e0.elapsed_time(e) for e in ends) / steps) if with_events else None
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, dev_ms

    def timed(step_fn, drain_fn, steps, warmup, bursts):
        """W warm-up steps, then bursts of K steps; returns the per-burst (seconds, device ms per step or None):
        the wall-clock bursts first (no events inside), then up to 5 more with HIP events for the device time."""
        for _ in range(max(warmup, P)):             # (at least one step per slot: verification reads them all)
            step_fn()
        drain_fn(None)
        first = burst(step_fn, drain_fn, steps)
        n_b = bursts if first[0] < 0.2 else max(3, min(bursts, int(2.0 / first[0])))
        runs_ = [first] + [burst(step_fn, drain_fn, steps) for _ in range(n_b - 1)]
        return runs_ + [burst(step_fn, drain_fn, steps, with_events=True) for _ in range(min(5, n_b))]

    def timed_scan(sc, steps, warmup, bursts):
        return timed(sc.step, sc.finish, steps, warmup, bursts)

    def summarise(runs, steps, rays_per_step):
        els = sorted(r[0] for r in runs if r[1] is None)         # the wall-clock bursts (no events inside)
        med = els[len(els) // 2] if len(els) % 2 else 0.5 * (els[len(els) // 2 - 1] + els[len(els) // 2])
        devs = sorted(r[1] for r in runs if r[1] is not None)    # the extra bursts with HIP events
        dev_med = devs[len(devs) // 2]
        v = lambda el: rays_per_step * steps / el / 1e6      # noqa: E731
        return {"value": v(med), "min": v(els[-1]), "max": v(els[0]), "ms_per_step": med / steps * 1e3,
                "dev_ms": dev_med, "bursts": len(els)}

    rays_per_step = n * B * world
    runs = timed_scan(scan, a.steps, a.warmup, a.bursts)
    res = summarise(runs, a.steps, rays_per_step)
    step_ms = res["dev_ms"]

    # ---------------------------------------------------------------- verification (untimed)
    # (1) every slot's buffer == a serial, one-ray-per-lane, whole-machine launch of the same poses;
    # (2) N>1: what was gathered == what the ranks hold; (3) with the CPU baseline: oracle subsample;
    # (4) crash / steer: every slot's last reduced result == the same reduction of the serial launch's ranges
    #     (crash: Car::isCrashed restated on the device in float64; steer: the FollowGap kernel on those ranges)
    #     on every rank's rows of what was exchanged.
    verification = {}
    ok = True
    d_ref = None
    literal_mode = {}           # the literal-arithmetic timing of the extras, quoted in verification.upstream_literal

    def reduced_reference(md, ranges):
        """The reduction of mode ``md`` over a (n*B,) float32 range tensor, independent of the fused path."""
        if md == "crash":
            hit = ((ranges.view(n, B).double() - d_edge.view(1, B)) < CRASH_THRESH).any(dim=1).view(n_groups, group)
            first = torch.where(hit.any(dim=1), hit.to(torch.int32).argmax(dim=1).to(torch.int32),
                                torch.full((n_groups,), -(group + 1), dtype=torch.int32, device=dev))
            return first
        ang = torch.empty(n, dtype=torch.float32, device=dev)
        fgap.eval_many_device(ranges.data_ptr(), n, B, ang.data_ptr(), stream=cur_stream)
        torch.cuda.synchronize()
        return ang

    def check_reduced(sc, md, refs):
        """refs[k]: the serial-launch ranges of slot k's batch (or None).  Returns (ok, detail)."""
        sc.finish()
        torch.cuda.synchronize()
        good = True
        for k, sl in enumerate(sc.slots):
            r = sc.results(sl)
            if r is None or refs[k] is None:
                continue
            want = reduced_reference(md, refs[k])
            mine = r[rank if r.shape[0] > 1 else 0, -1]
            good &= bool(torch.equal(mine, want))
            if sc.exchange and world > 1:
                allw = torch.empty((world,) + tuple(want.shape), dtype=want.dtype, device=dev)
                dist.all_gather_into_tensor(allw.view(-1), want.contiguous())
                good &= bool(torch.equal(r[:, -1], allw))
        if multi:
            flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            good = bool(flag.item())
        return good

    def reset_noise():
        # (a chunked noisy scan leaves the method at its last chunk's ray offset: ShardedScan.bind(noise=...))
        if w.noise_std > 0:
            meth.set_noise(w.noise_std, w.noise_seed, lo * B)

    slot_refs = [None] * P
    if not a.no_verify:
        torch.cuda.synchronize()
        reset_noise()
        if a.selftest_corrupt:
            scan.slots[-1].local[n * B // 2] += 1.0
        apply_schedule(False)
        if method in ("RM", "RMGPU"):
            meth.set_option("slots", 1)             # (auto may take two rays per lane: launch_plan.h)
        d_ref = torch.empty(n * B, dtype=torch.float32, device=dev)
        bad = []
        keep_refs = scan.reduced or (multi and not a.no_crash_line)
        for k, sl in enumerate(scan.slots):
            if keep_refs and k > 0 and n * B <= (1 << 27):
                d_ref = torch.empty(n * B, dtype=torch.float32, device=dev)
            meth.calc_range_fan_device(d_poses[k].data_ptr(), n, w.fov, B, d_ref.data_ptr(), stream=cur_stream)
            torch.cuda.synchronize()
            if not torch.equal(sl.local, d_ref):
                bad.append((k, int((sl.local != d_ref).sum().item())))
            if keep_refs and (k == 0 or n * B <= (1 << 27)):
                slot_refs[k] = d_ref
        ref_plan = meth.last_plan()
        verification["slots_equal_serial_launch"] = not bad
        verification["serial_launch"] = ref_plan["name"] + " grid %d" % ref_plan["grid"]
        if bad:
            ok = False
            verification["slots_differing"] = bad
        if scan.gather:
            # every rank's block of what this rank gathered against the owner's local buffer: exact for
            # 'ranges' / 'root', within the quantisation step for 'ranges_u16'
            g_ok = True
            for sl in scan.slots:
                mine = sl.local.view(torch.int32).to(torch.int64).sum().reshape(1)
                sums = torch.zeros(world, dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(sums, mine)
                if sl.gathered is None:
                    continue
                g = scan.global_order(sl).view(world, -1)
                if mode == "ranges_u16":
                    own = g[rank]
                    g_ok &= bool(((own - sl.local.clamp(0.0, max_range_m)).abs().max() <= max_range_m / 131070 * 1.01 + 2e-6).item())
                else:
                    got = g.view(torch.int32).to(torch.int64).sum(dim=1)
                    g_ok &= bool(torch.equal(got, sums))
            flag = torch.tensor([1 if g_ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verification["gathered_equals_local"] = bool(flag.item())
            ok &= bool(flag.item())
        if scan.reduced and not a.selftest_corrupt:
            good = check_reduced(scan, mode, slot_refs)
            verification["%s_results_equal_reference" % mode] = good
            ok &= good
        apply_schedule(True)

    def oracle_check(om, O):
        """64 poses of slot 0's batch: the device ranges (noise off) bit-equal to the CPU oracle."""
        sub = np.unique(np.linspace(0, n - 1, 64 if method != "GLT" else 16).astype(np.int64))
        poses = np.ascontiguousarray(batches[0][sub])
        nthr = O.max_threads()
        got = np.empty(len(sub) * B, np.float32)
        if w.noise_std > 0:
            meth.set_noise(0.0, 0, 0)
        try:
            for pipelined in (True, False):
                apply_schedule(pipelined)
                meth.calc_range_fan(poses, got, w.fov, B)
                if method in ("RM", "RMGPU") and variant_eff == 3:
                    # --variant 3: the upstream-literal arithmetic is what is timed — its statement is the checker's libm form
                    want = om.rm_fan_libm(poses, w.fov, B, step_coeff=1.0 if method == "RMGPU" else 0.999)[0]
                elif method in ("RM", "RMGPU"):
                    want = om.rm_fan(poses, w.fov, B, step_coeff=1.0 if method == "RMGPU" else 0.999, nthreads=nthr,
                                     want_hits=False, want_steps=False)[0]
                elif method == "BL":
                    want = om.bl_fan(poses, w.fov, B, nthreads=nthr)[0]
                elif method == "CDDT":
                    want = om.cddt_fan(theta_disc_of(w, method), poses, w.fov, B, nthreads=nthr)
                else:
                    # GiantLUT: the table (11.5 GB at cfg3) does not fit the host — the oracle's fan query reads the
                    # DEVICE table rows of the sampled poses' cells (the table itself is pinned by tests/)
                    td = theta_disc_of(w, method)
                    rr, cc = om.lut_pose_cells(poses)
                    rows = np.empty((len(sub), td), np.uint16)
                    cache = {}
                    for i, (r_, c_) in enumerate(zip(rr, cc)):
                        if int(r_) not in cache:
                            cache[int(r_)] = meth.table(int(r_), int(r_) + 1)[0]
                        rows[i] = cache[int(r_)][int(c_)]
                    want = om.lut_fan_rows(rows, poses, w.fov, B)
                if not np.array_equal(got, want):
                    return {"oracle_subsample": False, "differing_rays": int((got != want).sum())}
            if d_ref is not None and w.noise_std <= 0:
                # the serial full-batch launch above (== every slot's buffer) at the same poses
                apply_schedule(False)
                meth.calc_range_fan_device(d_poses[0].data_ptr(), n, w.fov, B, d_ref.data_ptr(), stream=cur_stream)
                torch.cuda.synchronize()
                rows_ = d_ref.view(n, B)[torch.from_numpy(sub).to(dev)].reshape(-1).cpu().numpy()
                if not np.array_equal(rows_, want):
                    return {"oracle_subsample": False, "differing_rays_full_batch": int((rows_ != want).sum())}
        finally:
            if w.noise_std > 0:
                meth.set_noise(w.noise_std, w.noise_seed, lo * B)
            apply_schedule(True)
        extra = {}
        if method in ("CDDT", "GLT", "BL"):
            # the variants' tolerance is "<= 1 cell vs oracle RM" — the distribution of the timed
            # method's error against EXACT ray marching on the same subsample goes on the line
            import bench_legs
            exact = om.rm_fan(poses, w.fov, B, step_coeff=1.0, nthreads=nthr, want_hits=False, want_steps=False)[0]
            extra["vs_exact_rm"] = dict(bench_legs.error_stats_cells(got, exact, gmap.resolution),
                                        what="|range - exact ray marching (oracle rm_fan, coefficient 1.0)| in cells, "
                                             "%d poses x %d beams of batch 0" % (len(sub), B))
        if method in ("RM", "RMGPU") and not a.selftest_corrupt and variant_eff != 3:
            # parity on the record (range_libc is absent: the oracle is UNPINNED): the same subsample
            # through the AUDIT mode (variant 3: upstream-literal arithmetic, glibc sinf / cosf on the device) must equal
            # the oracle's libm form bit for bit, and the line states how far the canonical default is from it
            sc = 1.0 if method == "RMGPU" else 0.999
            lit_r, lit_h = np.empty(len(sub) * B, np.float32), np.empty((len(sub) * B, 2), np.int32)
            can_h = np.empty_like(lit_h)
            if w.noise_std > 0:
                meth.set_noise(0.0, 0, 0)
            try:
                meth.calc_range_fan(poses, got, w.fov, B, hit_cells=can_h)
                meth.set_option("variant", 3)
                meth.calc_range_fan(poses, lit_r, w.fov, B, hit_cells=lit_h)
            finally:
                meth.set_option("variant", variant_eff)
                if w.noise_std > 0:
                    meth.set_noise(w.noise_std, w.noise_seed, lo * B)
            ref_r, ref_h, _ = om.rm_fan_libm(poses, w.fov, B, step_coeff=sc)
            if not (np.array_equal(lit_r, ref_r) and np.array_equal(lit_h, ref_h)):
                return {"oracle_subsample": False, "audit_mode_differing_rays": int((lit_r != ref_r).sum())}
            moved = (can_h != lit_h).any(axis=1)
            extra["upstream_literal"] = {
                "audit_mode": "variant 3 (range_libc's CPU arithmetic stated literally, glibc sinf / cosf) == the oracle's "
                              "libm form on %d rays: ranges and hit cells bit-equal" % lit_r.size,
                # the same schedule as `value` with the literal arithmetic (rm_fan_stream_kernel<.., LIT>), timed above
                "audit_mode_mrays_s": literal_mode.get("value"), "audit_mode_vs_canonical": literal_mode.get("vs_canonical"),
                "audit_mode_kernel": literal_mode.get("kernel"),
                "canonical_default_vs_literal": {"rays": int(lit_r.size), "rays_with_another_hit_cell": int(moved.sum()),
                                                 "max_range_difference_cells": round(float(np.abs(got - lit_r).max() / gmap.resolution), 4),
                                                 "ranges_bit_equal": int((got == lit_r).sum())},
                "oracle": "UNPINNED: range_libc is absent from the reference mount (restated from SURVEY.md Appendix A)"}
        if mode == "steer" and w.noise_std <= 0 and not a.selftest_corrupt:
            # the steering angles the timed loop left for batch 0 against FollowGap::eval restated on the CPU
            # (oracle/: bit-identical to the reference's compiled header) over the oracle's ranges
            scan.finish()
            torch.cuda.synchronize()
            r = scan.results(scan.slots[0])
            mine = r[rank if r.shape[0] > 1 else 0, -1].cpu().numpy()[sub]
            wr = want.reshape(len(sub), B)
            ref = np.array([O.followgap_eval(wr[i], 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004)
                            for i in range(len(sub))], np.float32)
            if not np.array_equal(mine, ref):
                return {"oracle_subsample": False, "differing_steering_angles": int((mine != ref).sum())}
            extra["oracle_followgap"] = "%d steering angles of batch 0 bit-equal to the CPU FollowGap" % len(sub)
        return dict({"oracle_subsample": True,
                     "oracle_sample": "%d poses x %d beams of batch 0, bit-equal (noise off)" % (len(sub), B)}, **extra)

    cddt_nbar = 0.0
    if method == "CDDT":
        # SURVEY.md section 8d prices a CDDT ray at one bisection of its bucket: 4 B x ceil(log2(n_bucket + 1)) with
        # the table's REAL mean bucket size (values / non-empty buckets, read back from the device table)
        vals, nonempty = meth.get_info("cddt_values"), meth.get_info("cddt_nonempty_buckets")
        cddt_nbar = vals / max(nonempty, 1)
    bpr, bpr_note = algorithmic_bytes_per_ray(method, mean_steps, B, w, cddt_nbar)
    out = {
        "metric": "million rays/sec, 1081-beam scans" if B == 1081 else
                  "million rays/sec, %d-beam scans" % B,
        "value": round(res["value"], 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(res["ms_per_step"], 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded maze + seeded free-space poses; maps/map.pgm is missing from "
                "the reference mount)" if "maze" in gmap.name else "reference map fixture + " + w.pose_note,
        "config": {"workload": w.describe(), "method": method, "poses_per_gpu": n,
                   "global_poses": n_global, "num_rays": B, "fov": w.fov,
                   "max_range_px": w.max_range_px, "map": "%dx%d" % (gmap.rows, gmap.cols),
                   "parallelism": "pose-batch dp%d" % world,
                   "pose_batches": "%d distinct seeded batches, one per step in flight" % P,
                   "pipeline": ("%d steps in flight on %d concurrent streams, grid_mult %d%s" % (
                       P, P, meth.get_info("grid_mult"), ", %d rays per lane" % plan["slots"] if method in ("RM", "RMGPU") else "")
                                if P > 1 else "serial (one stream), grid_mult %d" % meth.get_info("grid_mult")),
                   "kernel": plan["name"], "grid": plan["grid"], "block": plan["block"], "binning": plan["binning"],
                   "gather": {"none": "none",
                              "ranges": "all-gather ranges (4 B/ray), %d chunks per step, overlapped with "
                                        "the following steps' marches" % len(scan.chunks),
                              "ranges_u16": "all-gather of 16-bit fixed-point ranges (2 B/ray, LOSSY: <= %.3g mm), "
                                            "%d chunks per step" % (max_range_m / 131070 * 1e3, len(scan.chunks)),
                              "root": "gather of the ranges (4 B/ray) to rank 0 only, %d chunks per step" % len(scan.chunks),
                              "crash": "fused crash test per %d-pose roll-out (Car::isCrashed), all-gather of int32 "
                                       "crash indices in buckets of %d steps per slot" % (group, max(1, a.gather_every)),
                              "steer": "Follow-the-Gap per scan on the slot's stream, all-gather of float32 steering "
                                       "angles (4 B/pose) in buckets of %d steps per slot" % max(1, a.gather_every)}[mode]
                             + ("" if (multi or mode == "none") else " (one GPU: nothing to exchange)")},
        "bursts": res["bursts"], "value_min": round(res["min"], 2), "value_max": round(res["max"], 2),
        "value_is": "median of %d bursts of %d steps, each bracketed by barrier + device synchronisation (no HIP "
                    "event inside these bursts; step_ms_avg / roofline.launch_ms come from extra bursts with events)" % (
            res["bursts"], a.steps),
        "step_ms_avg": round(step_ms, 4),
        "mean_samples_per_ray": round(mean_steps, 3), "p99_samples_per_ray": round(p99_steps, 1),
        "max_samples_per_ray": round(max_steps, 1),
    }
    per_rank = n * B

    def exchange_bytes(md, wd):
        """(bytes all GPUs send per step, bytes the busiest GPU receives) of mode md on wd GPUs, this run's batch."""
        return _exchange_bytes(md, wd, per_rank, n, n_groups)

    def side_leg(md, steps):
        """Mode md on this run's streams, batches and schedule: its own ShardedScan and timed loop."""
        reset_noise()
        sc = make_scan(md)
        runs_ = timed_scan(sc, steps, min(a.warmup, 5), min(a.bursts, 7))
        r_ = summarise(runs_, steps, rays_per_step)
        leg = {"value": round(r_["value"], 2), "unit": "Mrays/s", "ms_per_step": round(r_["ms_per_step"], 4),
               "steps": steps, "bursts": r_["bursts"], "gather_bytes_per_step": exchange_bytes(md, world)[0],
               "schedule": out["config"]["pipeline"]}
        if md in ("crash", "steer") and not a.no_verify and not a.selftest_corrupt:
            leg["verified"] = check_reduced(sc, md, slot_refs)
        else:
            sc.finish()
            barrier()
        del sc
        return leg

    if multi:
        out["rccl_world"] = world
        # what the communicator itself reports, and which device every rank really used (gathered through it)
        out["rccl_ranks"] = dist.get_world_size()
        out["comm_backend"] = dist.get_backend()
        dev_ids = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(world)]
        dist.all_gather(dev_ids, torch.tensor([torch.cuda.current_device()], dtype=torch.int32, device=dev))
        out["rank_devices"] = [int(t.item()) for t in dev_ids]
        if world > 1 and not a.same_device:
            assert len(set(out["rank_devices"])) == world, "ranks share a device: %r" % (out["rank_devices"],)
        gb, ingress = exchange_bytes(mode, world)
        out["gather_bytes_per_step"] = gb
        # xGMI: what one GPU must RECEIVE per step (the busiest one: every GPU for an all-gather, rank 0 for
        # 'root') against 7 links x per-direction link rate
        links = min(XGMI_LINKS, max(world - 1, 1))
        peak = links * XGMI_LINK_GBS
        ach = ingress / (res["ms_per_step"] * 1e-3) / 1e9
        out["roofline_xgmi"] = {"bound": "xgmi", "ingress_bytes_per_gpu_per_step": ingress,
                                "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "GB/s",
                                "frac": round(ach / peak, 5),
                                "peak_is": "%d peer links x %.1f GB/s per direction (~153 GB/s per link both ways)" % (
                                    links, XGMI_LINK_GBS),
                                "floor_ms_per_step": round(ingress / (peak * 1e9) * 1e3, 4),
                                "measured_on": "same-device dry run (no xGMI)" if a.same_device else (
                                    "one rank (no peer)" if world == 1 else "%d GPUs" % world)}
        legs = {mode: {"value": out["value"], "ms_per_step": out["ms_per_step"]}}
        if mode != "none" and not a.no_extras:
            # the same steps with the shards left on their GPUs: what the ranks COMPUTE per second next to
            # `value`, which includes the exchange — a SCALE record then separates the march's scaling from
            # the xGMI bound of the exchange (its own timed loop, after the verified one)
            scan.finish()
            barrier()
            leg = side_leg("none", a.steps)
            leg["what"] = ("the same sharded steps without reduction or exchange (ranges stay on the GPU that "
                           "computed them); `value` includes the exchange")
            out["march_only"] = legs["none"] = leg
        if not a.no_crash_line:
            # the reduced exchanges the reference's consumers need (4 B per roll-out / per pose), each on the
            # same pipelined slot streams as `value`
            if mode != "crash":
                leg = side_leg("crash", a.steps)
                leg["what"] = ("fused crash test per %d-pose roll-out (scripts/mcts.py:237-245 consumes one index per "
                               "roll-out), all-gather of the int32 crash indices in buckets of %d steps per slot"
                               % (group, max(1, a.gather_every)))
                out["crash_mode"] = legs["crash"] = leg
            if mode != "steer" and fgap is not None:
                leg = side_leg("steer", a.steps)
                leg["what"] = ("scan + Follow-the-Gap per scan (scripts/mcts.py:262-267 consumes one steering angle "
                               "per scan), all-gather of the float32 angles in buckets of %d steps per slot"
                               % max(1, a.gather_every))
                out["steer_mode"] = legs["steer"] = leg
        # scaling_model: what each exchange mode can reach on 8 GPUs of one node.  A GPU computes at the rate it
        # was measured at in THIS run (per-GPU rays/s of the mode's local work: `march_only` for the modes that
        # move ranges — their exchange overlaps the marches —, the mode's own leg for crash / steer) unless the
        # xGMI ingress of the mode's bytes is slower; the speed-up is against ONE GPU marching without exchange.
        base = legs.get("none", legs.get(mode))
        r_none = base["value"] / world                         # Mrays/s per GPU, march only
        scale8 = 1.0 if scaling == "weak" else world / 8.0     # the per-GPU batch at 8 GPUs (strong: global batch / 8)
        model = scaling_model(r_none, {md: legs[md]["value"] / world for md in ("crash", "steer") if md in legs},
                              int(per_rank * scale8), int(n * scale8), max(1, int(n_groups * scale8)))
        for md, row in model.items():
            row["per_gpu_local_is"] = ("measured: %s leg of this run" % md) if (md in ("crash", "steer") and md in legs) \
                else "measured: march without exchange (this run)"
        out["scaling_model"] = {"modes": model, "per_gpu_march_mrays_s": round(r_none, 1),
                                "rays_per_gpu_per_step_at_8": int(per_rank * scale8),
                                "xgmi_peak_gbs": round(XGMI_LINKS * XGMI_LINK_GBS, 1),
                                "speedup_is": "8 x min(per-GPU local rate of the mode, rays per step / xGMI ingress floor) "
                                              "/ per-GPU march rate; link rate assumed (%d x %.1f GB/s per direction), "
                                              "local rates measured on %d GPU(s) in this run%s" % (
                                                  XGMI_LINKS, XGMI_LINK_GBS, world,
                                                  " (same-device dry run over gloo: host-paced, not representative)" if a.same_device else ""),
                                "reading": "`value` is the literal exchange BASELINE.json names (all-gather of every "
                                           "range): xGMI-bound at ~1x whatever the kernel does; crash / steer are what the "
                                           "reference's consumers read (scripts/mcts.py:237-245,262-267) and scale with the march"}
    if world == 1:
        # the dominant kernel.  `achieved` prices the ALGORITHMIC bytes of one launch (SURVEY.md section 8d's
        # per-ray figure x the rays of a launch) against the time a launch costs: `ms_per_step`, the wall clock
        # of the timed region / K — with P launches in flight that is the machine time a launch takes, not its
        # begin-to-end span (a kernel trace shows each launch ~P x longer, P of them overlapping).
        # `frac_device` is the same against the HIP-event time of the region / K (`launch_ms`, a few % shorter:
        # the events do not see the host's first launch latency).  `frac_hbm` is the MEASURED HBM traffic
        # (committed PMC pass of this launch shape) against the same peak: what the DRAM actually moves — K1b
        # gathers from an L2-resident table, so it is far below `frac`; `roofline_gather` / `roofline_valu` are
        # that kernel's real limiters.  `serial` is the same kernel alone on an idle machine (library events
        # around the march kernel on extra steps AFTER the timed region — a pair of event records per launch
        # costs ~12 us, so it stays out of `value`): the duration a kernel trace of `--pipeline 1` reports.
        reset_noise()
        wall_ms = res["ms_per_step"]
        achieved = bpr * n * B / (wall_ms * 1e-3) / 1e9
        achieved_dev = bpr * n * B / (step_ms * 1e-3) / 1e9
        apply_schedule(False)
        meth.set_option("timing", 2)
        ks = []
        solo_out = d_ref if d_ref is not None else torch.empty(n * B, dtype=torch.float32, device=dev)
        for _ in range(min(max(a.steps, 10), 30)):
            meth.calc_range_fan_device(d_poses[0].data_ptr(), n, w.fov, B, solo_out.data_ptr(), stream=cur_stream)
            ks.append(meth.last_kernel_ms())
        solo_plan = meth.last_plan()
        meth.set_option("timing", 0)
        apply_schedule(True)
        k_ms = float(np.median(ks))
        serial_ach = bpr * n * B / (k_ms * 1e-3) / 1e9
        pe = pmc_entry(a.workload, method, n, plan, "steer" if mode == "steer" else None)
        traffic = pe["bytes"] if pe else None
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "frac_is": "algorithmic bytes per launch / ms_per_step / peak (cache-served bytes count: "
                                      "can exceed what DRAM moves, see frac_hbm)",
                           "traffic_commit": pe.get("commit") if pe else None,
                           "achieved_device": round(achieved_dev, 2), "frac_device": round(achieved_dev / HBM_PEAK_GBS, 5),
                           "traffic": traffic,
                           "measured_hbm_gbs": round(traffic / (wall_ms * 1e-3) / 1e9, 1) if traffic else None,
                           "frac_hbm": round(traffic / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                           "traffic_source": (pe["profile"] + " (rocprofv3 --pmc passes of this kernel, grid and batch "
                                              "size, committed; not measured in this run)") if pe else
                                             "no committed PMC pass matches this launch shape",
                           "bytes_per_ray": round(bpr, 3), "bytes_per_ray_is": bpr_note,
                           "kernel": plan["name"], "grid": plan["grid"],
                           "launch_ms": round(step_ms, 4), "launches_in_flight": P,
                           "serial": {"kernel": solo_plan["name"], "grid": solo_plan["grid"],
                                      "kernel_ms": round(k_ms, 4), "achieved": round(serial_ach, 2),
                                      "frac": round(serial_ach / HBM_PEAK_GBS, 5),
                                      "what": "the march kernel alone on an idle machine (grid_mult %d), median of %d"
                                              % (default_gm, len(ks))}}
        if method == "CDDT":
            # SURVEY section 8(d) prices CDDT as a per-RAY bisection (44 B / ray); the kernels answer one look-up per (pose,
            # table bin) and ~10 beams share it, so that yardstick exceeds 1.  The line's `frac` is what the step's kernels
            # MOVE (the committed PMC pass of this launch shape) / time / peak; the bisection figure stays beside it
            rf = out["roofline"]
            rf["frac_bisection"], rf["achieved_bisection"] = rf["frac"], rf["achieved"]
            rf["frac"], rf["achieved"] = rf["frac_hbm"], rf["measured_hbm_gbs"]
            rf["frac_is"] = ("HBM bytes the step's kernels move (committed rocprofv3 PMC pass of this launch shape: `traffic`) / "
                             "ms_per_step / peak — null when no pass matches; frac_bisection: SURVEY 8(d)'s per-ray bisection bytes")
        if method in ("RM", "RMGPU") and mean_steps > 0 and not a.no_extras:
            # the kernel's real limiters, next to the contractual HBM object.  (1) the CU's scattered-gather
            # rate, probed in this run; (2) VALU issue: wave-level VALU instructions per launch (rocprofv3
            # SQ_INSTS_VALU of exactly this launch shape, committed) x 4 clocks each on 4 SIMDs per CU
            lanes, clk, ncu = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int(0)
            _lib.check(_lib.lib().rl_probe_gather_rate(local_rank, 46, ctypes.byref(lanes), ctypes.byref(clk),
                                                        ctypes.byref(ncu)))
            peak = lanes.value * ncu.value * clk.value
            # gathered samples: the statement's count less the t = 0 sample of every ray, which the kernel
            # reads once per pose with the pose record (pose_first_step)
            samples = max(mean_steps - 1.0, 0.0) * n * B
            out["roofline_gather"] = {"achieved_samples_per_s": round(samples / (wall_ms * 1e-3), 1),
                                      "peak": round(peak, 1), "frac": round(samples / (wall_ms * 1e-3) / peak, 5),
                                      "serial_frac": round(samples / (k_ms * 1e-3) / peak, 5),
                                      "probe": "%.2f active lanes/clk/CU x %d CUs x %.2f GHz (rl_probe_gather_rate, "
                                               "46 random lanes, this run)" % (lanes.value, ncu.value, clk.value / 1e9)}
        if pe and pe.get("valu_insts") and not a.no_extras:
            # VALU issue floor of any kernel with a committed SQ_INSTS_VALU pass: wave-level instructions x 4 clocks
            # on n_cu x 4 SIMDs
            prop = torch.cuda.get_device_properties(dev)
            clk_hz = float(getattr(prop, "clock_rate", 2400000)) * 1e3
            floor_ms = pe["valu_insts"] * 4.0 / (4 * prop.multi_processor_count * clk_hz) * 1e3
            out["roofline_valu"] = {"wave_valu_insts_per_launch": pe["valu_insts"], "floor_ms": round(floor_ms, 5),
                                    "frac": round(floor_ms / wall_ms, 5),
                                    "lanes_per_valu_inst": pe.get("lanes_per_valu"),
                                    "source": pe["profile"] + " (SQ_INSTS_VALU; not measured in this run)"}
        if not a.no_extras and P > 1 and not scan.reduced:
            # the same schedule with every step in flight scanning the SAME batch (what round 2 measured):
            # quantifies what identical cache lines in identical order are worth
            scan.bind(meth, [d_poses[0].data_ptr()] * P, w.fov)
            sb = summarise(timed_scan(scan, a.steps, min(a.warmup, 5), min(a.bursts, 9)), a.steps, rays_per_step)
            scan.bind(meth, [t.data_ptr() for t in d_poses], w.fov)
            out["same_batch"] = {"value": round(sb["value"], 2), "ms_per_step": round(sb["ms_per_step"], 4),
                                 "bursts": sb["bursts"],
                                 "what": "all %d steps in flight scan batch 0 (not the reported configuration)" % P}
        if not a.no_extras and is_rm and a.variant < 0 and variant_eff != 3 and not scan.reduced and not a.selftest_corrupt:
            # strict parity as a production mode: the SAME schedule — steps in flight, streams, pose
            # batches — with the upstream-literal arithmetic (option variant 3 -> rm_fan_stream_kernel<.., LIT>)
            meth.set_option("variant", 3)
            try:
                lm = summarise(timed_scan(scan, a.steps, min(a.warmup, 5), min(a.bursts, 9)), a.steps, rays_per_step)
                lp = meth.last_plan()
                literal_mode.update({"value": round(lm["value"], 2), "ms_per_step": round(lm["ms_per_step"], 4),
                                     "vs_canonical": round(lm["value"] / out["value"], 4), "kernel": lp["name"],
                                     "grid": lp["grid"], "bursts": lm["bursts"]})
                out["literal_mode"] = dict(literal_mode, what="the timed schedule with option variant 3: range_libc's CPU "
                                           "arithmetic stated literally (per-ray glibc sinf / cosf, un-fused march), bit-identical "
                                           "to the checker's libm form (verification.upstream_literal)")
            finally:
                meth.set_option("variant", 1)
        if not a.no_extras and method in ("RM", "RMGPU"):
            # what the reference's callers see: numpy in -> numpy out through ScanSimulator2D
            # (scripts/scan_simulator.py:88-135), scan() = the sim tick, scanMany(200) = one MCTS roll-out
            from pyracecarsimulator_amd import ScanSimulator2D
            sim = ScanSimulator2D(B, w.fov, 0.01, batch_size=min(200, n))
            sim.setMap(omap, w.max_range_px, gmap.resolution, gmap.origin)
            sim.setRaytracingMethod(method)
            hp = batches[0][:sim.batch_size]
            ts_one, ts_many = [], []
            for i in range(60):
                t0 = time.perf_counter()
                sim.scan(float(hp[i % len(hp), 0]), float(hp[i % len(hp), 1]), float(hp[i % len(hp), 2]))
                ts_one.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                sim.scanMany(hp)
                ts_many.append(time.perf_counter() - t0)
            out["end_to_end"] = {"scan_us": round(float(np.median(ts_one[10:])) * 1e6, 1),
                                 "scanMany_%d_us" % sim.batch_size: round(float(np.median(ts_many[10:])) * 1e6, 1),
                                 "what": "host wall clock, numpy in -> numpy out through ScanSimulator2D "
                                         "(PCIe and launch latency included), median of 50"}
            sim.scan_method.close()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(w, gmap, batches[0], method, a.cpu_seconds,
                          check=None if a.no_verify else oracle_check)
        v = cb.pop("_verification", None)
        if v is not None:
            verification.update(v)
            ok &= bool(v.get("oracle_subsample"))
        out["cpu_baseline"] = cb
    elif not a.no_verify:
        verification.setdefault("oracle_subsample", "not run (the oracle is loaded by the cpu_baseline leg only: "
                                                    "rank 0 of a 1-GPU run without --no-cpu-baseline)")
    if (rank == 0 and world == 1 and not multi and a.workload == "cfg2" and not a.method and not a.poses and
            mode == "none" and a.variant < 0 and not a.no_other_configs and not a.no_extras and not a.selftest_corrupt):
        # ONE driver command, every single-GPU configuration: short verified legs of the other
        # BASELINE.json configs and of the reduced modes, after the headline's measurement (bench_legs.py).  The oracle
        # is the checker of those legs, loaded only when the cpu_baseline leg of this run loads it anyway.
        import bench_legs
        O_ = None
        if not a.no_cpu_baseline and not a.no_verify:
            from oracle import oracle as O_
        only = set(x for x in a.only_configs.split(",") if x) or None
        legs = bench_legs.other_configs(torch, dev, local_rank, O=O_, pmc_lookup=pmc_entry, only=only,
                                            streams=streams if len(streams) > 1 else None)
        out["other_configs"] = legs
        bad_legs = [k for k, v in legs.items() if v.get("verified") is False]
        if not a.no_verify:
            verification["other_configs_verified"] = not bad_legs
            if bad_legs:
                verification["other_configs_failed"] = bad_legs
                ok = False
    if not a.no_verify:
        out["verified"] = bool(ok)
        out["verification"] = verification
    if multi:
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if not a.no_verify and not ok:
        print("bench.py: output verification FAILED: %s" % json.dumps(verification), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
