#!/usr/bin/env python3
"""Pipelining experiment on the GPU box: consecutive pose batches of one workload enqueued round
robin on P streams (one launch context each inside the library), for several grid sizes.  Prints
wall-clock us per batch; every batch is complete when the clock stops.  P = 1 is the serial
baseline (one stream: launch k+1 starts after launch k's last ray)."""
import argparse
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from pyracecarsimulator_amd import range_libc, workloads  # noqa: E402


def concurrent_streams(n, candidates=12, cycles=2_000_000):
    """n torch streams that run concurrently with each other: HIP multiplexes its streams onto a few
    hardware queues and two streams on one queue serialise, so the set is picked by measurement (a
    spin kernel on a pair of streams takes 1x the single time when they overlap, 2x when not)."""
    cand = [torch.cuda.Stream() for _ in range(candidates)]

    def spin(streams):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t

    spin(cand[:1])
    one = min(spin(cand[:1]) for _ in range(3))
    chosen = [cand[0]]
    for c in cand[1:]:
        if len(chosen) == n:
            break
        if all(min(spin([c, o]) for _ in range(2)) < 1.5 * one for o in chosen):
            chosen.append(c)
    return chosen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--poses", type=int, default=0)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--pipes", default="1,2,3,4")
    ap.add_argument("--grid-mults", default="8,4,2")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--sweep", action="store_true", help="cfg2 tuning grid: workgroup size x record source x grid x streams")
    a = ap.parse_args()
    w = workloads.CONFIGS[a.workload]()
    if a.poses:
        w.n_poses = a.poses
    omap = range_libc.PyOMap(w.gmap)
    dt = omap.distance_transform()
    poses = workloads.make_poses(w, dt=dt)
    n, B = len(poses), w.num_rays
    d_poses = torch.from_numpy(poses).cuda()
    meth = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
    for kv in a.opt:
        k, v = kv.split("=")
        meth.set_option(k, int(v))
    pmax = max(int(p) for p in a.pipes.split(","))
    streams = concurrent_streams(pmax)
    print('concurrent streams found: %d of %d wanted' % (len(streams), pmax), flush=True)
    pmax = len(streams)
    outs = [torch.empty(n * B, dtype=torch.float32, device="cuda") for _ in range(pmax)]
    ref = None
    if a.sweep:
        import itertools
        best = []
        for nt, oi in ((1024, 1), (1024, 0), (512, 0), (256, 0)):
            meth.set_option("wg_threads", nt)
            meth.set_option("order_inline", oi)
            for gm, P in itertools.product((2, 3, 4, 5, 6, 8), (2, 3, 4)):
                if P > pmax:
                    continue
                meth.set_option("grid_mult", gm)

                def run(k):
                    for i in range(k):
                        s_ = i % P
                        meth.calc_range_fan_device(d_poses.data_ptr(), n, w.fov, B, outs[s_].data_ptr(),
                                                   stream=streams[s_].cuda_stream)
                run(20)
                torch.cuda.synchronize()
                t = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    run(a.steps)
                    torch.cuda.synchronize()
                    t = min(t, (time.perf_counter() - t0) / a.steps)
                best.append((t, nt, oi, gm, P))
                print("wg %4d order_inline %d grid_mult %d streams %d  %7.2f us/batch" % (nt, oi, gm, P, t * 1e6), flush=True)
        best.sort()
        print("best:", ["%.2f us wg%d oi%d gm%d P%d" % (t * 1e6, nt, oi, gm, P) for t, nt, oi, gm, P in best[:6]])
        return
    for gm in (int(g) for g in a.grid_mults.split(",")):
        meth.set_option("grid_mult", gm)
        for P in (int(p) for p in a.pipes.split(",") if int(p) <= pmax):
            def run(k):
                for i in range(k):
                    s = i % P
                    meth.calc_range_fan_device(d_poses.data_ptr(), n, w.fov, B, outs[s].data_ptr(),
                                               stream=streams[s].cuda_stream)
            run(20)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                run(a.steps)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / a.steps)
            if ref is None:
                ref = outs[0].clone()
            same = all(torch.equal(o, ref) for o in outs[:P])
            print("grid_mult %2d  streams %d  %7.2f us/batch  %8.1f Mrays/s  identical=%s"
                  % (gm, P, best * 1e6, n * B / best / 1e6, same), flush=True)


if __name__ == "__main__":
    main()
