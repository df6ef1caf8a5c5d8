"""Stream helpers for callers that keep several scan batches in flight (bench.py, ShardedScan).

A ``*_device`` call of libscan_amd.so only enqueues work, and a method handle keeps its per-launch
scratch per stream (include/scanlib.h, "streams"), so consecutive pose batches enqueued on
different streams overlap on the GPU: batch k+1 fills the CUs that batch k's last long rays leave
idle.  HIP multiplexes its streams onto a few hardware queues, and two streams that share a queue
run strictly one after the other — so the set of streams is picked by measurement, not assumed.
"""
from __future__ import annotations

import time


def concurrent_streams(n: int, candidates: int = 16, cycles: int = 2_000_000):
    """Up to ``n`` torch streams on the current device that run concurrently with each other.

    A spin kernel (``torch.cuda._sleep``) on a pair of streams takes about the single time when
    the two overlap and twice that when they share a hardware queue; a candidate joins the set
    when it overlaps with every member.  Costs a few milliseconds, once."""
    import torch

    torch.cuda.init()                      # (a Stream made before torch's lazy initialisation fails)
    torch.zeros(1, device="cuda")
    cand = [torch.cuda.Stream() for _ in range(max(candidates, n))]

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
        if len(chosen) >= n:
            break
        if all(min(spin([c, o]) for _ in range(2)) < 1.5 * one for o in chosen):
            chosen.append(c)
    return chosen


__all__ = ["concurrent_streams"]
