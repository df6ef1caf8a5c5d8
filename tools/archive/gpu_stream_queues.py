#!/usr/bin/env python3
"""Which HIP streams run concurrently on this box?  Pairs of streams each get a spin kernel
(torch.cuda._sleep); a pair that takes ~1x the single time is on different hardware queues, ~2x
means the two streams share one (HIP multiplexes streams onto GPU_MAX_HW_QUEUES queues)."""
import ctypes as C
import time
import torch

hip = C.CDLL("libamdhip64.so")


def raw_stream(flags=1, prio=None):
    s = C.c_void_p()
    if prio is None:
        assert hip.hipStreamCreateWithFlags(C.byref(s), flags) == 0
    else:
        assert hip.hipStreamCreateWithPriority(C.byref(s), flags, prio) == 0
    return torch.cuda.ExternalStream(s.value)


def pair_time(a, b, cycles=20_000_000):
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t


def main():
    torch.cuda.init()
    x = torch.zeros(1, device="cuda")
    lo, hi = C.c_int(), C.c_int()
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
    print("priority range: least", lo.value, "greatest", hi.value)
    pool = [torch.cuda.Stream() for _ in range(6)]
    poolhi = [torch.cuda.Stream(priority=-1) for _ in range(2)]
    raw = [raw_stream() for _ in range(6)]
    rawhi = [raw_stream(prio=hi.value) for _ in range(2)]
    single = pair_time(pool[0], pool[0])
    print("same stream twice: %.1f ms" % (single * 1e3))
    names = {"pool": pool, "poolhi": poolhi, "raw": raw, "rawhi": rawhi}
    def show(na, i, nb, j):
        t = pair_time(names[na][i], names[nb][j])
        print("%s[%d] + %s[%d]: %.1f ms  -> %s" % (na, i, nb, j, t * 1e3, "CONCURRENT" if t < 0.75 * single else "serial"))
    for j in range(1, 6):
        show("pool", 0, "pool", j)
    for j in range(1, 6):
        show("raw", 0, "raw", j)
    show("raw", 1, "raw", 2)
    show("raw", 0, "pool", 0)
    show("pool", 0, "poolhi", 0)
    show("poolhi", 0, "poolhi", 1)
    show("raw", 0, "rawhi", 0)
    show("rawhi", 0, "rawhi", 1)
    cur = torch.cuda.current_stream()
    t = pair_time(cur, pool[0]); print("default + pool[0]: %.1f ms" % (t * 1e3))
    t = pair_time(cur, raw[0]); print("default + raw[0]: %.1f ms" % (t * 1e3))


if __name__ == "__main__":
    main()
