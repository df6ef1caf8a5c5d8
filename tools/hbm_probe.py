#!/usr/bin/env python3
"""What does HBM actually deliver on this box?  copy (read+write), fill (write only), sum (read only)
on 2 GiB float32 tensors, HIP-event timed — the practical ceiling next to the 8 TB/s spec that
bench.py's roofline uses."""
import torch

n = 512 * 1024 * 1024          # 2 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 4 / 1e9
t = timed(lambda: b.copy_(a))
print("copy  (read %.1f GB + write %.1f GB): %.0f GB/s" % (gb, gb, 2 * gb / t))
t = timed(lambda: b.fill_(1.0))
print("fill  (write only)              : %.0f GB/s" % (gb / t))
t = timed(lambda: a.sum())
print("sum   (read only)               : %.0f GB/s" % (gb / t))


# the same three patterns from hand-written 16-B-per-lane kernels (rl_probe_hbm): torch's elementwise copy /
# reduction kernels are not the ceiling — MI355X_MICROARCH.md's 6.29 TB/s is a float4 copy like this one
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyracecarsimulator_amd import _lib
del a, b
torch.cuda.empty_cache()
out = (ctypes.c_double * 5)()
_lib.check(_lib.lib().rl_probe_hbm(0, 2 << 30, out))
for name, v in zip(("copy", "read only", "write only", "copy, non-temporal stores", "fill, non-temporal stores"), out):
    print("rl_probe_hbm %-26s: %.0f GB/s" % (name, v))
out = (ctypes.c_double * 3)()
_lib.check(_lib.lib().rl_probe_hbm_nt(0, 2 << 30, out))
for name, v in zip(("read only, non-temporal loads", "copy, non-temporal loads", "copy, non-temporal loads + stores"), out):
    print("rl_probe_hbm_nt %-34s: %.0f GB/s" % (name, v))
