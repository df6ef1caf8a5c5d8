"""FollowGap kernel alone: 4096 scans x 1081 beams resident in HBM, HIP-event time per launch (median of 50), the scans
re-written before every launch by a copy kernel (cold-ish caches, as behind a march) and not."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd.followgap import PyFollowGap
n, B = 4096, 1081
rng = np.random.default_rng(1)
scans = torch.from_numpy(rng.uniform(0.2, 15.0, (n, B)).astype(np.float32)).cuda()
src = scans.clone()
ang = torch.empty(n, dtype=torch.float32, device="cuda")
fg = PyFollowGap(10, 15.0, 0.4189, 0.004)
for rewrite in (False, True):
    ts = []
    for _ in range(60):
        if rewrite: scans.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fg.eval_many_device(scans.data_ptr(), n, B, ang.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("followgap_kernel, %d scans x %d beams, scans %s: %.1f us (p10 %.1f)" % (n, B, "rewritten before each launch" if rewrite else "warm", np.median(ts[10:]), np.percentile(ts[10:], 10)))
