"""(round 5: the same experiment for cfg2 — 2049^2 maze, 4096 poses x 1081 beams: what would a step map of half / a quarter the bytes buy at most?)
What bounds cfg5 (4096^2 maze, 720 beams)?  A/B on one box, lone launches and four in flight, noise off:
the SAME map, kernel and pose count with the poses (a) over the whole map, (b) inside a central window of
1/4, 1/16 of the area — only the footprint of step-map lines the batch touches changes — and (c) noise on.
If (b) is much faster than (a) the launch is bound by lines that miss L2 (154 MB table, 4 MB of L2 per XCD);
if not, by what the rays do (VALU / gather rate)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import maps, range_libc, workloads
from pyracecarsimulator_amd.pipeline import concurrent_streams

w = workloads.cfg2()
g = w.gmap
omap = range_libc.PyOMap(g)
dt = omap.distance_transform()
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
n, B = 4096, w.num_rays
streams = concurrent_streams(4)

def poses_in(frac, seed):
    """n seeded free poses whose cell lies in the central window of `frac` of the map's area."""
    side = int(g.rows * frac ** 0.5)
    lo = (g.rows - side) // 2
    sub = maps.GridMap(np.ascontiguousarray(g.occ[lo:lo + side, lo:lo + side]), g.resolution,
                       (g.origin[0] + lo * g.resolution, g.origin[1] + lo * g.resolution, 0.0), "win")
    return maps.sample_free_poses(sub, n, seed, 2.0, np.ascontiguousarray(dt[lo:lo + side, lo:lo + side]))

d_o = [torch.empty(n * B, dtype=torch.float32, device="cuda") for _ in range(4)]
def lone(d_p, reps=20):
    for _ in range(3): m.calc_range_fan_device(d_p[0].data_ptr(), n, w.fov, B, d_o[0].data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): m.calc_range_fan_device(d_p[i % 4].data_ptr(), n, w.fov, B, d_o[0].data_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def piped(d_p, reps=40):
    import time
    for k in range(8): m.calc_range_fan_device(d_p[k % 4].data_ptr(), n, w.fov, B, d_o[k % 4].data_ptr(), stream=streams[k % 4].cuda_stream)
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(reps): m.calc_range_fan_device(d_p[k % 4].data_ptr(), n, w.fov, B, d_o[k % 4].data_ptr(), stream=streams[k % 4].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for frac in (1.0, 0.25, 1.0 / 16, 1.0 / 64):
    d_p = [torch.from_numpy(poses_in(frac, 40 + k)).cuda() for k in range(4)]
    d_steps = torch.empty(n * B, dtype=torch.int16, device="cuda")
    m.set_option("slots", 0); m.set_option("grid_mult", 8)
    m.calc_range_fan_device(d_p[0].data_ptr(), n, w.fov, B, d_o[0].data_ptr(), d_steps_ptr=d_steps.data_ptr()); torch.cuda.synchronize()
    mean_s = float(d_steps.to(torch.int32).bitwise_and(0xffff).float().mean())
    res = []
    for std in (0.0,):
        m.set_noise(std, w.noise_seed, 0)
        m.set_option("slots", 0); m.set_option("grid_mult", 8)
        a = lone(d_p)
        m.set_option("slots", 2); m.set_option("grid_mult", 3)
        b = piped(d_p)
        res.append((a, b)); res.append((a, b))
    side = int(g.rows * frac ** 0.5)
    print("poses in the central %4d^2 window (%5.1f %% of the map, step-map footprint ~%5.1f MB): mean samples %.2f | lone %.4f ms (%.0f Grays/s), 4 in flight %.4f ms (%.0f) | with noise: lone %.4f, 4 in flight %.4f"
          % (side, frac * 100, (side + 608) ** 2 * 4 / 1e6, mean_s, res[0][0], n * B / res[0][0] / 1e6, res[0][1], n * B / res[0][1] / 1e6, res[1][0], res[1][1]))
