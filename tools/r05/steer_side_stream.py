"""--gather steer at the crash price?  The FollowGap kernel is a second DEPENDENT launch on every slot stream
(8-12 us of a ~105-us slot cycle: 77 % of the plain pipelined rate).  A/B: FollowGap of step k on a SIDE stream per
slot (event-ordered behind the march, the slot's next march waits for it), so the slot stream goes straight on to
the next binning + march.  cfg2, 4096 poses x 1081 beams, four slots."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyracecarsimulator_amd import range_libc, workloads, racecar as RC
from pyracecarsimulator_amd.followgap import PyFollowGap
from pyracecarsimulator_amd.pipeline import concurrent_streams

w = workloads.cfg2()
omap = range_libc.PyOMap(w.gmap)
m = range_libc.PyRayMarchingGPU(omap, w.max_range_px)
B, n, P = w.num_rays, 4096, 4
dt = omap.distance_transform()
d_p = [torch.from_numpy(workloads.make_poses(w, dt=dt, seed=w.pose_seed + 7919 * k)).cuda() for k in range(P)]
outs = [torch.empty(n * B, dtype=torch.float32, device="cuda") for _ in range(P)]
ang = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(P)]
fg = PyFollowGap(10, 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004)
m.set_option("grid_mult", 3); m.set_option("slots", 2)
streams = concurrent_streams(P)
side = [torch.cuda.Stream() for _ in range(P)]
ev_scan = [torch.cuda.Event() for _ in range(P)]
ev_fg = [torch.cuda.Event() for _ in range(P)]
print("streams", len(streams), "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))

def run(mode, steps=300, bursts=9):
    m.set_option("nt_store", 1 if mode == "plain" else 0)
    ts = []
    for _ in range(bursts):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % P
            s = streams[k]
            if mode == "side":
                s.wait_event(ev_fg[k])
            m.calc_range_fan_device(d_p[k].data_ptr(), n, w.fov, B, outs[k].data_ptr(), stream=s.cuda_stream)
            if mode == "same":
                fg.eval_many_device(outs[k].data_ptr(), n, B, ang[k].data_ptr(), stream=s.cuda_stream)
            elif mode == "side":
                ev_scan[k].record(s)
                side[k].wait_event(ev_scan[k])
                fg.eval_many_device(outs[k].data_ptr(), n, B, ang[k].data_ptr(), stream=side[k].cuda_stream)
                ev_fg[k].record(side[k])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    ms = float(np.median(ts)) * 1e3
    print("%-6s %d steps: %.4f ms per step, %.1f Grays/s" % (mode, steps, ms, n * B / ms / 1e6), flush=True)
    return [a.clone() for a in ang]

for k in range(P):
    ev_fg[k].record(side[k])
for steps in (300, 20):
    run("plain", steps)
    a = run("same", steps)
    b = run("side", steps)
    print("  side == same:", all(bool(torch.equal(x, y)) for x, y in zip(a, b)))
