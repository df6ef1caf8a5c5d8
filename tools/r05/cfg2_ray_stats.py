"""CPU replay (the checker's arithmetic; runs in the build container, no GPU) behind round 5's kernel decisions on cfg2
(2049^2 maze, 4096 poses x 1081 beams): sample counts per ray (how heavy the tail is) and how evenly the static split of the
ray stream spreads them over the workgroups of a band (profiles/r05/cfg2_ray_stats.txt)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pyracecarsimulator_amd import workloads
from oracle import oracle as O

def cfg2_steps():
    w = workloads.cfg2(4096)
    om = O.OracleMap.from_gridmap(w.gmap, w.max_range_px)
    poses = workloads.make_poses(w, dt=om.dt)
    _, _, steps = om.rm_fan(poses, w.fov, w.num_rays, step_coeff=1.0, nthreads=O.max_threads())
    return w, om, poses, steps.reshape(len(poses), w.num_rays).astype(np.int64)

w, om, poses, steps = cfg2_steps()
g = w.gmap
print("mean samples per ray %.3f, max %d" % (steps.mean(), steps.max()))
for thr in (16, 24, 32, 48, 64, 96, 128):
    m = steps > thr
    print("rays > %3d samples: %.3f %% of the rays, %.1f %% of the samples" % (thr, 100 * m.mean(), 100 * steps[m].sum() / steps.sum()))
res, ox, oy = g.resolution, g.origin[0], g.origin[1]
gx, gy = (poses[:, 0] - ox) / res, (poses[:, 1] - oy) / res
key = (gy.astype(int) >> 6) * ((g.cols >> 6) + 1) + (gx.astype(int) >> 6)
order = np.argsort(key, kind="stable")
st = steps[order]
cpp, nb = 17, 8
blk = np.stack([st[:, b * 64:(b + 1) * 64].sum(axis=1) for b in range(cpp)], axis=1)
band = [blk[(4096 * b) // nb:(4096 * (b + 1)) // nb].sum() for b in range(nb)]
print("samples per XCD band: max / mean %.3f" % (max(band) / np.mean(band)))
for G in (64, 24):                         # workgroups per band: lone launch (grid 512), pipelined launch (grid 192)
    for rl in range(6):
        R = 1 << rl
        tots = []
        for b in range(nb):
            bl = blk[(4096 * b) // nb:(4096 * (b + 1)) // nb].reshape(-1)
            nruns = (len(bl) + R - 1) // R
            for q in range(G):
                idx = np.concatenate([np.arange(r * R, min((r + 1) * R, len(bl))) for r in range(q, nruns, G)])
                tots.append(bl[idx].sum())
        t = np.array(tots)
        print("workgroups per band %2d, runs of %2d blocks: samples per workgroup p10 / p50 / p90 / max relative to the mean: "
              "%.3f %.3f %.3f %.3f" % (G, R, *(np.percentile(t, [10, 50, 90, 100]) / t.mean())))
