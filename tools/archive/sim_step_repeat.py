"""CPU replay (no GPU): how often a long chain repeats its previous step, and samples per iteration of a k-way value-speculating loop — DESIGN.md section 4 "Round 3" (march_drain4)."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pyracecarsimulator_amd import workloads, maps
from scipy.ndimage import distance_transform_edt
w = workloads.cfg2(); g = w.gmap; occ = g.occ
dt = distance_transform_edt(occ == 0).astype(np.float32)
NP = 200
poses = maps.sample_free_poses(g, NP, 2, 2.0, dt)
B, fov, mr = 1081, 4.71, np.float32(300.0)
res = g.resolution; ox, oy, _ = g.origin
gx = ((poses[:,0]-ox)/res).astype(np.float32); gy = ((poses[:,1]-oy)/res).astype(np.float32)
alpha = (-fov/2 + np.arange(B)*fov/B)
ang = poses[:,2:3] + alpha[None,:]
DX = np.cos(ang).astype(np.float32).ravel(); DY = np.sin(ang).astype(np.float32).ravel()
GX = np.repeat(gx, B); GY = np.repeat(gy, B)
step = np.where(dt <= 0, np.inf, np.maximum(dt, 1.0)).astype(np.float32)
R, C = occ.shape
N = NP*B
t = np.zeros(N, np.float32); live = np.ones(N, bool); ns = np.zeros(N, int)
hist = [[] for _ in range(N)]
while live.any():
    idx = np.where(live)[0]
    x = GX[idx] + DX[idx]*t[idx]; y = GY[idx] + DY[idx]*t[idx]
    c = x.astype(np.int64); r = y.astype(np.int64)
    inb = (c>=0)&(c<C)&(r>=0)&(r<R)
    s = np.full(len(idx), np.float32(3e38)); s[inb] = step[r[inb], c[inb]]
    for i, si in zip(idx, s): hist[i].append(si)
    t[idx] = t[idx] + s; ns[idx] += 1
    live[idx] = t[idx] < mr
lens = np.array([len(h) for h in hist])
print("rays", N, "mean samples %.2f" % lens.mean(), "max", lens.max())
for thr in (20, 40, 80):
    sel = np.where(lens >= thr)[0]
    tot = 0; match = 0; iters4 = 0; iters2 = 0; itersA = 0
    for i in sel:
        h = hist[i]
        tot += len(h)
        m = [h[k] == h[k-1] for k in range(1, len(h))]
        match += sum(m)
        # 4-way value speculation: iterations needed (prediction g = last consumed step)
        def iters(W):
            k = 1; it = 0  # first sample known (d0 from pose), start at sample index 1 with g = h[0]
            gprev = h[0]
            while k < len(h):
                it += 1
                used = 1
                # sample k always valid; subsequent valid while previous consumed == g
                gg = gprev
                while used < W and k + used - 1 < len(h) and h[k + used - 1] == gg and k + used < len(h) + 0:
                    used += 1
                gprev = h[min(k + used - 1, len(h) - 1)]
                k += used
            return it
        iters4 += iters(4); iters2 += iters(2); itersA += iters(8)
    print("chains >= %d: %d rays, %d samples, repeat rate %.2f, samples/iter 2-way %.2f 4-way %.2f 8-way %.2f" % (thr, len(sel), tot, match/max(tot-len(sel),1), tot/iters2, tot/iters4, tot/itersA))
