"""CPU replay (no GPU): (quad, line) pairs and distinct lines per wave-load of the stream kernel for several tile shapes — DESIGN.md section 4 "Round 3"."""
"""Estimate TCP line accesses per wave-load for different tile shapes, emulating the stream kernel's refill:
one wave = 64 lanes fed consecutive rays (pose-major, beam-minor) of a stream; march while > LOW lanes live."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pyracecarsimulator_amd import workloads, maps
from scipy.ndimage import distance_transform_edt
w = workloads.cfg2()
g = w.gmap
occ = g.occ
dt = distance_transform_edt(occ == 0).astype(np.float32)
rng = np.random.default_rng(0)
NP = 24
poses = maps.sample_free_poses(g, NP, 2, 2.0, dt)
B, fov, mr = 1081, 4.71, 300.0
res = g.resolution; ox, oy, _ = g.origin
gx = ((poses[:,0]-ox)/res).astype(np.float32); gy = ((poses[:,1]-oy)/res).astype(np.float32)
alpha = (-fov/2 + np.arange(B)*fov/B)
ang = poses[:,2:3] + alpha[None,:]
DX = np.cos(ang).astype(np.float32).ravel(); DY = np.sin(ang).astype(np.float32).ravel()
GX = np.repeat(gx, B); GY = np.repeat(gy, B)
step = np.where(dt <= 0, np.inf, np.maximum(dt, 1.0)).astype(np.float32)
R, C = occ.shape
def sample(x, y):
    c = x.astype(np.int64); r = y.astype(np.int64)
    inb = (c >= 0) & (c < C) & (r >= 0) & (r < R)
    s = np.full(x.shape, 3e38, np.float32)
    s[inb] = step[r[inb], c[inb]]
    return s, r, c
N = NP * B
shapes = {"4x8 f32": (4, 8), "8x8 u16": (8, 8), "8x16 u8": (8, 16), "16x8 u8": (16, 8), "4x32 u8":(4,32)}
tot = {k: 0 for k in shapes}; totq = {k: 0 for k in shapes}; loads = 0; lanes_sum = 0
LOW = 12
nxt = 0
# lane state
t = np.full(64, np.inf, np.float32); ray = np.full(64, -1)
# first sample at origin handled: t0 = step at origin
def refill():
    global nxt
    idle = np.where(~(t < mr))[0]
    for l in idle:
        if nxt >= N:
            ray[l] = -1; continue
        ray[l] = nxt
        s, _, _ = sample(GX[nxt:nxt+1], GY[nxt:nxt+1])
        t[l] = s[0] if np.isfinite(s[0]) and s[0] < 1e38 else np.inf
        nxt += 1
while True:
    refill()
    live = t < mr
    if not live.any(): break
    while True:
        live = t < mr
        n = live.sum()
        if n <= (LOW if nxt < N else 0): break
        idx = ray[live]
        x = GX[idx] + DX[idx]*t[live]; y = GY[idx] + DY[idx]*t[live]
        s, r, c = sample(x, y)
        loads += 1; lanes_sum += n
        lanes = np.where(live)[0]
        for k, (th, tw) in shapes.items():
            tid = (r // th) * 100000 + (c // tw)
            tot[k] += len(np.unique(tid))
            # per quad distinct
            q = lanes // 4
            totq[k] += len(np.unique(q * 10**12 + tid))
        t[live] = t[live] + s
print("wave-loads", loads, "lanes/load %.1f" % (lanes_sum/loads))
for k in shapes:
    print("%-10s distinct lines/load %.1f   (quad,line) pairs/load %.1f" % (k, tot[k]/loads, totq[k]/loads))
