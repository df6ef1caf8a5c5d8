"""CPU replay (the checker's arithmetic; runs in the build container, no GPU) behind round 5's kernel decisions on cfg2
(2049^2 maze, 4096 poses x 1081 beams): round trips a long ray costs under the drain policies (value speculation on the step,
4 / 8 / 16 deep, with and without plain stretches) and how often a step repeats its predecessor (profiles/r05/drain_depth_model.txt)."""
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
g = w.gmap; dt = om.dt
B = w.num_rays; res = g.resolution; ox, oy = g.origin[0], g.origin[1]
def march(p,j):
    gx=np.float32((poses[p,0]-ox)/res); gy=np.float32((poses[p,1]-oy)/res)
    th=poses[p,2]; a=-w.fov/2+j*(w.fov/B)
    dx,dy=np.float32(np.cos(th+a)),np.float32(np.sin(th+a)); t=np.float32(0); seq=[]
    while t<300:
        c=int(np.float32(gx+dx*t)); r=int(np.float32(gy+dy*t))
        if c<0 or r<0 or c>=g.cols or r>=g.rows: break
        d=dt[r,c]
        if d<=0: break
        s=np.float32(max(d,1.0)); seq.append(float(s)); t=np.float32(t+s)
    return seq
def trips_spec(seq, D, stretch):
    # round trips to consume seq[1:] given seq[0] known as g; policy: `stretch` plain samples, then speculate D-deep until first prediction fails
    i=1; n=len(seq); trips=0
    while i<n:
        k=0
        while k<stretch and i<n: i+=1; trips+=1; k+=1
        while i<n:
            gprev=seq[i-1]; trips+=1
            # consume sample i always; then while it equals g continue up to D
            c=1; first_hit = (seq[i]==gprev)
            while c<D and i+c<n and seq[i+c-1]==gprev: c+=1
            i+=c
            if not first_hit: break
    return trips
def trips_always(seq,D): return trips_spec(seq,D,0) if False else trips_always2(seq,D)
def trips_always2(seq,D):
    i=1;n=len(seq);trips=0
    while i<n:
        gprev=seq[i-1]; trips+=1; c=1
        while c<D and i+c<n and seq[i+c-1]==gprev: c+=1
        i+=c
    return trips
def trips_stride(seq,D):
    # speculate on the last DIFFERENCE pattern: predict s[i+k] = s[i-1] (same as always) but also accept period-2 alternation: predict s[i+k]=s[i-2+ (k%2)]
    i=2;n=len(seq);trips=2
    while i<n:
        trips+=1;c=0
        while c<D and i+c<n:
            pred = seq[i+c-2]
            c+=1
            if seq[i+c-1]!=pred: break
        i+=max(c,1)
    return trips
rng=np.random.default_rng(1)
for lo,hi in ((32,48),(48,80),(80,400)):
    idx=np.argwhere((steps>=lo)&(steps<hi)); sel=idx[rng.choice(len(idx),min(250,len(idx)),replace=False)]
    seqs=[march(p,j) for p,j in sel]; seqs=[s for s in seqs if len(s)>8]
    n=np.array([len(s) for s in seqs])
    out=["%d..%d samples (mean %.0f):"%(lo,hi,n.mean())]
    out.append("plain %.0f"%n.mean())
    out.append("cur(stretch8,D4) %.1f"%np.mean([trips_spec(s,4,8) for s in seqs]))
    for D in (4,8,16):
        out.append("always D%d %.1f"%(D,np.mean([trips_always2(s,D) for s in seqs])))
    for D in (4,8,16):
        out.append("period2 D%d %.1f"%(D,np.mean([trips_stride(s,D) for s in seqs])))
    print("  ".join(out))
