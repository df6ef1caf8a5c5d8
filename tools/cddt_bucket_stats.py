"""CPU (no GPU): CDDT bucket size statistics of the bench maps — DESIGN.md section 4 "K3b CDDT, round 3"."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pyracecarsimulator_amd import workloads, maps
for name, w in (("cfg2", workloads.cfg2()), ("cfg3", workloads.cfg3()), ("colombia", workloads.cfg4())):
    occ = w.gmap.occ != 0
    R, C = occ.shape
    pad = np.pad(occ, 1, constant_values=False)
    free_nb = (~pad[:-2,1:-1]) | (~pad[2:,1:-1]) | (~pad[1:-1,:-2]) | (~pad[1:-1,2:])
    border = np.zeros_like(occ); border[0,:]=border[-1,:]=border[:,0]=border[:,-1]=True
    edge = occ & (free_nb | border)
    rr, cc = np.nonzero(edge)
    td = 108; nb = td//2
    sizes_all = []
    for a in (0, 7, 13, 27, 40):
        th = a * 2*np.pi/td
        cs, sn = np.float32(np.cos(th)), np.float32(np.sin(th))
        px = cc + 0.5; py = rr + 0.5
        ly = px*sn + py*cs
        H, W = R, C
        tr = max(0.0, -min(H*cs, W*sn + H*cs, W*sn))
        ly = ly + tr
        half = (abs(sn)+abs(cs))*0.5
        lower = np.floor(ly - half + 1e-5).astype(int); upper = np.floor(ly + half - 1e-5).astype(int)
        wdt = int(np.ceil(abs(W*sn)+abs(H*cs))) + 1
        cnt = np.zeros(wdt+2, int)
        for k in range(0, 3):
            idx = lower + k
            ok = idx <= upper
            np.add.at(cnt, np.clip(idx[ok], 0, wdt), 1)
        sizes_all.append(cnt[:wdt])
    s = np.concatenate(sizes_all)
    print(name, "edges", len(rr), "buckets sampled", len(s), "mean %.1f median %d p90 %d p99 %d max %d" % (s.mean(), np.median(s), np.percentile(s,90), np.percentile(s,99), s.max()),
          "frac<=28: %.3f  <=31: %.3f <=60: %.3f" % ((s<=28).mean(), (s<=31).mean(), (s<=60).mean()), "values-weighted frac in buckets<=31: %.3f" % (s[s<=31].sum()/s.sum()))
