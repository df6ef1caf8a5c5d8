#!/usr/bin/env python3
"""Would a COMPACT step map (one byte per cell: a code into a 256-entry table of step values held in LDS, an escape code
for the rest) pay?  CPU-only model on the checker's sample positions (numpy statement of the march), per workload:
 1. how many distinct step values the map holds and what share of the SAMPLES the 253 most frequent ones cover
    (codes 253 / 254 = the two stop codes, 255 = escape: the lane reads the f32 map as today);
 2. counted L1 accesses per wave-load under the TCP's per-quad rule (profiles/r03/tcp_counter_probe: a quad of lanes on one
    128-B line costs 0.5, a divergent quad 0.5 per lane) for the f32 map's 4 x 8-cell lines against 8 x 16- and 16 x 8-cell
    lines of bytes and 8 x 8 of u16, same rank refill as the stream kernel;
 3. distinct lines touched per launch sample (footprint a launch pulls through the L2).
usage: python tools/r05/compact_map_model.py   (a few minutes on 8 cores; no GPU)"""
import os, sys, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pyracecarsimulator_amd import workloads
from oracle import oracle as O, np_statement as NS
f32 = np.float32


def march(w, n_poses):
    w = type(w)(**{**w.__dict__, "n_poses": n_poses})
    g = w.gmap
    dt = O.OracleMap.from_gridmap(g, w.max_range_px).dt
    poses = workloads.make_poses(w, dt=dt)
    rows, cols = dt.shape
    gx, gy, th = NS._pose_grid(g.resolution, g.origin, poses)
    st, ct = NS.sincosf(th)
    B = w.num_rays
    alpha = NS.fma(np.arange(B, dtype=f32), f32(f32(w.fov) / f32(B)), f32(f32(-0.5) * f32(w.fov)))
    sa, ca = NS.sincosf(alpha)
    dx = NS.fma(ct[:, None], ca[None, :], -(st[:, None] * sa[None, :]).astype(f32)).ravel()
    dy = NS.fma(st[:, None], ca[None, :], (ct[:, None] * sa[None, :]).astype(f32)).ravel()
    GX, GY = np.repeat(gx, B), np.repeat(gy, B)
    n = GX.size
    t = np.zeros(n, f32)
    live = np.ones(n, bool)
    ts, ids, prs, pcs, ds, first = [], [], [], [], [], []
    k = 0
    while True:
        live &= t < f32(w.max_range_px)
        idx = np.nonzero(live)[0]
        if idx.size == 0:
            break
        fx, fy = NS.fma(dx[idx], t[idx], GX[idx]), NS.fma(dy[idx], t[idx], GY[idx])
        inb = (fx > -1) & (fx < cols) & (fy > -1) & (fy < rows)
        live[idx[~inb]] = False
        idx, fx, fy = idx[inb], fx[inb], fy[inb]
        pc, pr = np.trunc(fx).astype(np.int64), np.trunc(fy).astype(np.int64)
        d = dt[pr, pc]
        ts.append(t[idx].copy()); ids.append(idx); prs.append(pr); pcs.append(pc); ds.append(d.copy())
        first.append(np.full(idx.size, k == 0))
        hit = d <= 0
        live[idx[hit]] = False
        go = idx[~hit]
        t[go] = (t[go] + np.maximum(d[~hit], f32(1.0))).astype(f32)
        k += 1
    cat = np.concatenate
    return dt, n, cat(ts), cat(ids), cat(prs), cat(pcs), cat(ds), cat(first)


def quad_cost(lines_per_wave):
    """lines_per_wave: (64,) line ids, -1 = idle lane -> counted accesses"""
    L = lines_per_wave.reshape(16, 4)
    s = 0.0
    for q in range(16):
        v = L[q][L[q] >= 0]
        if v.size:
            s += 0.5 if (v == v[0]).all() else 0.5 * v.size
    return s


def replay(n, ts, ids, layouts, low_water=20, NW=16, policy="rank"):
    order = np.lexsort((ts, ids))
    ids = ids[order]
    lay = {k: v[order] for k, v in layouts.items()}
    cnt = np.bincount(ids, minlength=n)
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    nxt, wl, lanes = 0, 0, 0
    acc = {k: 0.0 for k in lay}
    dl = {k: 0 for k in lay}
    waves = [(np.full(64, -1, np.int64), np.zeros(64, np.int64)) for _ in range(NW)]
    active = True
    while active:
        active = False
        for ray, pos in waves:
            live = ray >= 0
            if nxt < n and int(live.sum()) <= low_water:
                free = np.nonzero(~live)[0] if policy == "rank" else np.nonzero(np.repeat((~live).reshape(16, 4).all(1), 4))[0]
                k = min(free.size, n - nxt)
                if k:
                    ray[free[:k]] = np.arange(nxt, nxt + k); pos[free[:k]] = 0; nxt += k
                live = ray >= 0
                # (rays without a single load: finished at once)
                ray[live & (cnt[np.maximum(ray, 0)] == 0)] = -1
                live = ray >= 0
            if not live.any():
                continue
            active = True
            at = np.minimum(start[np.maximum(ray, 0)] + pos, ids.size - 1)
            for k_, v in lay.items():
                L = np.where(live, v[at], -1)
                acc[k_] += quad_cost(L)
                dl[k_] += np.unique(L[L >= 0]).size
            wl += 1; lanes += int(live.sum())
            pos[live] += 1
            ray[live & (pos >= cnt[np.maximum(ray, 0)])] = -1
    return wl, lanes / wl, {k: acc[k] / wl for k in lay}, {k: dl[k] / wl for k in lay}


def main():
    for name, npos in (("cfg2", 96), ("cfg5", 64)):
        w = getattr(workloads, name)()
        dt, n, ts, ids, pr, pc, d, first = march(w, npos)
        step = np.where(d <= 0, f32(np.inf), np.maximum(d, f32(1.0))).astype(f32)
        cells = np.where(dt <= 0, f32(np.inf), np.maximum(dt, f32(1.0))).astype(f32)
        vals, cc = np.unique(cells, return_counts=True)
        top = vals[np.argsort(-cc)[:253]]
        loads = ~first
        cov_samples = np.isin(step[loads], top).mean()
        cov_cells = np.isin(cells, top).mean()
        # the smallest 253 values instead of the most frequent ones (a code is then monotone in the step)
        low = vals[:253]
        print("%s (%s %dx%d, %d poses x %d beams): %.2f loads per ray; %d distinct step values in the map; the 253 most frequent cover "
              "%.2f %% of the cells and %.2f %% of the loads (the 253 smallest: %.2f %% of the loads, up to %.1f cells)"
              % (name, w.gmap.name, dt.shape[0], dt.shape[1], npos, w.num_rays, loads.sum() / n, vals.size, 100 * cov_cells,
                 100 * cov_samples, 100 * np.isin(step[loads], low).mean(), low[-1] if np.isfinite(low[-1]) else low[-2]), flush=True)
        lay = {"f32 4x8": (pr >> 2) * 16384 + (pc >> 3), "u8 8x16": (pr >> 3) * 16384 + (pc >> 4),
               "u8 16x8": (pr >> 4) * 16384 + (pc >> 3), "u16 8x8": (pr >> 3) * 16384 + (pc >> 3),
               "u8 4x16 (64-B sectors)": (pr >> 2) * 16384 + (pc >> 4)}
        sel = loads
        for policy, lw in (("rank", 20), ("quad", 32), ("quad", 44)):
            wl, ml, acc, dl = replay(n, ts[sel], ids[sel], {k: v[sel] for k, v in lay.items()}, low_water=lw, policy=policy)
            print("  %s refill at <= %d live lanes (16 waves on one stream): %d wave loads, %.1f live lanes each" % (policy, lw, wl, ml))
            for k in lay:
                print("    %-24s %5.2f counted accesses per wave-load (%7.0f in all), %5.2f distinct lines per wave-load; %7d distinct lines in all"
                      % (k, acc[k], acc[k] * wl, dl[k], np.unique(lay[k][sel]).size), flush=True)


if __name__ == "__main__":
    main()
