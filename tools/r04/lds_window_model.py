#!/usr/bin/env python3
"""What would staging the STEP MAP in LDS (north_star's "grid tile in LDS", which occ_fan_lds has for occupancy) buy the
EXACT march?  CPU-only model on the oracle's sample positions (numpy statement of the march, oracle/np_statement.py):

 1. a window per POSE (cfg2: one pose per 1024 cells): share of a ray's step-map loads with t < R (a window of 2R x 2R
    cells holds them whatever the direction), rays that finish inside, staged cells per LDS read;
 2. a window per TILE of poses (dense batches, cfg3 / cfg5): share of the loads inside a tile-centred window;
 3. the refill policy against the L1's per-quad rule (profiles/r03/tcp_counter_probe: a quad of lanes on one 128-B line
    costs 0.5 counted accesses, a divergent quad 0.5 per lane): rank refill (the stream kernel's), quad-granular
    refill, lock step.
usage: python tools/r04/lds_window_model.py          (about two minutes on 8 cores; needs no GPU)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyracecarsimulator_amd import workloads
from oracle import oracle as O, np_statement as NS
f32 = np.float32


def march(w, n_poses):
    """-> per-ray arrays of (t, cell offset from the pose's tile centre, 128-B line id) for every map sample."""
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
    T = 32
    tcx, tcy = np.floor(GX / T) * T + T / 2, np.floor(GY / T) * T + T / 2
    t = np.zeros(n, f32)
    live = np.ones(n, bool)
    ts, offs, ids, lines, first = [], [], [], [], []
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
        ts.append(t[idx].copy()); ids.append(idx); first.append(np.full(idx.size, k == 0))
        offs.append(np.maximum(np.abs(pc - tcx[idx]), np.abs(pr - tcy[idx])))
        lines.append((pr >> 2) * 8192 + (pc >> 3))           # the tiled step map: 4 rows x 8 columns per 128-B line
        hit = d <= 0
        live[idx[hit]] = False
        go = idx[~hit]
        t[go] = (t[go] + np.maximum(d[~hit], f32(1.0))).astype(f32)
        k += 1
    cat = np.concatenate
    return n, cat(ts), cat(offs), cat(ids), cat(lines), cat(first)


def per_pose_window():
    w = workloads.cfg2()
    n, ts, _, ids, _, first = march(w, 256)
    loads = ~first                                           # the sample at t = 0 is read once per pose, with its record
    print("cfg2, 256 poses: %.2f map samples per ray, %.2f loads (the first is the pose's)" % (ts.size / n, loads.sum() / n))
    for R in (28, 44, 60, 90):
        ins = ts < R
        out_per_ray = np.bincount(ids[~ins], minlength=n)
        cells = (2 * (R + 4)) ** 2
        print("  window of %3d x %3d cells (%3d KB): %4.1f %% of the loads inside, %4.1f %% of the rays finish inside; "
              "%6d cells staged for %5.0f LDS reads per pose" % (2 * (R + 4), 2 * (R + 4), cells * 4 // 1024,
              100 * (ins & loads).sum() / loads.sum(), 100 * (out_per_ray == 0).mean(), cells, (ins & loads).sum() / 256))
    print("  4096 windows of 128 x 128 cells = %.0f MB of L2 -> LDS traffic per launch (the gather moves 46-56 MB)" % (4096 * 65536 / 1e6))
    return n, ts, ids


def per_tile_window():
    for name, npos in (("cfg3", 192), ("cfg5", 128)):
        n, ts, offs, ids, _, first = march(getattr(workloads, name)(), npos)
        print("%s (16 poses per 32 x 32-cell tile at full size): %.2f loads per ray" % (name, (~first).sum() / n))
        for H in (48, 64, 80):
            print("  tile-centred window of %3d x %3d cells (%3d KB): %4.1f %% of the loads inside" % (
                2 * H, 2 * H, (2 * H) ** 2 * 4 // 1024, 100 * (offs[~first] < H).mean()))


def refill_policies():
    w = workloads.cfg2()
    n, ts, _, ids, lines, _ = march(w, 96)
    order = np.lexsort((ts, ids))
    ids, lines = ids[order], lines[order]
    cnt = np.bincount(ids, minlength=n)
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    NW = 16

    def cost(ray, pos):
        live = ray >= 0
        L = np.where(live, lines[np.minimum(start[np.maximum(ray, 0)] + pos, lines.size - 1)], -1).reshape(16, 4)
        s = 0.0
        for q in range(16):
            v = L[q][L[q] >= 0]
            if v.size:
                s += 0.5 if len(set(v.tolist())) == 1 else 0.5 * v.size
        return int(live.sum()), s

    def simulate(policy, low_water):
        nxt, wl, lanes, acc = 0, 0, 0, 0.0
        waves = [(np.full(64, -1, np.int64), np.zeros(64, np.int64)) for _ in range(NW)]
        active = True
        while active:
            active = False
            for ray, pos in waves:
                live = ray >= 0
                if nxt < n and int(live.sum()) <= low_water:
                    if policy == "rank":
                        free = np.nonzero(~live)[0]
                    elif policy == "quad":
                        free = np.nonzero(np.repeat((~live).reshape(16, 4).all(1), 4))[0]
                    else:
                        free = np.nonzero(~live)[0] if not live.any() else np.array([], np.int64)
                    k = min(free.size, n - nxt)
                    if k:
                        ray[free[:k]] = np.arange(nxt, nxt + k); pos[free[:k]] = 0; nxt += k
                    live = ray >= 0
                if not live.any():
                    continue
                active = True
                a, s = cost(ray, pos)
                wl += 1; lanes += a; acc += s
                pos[live] += 1
                ray[live & (pos >= cnt[np.maximum(ray, 0)])] = -1
        return wl, lanes / wl, acc / wl, acc

    print("refill policy against the L1's per-quad rule (96 cfg2 poses, 16 waves sharing one ray stream):")
    for policy, lw in (("rank", 20), ("quad", 32), ("quad", 52), ("lockstep", 0)):
        wl, ml, a, tot = simulate(policy, lw)
        print("  %-8s refill at <= %2d live lanes: %6d wave loads, %4.1f lanes each, %5.2f counted accesses per load, %7.0f in all"
              % (policy, lw, wl, ml, a, tot))


if __name__ == "__main__":
    per_pose_window()
    per_tile_window()
    refill_policies()
