"""Independent NumPy/SciPy statement of the sphere-tracing scan — TEST INFRASTRUCTURE ONLY.

A second, differently-structured statement of SURVEY.md Appendix A used to
cross-check oracle/rangelib_oracle.c: the distance transform
comes from scipy.ndimage, the deterministic trig is re-derived here in float32
NumPy with an emulated single-rounding fma (float64 product + sum is exact
enough: a float32*float32 product is exact in float64, and the float64 add
followed by one rounding to float32 differs from a true fma only in
double-rounding corner cases, which the cross-check would surface), and the
march is vectorised over rays instead of looping over them.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def fma(a, b, c):
    """float32 fused multiply-add emulated through float64."""
    return (np.asarray(a, f32).astype(np.float64) * np.asarray(b, f32).astype(np.float64)
            + np.asarray(c, f32).astype(np.float64)).astype(f32)


def sincosf(x):
    x = np.asarray(x, f32)
    two_over_pi = f32(float.fromhex("0x1.45f306p-1"))
    p1 = f32(float.fromhex("0x1.921fb6p+0"))
    p2 = f32(float.fromhex("-0x1.777a5cp-25"))
    p3 = f32(float.fromhex("-0x1.ee59dap-50"))
    k = np.rint(x * two_over_pi).astype(f32)
    r = fma(-k, p1, x)
    r = fma(-k, p2, r)
    r = fma(-k, p3, r)
    z = (r * r).astype(f32)
    ps = fma(z, f32(-1.9515295891e-4), f32(8.3321608736e-3))
    ps = fma(z, ps, f32(-1.6666654611e-1))
    sr = fma((r * z).astype(f32), ps, r)
    pc = fma(z, f32(2.443315711809948e-5), f32(-1.388731625493765e-3))
    pc = fma(z, pc, f32(4.166664568298827e-2))
    cr = fma((z * z).astype(f32), pc, fma(z, f32(-0.5), f32(1.0)))
    q = k.astype(np.int64) & 3
    s = np.where(q & 1, cr, sr)
    c = np.where(q & 1, sr, cr)
    c = np.where((q == 1) | (q == 2), -c, c)
    s = np.where(q >= 2, -s, s)
    return s.astype(f32), c.astype(f32)


def edt(occ):
    """Exact EDT in cells as float32 (scipy works in float64; d^2 is an integer)."""
    from scipy import ndimage
    occ = np.asarray(occ) != 0
    if not occ.any():
        return np.full(occ.shape, 1e10, dtype=f32)
    d = ndimage.distance_transform_edt(~occ)
    d2 = np.rint(d * d)                    # exact integer squared distance
    return np.sqrt(d2.astype(f32)).astype(f32)


def rm_fan(occ, resolution, origin, max_range_px, poses, fov, num_rays, step_coeff=0.999,
           dt=None):
    """Vectorised sphere tracing; returns (ranges f32[P*B], hits i32[P*B,2], steps u16[P*B])."""
    occ = np.asarray(occ)
    rows, cols = occ.shape
    dt = edt(occ) if dt is None else np.asarray(dt, f32)
    poses = np.asarray(poses, f32).reshape(-1, 3)
    res, ox, oy, yaw = f32(resolution), f32(origin[0]), f32(origin[1]), f32(origin[2])
    wa = f32(-yaw)
    wsin, wcos = sincosf(wa)
    inv = f32(1.0 / float(res))
    x = ((poses[:, 0] - ox) * inv).astype(f32)
    y = ((poses[:, 1] - oy) * inv).astype(f32)
    gx = fma(wcos, x, -(wsin * y).astype(f32))
    gy = fma(wsin, x, (wcos * y).astype(f32))
    st, ct = sincosf((poses[:, 2] + wa).astype(f32))
    j = np.arange(num_rays, dtype=f32)
    alpha = fma(j, f32(f32(fov) / f32(num_rays)), f32(f32(-0.5) * f32(fov)))
    sa, ca = sincosf(alpha)
    dx = fma(ct[:, None], ca[None, :], -(st[:, None] * sa[None, :]).astype(f32)).ravel()
    dy = fma(st[:, None], ca[None, :], (ct[:, None] * sa[None, :]).astype(f32)).ravel()
    gx = np.repeat(gx, num_rays)
    gy = np.repeat(gy, num_rays)
    n = gx.size
    mr = f32(max_range_px)
    t = np.zeros(n, f32)
    out = np.full(n, mr, f32)
    hits = np.full((n, 2), -1, np.int32)
    steps = np.zeros(n, np.int64)
    live = np.ones(n, bool)
    while True:
        live &= t < mr
        idx = np.nonzero(live)[0]
        if idx.size == 0:
            break
        fx = fma(dx[idx], t[idx], gx[idx])
        fy = fma(dy[idx], t[idx], gy[idx])
        inb = (fx > -1) & (fx < cols) & (fy > -1) & (fy < rows)
        live[idx[~inb]] = False
        idx, fx, fy = idx[inb], fx[inb], fy[inb]
        pc = np.trunc(fx).astype(np.int64)
        pr = np.trunc(fy).astype(np.int64)
        d = dt[pr, pc]
        steps[idx] += 1
        hit = d <= 0
        hi = idx[hit]
        xd = (pc[hit].astype(f32) - gx[hi]).astype(f32)
        yd = (pr[hit].astype(f32) - gy[hi]).astype(f32)
        out[hi] = np.sqrt(fma(xd, xd, (yd * yd).astype(f32))).astype(f32)
        hits[hi, 0] = pc[hit]
        hits[hi, 1] = pr[hit]
        live[hi] = False
        go = idx[~hit]
        t[go] = (t[go] + np.maximum((d[~hit] * f32(step_coeff)).astype(f32), f32(1.0))).astype(f32)
    return (out * res).astype(f32), hits, np.minimum(steps, 65535).astype(np.uint16)


# ----------------------------------------------------------------------------------------------
# Second statements of the other casters (round 2): BresenhamsLine, GiantLUTCast, CDDTCast, written
# array-at-a-time from SURVEY.md rows a12-a14 / Appendix A, to cross-check rangelib_oracle.c bit for
# bit (tests/test_oracle.py) — until then those three were pinned by properties only.
# ----------------------------------------------------------------------------------------------
def _pose_grid(resolution, origin, poses):
    """world poses -> (gx, gy, grid heading) float32 (row a9)."""
    poses = np.asarray(poses, f32).reshape(-1, 3)
    res, ox, oy, yaw = f32(resolution), f32(origin[0]), f32(origin[1]), f32(origin[2])
    wa = f32(-yaw)
    wsin, wcos = sincosf(wa)
    inv = f32(1.0 / float(res))
    x = ((poses[:, 0] - ox) * inv).astype(f32)
    y = ((poses[:, 1] - oy) * inv).astype(f32)
    gx = fma(wcos, x, -(wsin * y).astype(f32))
    gy = fma(wsin, x, (wcos * y).astype(f32))
    return gx, gy, (poses[:, 2] + wa).astype(f32)


def _fan_alpha(fov, num_rays):
    j = np.arange(num_rays, dtype=f32)
    return fma(j, f32(f32(fov) / f32(num_rays)), f32(f32(-0.5) * f32(fov)))


def _fan_dirs(resolution, origin, poses, fov, num_rays):
    gx, gy, thg = _pose_grid(resolution, origin, poses)
    st, ct = sincosf(thg)
    sa, ca = sincosf(_fan_alpha(fov, num_rays))
    dx = fma(ct[:, None], ca[None, :], -(st[:, None] * sa[None, :]).astype(f32)).ravel()
    dy = fma(st[:, None], ca[None, :], (ct[:, None] * sa[None, :]).astype(f32)).ravel()
    return np.repeat(gx, num_rays), np.repeat(gy, num_rays), dx, dy


def _trunc_i(v):
    return np.trunc(v).astype(np.int64)


def bl_fan(occ, resolution, origin, max_range_px, poses, fov, num_rays):
    """BresenhamsLine over a fan: (ranges f32, hits i32[n,2], steps u16), all rays stepped together."""
    occ = np.asarray(occ) != 0
    rows, cols = occ.shape
    gx, gy, dx, dy = _fan_dirs(resolution, origin, poses, fov, num_rays)
    n = gx.size
    mr = f32(max_range_px)
    out = np.full(n, mr, f32)
    hits = np.full((n, 2), -1, np.int32)
    steps = np.zeros(n, np.int64)
    with np.errstate(invalid="ignore", over="ignore"):
        sane = (np.abs(gx) < f32(1e9)) & (np.abs(gy) < f32(1e9)) & (((dx - dx) + (dy - dy)) == 0)
        inmap = sane & (gx > -1) & (gx < cols) & (gy > -1) & (gy < rows)
        sc, sr = _trunc_i(np.where(inmap, gx, 0)), _trunc_i(np.where(inmap, gy, 0))
        start_occ = inmap & occ[sr, sc]
        out[start_occ] = 0
        hits[start_occ, 0] = sc[start_occ]
        hits[start_occ, 1] = sr[start_occ]
        walk = sane & ~start_occ
        x0, y0 = gx.copy(), gy.copy()
        x1, y1 = fma(mr, dx, gx), fma(mr, dy, gy)
        steep = np.abs((y1 - y0).astype(f32)) > np.abs((x1 - x0).astype(f32))
        x0, y0 = np.where(steep, y0, x0).astype(f32), np.where(steep, x0, y0).astype(f32)
        x1, y1 = np.where(steep, y1, x1).astype(f32), np.where(steep, x1, y1).astype(f32)
        lim_major = np.where(steep, f32(rows), f32(cols)).astype(f32)
        lim_minor = np.where(steep, f32(cols), f32(rows)).astype(f32)
        deltax, deltay = np.abs((x1 - x0).astype(f32)), np.abs((y1 - y0).astype(f32))
        error = np.zeros(n, f32)
        _x, _y = x0.copy(), y0.copy()
        xstep = np.where(x0 < x1, f32(1), f32(-1)).astype(f32)
        ystep = np.where(y0 < y1, f32(1), f32(-1)).astype(f32)
        end = _trunc_i(np.where(walk, (x1 + xstep).astype(f32), 0))
        cap = np.full(n, int(np.trunc(float(mr))) + 3, np.int64)
        live = walk.copy()
        while True:
            go = live & (_trunc_i(np.where(live, _x, 0)) != end) & (cap > 0)
            cap[live & (_trunc_i(np.where(live, _x, 0)) != end)] -= 1
            live = go
            if not live.any():
                break
            i = np.nonzero(live)[0]
            _x[i] = (_x[i] + xstep[i]).astype(f32)
            error[i] = (error[i] + deltay[i]).astype(f32)
            bump = (error[i] * f32(2)).astype(f32) >= deltax[i]
            ib = i[bump]
            _y[ib] = (_y[ib] + ystep[ib]).astype(f32)
            error[ib] = (error[ib] - deltax[ib]).astype(f32)
            steps[i] += 1
            inb = (_x[i] >= 0) & (_x[i] < lim_major[i]) & (_y[i] >= 0) & (_y[i] < lim_minor[i])
            k = i[inb]
            col = np.where(steep[k], _trunc_i(_y[k]), _trunc_i(_x[k]))
            row = np.where(steep[k], _trunc_i(_x[k]), _trunc_i(_y[k]))
            h = occ[row, col]
            kh = k[h]
            xd, yd = (_x[kh] - x0[kh]).astype(f32), (_y[kh] - y0[kh]).astype(f32)
            out[kh] = np.sqrt(fma(xd, xd, (yd * yd).astype(f32))).astype(f32)
            hits[kh, 0] = col[h]
            hits[kh, 1] = row[h]
            live[kh] = False
    return (out * f32(resolution)).astype(f32), hits, np.minimum(steps, 65535).astype(np.uint16)


def _theta_bin(th, theta_disc):
    """nearest bin of th in [0, theta_disc): rint(th * theta_disc / 2pi) mod theta_disc."""
    bpr = f32(f32(theta_disc) * f32(0.15915494309189535))
    with np.errstate(invalid="ignore", over="ignore"):
        u = np.rint((np.asarray(th, f32) * bpr).astype(f32))
        u = np.where((u > -1e9) & (u < 1e9), u, 0)
    return np.mod(u.astype(np.int64), theta_disc)


def lut_fan(lut, occ_shape, resolution, origin, max_range_px, poses, fov, num_rays):
    """GiantLUTCast query over a fan; ``lut`` uint16 (rows, cols, theta_disc)."""
    rows, cols = occ_shape
    td = lut.shape[2]
    gx, gy, thg = _pose_grid(resolution, origin, poses)
    alpha = _fan_alpha(fov, num_rays)
    th = (thg[:, None] + alpha[None, :]).astype(f32)
    with np.errstate(invalid="ignore"):
        inb = (gx >= 0) & (gx < cols) & (gy >= 0) & (gy < rows)
    r = _trunc_i(np.where(inb, gy, 0))
    c = _trunc_i(np.where(inb, gx, 0))
    b = _theta_bin(th, td)
    q = lut[r[:, None], c[:, None], b].astype(f32)
    mr, res = f32(max_range_px), f32(resolution)
    val = ((q * f32(mr / f32(65535.0))).astype(f32) * res).astype(f32)
    return np.where(inb[:, None], val, f32(mr * res)).astype(f32).ravel()


class CddtTable:
    """CDDTCast table (row a13): per theta bin in [0, pi), every edge cell's centre projected into the
    bin's rotated frame; bucket = integer rotated row (a cell covers every bucket its half-width
    reaches); each bucket holds the sorted, de-duplicated rotated x of the cells it covers."""
    EPS = f32(1e-5)

    def __init__(self, occ, theta_disc):
        occ = np.asarray(occ) != 0
        rows, cols = occ.shape
        self.rows, self.cols, self.td = rows, cols, int(theta_disc)
        self.nb = (self.td + 1) // 2
        pad = np.pad(occ, 1, constant_values=False)
        all4 = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
        border = np.zeros_like(occ)
        border[0, :] = border[-1, :] = border[:, 0] = border[:, -1] = True
        edge = occ & (~all4 | border)
        er, ec = np.nonzero(edge)
        px, py = (ec.astype(f32) + f32(0.5)).astype(f32), (er.astype(f32) + f32(0.5)).astype(f32)
        ang = (np.arange(self.nb, dtype=f32) * f32(f32(6.283185307179586) / f32(self.td))).astype(f32)
        self.sin, self.cos = sincosf(ang)
        W, H = f32(cols), f32(rows)
        height = (np.abs((W * self.sin).astype(f32)) + np.abs((H * self.cos).astype(f32))).astype(f32)
        self.width = (np.ceil((height - self.EPS).astype(f32)).astype(np.int64) + 1)
        lt = (H * self.cos).astype(f32)
        rt = fma(W, self.sin, lt)
        rb = (W * self.sin).astype(f32)
        mn = np.minimum(lt, np.minimum(rt, rb))
        self.trans = np.maximum(f32(0), ((-mn).astype(f32) - self.EPS).astype(f32)).astype(f32)
        self.buckets = []
        for a in range(self.nb):
            cs, sn = self.cos[a], self.sin[a]
            half = f32((abs(sn) + abs(cs)) * f32(0.5))
            lx = fma(px, cs, -(py * sn).astype(f32))
            ly = (fma(px, sn, (py * cs).astype(f32)) + self.trans[a]).astype(f32)
            upper = _trunc_i(((ly + half).astype(f32) - self.EPS).astype(f32))
            lower = _trunc_i(((ly - half).astype(f32) + self.EPS).astype(f32))
            lower = np.maximum(lower, 0)
            upper = np.minimum(upper, self.width[a] - 1)
            per = [[] for _ in range(int(self.width[a]))]
            for k in range(len(lx)):
                for bkt in range(int(lower[k]), int(upper[k]) + 1):
                    per[bkt].append(lx[k])
            self.buckets.append([np.unique(np.asarray(v, f32)) for v in per])

    def query(self, gx, gy, th, max_range):
        out = np.full(len(gx), f32(max_range), f32)
        b = _theta_bin((-np.asarray(th, f32)).astype(f32), self.td)
        flipped = b >= self.nb
        b = np.where(flipped, b - self.td // 2, b)
        b = np.minimum(b, self.nb - 1)
        for i in range(len(gx)):
            a = int(b[i])
            cs, sn = self.cos[a], self.sin[a]
            lx = fma(gx[i], cs, -f32(gy[i] * sn))
            ly = f32(fma(gx[i], sn, f32(gy[i] * cs)) + self.trans[a])
            if not (ly >= 0 and ly < f32(self.width[a])):
                continue
            xs = self.buckets[a][int(np.trunc(ly))]
            if not flipped[i]:
                k = np.searchsorted(xs, lx, side="left")            # first stored x >= lx
                if k < len(xs):
                    out[i] = min(f32(xs[k] - lx), f32(max_range))
            else:
                k = np.searchsorted(xs, lx, side="right")           # last stored x <= lx
                if k > 0:
                    out[i] = min(f32(lx - xs[k - 1]), f32(max_range))
        return out


def cddt_fan(table, resolution, origin, max_range_px, poses, fov, num_rays):
    gx, gy, thg = _pose_grid(resolution, origin, poses)
    alpha = _fan_alpha(fov, num_rays)
    th = (thg[:, None] + alpha[None, :]).astype(f32).ravel()
    r = table.query(np.repeat(gx, num_rays), np.repeat(gy, num_rays), th, max_range_px)
    return (r * f32(resolution)).astype(f32)
