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
