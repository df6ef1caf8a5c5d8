// literal_kernels.h — the AUDIT mode of the ray-marching methods (option variant 3): range_libc's CPU arithmetic as
// literally as it can be stated without its source (the reference reaches it at
// scripts/scan_simulator.py:103-106,130-133 and scripts/two_player/scan.py:57-70):
//   theta' = -theta_w + (-world_angle - 3pi/2);  calc_range(y, x, theta') marches (row, col) along
//   (cosf, sinf)(theta') with libm trig, every product and sum its own float32 rounding (no fma), per-ray angles
//   theta_p + (-fov/2 + j * (fov / num_rays)) rounded to float32 before the trig.
// The production kernels (rm_kernels.h) compute the canonical form instead — one deterministic sincos per pose, an
// angle-addition per beam, fused position updates — which lands in another hit cell on <= 1e-4 of the rays
// (DESIGN.md section 2).  This file exists so that the distance can be closed on demand: its results are
// bit-identical to the CPU checker's upstream-literal statement (orc_rm_fan_libm / orc_rm_rays_libm) on a host whose libm is glibc
// >= 2.28 with FMA (x86-64: the __sinf_fma / __cosf_fma variants), because lit_sinf / lit_cosf below are that
// libm's algorithm (ARM optimized routines' sinf.c / cosf.c / sincosf.h: double-precision reduction and
// polynomials, one rounding to float at the end) with its contractions written as explicit fma —
// the checker walks the same statement against the host's sinf / cosf over EVERY finite float (orc_libm_restatement_check:
// 0 mismatches on glibc 2.35).  One lane per ray, no schedule: 3-5x slower than K1b, never a default.
#pragma once
#include "scan_device.h"

namespace scan {

struct LiteralParams {
    float rotation_const;     // (float)(-world_angle - 3pi/2), double arithmetic on the host
    float wsin, wcos;         // (float)sin / cos of world_angle, the host's libm in double
};

namespace lit {
constexpr double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0, PI63 = 0x1.921FB54442D18p-62;
constexpr double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16, S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
                 S3 = -0x1.994eb3774cf24p-13;

__device__ __forceinline__ uint32_t abstop12(float x) { return (__builtin_bit_cast(uint32_t, x) >> 20) & 0x7ffu; }

// sinf_poly: n even -> sine polynomial of x (|x| <= pi/4), n odd -> cosine; `neg`: the table with negated cosine
// coefficients (quadrants 2, 3)
__device__ __forceinline__ float poly(double x, double x2, bool neg, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = __builtin_fma(x2, S3, S2);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, S1, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    const double sg = neg ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, sg * C4, sg * C3);
    const double c1 = __builtin_fma(x2, sg * C1, sg * C0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, sg * C2, c1);
    return (float)__builtin_fma(x6, c2, c);
}

__device__ __forceinline__ double reduce_fast(double x, int &n)
{
    const double r = x * HPI_INV;
    n = ((int32_t)r + 0x800000) >> 24;
    return __builtin_fma(-(double)n, HPI, x);
}

// |x| >= 120: 4/pi to 192 bits, 32-bit windows a byte apart
__device__ __forceinline__ double reduce_large(uint32_t xi, int &np)
{
    const uint32_t inv_pio4[24] = {0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u,
                                   0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u,
                                   0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u,
                                   0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (int)((xi >> 23) & 7);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    np = (int)n;
    return (double)(int64_t)res0 * PI63;
}

__device__ __forceinline__ bool flip(int q) { return ((q + 1) & 2) != 0; }      // sign[] = {1, -1, -1, 1}
}  // namespace lit

__device__ __forceinline__ float lit_sinf(float y)
{
    using namespace lit;
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return poly(x, x * x, false, 0);
    }
    if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, n);
        const double s = flip(n & 3) ? -1.0 : 1.0;
        return poly(x * s, x * x, (n & 2) != 0, n);
    }
    if (abstop12(y) < abstop12(__builtin_inff())) {
        const uint32_t xi = __builtin_bit_cast(uint32_t, y);
        const int sign = (int)(xi >> 31);
        x = reduce_large(xi, n);
        const double s = flip((n + sign) & 3) ? -1.0 : 1.0;
        return poly(x * s, x * x, ((n + sign) & 2) != 0, n);
    }
    return y - y;
}

__device__ __forceinline__ float lit_cosf(float y)
{
    using namespace lit;
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return poly(x, x * x, false, 1);
    }
    if (abstop12(y) < abstop12(120.0f))
        x = reduce_fast(x, n);
    else if (abstop12(y) < abstop12(__builtin_inff()))
        x = reduce_large(__builtin_bit_cast(uint32_t, y), n);
    else
        return y - y;
    const double s = flip((n + 1) & 3) ? -1.0 : 1.0;
    return poly(x * s, x * x, ((n + 1) & 2) != 0, n ^ 1);
}

// one upstream-literal cast from a world pose (the checker's rm_cast_libm)
__device__ __forceinline__ RayResult literal_cast(const MapParams &m, const LiteralParams &lp, float max_range,
                                                  float step_coeff, float xw, float yw, float thw)
{
    RayResult res;
    res.range_px = max_range;
    res.hit_c = -1;
    res.hit_r = -1;
    res.steps = 0;
    const float theta = -thw + lp.rotation_const;
    float x = (xw - m.ox) * m.inv_res;
    float y = (yw - m.oy) * m.inv_res;
    const float temp = x;
    x = lp.wcos * x - lp.wsin * y;
    y = lp.wsin * temp + lp.wcos * y;
    const float x0 = y, y0 = x;                       // calc_range(y, x, theta): the first coordinate indexes rows
    const float rdx = lit_cosf(theta), rdy = lit_sinf(theta);
    float t = 0.0f;
    while (t < max_range) {
        const float fx = x0 + rdx * t, fy = y0 + rdy * t;
        // (int) of a NaN is INT_MIN on the host (out of the map), 0 on the device: leave like the host does
        if (fx != fx || fy != fy) break;
        const int px = (int)fx, py = (int)fy;
        if (px >= m.rows || px < 0 || py < 0 || py >= m.cols) break;
        const float d = m.dt[(size_t)px * m.cols + py];
        ++res.steps;
        if (d <= 0.0f) {
            const float xd = (float)px - x0, yd = (float)py - y0;
            res.range_px = __builtin_sqrtf(xd * xd + yd * yd);
            res.hit_c = py;
            res.hit_r = px;
            break;
        }
        const float st = d * step_coeff;
        t += st > 1.0f ? st : 1.0f;
    }
    return res;
}

// fan form (the fork's 4-argument calc_range_many): beam j of pose p at theta_p + (amin + j * inc), the sum and the
// product each rounded to float32.  RAYS: the upstream 2-argument form, one (x, y, theta) row per ray.
template <bool AUX, bool RAYS>
__global__ __launch_bounds__(256) void rm_literal_kernel(MapParams m, FanParams f, LiteralParams lp,
                                                         const float *__restrict__ poses, long n_rays,
                                                         float *__restrict__ out, int32_t *__restrict__ hits,
                                                         uint16_t *__restrict__ steps)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += stride) {
        float xw, yw, th;
        if (RAYS) {
            xw = poses[3 * i]; yw = poses[3 * i + 1]; th = poses[3 * i + 2];
        } else {
            const long p = i / f.num_rays;
            const int j = (int)(i - p * f.num_rays);
            const float aj = (float)j * f.inc;
            xw = poses[3 * p]; yw = poses[3 * p + 1];
            th = poses[3 * p + 2] + (f.amin + aj);
        }
        const RayResult r = literal_cast(m, lp, f.max_range, f.step_coeff, xw, yw, th);
        float v = r.range_px * m.res;
        if (f.noise_std > 0.0f) v += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (uint64_t)i);
        if (out) out[i] = v;
        if (AUX) {
            if (hits) { hits[2 * i] = r.hit_c; hits[2 * i + 1] = r.hit_r; }
            if (steps) steps[i] = (uint16_t)(r.steps > 65535u ? 65535u : r.steps);
        }
    }
}

// diagnostics: lit_sinf / lit_cosf of an array (rl_probe_literal_sincosf: the GPU test holds it against the host's libm)
__global__ __launch_bounds__(256) void literal_sincosf_kernel(const float *__restrict__ x, long n, float *__restrict__ s,
                                                              float *__restrict__ c)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        s[i] = lit_sinf(x[i]);
        c[i] = lit_cosf(x[i]);
    }
}

}  // namespace scan
