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
#include "literal_math.h"

namespace scan {

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

}  // namespace scan
