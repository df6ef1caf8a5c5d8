// scan_device.h — device-side arithmetic of the lidar scan path (gfx950 only).
//
// The float32 arithmetic here is the canonical form of SURVEY.md Appendix A:
// every operation that feeds a float->int truncation is a single IEEE operation
// or an explicit fma, the translation unit is built with -ffp-contract=off, and
// no libm/OCML trig is used, so results are bit-identical to the CPU statement
// the parity tests check against.  Behaviour restated (not ported) from
// range_libc's RayMarching::calc_range / kernels.cu cuda_ray_marching
// (SURVEY.md rows a8, a9, a11), reached from the reference at
// scripts/scan_simulator.py:103-106,130-133.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scan {

struct MapParams {
    const float *dt;          // exact EDT in cells, row-major [rows][cols]
    const uint32_t *bits;     // bit-packed occupancy, bits_stride words per row
    int bits_stride;
    int rows, cols;
    float frows, fcols;
    float res, inv_res;       // world_scale and (float)(1.0/res)
    float ox, oy;             // world origin
    float wa, wa_cos, wa_sin; // world_angle = -yaw and its det_sincosf
};

struct FanParams {
    int n_poses, num_rays;
    float amin, inc;          // -fov/2 and fov/num_rays (computed on the host, IEEE)
    float max_range;          // cells
    float step_coeff;         // 0.999f (RayMarching) or 1.0f (kernels.cu)
    float noise_std;          // <= 0: off
    uint64_t noise_seed, ray_offset;
};

// ---- deterministic sin/cos: 3-term Cody-Waite by pi/2 + minimax polynomials ----
__device__ __forceinline__ void det_sincosf(float x, float &s, float &c)
{
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float P1 = 0x1.921fb6p+0f, P2 = -0x1.777a5cp-25f, P3 = -0x1.ee59dap-50f;
    float k = __builtin_rintf(x * TWO_OVER_PI);
    float r = __builtin_fmaf(-k, P1, x);
    r = __builtin_fmaf(-k, P2, r);
    r = __builtin_fmaf(-k, P3, r);
    float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    float sr = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    float cr = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    float ss = (q & 1) ? cr : sr;
    float cc = (q & 1) ? sr : cr;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    s = ss;
    c = cc;
}

// world pose -> grid position (col,row units) and grid heading (row a9)
__device__ __forceinline__ void world_to_grid(const MapParams &m, float xw, float yw, float thw,
                                              float &gx, float &gy, float &thg)
{
    float x = (xw - m.ox) * m.inv_res;
    float y = (yw - m.oy) * m.inv_res;
    gx = __builtin_fmaf(m.wa_cos, x, -(m.wa_sin * y));
    gy = __builtin_fmaf(m.wa_sin, x, m.wa_cos * y);
    thg = thw + m.wa;
}

// beam j of the fan: alpha_j = -fov/2 + j*fov/num_rays (scripts/ros_interface.py:342-344)
__device__ __forceinline__ float fan_alpha(const FanParams &f, int j)
{
    return __builtin_fmaf((float)j, f.inc, f.amin);
}

// ---- counter-based Gaussian noise (row a15): Philox-2x32-10 + Box-Muller ----------
// One normal per ray needs 2 x 24 random bits: the 2x32 member of the Philox family (Salmon et
// al., "Parallel random numbers: as easy as 1, 2, 3", 10 rounds, BigCrush-clean) yields exactly
// that for 4 integer multiplies per round — half of Philox-4x32.  Counter = global ray id, key =
// the 64-bit seed folded to 32 bits, so a sharded batch reproduces the unsharded one.
__device__ __forceinline__ float gauss_noise(uint64_t seed, uint64_t ray_id)
{
    uint32_t c0 = (uint32_t)ray_id, c1 = (uint32_t)(ray_id >> 32);
    uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // (the 64-bit product in one v_mad_u64_u32 instead of v_mul_hi_u32 + v_mul_lo_u32)
        const uint64_t prod = (uint64_t)0xD256D193u * c0;
        c0 = (uint32_t)(prod >> 32) ^ k ^ c1;
        c1 = (uint32_t)prod;
        k += 0x9E3779B9u;
    }
    // two uniforms, u1 in (0,1], u2 in [0,1): Box-Muller with the hardware log / cos / sqrt estimates
    const float u1 = ((float)(c0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(c1 >> 8) * (1.0f / 16777216.0f);
    return __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// ---- sphere tracing on the float32 distance transform (rows a8 / a11) -------------
// Correctly rounded sqrt for the hit distance sqrt(xd^2 + yd^2): the argument is 0 or >= 2^-46
// (xd, yd are differences of cell indices and in-map coordinates), never denormal, never inf, so the
// compiler's general sequence (range scaling in front, class test behind: 16 instructions) reduces
// to the hardware estimate (<= 1 ulp) and the two-sided residual fix-up (9 instructions).  Same bits
// as sqrtf for every such argument (tests compare against the CPU's IEEE sqrtf).
__device__ __forceinline__ float hit_sqrtf(float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    const float yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    const float rm = __builtin_fmaf(-ym, y, x);
    const float rp = __builtin_fmaf(-yp, y, x);
    float r = rm <= 0.0f ? ym : y;
    r = rp > 0.0f ? yp : r;
    return r;
}

// The range of a finished ray.  The stream kernels store it with one scattered 4-B store per lane, write-once and
// never read back by the launch: a NON-TEMPORAL store keeps the 4 B per ray (17.7 MB per cfg2 launch, four launches
// in flight) from displacing the band of the step map the XCD's L2 is supposed to hold — cfg2 serial +6 % (lone
// kernel 44.3 -> 42.2 us), cfg3 RMGPU +2.8 %, cfg5 shard +2 %, cfg4 shard +1.3 %, cfg2 pipelined +0.8 %
// (profiles/r04/nt_store_ab.txt; -DRL_PLAIN_STORE rebuilds the other side of the A/B).  A kernel that reads the ranges
// back at once on the same stream (FollowGap: --gather steer -4.6 %) wants them in the L2: option nt_store 0.
__device__ __forceinline__ void range_store(float *base, uint32_t byte_off, float r, int plain = 0)
{
    // (base + 32-bit byte offset formed inside each branch: the store keeps its SGPR-base addressing)
#ifdef RL_PLAIN_STORE
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = r;
#else
    if (plain) {      // wave-uniform (a launch parameter)
        *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = r;
    } else {
        // (the empty asm statements keep the two stores apart — hoisting and sinking both stop at them —: two stores
        //  of one value to one address that differ only in their non-temporal hint become ONE PLAIN store in the
        //  optimiser — the first build with this branch had no `nt` store left in it.  The store itself stays the
        //  compiler's: written as inline asm it missed the wait states a VALU-written SGPR base needs in front of a
        //  VMEM instruction, and faulted in the AUX kernels, which keep `out` in a spill lane)
        asm volatile("" ::: "memory");
        __builtin_nontemporal_store(r, reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off));
        asm volatile("" ::: "memory");
    }
#endif
}

struct RayResult {
    float range_px;
    int hit_c, hit_r;
    unsigned steps;
};

__device__ __forceinline__ RayResult rm_march(const MapParams &m, float max_range,
                                              float step_coeff, float gx, float gy, float dx,
                                              float dy)
{
    RayResult res;
    res.range_px = max_range;
    res.hit_c = -1;
    res.hit_r = -1;
    res.steps = 0;
    float t = 0.0f;
    while (t < max_range) {
        float fx = __builtin_fmaf(dx, t, gx);
        float fy = __builtin_fmaf(dy, t, gy);
        // same cell set as (int)fx in [0,cols) && (int)fy in [0,rows); NaN/huge -> miss
        if (!(fx > -1.0f && fx < m.fcols && fy > -1.0f && fy < m.frows)) break;
        int pc = (int)fx, pr = (int)fy;
        float d = m.dt[(size_t)pr * m.cols + pc];
        ++res.steps;
        if (d <= 0.0f) {
            float xd = (float)pc - gx;
            float yd = (float)pr - gy;
            res.range_px = __builtin_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
            res.hit_c = pc;
            res.hit_r = pr;
            break;
        }
        t += __builtin_fmaxf(d * step_coeff, 1.0f);
    }
    return res;
}

}  // namespace scan
