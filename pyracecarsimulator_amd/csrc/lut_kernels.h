// lut_kernels.h — K3: GiantLUTCast, table build and the fan kernels.  Part of scan_kernels.h.
#pragma once
#include "scan_device.h"
#include "scan_params.h"
#include "rm_kernels.h"

// ==============================================================================
// K3: GiantLUTCast — the bandwidth-bound variant.
// Table: uint16 lut[row][col][theta_bin], so the fan of one pose is ONE contiguous
// run of ~num_rays entries (bin spacing ~ beam spacing when theta_disc ~ 2pi*B/fov):
// a query streams ~2 B/ray in and 4 B/ray out, nothing else.  2000^2 x 1442 bins =
// 11.5 GB of the 288 GB HBM.  Built on the device with the K1 march from every cell
// corner (range_libc seeds its table with RayMarching the same way).
// ==============================================================================
namespace scan {


__device__ __forceinline__ int lut_bin(float th, const LutParams &lp)
{
    float u = __builtin_rintf(th * lp.bins_per_rad);
    if (!(u > -1e9f && u < 1e9f)) u = 0.0f;
    int b = (int)u % lp.theta_disc;
    return b < 0 ? b + lp.theta_disc : b;
}

// one workgroup per (row, 4-column group); lane = theta bin
__global__ __launch_bounds__(256) void lut_build_kernel(MapParams m, LutParams lp, float max_range,
                                                        float step_coeff, int row0, int row1)
{
    const long cells = (long)(row1 - row0) * m.cols;
    for (long cell = blockIdx.x; cell < cells; cell += gridDim.x) {
        const int r = row0 + (int)(cell / m.cols), c = (int)(cell % m.cols);
        uint16_t *dst = lp.lut + ((size_t)r * m.cols + c) * lp.theta_disc;
        for (int b = threadIdx.x; b < lp.theta_disc; b += blockDim.x) {
            float dx, dy;
            det_sincosf((float)b * lp.bin_width, dy, dx);
            RayResult rr = rm_march(m, max_range, step_coeff, (float)c, (float)r, dx, dy);
            float q = __builtin_rintf(__builtin_fminf(rr.range_px, max_range) * lp.quant);
            dst[b] = (uint16_t)q;
        }
    }
}

// nearest-bin index without an integer division: u is an integer-valued float; for
// |u| < 2^23 the float wrap below is exact and equals ((int)u % td + td) % td
__device__ __forceinline__ int lut_bin_fast(float th, const LutParams &lp, float td_f, float inv_td)
{
    const float u = __builtin_rintf(th * lp.bins_per_rad);
    if (!(__builtin_fabsf(u) < 8388608.0f)) return lut_bin(th, lp);   // huge headings: integer path
    const float q = __builtin_floorf(u * inv_td);
    float b = __builtin_fmaf(-q, td_f, u);
    b = b < 0.0f ? b + td_f : b;
    b = b >= td_f ? b - td_f : b;
    return (int)b;
}

// fan query: ONE WAVE PER POSE, lane = beam within a 64-beam chunk.  The kernel is a pure
// stream (2 B/ray in, 4 B/ray out), so what matters is bytes in flight: all CH chunks of a pose
// (CH independent 2-byte loads per lane, ~2 KiB per wave) are issued before the first use, and
// the loop is software-pipelined across poses — the loads of pose n+1 are issued BEFORE the
// stores of pose n, because gfx950's vmcnt retires loads and stores in issue order and a load
// issued behind 17 stores would wait for their write acknowledgements.
// out[pose*num_rays + j] metres.
template <int CH>
__global__ __launch_bounds__(256) void lut_fan_kernel(MapParams m, FanParams f, LutParams lp,
                                                      const float *__restrict__ poses,
                                                      float *__restrict__ out)
{
    const float miss = f.max_range * m.res;
    const float td_f = (float)lp.theta_disc, inv_td = 1.0f / (float)lp.theta_disc;
    const float scale = lp.dequant;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    const int cpp = (f.num_rays + 63) >> 6;

    auto issue = [&](int pose, uint16_t (&q)[CH], bool &inb, int k_lo) {
        float gx, gy, thg;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                      poses[3 * (size_t)pose + 2], gx, gy, thg);
        inb = gx >= 0.0f && gx < m.fcols && gy >= 0.0f && gy < m.frows;
        const uint16_t *row = lp.lut + (inb ? ((size_t)(int)gy * m.cols + (int)gx) * lp.theta_disc : 0);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = ((k_lo + k) << 6) + lane;
            q[k] = 0;
            if (inb && j < f.num_rays && !(lp.debug & 1)) q[k] = row[lut_bin_fast(thg + fan_alpha(f, j), lp, td_f, inv_td)];
        }
    };
    auto retire = [&](int pose, const uint16_t (&q)[CH], bool inb, int k_lo) {
        float *dst = out + (size_t)pose * f.num_rays;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = ((k_lo + k) << 6) + lane;
            if (j < f.num_rays) {
                float r = inb ? (float)q[k] * scale * m.res : miss;
                if (f.noise_std > 0.0f)
                    r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (size_t)pose * f.num_rays + j);
                if (!(lp.debug & 2) || r < 0.0f) dst[j] = r;
            }
        }
    };

    // work items: (pose, group of CH chunks); rounds per pose = ceil(cpp / CH)
    const int rpp = (cpp + CH - 1) / CH;
    const long n_items = (long)f.n_poses * rpp;
    long it = wave;
    if (it >= n_items) return;
    uint16_t qa[CH], qb[CH];
    bool ia, ib;
    int pa = (int)(it / rpp), ka = (int)(it % rpp) * CH;
    issue(pa, qa, ia, ka);
    for (it += n_waves; it < n_items; it += n_waves) {
        const int pb = (int)(it / rpp), kb = (int)(it % rpp) * CH;
        issue(pb, qb, ib, kb);            // next item's loads first ...
        retire(pa, qa, ia, ka);           // ... then this item's stores
#pragma unroll
        for (int k = 0; k < CH; ++k) qa[k] = qb[k];
        ia = ib;
        pa = pb;
        ka = kb;
    }
    retire(pa, qa, ia, ka);
}

// The production fan query.  Measured on MI355X: with one 2-byte load per beam the table read
// ran at only ~2 TB/s even when the poses' rows fit the Infinity Cache — the limit is requests in
// flight, not bytes (a wave-load covered just 128 B).  So the wave fetches the pose's WHOLE theta
// row (theta_disc*2 B, e.g. 2884 B) with NL 16-byte-per-lane loads (1 KiB per wave-instruction),
// parks it in LDS, and the beams gather their bins from LDS.  Rows are read 1.33x wider than the
// fan needs (fov/2pi of the row), which costs less than narrow requests do.  Software-pipelined:
// the next pose's row is in flight while the current one is gathered and stored.
template <int NL, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(88)))     // (+ VCC etc. <= 96: the eighth wave per SIMD)
void lut_fan_lds_kernel(MapParams m, FanParams f, LutParams lp,
                                                          const float *__restrict__ poses,
                                                          float *__restrict__ out)
{
    extern __shared__ uint32_t lds_rows[];                   // per wave: NL*256 dwords
    uint32_t *my = lds_rows + (threadIdx.x >> 6) * (NL * 256);
    const uint16_t *my16 = reinterpret_cast<const uint16_t *>(my);
    const float miss = f.max_range * m.res;
    const float td_f = (float)lp.theta_disc, inv_td = 1.0f / (float)lp.theta_disc;
    const float scale = lp.dequant;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    const int D = lp.theta_disc >> 1;                        // dwords per row (theta_disc even)

    uint4 regs[NL];
    // fast: the fan's bins are ONE ascending circular run shorter than a row (fov >= 0, span < theta_disc,
    // |bin index before the wrap| < 2^23): beam j's bin is then (u_j - ubase) with at most one wrap, ubase =
    // u_0 - bin_0 a multiple of theta_disc — the same integer as the statement's ((int)u % td + td) % td
    // (every float involved is an exactly represented integer), for 8 instead of ~20 instructions per beam
    auto issue = [&](int pose, float &thg, bool &inb, bool &fast, float &ubase) {
        float gx, gy;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                      poses[3 * (size_t)pose + 2], gx, gy, thg);
        inb = gx >= 0.0f && gx < m.fcols && gy >= 0.0f && gy < m.frows;
        const uint32_t *row = reinterpret_cast<const uint32_t *>(
            lp.lut + (inb ? ((size_t)(int)gy * m.cols + (int)gx) * lp.theta_disc : 0));
        // only the bins the fan can touch: beam angles grow with j, so the bins are the circular run
        // from the first beam's bin over `span` bins (fov 4.71 at theta_disc 1442: 1081 of 1442 —
        // a quarter of the row's bytes stay in HBM)
        const float u0 = __builtin_rintf((thg + fan_alpha(f, 0)) * lp.bins_per_rad);
        const float u1 = __builtin_rintf((thg + fan_alpha(f, f.num_rays - 1)) * lp.bins_per_rad);
        const float spanf = u1 - u0;
        const bool all = !(spanf >= 0.0f && spanf < td_f - 8.0f) || !(__builtin_fabsf(u0) < 8388608.0f);
        const int span = all ? 0 : (int)spanf;
        const int b0 = all ? 0 : lut_bin_fast(thg + fan_alpha(f, 0), lp, td_f, inv_td);
        fast = f.inc >= 0.0f && spanf >= 0.0f && spanf < td_f && __builtin_fabsf(u0) < 4194304.0f &&
               __builtin_fabsf(u1) < 4194304.0f;
        ubase = u0 - (float)(all ? lut_bin_fast(thg + fan_alpha(f, 0), lp, td_f, inv_td) : b0);
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int idx = (n * 64 + lane) * 4;               // dword index; bins 2*idx .. 2*idx+7
            regs[n] = make_uint4(0, 0, 0, 0);
            int d = 2 * idx - b0;                              // chunk start relative to the first bin
            d = d < 0 ? d + lp.theta_disc : d;
            const bool need = all || d <= span || d >= lp.theta_disc - 7;
            if (inb && idx < D && need) {
                // NON-TEMPORAL: a row is read once per pose out of a table (11.5 GB at cfg3) that no cache holds —
                // loads that do not allocate in the L2 / Infinity Cache: the lone launch 0.108 -> 0.089 ms,
                // 677 -> 840 Grays/s (profiles/r04/lut_nt_ab.txt; lut_debug bit 4 = plain loads, the A/B partner)
                if (!(lp.debug & 16)) {
                    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(row + idx));
                    regs[n] = make_uint4(v.x, v.y, v.z, v.w);
                } else
                    regs[n] = *reinterpret_cast<const uint4 *>(row + idx);
            }
        }
    };

    int pose = wave;
    if (pose >= f.n_poses) return;
    float thg, thg_n = 0.0f, ubase, ubase_n = 0.0f;
    bool inb, inb_n = false, fast, fast_n = false;
    issue(pose, thg, inb, fast, ubase);
    const uint32_t td_u = (uint32_t)lp.theta_disc;
    for (;;) {
#pragma unroll
        for (int n = 0; n < NL; ++n) *reinterpret_cast<uint4 *>(my + (n * 64 + lane) * 4) = regs[n];
        const int next = pose + n_waves;
        if (next < f.n_poses) issue(next, thg_n, inb_n, fast_n, ubase_n);      // in flight during the gather
        float *dst = out + (size_t)pose * f.num_rays;
        if (fast && inb && !(f.noise_std > 0.0f) && !(lp.debug & 2)) {       // wave-uniform
            // (groups of four chunks: the whole fan unrolled at once keeps 17 bins + 17 ranges live and costs
            //  the kernel three of its eight waves per SIMD)
#pragma unroll 1
            for (int k0 = 0; k0 < CH; k0 += 4) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int j = ((k0 + kk) << 6) + lane;
                    if (k0 + kk < CH && j < f.num_rays) {
                        const float u = __builtin_rintf((thg + fan_alpha(f, j)) * lp.bins_per_rad);
                        const uint32_t b = (uint32_t)(int)(u - ubase);
                        const uint32_t bw = min(b, b - td_u);                   // one wrap at most
                        const float r = (float)my16[bw] * scale * m.res;
                        if (lp.debug & 8) __builtin_nontemporal_store(r, dst + j); else dst[j] = r;
                    }
                }
            }
        } else {
            // the general statement (poses outside the map, noise, fov < 0, fans as long as a row, headings
            // beyond 2^22 bins): rare — kept rolled so that it does not set the kernel's register count
#pragma unroll 1
            for (int k = 0; k < CH; ++k) {
                const int j = (k << 6) + lane;
                if (j < f.num_rays) {
                    float r = miss;
                    if (inb) r = (float)my16[lut_bin_fast(thg + fan_alpha(f, j), lp, td_f, inv_td)] * scale * m.res;
                    if (f.noise_std > 0.0f)
                        r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (size_t)pose * f.num_rays + j);
                    if (!(lp.debug & 2) || r < 0.0f) dst[j] = r;
                }
            }
        }
        if (next >= f.n_poses) break;
        pose = next;
        thg = thg_n;
        inb = inb_n;
        fast = fast_n;
        ubase = ubase_n;
    }
}

__global__ __launch_bounds__(256) void lut_rays_kernel(MapParams m, FanParams f, LutParams lp,
                                                       const float *__restrict__ ins, long n,
                                                       float *__restrict__ out)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gx, gy, thg;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], gx, gy, thg);
        float r = f.max_range * m.res;
        if (gx >= 0.0f && gx < m.fcols && gy >= 0.0f && gy < m.frows)
            r = (float)lp.lut[((size_t)(int)gy * m.cols + (int)gx) * lp.theta_disc + lut_bin(thg, lp)] *
                lp.dequant * m.res;
        if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
        out[i] = r;
    }
}

}  // namespace scan
