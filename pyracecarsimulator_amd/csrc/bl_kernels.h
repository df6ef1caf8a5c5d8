// bl_kernels.h — walks on the bit-packed occupancy: K2 / K2b BresenhamsLine and occ_fan_lds,
// the north_star kernel shape (unit steps on an LDS window).  Part of scan_kernels.h.
#pragma once
#include "scan_device.h"
#include "scan_params.h"
#include "rm_kernels.h"

// ==============================================================================
// K2: BresenhamsLine on an LDS-resident occupancy tile.
// One workgroup per pose.  The (2R+1)^2 window of the BIT-PACKED occupancy around the
// pose (R = max_range + 3; 607 rows x 21 words = 51 KB for 300 px) is staged into LDS
// with coalesced row loads — that is all the map traffic of the pose: 47 B per ray —
// together with the per-beam (cos, sin) fan.  Each lane then walks one beam cell by
// cell entirely in LDS; a wave leaves the walk as soon as all of its lanes have hit
// or run out (EXEC-mask early termination).  Bit-exact to the CPU statement.
// ==============================================================================
namespace scan {

struct BlParams {
    int R;            // window radius in cells
    int ww;           // window row stride in 32-bit words (odd)
    int use_lds;      // 0: window too large for LDS -> read the global bit map directly
};

template <bool AUX>
__global__ __launch_bounds__(256) void bl_fan_kernel(MapParams m, FanParams f, BlParams bp,
                                                     const float *__restrict__ poses,
                                                     float *__restrict__ out,
                                                     int32_t *__restrict__ hits,
                                                     uint16_t *__restrict__ steps)
{
    extern __shared__ uint32_t lds_u[];
    float2 *fan_cs = reinterpret_cast<float2 *>(lds_u);                  // num_rays float2
    uint32_t *win = lds_u + 2 * (size_t)f.num_rays;                      // (2R+1) * ww words
    for (int j = threadIdx.x; j < f.num_rays; j += blockDim.x) {
        float s, c;
        det_sincosf(fan_alpha(f, j), s, c);
        fan_cs[j] = make_float2(c, s);
    }
    const int WH = 2 * bp.R + 1;
    const float miss = f.max_range;

    for (int pose = blockIdx.x; pose < f.n_poses; pose += gridDim.x) {
        float gx, gy, thg, st, ct;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                      poses[3 * (size_t)pose + 2], gx, gy, thg);
        det_sincosf(thg, st, ct);
        // poses that cannot index the grid (non-finite / absurdly far) miss without walking
        const bool sane = fabsf(gx) < 1e9f && fabsf(gy) < 1e9f && (ct - ct) + (st - st) == 0.0f;
        // window origin: word-aligned column, row; may lie outside the map (zeros there)
        const int cx = sane ? (int)gx : 0, cy = sane ? (int)gy : 0;
        const int wx0 = ((cx - bp.R) >> 5) << 5;        // arithmetic shift: floor to a word
        const int wy0 = cy - bp.R;
        __syncthreads();                                // previous pose's walkers are done
        if (bp.use_lds) {
            for (int i = threadIdx.x; i < WH * bp.ww; i += blockDim.x) {
                const int wr = i / bp.ww, wc = i - wr * bp.ww;
                const int r = wy0 + wr, w = (wx0 >> 5) + wc;
                uint32_t v = 0;
                if (r >= 0 && r < m.rows && w >= 0 && w < m.bits_stride)
                    v = m.bits[(size_t)r * m.bits_stride + w];
                win[i] = v;
            }
        }
        __syncthreads();
        auto occupied = [&](int col, int row) -> bool {
            if (bp.use_lds) {
                const int x = col - wx0, y = row - wy0;
                return (win[y * bp.ww + (x >> 5)] >> (x & 31)) & 1u;
            }
            return (m.bits[(size_t)row * m.bits_stride + (col >> 5)] >> (col & 31)) & 1u;
        };
        for (int j = threadIdx.x; j < f.num_rays; j += blockDim.x) {
            const float2 cs = fan_cs[j];
            const float dx = __builtin_fmaf(ct, cs.x, -(st * cs.y));
            const float dy = __builtin_fmaf(st, cs.x, ct * cs.y);
            float range = miss;
            int hc = -1, hr = -1;
            unsigned n = 0;
            if (sane) {
                if (gx > -1.0f && gx < m.fcols && gy > -1.0f && gy < m.frows &&
                    occupied((int)gx, (int)gy)) {
                    range = 0.0f;                       // start cell occupied
                    hc = (int)gx;
                    hr = (int)gy;
                } else {
                    float x0 = gx, y0 = gy;
                    float x1 = __builtin_fmaf(f.max_range, dx, gx);
                    float y1 = __builtin_fmaf(f.max_range, dy, gy);
                    const bool steep = fabsf(y1 - y0) > fabsf(x1 - x0);
                    if (steep) {
                        float tmp = x0; x0 = y0; y0 = tmp;
                        tmp = x1; x1 = y1; y1 = tmp;
                    }
                    const float lim_major = steep ? m.frows : m.fcols;
                    const float lim_minor = steep ? m.fcols : m.frows;
                    const float deltax = fabsf(x1 - x0), deltay = fabsf(y1 - y0);
                    float error = 0.0f, _x = x0, _y = y0;
                    const float xstep = x0 < x1 ? 1.0f : -1.0f;
                    const float ystep = y0 < y1 ? 1.0f : -1.0f;
                    const int end = (int)(x1 + xstep);
                    int cap = (int)f.max_range + 3;
                    while ((int)_x != end && cap-- > 0) {
                        _x += xstep;
                        error += deltay;
                        if (error * 2.0f >= deltax) {
                            _y += ystep;
                            error -= deltax;
                        }
                        ++n;
                        if (_x >= 0.0f && _x < lim_major && _y >= 0.0f && _y < lim_minor) {
                            const int col = steep ? (int)_y : (int)_x;
                            const int row = steep ? (int)_x : (int)_y;
                            if (occupied(col, row)) {
                                const float xd = _x - x0, yd = _y - y0;
                                range = __builtin_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                                hc = col;
                                hr = row;
                                break;
                            }
                        }
                    }
                }
            }
            const size_t i = (size_t)pose * f.num_rays + j;
            float r = range * m.res;
            if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
            out[i] = r;
            if (AUX) {
                if (hits) { hits[2 * i] = hc; hits[2 * i + 1] = hr; }
                if (steps) steps[i] = (uint16_t)n;
            }
        }
    }
}

// ------------------------------------------------------------------------------
// occ_fan_lds (variant 2 of the ray-marching methods): the kernel shape
// BASELINE.json's north_star spells out — bit-packed occupancy window of the pose and the angle fan
// staged in LDS, wave ballot for early-hit termination — as an A/B partner of K1b.  There is no
// distance field here, so the march takes UNIT steps: the 64 lanes of a wave test 64 consecutive
// samples t = t0 .. t0+63 of ONE ray against the LDS window, and ballot + ffs picks the first event
// (occupied cell -> hit at that cell, sample outside the map -> miss).  A ray costs one wave pass per
// 64 cells of range.  Samples are denser than sphere tracing's, so results are NOT bit-identical to
// RayMarching: ranges agree within one cell on all but corner-grazing rays (acceptance of step 5).
// Measured against K1b in profiles/r02/ab_occ_lds.txt (4096 poses x 1081 beams): 1380 us against 58 us
// on the 2049^2 maze, 1317 us against 35 us on colombia — a wave pass (~40 instructions) per ray and
// per 64 cells of range here, against ~5 wave instructions per ray for 64 rays sphere-tracing side by
// side on the cache-resident step map, plus 51 KB of window staging per pose — which is why the
// product's default stays K1b.
// ------------------------------------------------------------------------------
template <bool AUX>
__global__ __launch_bounds__(256) void occ_fan_lds_kernel(MapParams m, FanParams f, BlParams bp,
                                                          const float *__restrict__ poses,
                                                          float *__restrict__ out,
                                                          int32_t *__restrict__ hits,
                                                          uint16_t *__restrict__ steps)
{
    extern __shared__ uint32_t lds_u[];
    float2 *fan_cs = reinterpret_cast<float2 *>(lds_u);                  // num_rays float2
    uint32_t *win = lds_u + 2 * (size_t)f.num_rays;                      // (2R+1) * ww words
    for (int j = threadIdx.x; j < f.num_rays; j += blockDim.x) {
        float s, c;
        det_sincosf(fan_alpha(f, j), s, c);
        fan_cs[j] = make_float2(c, s);
    }
    const int WH = 2 * bp.R + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float flane = (float)lane;
    for (int pose = blockIdx.x; pose < f.n_poses; pose += gridDim.x) {
        float gx, gy, thg, st, ct;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                      poses[3 * (size_t)pose + 2], gx, gy, thg);
        det_sincosf(thg, st, ct);
        const bool inb = gx > -1.0f && gx < m.fcols && gy > -1.0f && gy < m.frows && (ct - ct) + (st - st) == 0.0f;
        const int cx = inb ? (int)gx : 0, cy = inb ? (int)gy : 0;
        const int wx0 = ((cx - bp.R) >> 5) << 5;        // word-aligned window origin (may lie outside: zeros)
        const int wy0 = cy - bp.R;
        __syncthreads();                                // the previous pose's rays are done
        for (int i = threadIdx.x; i < WH * bp.ww; i += blockDim.x) {
            const int wr = i / bp.ww, wc = i - wr * bp.ww;
            const int r = wy0 + wr, w = (wx0 >> 5) + wc;
            uint32_t v = 0;
            if (r >= 0 && r < m.rows && w >= 0 && w < m.bits_stride) v = m.bits[(size_t)r * m.bits_stride + w];
            win[i] = v;
        }
        __syncthreads();
        // a wave takes blocks of 64 consecutive beams; lane k keeps beam k's result for one coalesced store
        for (int j0 = wave * 64; j0 < f.num_rays; j0 += (int)(blockDim.x >> 6) * 64) {
            float my_r = f.max_range;
            int my_c = -1, my_rw = -1;
            unsigned my_n = 0;
            const int jn = min(64, f.num_rays - j0);
            for (int k = 0; k < jn; ++k) {
                const float2 cs = fan_cs[j0 + k];                          // (broadcast read)
                const float dx = __builtin_fmaf(ct, cs.x, -(st * cs.y));
                const float dy = __builtin_fmaf(st, cs.x, ct * cs.y);
                float range = f.max_range;
                int hc = -1, hr = -1;
                unsigned n = 0;
                if (inb) {
                    for (float t0 = 0.0f; t0 < f.max_range; t0 += 64.0f) {
                        const float t = t0 + flane;
                        const float fx = __builtin_fmaf(dx, t, gx), fy = __builtin_fmaf(dy, t, gy);
                        const bool live = t < f.max_range;
                        const bool inside = fx > -1.0f && fx < m.fcols && fy > -1.0f && fy < m.frows;
                        const int pc = (int)fx, pr = (int)fy;
                        bool occ = false;
                        if (live && inside) {
                            const int x = pc - wx0, y = pr - wy0;
                            occ = (win[y * bp.ww + (x >> 5)] >> (x & 31)) & 1u;
                        }
                        const unsigned long long ev = __ballot(live && (occ || !inside));
                        if (ev) {
                            const int first = __ffsll((long long)ev) - 1;
                            const int f_occ = __shfl((int)occ, first);
                            n += (unsigned)first + (f_occ ? 1u : 0u);
                            if (f_occ) {
                                hc = __shfl(pc, first);
                                hr = __shfl(pr, first);
                                const float xd = (float)hc - gx, yd = (float)hr - gy;
                                range = hit_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                            }
                            break;
                        }
                        n += 64u;
                    }
                }
                if (lane == k) { my_r = range; my_c = hc; my_rw = hr; my_n = n; }
            }
            if (lane < jn) {
                const size_t i = (size_t)pose * f.num_rays + j0 + lane;
                float r = my_r * m.res;
                if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
                out[i] = r;
                if (AUX) {
                    if (hits) { hits[2 * i] = my_c; hits[2 * i + 1] = my_rw; }
                    if (steps) steps[i] = (uint16_t)(my_n > 65535u ? 65535u : my_n);
                }
            }
        }
    }
}

// K2b: the same walk on the K1b schedule (tile-ordered poses, XCD bands, a workgroup's waves
// sharing one ray stream with lane refill), reading a bit-packed map straight through L1/L2
// (2049^2 cells = 0.5 MB: the whole map is cache resident).  Staging a per-pose LDS window
// (bl_fan_kernel above) ties 1081 rays to one workgroup and makes every pose end with its
// slowest ray (up to 303 steps against a mean of 46); the stream form has no such join.
//
// Round 2: the walk itself is a hand-scheduled loop (bl_march_loop), made possible by two padded
// copies of the bit map (bl_pad_bits_kernel):
//  * a border of free cells as wide as a walk can get away from the map, so the four in-bounds
//    tests of every step disappear (cells out there ARE free, and floor-conversion of a negative
//    coordinate lands in the border exactly where the statement's `_x >= 0` test says "outside");
//  * a TRANSPOSED copy for steep rays (major axis = rows): the walk's (major, minor) pair addresses
//    either copy with the same formula — bit `major & 31` of word `minor * stride + (major >> 5)` —
//    so the per-step "steep ? .. : .." selects disappear; base offset and stride are per-lane values.
// 19 VALU + 1 load per step (EXEC = lanes still walking), against ~40 compiler-scheduled before.
// Origins so far outside that the padded copies do not cover their walk never reach the map: they
// run the stepping arithmetic without map reads when they are claimed (their step count is still
// the statement's).  Bit-identical to the CPU statement (ranges, hit cells, step counts).

// out[(pr) * stride + w]: 32 cells of the padded view; view(rr, cc) = occ[rr][cc] or, transposed, occ[cc][rr]
__global__ __launch_bounds__(256) void bl_pad_bits_kernel(const uint8_t *__restrict__ occ, int rows, int cols,
                                                          int transposed, int pad_minor, int pad_major32,
                                                          int stride, int prow_count, uint32_t *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x, pr = blockIdx.y;
    if (w >= stride || pr >= prow_count) return;
    const int vrows = transposed ? cols : rows, vcols = transposed ? rows : cols;
    const int rr = pr - pad_minor;
    uint32_t word = 0;
    if (rr >= 0 && rr < vrows) {
        const int c0 = (w - pad_major32) * 32;
#pragma unroll 4
        for (int k = 0; k < 32; ++k) {
            const int cc = c0 + k;
            if (cc >= 0 && cc < vcols) {
                const uint8_t v = transposed ? occ[(size_t)cc * cols + rr] : occ[(size_t)rr * cols + cc];
                if (v) word |= 1u << k;
            }
        }
    }
    out[(size_t)pr * stride + w] = word;
}

// The walk: x is the major coordinate (advances by xstep = +-1 every step), y the minor one.
//   top:    leave when (int)x == end or the step budget is used up        (the statement's while test)
//   step:   x += xstep; err += deltay; if (2 err >= deltax) { y += ystep; err -= deltax }
//           (the conditional pair as m = 0/1 and two fmas: y + m*ystep and err - m*deltax round once,
//            exactly like the add / subtract they stand for)
//   probe:  bit (floor x & 31) of word [floor y][floor x >> 5] of the lane's padded copy; a set bit ends the walk
__device__ __forceinline__ void bl_march_loop(float &x, float &y, float &err, uint32_t &n, int &ix, int &iy,
                                              uint32_t &bit, uint32_t &live, float xstep, float ystep,
                                              float deltax, float deltay, int end, int stride, uint32_t basek,
                                              const uint32_t *bits, uint32_t cap0, uint32_t low)
{
    unsigned long long save, tmp;
    uint32_t cnt;
    float e2, m;
    int it, a;
    uint32_t word;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_ne_u32_e32 0, %[live]\n"
        "L_blwalk_%=:\n\t"
        "v_cvt_i32_f32_e32 %[it], %[x]\n\t"
        "v_cmpx_ne_i32_e32 %[it], %[end]\n\t"
        "v_cmpx_gt_u32_e32 %[cap0], %[n]\n\t"
        "v_add_f32_e32 %[x], %[x], %[xstep]\n\t"
        "v_add_f32_e32 %[err], %[err], %[deltay]\n\t"
        "v_add_f32_e32 %[e2], %[err], %[err]\n\t"
        "v_cmp_ge_f32_e32 vcc, %[e2], %[deltax]\n\t"
        "v_cndmask_b32_e64 %[m], 0, 1.0, vcc\n\t"
        "v_fma_f32 %[y], %[m], %[ystep], %[y]\n\t"
        "v_fma_f32 %[err], %[m], -%[deltax], %[err]\n\t"
        "v_add_u32_e32 %[n], 1, %[n]\n\t"
        "v_cvt_flr_i32_f32_e32 %[ix], %[x]\n\t"
        "v_cvt_flr_i32_f32_e32 %[iy], %[y]\n\t"
        "v_ashrrev_i32_e32 %[a], 5, %[ix]\n\t"
        "v_mad_i32_i24 %[a], %[iy], %[stride], %[a]\n\t"
        "v_lshl_add_u32 %[a], %[a], 2, %[basek]\n\t"
        "global_load_dword %[word], %[a], %[bits]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_bfe_u32 %[bit], %[word], %[ix], 1\n\t"
        "v_cmpx_eq_u32_e32 0, %[bit]\n\t"
        "s_bcnt1_i32_b64 %[cnt], exec\n\t"
        "s_cmp_gt_u32 %[cnt], %[low]\n\t"
        "s_cbranch_scc1 L_blwalk_%=\n\t"
        "s_mov_b64 %[tmp], exec\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "v_mov_b32_e32 %[live], 0\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "v_mov_b32_e32 %[live], 1\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [x] "+v"(x), [y] "+v"(y), [err] "+v"(err), [n] "+v"(n), [ix] "+v"(ix), [iy] "+v"(iy), [bit] "+v"(bit),
          [live] "+v"(live), [e2] "=&v"(e2), [m] "=&v"(m), [it] "=&v"(it), [a] "=&v"(a), [word] "=&v"(word),
          [save] "=&s"(save), [tmp] "=&s"(tmp), [cnt] "=&s"(cnt)
        : [xstep] "v"(xstep), [ystep] "v"(ystep), [deltax] "v"(deltax), [deltay] "v"(deltay), [end] "v"(end),
          [stride] "v"(stride), [basek] "v"(basek), [bits] "s"(bits), [cap0] "s"(cap0), [low] "s"(low)
        : "vcc", "scc", "memory");
}

template <bool AUX, int NT>
__global__ __launch_bounds__(NT) void bl_fan_stream_kernel(MapParams m, FanParams f, StreamParams sp, BlPad bp,
                                                          float *__restrict__ out,
                                                          int32_t *__restrict__ hits,
                                                          uint16_t *__restrict__ steps)
{
    extern __shared__ float lds_f[];
    uint32_t *q_next = reinterpret_cast<uint32_t *>(lds_f);
    float2 *fan_cs = reinterpret_cast<float2 *>(lds_f + 2);
    if (threadIdx.x == 0) *q_next = 0;
    for (int j = threadIdx.x; j < f.num_rays; j += NT) {
        float s, c;
        det_sincosf(fan_alpha(f, j), s, c);
        fan_cs[j] = make_float2(c, s);
    }
    __syncthreads();
    const int nb = sp.n_bands;
    const int band = (int)(blockIdx.x % (unsigned)nb);
    const uint32_t g = blockIdx.x / (unsigned)nb;
    const uint32_t G = ((uint32_t)gridDim.x - (uint32_t)band + (uint32_t)nb - 1) / (uint32_t)nb;
    const uint32_t seg_lo = (uint32_t)(((long)f.n_poses * band) / nb);
    const uint32_t seg_hi = (uint32_t)(((long)f.n_poses * (band + 1)) / nb);
    const uint32_t seg_rays = (seg_hi - seg_lo) * (uint32_t)f.num_rays;
    const uint32_t seg_chunks = (seg_rays + 63u) >> 6;
    const uint32_t rl = (uint32_t)sp.run_log2, rmask = (1u << rl) - 1u;
    const uint32_t seg_runs = (seg_chunks + rmask) >> rl;
    const uint32_t K = (g < seg_runs ? (seg_runs - g + G - 1) / G : 0) << rl;
    const uint32_t total = K << 6;
    // i-th block of this workgroup's stream -> first ray of the block
    auto blk_of = [&](uint32_t i) { return (((g + (i >> rl) * G) << rl) + (i & rmask)) << 6; };
    const unsigned lane = threadIdx.x & 63;
    auto occupied = [&](int col, int row) -> bool {
        return (m.bits[(size_t)row * m.bits_stride + (col >> 5)] >> (col & 31)) & 1u;
    };
    const uint32_t cap0 = (uint32_t)((int)f.max_range + 3);

    bool exhausted = total == 0;
    bool has_ray = false, steep = false;
    uint32_t live = 0, bit = 0, nstep = 0, oidx = 0, basek = 0;
    float x0 = 0, y0 = 0, _x = 0, _y = 0, error = 0, deltax = 0, deltay = 0, xstep = 0, ystep = 0;
    float range0 = 0;                 // range of a ray that never walks (start cell occupied: 0)
    int end = 0, ix = -1, iy = -1, stride = 0;
    bool start_hit = false;

    for (;;) {
        const unsigned long long idle = __ballot(live == 0);
        if (idle) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (live == 0 && has_ray) {
                float range = f.max_range;
                int hc = -1, hr = -1;
                if (start_hit) {
                    range = range0;
                    hc = ix;
                    hr = iy;
                } else if (bit) {
                    const float xd = _x - x0, yd = _y - y0;
                    range = __builtin_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                    hc = steep ? iy : ix;
                    hr = steep ? ix : iy;
                }
                float r = range * m.res;
                if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + oidx);
                range_store(out, (uint32_t)oidx << 2, r, sp.plain_store);
                if (AUX) {
                    if (hits) { hits[2 * (size_t)oidx] = hc; hits[2 * (size_t)oidx + 1] = hr; }
                    if (steps) steps[oidx] = (uint16_t)nstep;
                }
                has_ray = false;
            }
            if (!exhausted) {
                const uint32_t cnt = (uint32_t)__popcll(idle);
                uint32_t qb = 0;
                if (lane == 0) qb = atomicAdd(q_next, cnt);
                qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
                exhausted = qb + cnt >= total;
                const uint32_t q = qb + rank;
                const uint32_t ray = blk_of(q >> 6) + (q & 63);
                if (live == 0 && q < total && ray < seg_rays) {
                    const uint32_t spose = fast_div(ray, sp.div_B);
                    const int j = (int)(ray - spose * (uint32_t)f.num_rays);
                    const uint32_t po = sp.order[seg_lo + spose];
                    const PoseRec pr_ = sp.rec[seg_lo + spose];
                    const float2 cs = fan_cs[j];
                    const float gx = pr_.gx, gy = pr_.gy;
                    const float dx = __builtin_fmaf(pr_.ct, cs.x, -(pr_.st * cs.y));
                    const float dy = __builtin_fmaf(pr_.st, cs.x, pr_.ct * cs.y);
                    oidx = (po & ~POSE_INVALID) * (uint32_t)f.num_rays + (uint32_t)j;
                    has_ray = true;
                    nstep = 0;
                    bit = 0;
                    start_hit = false;
                    if (!(po & POSE_INVALID)) {
                        if (gx > -1.0f && gx < m.fcols && gy > -1.0f && gy < m.frows &&
                            occupied((int)gx, (int)gy)) {
                            start_hit = true;                   // start cell occupied: range 0
                            range0 = 0.0f;
                            ix = (int)gx;
                            iy = (int)gy;
                        } else {
                            x0 = gx;
                            y0 = gy;
                            float x1 = __builtin_fmaf(f.max_range, dx, gx);
                            float y1 = __builtin_fmaf(f.max_range, dy, gy);
                            steep = fabsf(y1 - y0) > fabsf(x1 - x0);
                            if (steep) {
                                float tmp = x0; x0 = y0; y0 = tmp;
                                tmp = x1; x1 = y1; y1 = tmp;
                            }
                            deltax = fabsf(x1 - x0);
                            deltay = fabsf(y1 - y0);
                            error = 0.0f;
                            _x = x0;
                            _y = y0;
                            xstep = x0 < x1 ? 1.0f : -1.0f;
                            ystep = y0 < y1 ? 1.0f : -1.0f;
                            end = (int)(x1 + xstep);
                            stride = steep ? bp.stride_t : bp.stride_n;
                            basek = steep ? bp.k_t : bp.k_n;
                            const bool near = gx > -bp.near && gx < m.fcols + bp.near && gy > -bp.near &&
                                              gy < m.frows + bp.near;
                            if (near) {
                                live = 1;
                            } else {
                                // too far outside for the padded copies: this walk never meets the map; only
                                // its step count is left to find (same arithmetic, no map reads)
                                uint32_t cap = cap0;
                                while ((int)_x != end && cap-- > 0) {
                                    _x += xstep;
                                    ++nstep;
                                }
                            }
                        }
                    }
                }
            }
        }
        if (exhausted && !__ballot(live != 0) && !__ballot(has_ray)) break;
        bl_march_loop(_x, _y, error, nstep, ix, iy, bit, live, xstep, ystep, deltax, deltay, end, stride, basek,
                      bp.bits, cap0, exhausted ? 0u : (uint32_t)sp.low_water);
    }
}

// one world (x, y, theta) row per ray, straight from the global bit map
__global__ __launch_bounds__(256) void bl_rays_kernel(MapParams m, FanParams f,
                                                      const float *__restrict__ ins, long n_rays,
                                                      float *__restrict__ out)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += stride) {
        float gx, gy, thg, dx, dy;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], gx, gy, thg);
        det_sincosf(thg, dy, dx);
        auto occupied = [&](int col, int row) -> bool {
            return (m.bits[(size_t)row * m.bits_stride + (col >> 5)] >> (col & 31)) & 1u;
        };
        float range = f.max_range;
        const bool sane = fabsf(gx) < 1e9f && fabsf(gy) < 1e9f && (dx - dx) + (dy - dy) == 0.0f;
        if (sane) {
            if (gx > -1.0f && gx < m.fcols && gy > -1.0f && gy < m.frows && occupied((int)gx, (int)gy)) {
                range = 0.0f;
            } else {
                float x0 = gx, y0 = gy;
                float x1 = __builtin_fmaf(f.max_range, dx, gx);
                float y1 = __builtin_fmaf(f.max_range, dy, gy);
                const bool steep = fabsf(y1 - y0) > fabsf(x1 - x0);
                if (steep) {
                    float tmp = x0; x0 = y0; y0 = tmp;
                    tmp = x1; x1 = y1; y1 = tmp;
                }
                const float lim_major = steep ? m.frows : m.fcols;
                const float lim_minor = steep ? m.fcols : m.frows;
                const float deltax = fabsf(x1 - x0), deltay = fabsf(y1 - y0);
                float error = 0.0f, _x = x0, _y = y0;
                const float xstep = x0 < x1 ? 1.0f : -1.0f;
                const float ystep = y0 < y1 ? 1.0f : -1.0f;
                const int end = (int)(x1 + xstep);
                int cap = (int)f.max_range + 3;
                while ((int)_x != end && cap-- > 0) {
                    _x += xstep;
                    error += deltay;
                    if (error * 2.0f >= deltax) {
                        _y += ystep;
                        error -= deltax;
                    }
                    if (_x >= 0.0f && _x < lim_major && _y >= 0.0f && _y < lim_minor) {
                        const int col = steep ? (int)_y : (int)_x;
                        const int row = steep ? (int)_x : (int)_y;
                        if (occupied(col, row)) {
                            const float xd = _x - x0, yd = _y - y0;
                            range = __builtin_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                            break;
                        }
                    }
                }
            }
        }
        float r = range * m.res;
        if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
        out[i] = r;
    }
}

}  // namespace scan
