// rm_kernels.h — K1 / K1b: ray marching on the float32 EDT (RayMarching / RayMarchingGPU):
// the chunk-per-wave kernels, the pose binning passes, the step map, the hand-scheduled march / drain loops and
// the persistent stream kernel rm_fan_stream_kernel.  Part of scan_kernels.h.
#pragma once
#include "scan_device.h"
#include "scan_params.h"
#include "literal_math.h"

namespace scan {

constexpr int WG = 256;                 // 4 waves
constexpr int WAVES_PER_WG = WG / 64;

// ------------------------------------------------------------------------------
// K1: fan-expanding ray marching.  out[pose*num_rays + j] in metres.
// ------------------------------------------------------------------------------
struct CrashParams {
    const double *edge;      // num_rays doubles (Car::setCarEdgeDistances) or nullptr
    double thresh;
    int *first_crashed;      // group > 0: atomicMin targets, one per group, initialised to INT_MAX
                             // group == 0: one word per POSE, a crashed pose gets `mark` stored
    int group;               // poses per group (roll-out), or 0 = per-pose marks
    int mark;                // group == 0: this launch's mark (the caller's epoch: no clearing pass)
};

// A crashed pose is recorded.  Per-pose marks (group == 0) are plain idempotent stores — what the
// batched paths use, followed by crash_reduce_kernel.  The single-word form (small single roll-outs)
// only sends its atomic when it can still lower the value: a pose scraping a wall crashes on
// hundreds of beams and same-word atomics retire ~10 per us.
__device__ __forceinline__ void crash_note(const CrashParams &cp, uint32_t pose)
{
    if (cp.group == 0) {
        cp.first_crashed[pose] = cp.mark;
        return;
    }
    int *slot = &cp.first_crashed[pose / (uint32_t)cp.group];
    const int idx = (int)(pose % (uint32_t)cp.group);
    if (idx < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, idx);
}

template <bool AUX, bool CRASH>
__global__ __launch_bounds__(WG) void rm_fan_kernel(MapParams m, FanParams f,
                                                    const float *__restrict__ poses,
                                                    float *__restrict__ out,
                                                    int32_t *__restrict__ hits,
                                                    uint16_t *__restrict__ steps, CrashParams cp)
{
    extern __shared__ float2 fan_cs[];   // per-beam (cos a_j, sin a_j), staged once per WG
    for (int j = threadIdx.x; j < f.num_rays; j += WG) {
        float s, c;
        det_sincosf(fan_alpha(f, j), s, c);
        fan_cs[j] = make_float2(c, s);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * WAVES_PER_WG;
    const int cpp = (f.num_rays + 63) >> 6;                 // chunks per pose
    const long n_chunks = (long)f.n_poses * cpp;

    for (long ch = (long)blockIdx.x * WAVES_PER_WG + wave_in_wg; ch < n_chunks; ch += n_waves) {
        const int pose = (int)(ch / cpp);
        const int j = ((int)(ch - (long)pose * cpp) << 6) + lane;
        // wave-uniform pose constants
        float gx, gy, thg, st, ct;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                      poses[3 * (size_t)pose + 2], gx, gy, thg);
        det_sincosf(thg, st, ct);
        if (j < f.num_rays) {
            const float2 cs = fan_cs[j];
            const float dx = __builtin_fmaf(ct, cs.x, -(st * cs.y));
            const float dy = __builtin_fmaf(st, cs.x, ct * cs.y);
            RayResult rr = rm_march(m, f.max_range, f.step_coeff, gx, gy, dx, dy);
            const size_t i = (size_t)pose * f.num_rays + j;
            float r = rr.range_px * m.res;
            if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
            if (out) out[i] = r;
            if (AUX) {
                if (hits) { hits[2 * i] = rr.hit_c; hits[2 * i + 1] = rr.hit_r; }
                if (steps) steps[i] = (uint16_t)(rr.steps > 65535u ? 65535u : rr.steps);
            }
            if (CRASH) {
                // Car::isCrashed racecar/src/racecar.cpp:320: (rays - edge[j]) < CRASH_THRESH
                const bool crashed = ((double)r - cp.edge[j]) < cp.thresh;
                if (__ballot(crashed)) {
                    if (lane == __ffsll((long long)__ballot(crashed)) - 1)
                        crash_note(cp, (uint32_t)pose);
                }
            }
        }
    }
}

// one world (x, y, theta) row per ray: upstream calc_range_many(ins, outs)
__global__ __launch_bounds__(WG) void rm_rays_kernel(MapParams m, FanParams f,
                                                     const float *__restrict__ ins, long n,
                                                     float *__restrict__ out,
                                                     int32_t *__restrict__ hits,
                                                     uint16_t *__restrict__ steps)
{
    const long stride = (long)gridDim.x * WG;
    for (long i = (long)blockIdx.x * WG + threadIdx.x; i < n; i += stride) {
        float gx, gy, thg, dx, dy;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], gx, gy, thg);
        det_sincosf(thg, dy, dx);
        RayResult rr = rm_march(m, f.max_range, f.step_coeff, gx, gy, dx, dy);
        float r = rr.range_px * m.res;
        if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
        out[i] = r;
        if (hits) { hits[2 * i] = rr.hit_c; hits[2 * i + 1] = rr.hit_r; }
        if (steps) steps[i] = (uint16_t)(rr.steps > 65535u ? 65535u : rr.steps);
    }
}


// ==============================================================================
// K1b: the MI355X-shaped ray-marching path (variant 1, default).
//
//  (0) pad_dt_tiled_kernel — per method: the STEP MAP.  The float32 EDT with a border of
//      ceil(max_range)+2 cells, holding what the march adds to t at that cell: free cells
//      max(d*coeff, 1), occupied cells +inf, border 3e38.  A ray whose origin is inside the
//      map stays within max_range of it while it is live, so the march loop needs no bounds
//      test, no address clamp, no hit test and no per-sample max: leaving the map or hitting
//      adds a huge step and t leaves the [0, max_range) window.  (Origins outside the map are
//      misses before the first sample — decided once per pose.)  Rows are interleaved in
//      groups of 4 so that a 128-B line is a 4x8 block of cells (see pdt_tiled_index).
//  (1) pose binning — pose_bin_small_kernel (one 1024-lane workgroup, < 8192 poses) or
//      pose_prep/tile_scan_a,b/pose_scatter (grid-wide): per-pose records (gx, gy, cos th, sin th)
//      ordered by the map tile the pose stands in (LDS histogram -> scan -> scatter).  Small
//      batches and maps that fit every XCD's L2 skip it: the march kernel derives the records
//      of its own ray blocks into LDS (INLINE).
//  (2) rm_fan_stream_kernel.  The tile-ordered pose list is cut into 8 contiguous BANDS,
//      band x marched only by workgroups with blockIdx % 8 == x — one XCD under
//      round-robin dispatch (speed only, never correctness) — so each XCD's 4 MiB L2
//      holds one band of the map.  Inside a band, workgroup g owns runs of 2^k consecutive
//      64-ray blocks, interleaved with the band's other workgroups (every workgroup sees the
//      band's average cost; a run keeps it on one pose's fan for a while).
//      A workgroup's 16 waves share ONE stream of ray slots through an LDS counter: a
//      wave marches while more than `low_water` of its lanes are live, then every
//      finished lane stores its range and claims the next slot (ballot + mbcnt ranks,
//      one LDS atomic per wave).  Lanes stay busy although samples-per-ray is ~7 on
//      average and ~25 at the wave maximum.  (A global work counter per band was tried
//      first and rejected: returning atomics on one contended word retire at ~10/us on
//      MI355X, which made the launch atomic-bound.)
//      The first version was instruction-issue bound (~60 VALU+SALU per sample); the march
//      loop is now hand-scheduled assembly with EXEC as the live mask:
//      9 VALU + 1 load + 4 SALU per sample (march_loop below), two or three rays per lane
//      (march_loop2/3).  What bounds it today — VALU issue and the CU's gather rate, both at
//      ~65 % (cfg2) to ~85 % (32 k poses) — is in DESIGN.md section 4.
// Results are bit-identical to K1 (same arithmetic; only the schedule differs).
// ==============================================================================
struct PoseRec {
    float gx, gy, ct, st;
};

constexpr uint32_t POSE_INVALID = 0x80000000u;   // order[] flag: origin outside the map / non-finite

// Stop codes stored in the step map instead of 0 / "outside": adding them to t ends the march
// through the ordinary `t < max_range` test, so the loop needs no separate hit test.
#define PDT_HIT __builtin_inff()        /* occupied cell (EDT 0)          */
#define PDT_OUTSIDE 3.0e38f             /* border: the ray left the map   */

// tiled layout (TILED march): groups of 4 rows interleaved element-wise, so that one 128-B line holds a
// 4-row x 8-column block of cells, with a POWER-OF-TWO group pitch: in bytes, with r' = r + pad + 4 >= 0 and
// c' = c + pad >= 0 (pad: border width, a multiple of 8; 4 more rows of slack in front),
//   byte(r, c) = ((r' >> 2) << K) | (c' << 4) | ((r' & 3) << 2),     2^K = 16 * pcol2 >= 16 * max(padded cols, rows)
// which the march computes in THREE instructions (round 2's pitch of 4*pcol bytes took four):
//   a = r * M + padM          M = 4 + 2^(K-2): both copies of r' the address needs, (r'<<2) and (r'<<(K-2)),
//                             from one 24-bit multiply-add (padM = (pad+4) * M sits in a VGPR: one SGPR
//                             operand per VALU instruction on gfx9)
//   a = a & MASK              MASK = 0xC | (~0 << K): keeps (r'&3)<<2 and (r'>>2)<<K — the copies do not overlap
//                             because 2^(K-4) >= padded rows
//   a = (c << 4) + a          the column bias pad<<4 is folded into the SGPR base; a >= 0 because the slack
//                             group makes (r'>>2) >= 1 and 16*pad < 2^K
// Columns [cols + 2*pad, pcol2) of a group are never written or read: the table is larger (2049^2: 44 MB
// instead of 28 MB), the touched lines are the same.
struct TiledGeom {
    int K;                    // log2 of the group pitch in bytes
    int pad, padr;            // column bias, row bias (pad + 4)
    int pcols, prows;         // padded extent that holds data: cols + 2*pad, rows + 2*pad + 4 (multiple of 4)
};

__device__ __host__ __forceinline__ size_t pdt_tiled_byte(int rp, int cp, int K)
{
    return ((size_t)(rp >> 2) << K) | ((size_t)cp << 4) | ((size_t)(rp & 3) << 2);
}

// Both padded copies hold the march's STEP, not the distance: free cells max(d*coeff, 1) (the
// two roundings of rm_march, done once per map instead of once per sample), occupied cells +inf,
// border 3e38 — the stop codes survive because t + code >= max_range either way.
__global__ __launch_bounds__(256) void pad_dt_tiled_kernel(const float *__restrict__ dt, int rows, int cols,
                                                           float *__restrict__ pdt, TiledGeom tg, float coeff)
{
    const int pr = blockIdx.y;                      // r' (biased row)
    const int r = pr - tg.padr;
    for (int pc = blockIdx.x * blockDim.x + threadIdx.x; pc < tg.pcols; pc += gridDim.x * blockDim.x) {
        const int c = pc - tg.pad;
        float v = PDT_OUTSIDE;
        if (r >= 0 && r < rows && c >= 0 && c < cols) {
            v = dt[(size_t)r * cols + c];
            v = v <= 0.0f ? PDT_HIT : __builtin_fmaxf(v * coeff, 1.0f);
        }
        *reinterpret_cast<float *>(reinterpret_cast<char *>(pdt) + pdt_tiled_byte(pr, pc, tg.K)) = v;
    }
}

__global__ __launch_bounds__(256) void pad_dt_kernel(const float *__restrict__ dt, int rows, int cols,
                                                     float *__restrict__ pdt, int pad, int stride,
                                                     float coeff)
{
    const int pr = blockIdx.y;                      // padded row
    const int r = pr - pad;
    for (int pc = blockIdx.x * blockDim.x + threadIdx.x; pc < stride; pc += gridDim.x * blockDim.x) {
        const int c = pc - pad;
        float v = PDT_OUTSIDE;
        if (r >= 0 && r < rows && c >= 0 && c < cols) {
            v = dt[(size_t)r * cols + c];
            v = v <= 0.0f ? PDT_HIT : __builtin_fmaxf(v * coeff, 1.0f);
        }
        pdt[(size_t)pr * stride + pc] = v;
    }
}

// ---- the CODE map: palette of the map's distinct steps + the tiled map of their indices (see RM_LOAD / RM_DECODE) ----
// Layout: groups of 8 rows interleaved element-wise, power-of-two group pitch, like the float32 map with
//   byte(r, c) = ((r' >> 3) << K) | (c' << cs) | ((r' & 7) << es)        es = 1 (u16), 0 (u8);  cs = 3 + es
// so a 128-B line is 8 rows x 8 (u16) / 16 (u8) columns, and the march's three address instructions stay:
//   a = r * M + padM,  M = 2^es + 2^(K-3);   a &= MASK,  MASK = (7 << es) | (~0 << K);   a = (c << cs) + a.
__device__ __host__ __forceinline__ size_t code_tiled_byte(int rp, int cp, int K, int es)
{
    return ((size_t)(rp >> 3) << K) | ((size_t)cp << (3 + es)) | ((size_t)(rp & 7) << es);
}

// (1) every free cell whose step stays below max_range leaves its EDT value in slot d^2 of `val` (the EDT is
//     sqrtf of an integer: cells of one slot hold the same float, the store is idempotent; d * d rounds to d^2
//     exactly far beyond the 2^21 slots a palette may have)
__global__ __launch_bounds__(256) void code_mark_kernel(const float *__restrict__ dt, size_t n_cells,
                                                        float *__restrict__ val, uint32_t nb, float coeff,
                                                        float max_range)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
        const float d = dt[i];
        if (d > 0.0f && __builtin_fmaxf(d * coeff, 1.0f) < max_range) {
            const uint32_t d2 = (uint32_t)__builtin_fmaf(d, d, 0.5f);
            if (d2 < nb) val[d2] = d;
        }
    }
}

// (2) one workgroup: exclusive scan over the occupied slots -> idx[d^2] = palette index, tab[index] = the march's
//     step of that EDT value (the two roundings of pad_dt_tiled_kernel), then the two stop codes; n_out[0] = palette
//     size with the stop codes, n_out[1] = 1 when it does not fit `cap`.
__global__ __launch_bounds__(1024) void code_scan_kernel(const float *__restrict__ val, uint32_t nb, float coeff,
                                                         uint32_t *__restrict__ idx, float *__restrict__ tab,
                                                         uint32_t cap, uint32_t *__restrict__ n_out)
{
    __shared__ uint32_t part[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t lo = tid * per, hi = min(nb, lo + per);
    uint32_t local = 0;
    for (uint32_t i = lo; i < hi; ++i) local += val[i] > 0.0f ? 1u : 0u;
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < wave) before += part[w];
        total += part[w];
    }
    uint32_t k = before + incl - local;
    for (uint32_t i = lo; i < hi; ++i) {
        const float d = val[i];
        if (d > 0.0f) {
            idx[i] = k;
            if (k < cap) tab[k] = __builtin_fmaxf(d * coeff, 1.0f);
            ++k;
        }
    }
    if (tid == 0) {
        if (total + 2u <= cap) {
            tab[total] = PDT_HIT;
            tab[total + 1] = PDT_OUTSIDE;
        }
        n_out[0] = total + 2u;
        n_out[1] = total + 2u <= cap ? 0u : 1u;
    }
}

// (3) the padded, tiled map of codes.  u16: 4 * index (the LDS byte offset of the step); u8: the index.
//     hit = n_free, outside / a step of max_range or more = n_free + 1.
template <int ES>
__global__ __launch_bounds__(256) void pad_code_tiled_kernel(const float *__restrict__ dt, int rows, int cols,
                                                             void *__restrict__ cmap, TiledGeom tg, float coeff,
                                                             float max_range, const uint32_t *__restrict__ idx,
                                                             uint32_t nb, const uint32_t *__restrict__ n_pal)
{
    const uint32_t n_free = n_pal[0] - 2u;
    const int pr = blockIdx.y;
    const int r = pr - tg.padr;
    for (int pc = blockIdx.x * blockDim.x + threadIdx.x; pc < tg.pcols; pc += gridDim.x * blockDim.x) {
        const int c = pc - tg.pad;
        uint32_t code = n_free + 1u;
        if (r >= 0 && r < rows && c >= 0 && c < cols) {
            const float d = dt[(size_t)r * cols + c];
            if (d <= 0.0f) {
                code = n_free;
            } else if (__builtin_fmaxf(d * coeff, 1.0f) < max_range) {
                const uint32_t d2 = (uint32_t)__builtin_fmaf(d, d, 0.5f);
                code = d2 < nb ? idx[d2] : n_free + 1u;
            }
        }
        char *at = reinterpret_cast<char *>(cmap) + code_tiled_byte(pr, pc, tg.K, ES);
        if (ES == 1) *reinterpret_cast<uint16_t *>(at) = (uint16_t)(code << 2);
        else *reinterpret_cast<uint8_t *>(at) = (uint8_t)code;
    }
}

// The FIRST sample of a ray is taken at t = 0, i.e. at the pose's own cell, whatever the beam: it is
// read once per pose (with the record) instead of once per ray, and the ray starts at t = first step.
//   free origin cell   -> its step max(d*coeff, 1): the ray starts there with one sample counted
//   occupied origin    -> 0: the ray starts at t = 0 and finds the hit itself (KAT-2: its range is
//                         the distance to the cell corner, computed from the sampled cell)
//   no ray (origin outside the map / non-finite pose) -> PDT_NO_RAY: born finished, a miss
#define PDT_NO_RAY 2.5e38f
__device__ __forceinline__ float pose_first_step(const MapParams &m, float gx, float gy, uint32_t flags,
                                                 float coeff)
{
    if (flags & POSE_INVALID) return PDT_NO_RAY;
    const float v = m.dt[(size_t)(int)gy * m.cols + (int)gx];
    return v <= 0.0f ? 0.0f : __builtin_fmaxf(v * coeff, 1.0f);
}

// Tile (ty, tx) -> its rank in the binning order.  `tiles_xs` = tiles_x | stripe << 16.  stripe == 0: row-major.  Else the
// tile rows are grouped in stripes of `stripe` rows and a stripe is walked COLUMN by column: a band of the sorted pose
// list (an XCD's share) then sweeps its part of the map once, left to right, with a window one stripe tall — in row-major
// order every tile row of the band sweeps the whole map width again and re-reads what the row above it read (the step
// map of one sweep plus its max_range halo is larger than what an L2 shared by several launches keeps).
__device__ __forceinline__ uint32_t tile_key(int ty, int tx, int tiles_xs, int n_tiles)
{
    const int tiles_x = tiles_xs & 0xffff, rps = tiles_xs >> 16;
    if (rps == 0) return (uint32_t)(ty * tiles_x + tx);
    const int tiles_y = n_tiles / tiles_x;
    const int y0 = (ty / rps) * rps, h = min(rps, tiles_y - y0);
    return (uint32_t)(y0 * tiles_x + tx * h + (ty - y0));
}

__device__ __forceinline__ uint32_t pose_record(const MapParams &m, const float *__restrict__ poses,
                                                int p, int tile_shift, int tiles_x, int n_tiles,
                                                PoseRec &r, bool walk_outside = false)
{
    float thg;
    world_to_grid(m, poses[3 * (size_t)p], poses[3 * (size_t)p + 1], poses[3 * (size_t)p + 2], r.gx,
                  r.gy, thg);
    det_sincosf(thg, r.st, r.ct);
    const bool fin = (r.ct - r.ct) + (r.st - r.st) == 0.0f;
    const bool inb = r.gx > -1.0f && r.gx < m.fcols && r.gy > -1.0f && r.gy < m.frows;
    if (walk_outside) {
        // Bresenham keeps walking from an origin outside the map (cells out there are free); only
        // poses that cannot index the grid at all are dropped
        const bool sane = fin && __builtin_fabsf(r.gx) < 1e9f && __builtin_fabsf(r.gy) < 1e9f;
        if (!sane) {
            r.gx = 0.0f; r.gy = 0.0f; r.ct = 1.0f; r.st = 0.0f;
            return ((uint32_t)n_tiles - 1) | POSE_INVALID;
        }
        if (!inb) return (uint32_t)n_tiles - 1;
    } else if (!(fin && inb)) {
        r.gx = 0.0f; r.gy = 0.0f; r.ct = 1.0f; r.st = 0.0f;
        return ((uint32_t)n_tiles - 1) | POSE_INVALID;
    }
    return tile_key((int)r.gy >> tile_shift, (int)r.gx >> tile_shift, tiles_x, n_tiles);
}

// The record of a pose in the UPSTREAM-LITERAL arithmetic (variant 3, stream kernel template argument LIT):
// RangeMethod::numpy_calc_range's world -> grid with un-fused products and the double-precision sin / cos of the world
// angle rounded once (the checker's rm_cast_libm; literal_kernels.h::literal_cast).  calc_range(y, x, theta') marches
// (row, col): the record keeps the COLUMN coordinate in gx and the ROW coordinate in gy, as every loop here expects,
// and the pose's world heading in ct — the direction is computed per RAY at claim time (libm sinf / cosf of
// theta_p + alpha_j).  A pose the literal march would leave before its first sample (outside the map, non-finite) is
// flagged like the canonical one: every beam a miss.
__device__ __forceinline__ uint32_t pose_record_lit(const MapParams &m, const LiteralParams &lp, const float *__restrict__ poses,
                                                    int p, PoseRec &r)
{
    const float xw = poses[3 * (size_t)p], yw = poses[3 * (size_t)p + 1], thw = poses[3 * (size_t)p + 2];
    float x = (xw - m.ox) * m.inv_res;
    float y = (yw - m.oy) * m.inv_res;
    const float temp = x;
    x = lp.wcos * x - lp.wsin * y;
    y = lp.wsin * temp + lp.wcos * y;
    r.gx = x;
    r.gy = y;
    r.ct = thw;
    r.st = 0.0f;
    const bool fin = (thw - thw) == 0.0f;
    const bool inb = x > -1.0f && x < m.fcols && y > -1.0f && y < m.frows;       // (int) truncation: (-1, 0) is cell 0
    if (!(fin && inb)) {
        r.gx = 0.0f; r.gy = 0.0f; r.ct = 0.0f;
        return POSE_INVALID;
    }
    return 0u;
}

__global__ __launch_bounds__(1024) void pose_bin_kernel(MapParams m, const float *__restrict__ poses,
                                                        int n, PoseRec *__restrict__ rec,
                                                        PoseRec *__restrict__ rec_sorted,
                                                        uint32_t *__restrict__ order,
                                                        uint32_t *__restrict__ keys, int tile_shift,
                                                        int tiles_x, int n_tiles, int do_sort, int walk_outside,
                                                        float *__restrict__ d0, float coeff)
{
    extern __shared__ uint32_t hist[];          // n_tiles counters, then 1024 scan partials
    uint32_t *part = hist + n_tiles;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_tiles; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int p = tid; p < n; p += 1024) {
        PoseRec r;
        const uint32_t kf = pose_record(m, poses, p, tile_shift, tiles_x, n_tiles, r, walk_outside != 0);
        const uint32_t flag = kf & POSE_INVALID, key = kf & ~POSE_INVALID;
        if (do_sort) {
            rec[p] = r;
            keys[p] = key | flag;
            atomicAdd(&hist[key], 1u);
        } else {
            rec_sorted[p] = r;
            order[p] = (uint32_t)p | flag;
            if (d0) d0[p] = pose_first_step(m, r.gx, r.gy, flag, coeff);
        }
    }
    if (!do_sort) return;
    __syncthreads();
    // exclusive scan of hist[0..n_tiles): each lane owns E consecutive counters
    const int E = (n_tiles + 1023) / 1024;
    uint32_t local = 0;
    for (int e = 0; e < E; ++e) {
        int i = tid * E + e;
        if (i < n_tiles) local += hist[i];
    }
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t base = part[tid] - local;
    for (int e = 0; e < E; ++e) {
        int i = tid * E + e;
        if (i < n_tiles) {
            uint32_t c = hist[i];
            hist[i] = base;
            base += c;
        }
    }
    __syncthreads();
    for (int p = tid; p < n; p += 1024) {
        const uint32_t kf = keys[p];
        const uint32_t slot = atomicAdd(&hist[kf & ~POSE_INVALID], 1u);
        order[slot] = (uint32_t)p | (kf & POSE_INVALID);
        const PoseRec r = rec[p];
        rec_sorted[slot] = r;
        if (d0) d0[slot] = pose_first_step(m, r.gx, r.gy, kf, coeff);
    }
}

// Up to 8192 poses: the same binning with every lane keeping its (up to 8) pose records in
// registers between the histogram and the scatter pass — no scratch round trip through memory,
// and the 8 pose loads of a lane are in flight together.
// KEYS_ONLY: only the tile order is produced (order[slot] = pose id) — no sincos, no records: the
// march kernel derives the records of the blocks it owns itself (INLINE prologue, pose ids from
// `order`), so the ~100 instructions per pose of the record leave this one-workgroup critical path.
template <bool KEYS_ONLY>
__global__ __launch_bounds__(1024) void pose_bin_small_kernel(MapParams m, const float *__restrict__ poses,
                                                              int n, PoseRec *__restrict__ rec_sorted,
                                                              uint32_t *__restrict__ order,
                                                              int tile_shift, int tiles_x, int n_tiles,
                                                              int walk_outside, float *__restrict__ d0, float coeff)
{
    extern __shared__ uint32_t hist[];          // n_tiles counters, then 1024 scan partials
    uint32_t *part = hist + n_tiles;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_tiles; i += 1024) hist[i] = 0;
    __syncthreads();
    PoseRec r[8];
    uint32_t kf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = tid + u * 1024;
        kf[u] = 0;
        if (p < n) {
            if (KEYS_ONLY) {
                float gx, gy, thg;
                world_to_grid(m, poses[3 * (size_t)p], poses[3 * (size_t)p + 1], 0.0f, gx, gy, thg);
                const bool inb = gx > -1.0f && gx < m.fcols && gy > -1.0f && gy < m.frows;   // (NaN -> false)
                kf[u] = inb ? tile_key((int)gy >> tile_shift, (int)gx >> tile_shift, tiles_x, n_tiles) : (uint32_t)n_tiles - 1;
            } else {
                kf[u] = pose_record(m, poses, p, tile_shift, tiles_x, n_tiles, r[u], walk_outside != 0);
            }
            atomicAdd(&hist[kf[u] & ~POSE_INVALID], 1u);
        }
    }
    __syncthreads();
    const int E = (n_tiles + 1023) / 1024;
    uint32_t local = 0;
    for (int e = 0; e < E; ++e) {
        int i = tid * E + e;
        if (i < n_tiles) local += hist[i];
    }
    // exclusive scan of the 1024 per-lane sums: shuffle scan inside each wave, the 16 wave totals
    // through LDS (2 barriers instead of the 20 of a Hillis-Steele pass over `part`)
    uint32_t incl = local;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    {
        const uint32_t v = lane < 16 ? part[lane] : 0u;
        uint32_t wi = v;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)wi, off);
            if (lane >= off) wi += o;
        }
        wave_base = (uint32_t)__shfl((int)(wi - v), wave);
    }
    uint32_t base = wave_base + incl - local;
    for (int e = 0; e < E; ++e) {
        int i = tid * E + e;
        if (i < n_tiles) {
            uint32_t c = hist[i];
            hist[i] = base;
            base += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = tid + u * 1024;
        if (p < n) {
            const uint32_t slot = atomicAdd(&hist[kf[u] & ~POSE_INVALID], 1u);
            order[slot] = (uint32_t)p | (kf[u] & POSE_INVALID);
            if (!KEYS_ONLY) {
                rec_sorted[slot] = r[u];
                if (d0) d0[slot] = pose_first_step(m, r[u].gx, r[u].gy, kf[u], coeff);
            }
        }
    }
}

// Large batches: the same binning as three grid-wide kernels (one lane per pose, tile histogram
// and cursors in global memory), because one workgroup walking 10^5..10^6 poses would serialise
// hundreds of microseconds in front of the march.
// Workgroup w owns poses [w*POSES_PER_WG, ...): per-workgroup tile histograms in LDS (no contended
// global atomics — clustered roll-out poses would serialise on a few words), written tile-major
// as hist_all[tile * n_wg + w]; one scan over that array then gives every (tile, workgroup) pair
// its base slot, and the scatter pass hands out slots from LDS cursors.

__global__ __launch_bounds__(256) void pose_prep_kernel(MapParams m, const float *__restrict__ poses,
                                                        int n, PoseRec *__restrict__ rec,
                                                        uint32_t *__restrict__ keys,
                                                        uint32_t *__restrict__ hist_all, int n_wg,
                                                        int tile_shift, int tiles_x, int n_tiles,
                                                        uint32_t *__restrict__ order_if_unsorted,
                                                        int walk_outside, int poses_per_wg,
                                                        float *__restrict__ d0_if_unsorted, float coeff)
{
    extern __shared__ uint32_t lhist[];            // n_tiles
    const int w = blockIdx.x;
    if (!order_if_unsorted) {
        for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) lhist[i] = 0;
        __syncthreads();
    }
    const int p_end = min(n, (w + 1) * poses_per_wg);
    for (int p = w * poses_per_wg + threadIdx.x; p < p_end; p += blockDim.x) {
        PoseRec r;
        const uint32_t kf = pose_record(m, poses, p, tile_shift, tiles_x, n_tiles, r, walk_outside != 0);
        rec[p] = r;
        if (order_if_unsorted) {                   // keep the caller's pose order
            order_if_unsorted[p] = (uint32_t)p | (kf & POSE_INVALID);
            if (d0_if_unsorted) d0_if_unsorted[p] = pose_first_step(m, r.gx, r.gy, kf, coeff);
        } else {
            keys[p] = kf;
            atomicAdd(&lhist[kf & ~POSE_INVALID], 1u);
        }
    }
    if (order_if_unsorted) return;
    __syncthreads();
    for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) hist_all[(size_t)i * n_wg + w] = lhist[i];
}

// exclusive scan of the 256 values a workgroup of 256 holds (one per lane) + their total
__device__ __forceinline__ uint32_t wg256_excl_scan(uint32_t v, uint32_t *part /* 4 */, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += o;
    }
    __syncthreads();                                   // (part[] of the previous tile has been read)
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += part[w];
    total = part[0] + part[1] + part[2] + part[3];
    return before + incl - v;
}

// The (tile, workgroup) counters of the grid-wide binning, hist_all[tile * n_wg + w], scanned by one
// workgroup PER TILE — inside the tile's own run of n_wg counters (coalesced 256-wide pieces, running carry) +
// the tile's total; the totals of the tiles in front are added by the scatter kernel — instead
// of one workgroup walking all tiles x workgroups counters with a lane-strided pattern (131 072 counters at
// 262 144 poses: the single-workgroup scan was the longest of the three binning kernels).
__global__ __launch_bounds__(256) void tile_scan_a_kernel(uint32_t *__restrict__ hist_all, int n_wg,
                                                          uint32_t *__restrict__ tile_total)
{
    __shared__ uint32_t part[4];
    uint32_t *row = hist_all + (size_t)blockIdx.x * n_wg;
    uint32_t carry = 0;
    for (int i0 = 0; i0 < n_wg; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        const uint32_t v = i < n_wg ? row[i] : 0u;
        uint32_t tot;
        const uint32_t ex = wg256_excl_scan(v, part, tot);
        if (i < n_wg) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) tile_total[blockIdx.x] = carry;
}

// (round 4: the scan of the tile totals — tile_scan_b_kernel, a launch of its own until then, ~4.6 us of launch
//  latency for a microsecond of work — happens here: every scatter workgroup scans the <= 1024 totals itself)
__global__ __launch_bounds__(256) void pose_scatter_kernel(int n, const PoseRec *__restrict__ rec,
                                                           const uint32_t *__restrict__ keys,
                                                           const uint32_t *__restrict__ base_all,
                                                           const uint32_t *__restrict__ tile_total,
                                                           int n_wg, int n_tiles,
                                                           PoseRec *__restrict__ rec_sorted,
                                                           uint32_t *__restrict__ order, int poses_per_wg,
                                                           MapParams m, float *__restrict__ d0, float coeff)
{
    extern __shared__ uint32_t cursor[];           // n_tiles
    __shared__ uint32_t part[4];
    const int w = blockIdx.x;
    {
        // cursor[tile] = slots of this tile in front of this workgroup's poses (tile_scan_a) + poses in the tiles
        // in front of it (exclusive scan of the tile totals: each lane owns E consecutive tiles)
        const int E = (n_tiles + 255) / 256;
        uint32_t local = 0;
        for (int e = 0; e < E; ++e) {
            const int i = (int)threadIdx.x * E + e;
            if (i < n_tiles) local += tile_total[i];
        }
        uint32_t tot;
        uint32_t base = wg256_excl_scan(local, part, tot);
        for (int e = 0; e < E; ++e) {
            const int i = (int)threadIdx.x * E + e;
            if (i < n_tiles) {
                cursor[i] = base_all[(size_t)i * n_wg + w] + base;
                base += tile_total[i];
            }
        }
    }
    __syncthreads();
    const int p_end = min(n, (w + 1) * poses_per_wg);
    for (int p = w * poses_per_wg + threadIdx.x; p < p_end; p += blockDim.x) {
        const uint32_t kf = keys[p];
        const uint32_t slot = atomicAdd(&cursor[kf & ~POSE_INVALID], 1u);
        order[slot] = (uint32_t)p | (kf & POSE_INVALID);
        const PoseRec r = rec[p];
        rec_sorted[slot] = r;
        if (d0) d0[slot] = pose_first_step(m, r.gx, r.gy, kf, coeff);
    }
}

// (cos, sin) of every beam angle of a fan: the table the stream kernels stage into LDS
__global__ __launch_bounds__(256) void fan_table_kernel(FanParams f, float2 *__restrict__ tab)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < f.num_rays) {
        float s, c;
        det_sincosf(fan_alpha(f, j), s, c);
        tab[j] = make_float2(c, s);
    }
}

// unsigned division by a launch-time constant (round-up method, any 32-bit dividend)
struct FastDiv {
    uint32_t mul, sh1, sh2, d;
};
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv &f)
{
    uint32_t t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}


// position of a sample: (origin + direction x t) for both coordinates.  Canonical form: ONE packed fma (both halves
// IEEE-fused: the CPU statement's fmaf).  LIT (upstream-literal arithmetic, variant 3: range_libc computes x0 + dx * t with the
// product and the sum each rounded to float32): packed multiply, then packed add — one VALU instruction more per sample.
// Operands as register-pair strings: D <- DIR(hi, lo crossed by op_sel) x T(lo, lo) [+] ORG.
#define RM_POS(D, DIR, T, ORG)                                                                             \
    ".if %[lit]\n\t"                                                                                       \
    "v_pk_mul_f32 " D ", " DIR ", " T " op_sel:[1,0] op_sel_hi:[0,0]\n\t"                                    \
    "v_pk_add_f32 " D ", " D ", " ORG "\n\t"                                                                 \
    ".else\n\t"                                                                                            \
    "v_pk_fma_f32 " D ", " DIR ", " T ", " ORG " op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"                        \
    ".endif\n\t"

// LDS header of the stream kernels in floats: [0] slot counter, [1] spare, [2..66) crash_seen.  CODE launches keep
// the code -> step table right behind it, at a byte offset the march loops name as an immediate (RM_DECODE).
constexpr int STREAM_HDR = 66;

// ------------------------------------------------------------------------------
// The CODE map (round 6): the step map in 16-bit codes (CODE = 2) or 8-bit codes (CODE = 1) instead of float32 steps.
// A map holds few DISTINCT steps — the EDT is sqrtf of an integer, a maze of 40-cell corridors has 624 of them, a
// 4096^2 one 1591 —, so a cell stores the index of its step in the map's palette and the workgroup keeps the palette
// (exact float32 steps, +inf for "hit", 3e38 for "left the map or a step past max_range") in LDS: the sample sequence
// and every bit of the result are those of the float32 step map, a 128-B line holds 8 x 8 (u16) or 8 x 16 (u8) cells
// instead of 4 x 8, and a band of the map is half / a quarter of the bytes in the XCD's L2.
//   u16: the cell holds 4 * index — the LDS byte offset: sample = global_load_ushort + ds_read_b32, no VALU more;
//   u8:  the cell holds the index: one shift more per sample.
// RM_LOAD fetches a sample's cell, RM_DECODE (behind the s_waitcnt that returned it) turns it into the step in place.
// ------------------------------------------------------------------------------
#define RM_LOAD(D, A)                                                                                      \
    ".if %[code] == 2\n\t"                                                                                 \
    "global_load_ushort " D ", " A ", %[base]\n\t"                                                          \
    ".elseif %[code] == 1\n\t"                                                                             \
    "global_load_ubyte " D ", " A ", %[base]\n\t"                                                           \
    ".else\n\t"                                                                                            \
    "global_load_dword " D ", " A ", %[base]\n\t"                                                           \
    ".endif\n\t"
#define RM_DECODE(D)                                                                                       \
    ".if %[code] == 1\n\t"                                                                                 \
    "v_lshlrev_b32_e32 " D ", 2, " D "\n\t"                                                                 \
    ".endif\n\t"                                                                                           \
    ".if %[code]\n\t"                                                                                      \
    "ds_read_b32 " D ", " D " offset:%[taboff]\n\t"                                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                             \
    ".endif\n\t"
// one slot of march_loop2 advances by the sample in d: t += d, drop the lane when t left the range window, next cell's address, its load
#define RM2_ADVANCE_A                                                                                      \
    "v_add_f32_e32 v20, v20, %[dA]\n\t"                                                                    \
    "v_cmpx_gt_f32_e32 %[mx], v20\n\t"                                                                     \
    "s_mov_b64 %[mA], exec\n\t"                                                                            \
    RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")                                                 \
    "v_cvt_i32_f32_e32 %[cA], v26\n\t"                                                                     \
    "v_cvt_i32_f32_e32 %[rA], v27\n\t"                                                                     \
    ".if %[tiled]\n\t"                                                                                     \
    "v_mad_i32_i24 v26, %[rA], %[stride], %[k4]\n\t"                                                       \
    "v_and_b32_e32 v26, %[nstride], v26\n\t"                                                               \
    "v_lshl_add_u32 v26, %[cA], %[cs], v26\n\t"                                                            \
    ".else\n\t"                                                                                            \
    "v_mad_i32_i24 v26, %[rA], %[stride], %[cA]\n\t"                                                       \
    "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"                                                                \
    ".endif\n\t"                                                                                           \
    RM_LOAD("%[dA]", "v26")
#define RM2_ADVANCE_B                                                                                      \
    "v_add_f32_e32 v28, v28, %[dB]\n\t"                                                                    \
    "v_cmpx_gt_f32_e32 %[mx], v28\n\t"                                                                     \
    "s_mov_b64 %[mB], exec\n\t"                                                                            \
    RM_POS("v[34:35]", "v[30:31]", "v[28:29]", "v[32:33]")                                                 \
    "v_cvt_i32_f32_e32 %[cB], v34\n\t"                                                                     \
    "v_cvt_i32_f32_e32 %[rB], v35\n\t"                                                                     \
    ".if %[tiled]\n\t"                                                                                     \
    "v_mad_i32_i24 v34, %[rB], %[stride], %[k4]\n\t"                                                       \
    "v_and_b32_e32 v34, %[nstride], v34\n\t"                                                               \
    "v_lshl_add_u32 v34, %[cB], %[cs], v34\n\t"                                                            \
    ".else\n\t"                                                                                            \
    "v_mad_i32_i24 v34, %[rB], %[stride], %[cB]\n\t"                                                       \
    "v_lshl_add_u32 v34, v34, 2, %[k4]\n\t"                                                                \
    ".endif\n\t"                                                                                           \
    RM_LOAD("%[dB]", "v34")
// (the column shift of the tiled address: bytes a column takes inside a row group — 4 rows x 4 B, 8 rows x 2 B: 16; 8 rows x 1 B: 8)
#define RM_CODE_OPERANDS [code] "n"(CODE), [cs] "n"(CODE == 1 ? 3 : 4), [taboff] "n"(STREAM_HDR * 4)

// ------------------------------------------------------------------------------
// The march loop of K1b, hand-scheduled for gfx950.  EXEC holds the live lanes
// (v_cmpx drops a lane the moment its t reaches max_range, hits, or leaves the map),
// so finished lanes cost nothing but their slot and keep (c, r, d) of their last
// sample; the loop leaves when at most `low` lanes are still live.
// Per sample: 9 VALU (the two position fmas are one packed instruction) + 1 global load + 4 SALU
// (either step coefficient).
//   fx = fma(dx,t,gx); fy = fma(dy,t,gy); c = (int)fx; r = (int)fy      (Appendix A "march")
//   d  = step map at (r, c)              = max(dt*coeff, 1) | +inf (occupied) | 3e38 (border)
//   t += d                               => a hit / leaving the map pushes t past max_range
// ------------------------------------------------------------------------------
template <bool AUX, bool TILED, bool LIT = false, int CODE = 0>
__device__ __forceinline__ void march_loop(float dx, float dy, float gx, float gy, float &t, int &c,
                                           int &r, float &d, uint32_t &nstep, const float *pdt,
                                           int stride, int nstride, uint32_t k4, float max_range,
                                           uint32_t low)
{
    // TILED: stride = M, nstride = MASK, k4 = padM (see pdt_tiled_byte): 3 address instructions instead of
    // 2, but the samples of a wave fall into fewer 128-B lines (4x8-cell blocks instead of 1x32-cell row pieces)
    // The two position fmas are ONE packed instruction (v_pk_fma_f32: both halves IEEE-fused, the same
    // bits as two v_fma_f32).  Packed operands are even-aligned register pairs, and inline asm cannot
    // name the halves of a 64-bit operand, so the pairs are fixed registers, in the order the refill
    // code leaves the values in (no copies in front of the block): direction (dy, dx) v[22:23] — its
    // halves are crossed by op_sel —, origin (gx, gy) v[24:25], t v20 broadcast to both halves (v21 is
    // named by the encoding, never read), position / address scratch v[26:27].
    unsigned long long save;
    uint32_t n;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n"
        "L_march_%=:\n\t"
        RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")
        "v_cvt_i32_f32_e32 %[c], v26\n\t"
        "v_cvt_i32_f32_e32 %[r], v27\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v26, %[nstride], v26\n\t"
        "v_lshl_add_u32 v26, %[c], %[cs], v26\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[c]\n\t"
        "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"
        ".endif\n\t"
        RM_LOAD("%[d]", "v26")
        ".if %[aux]\n\t"
        "v_add_u32_e32 %[ns], 1, %[ns]\n\t"
        ".endif\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        RM_DECODE("%[d]")
        "v_add_f32_e32 v20, v20, %[d]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_bcnt1_i32_b64 %[n], exec\n\t"
        "s_cmp_gt_u32 %[n], %[low]\n\t"
        "s_cbranch_scc1 L_march_%=\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [t] "+{v20}"(t), [c] "+v"(c), [r] "+v"(r), [d] "+v"(d), [ns] "+v"(nstep),
          [save] "=&s"(save), [n] "=&s"(n)
        : [dy] "{v22}"(dy), [dx] "{v23}"(dx), [gx] "{v24}"(gx), [gy] "{v25}"(gy),
          [mx] "s"(max_range), [stride] "s"(stride), [nstride] "s"(nstride), [k4] "v"(k4),
          [base] "s"(pdt), [lit] "n"(LIT ? 1 : 0), RM_CODE_OPERANDS, [low] "s"(low), [aux] "n"(AUX ? 1 : 0), [tiled] "n"(TILED ? 1 : 0)
        : "v26", "v27", "vcc", "scc", "memory");
}


// march_loop with an iteration cap (drain phase: a bounded stretch of the plain loop between two attempts of
// the speculating loop).  Leaves when no lane is live or after `iters` samples per lane.
template <bool TILED, bool LIT = false, int CODE = 0>
__device__ __forceinline__ void march_loop_capped(float dx, float dy, float gx, float gy, float &t, int &c, int &r,
                                                  float &d, const float *pdt, int stride, int nstride, uint32_t k4,
                                                  float max_range, uint32_t iters)
{
    unsigned long long save;
    uint32_t n = iters;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_cbranch_execz L_cap_done_%=\n"
        "L_cap_%=:\n\t"
        RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")
        "v_cvt_i32_f32_e32 %[c], v26\n\t"
        "v_cvt_i32_f32_e32 %[r], v27\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v26, %[nstride], v26\n\t"
        "v_lshl_add_u32 v26, %[c], %[cs], v26\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[c]\n\t"
        "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"
        ".endif\n\t"
        RM_LOAD("%[d]", "v26")
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        RM_DECODE("%[d]")
        "v_add_f32_e32 v20, v20, %[d]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_cbranch_execz L_cap_done_%=\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 L_cap_%=\n"
        "L_cap_done_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [t] "+{v20}"(t), [c] "+v"(c), [r] "+v"(r), [d] "+v"(d), [save] "=&s"(save), [n] "+s"(n)
        : [dy] "{v22}"(dy), [dx] "{v23}"(dx), [gx] "{v24}"(gx), [gy] "{v25}"(gy), [mx] "s"(max_range),
          [stride] "s"(stride), [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [lit] "n"(LIT ? 1 : 0), RM_CODE_OPERANDS, [tiled] "n"(TILED ? 1 : 0)
        : "v26", "v27", "vcc", "scc", "memory");
}

// ------------------------------------------------------------------------------
// The DRAIN loop of the one-ray-per-lane kernel: value speculation on the step.
// When a workgroup's stream has run dry, what is left are single long rays — rays sliding along a wall
// take 70..240 samples (mean 6.9), each a dependent load (~115 ns), and the launch ends with the longest
// of them.  Such a ray sees the same step again and again (93 % of the steps of chains >= 80 samples repeat
// their predecessor, tools: /tmp-free CPU replay in DESIGN.md section 4), so the samples at t, t+g, t+2g,
// t+3g (g = the last step) are loaded TOGETHER and the k-th is consumed only if the march really arrived
// at that t: t_k = t_{k-1} + g bit for bit when sample k-1 returned g.  Same t sequence, same cells, same
// results as march_loop — 1 memory round trip per up to 4 samples instead of per sample.
// ~51 VALU per iteration: only worth it when few lanes are live and the SIMD is idle (drain phase).
// Speculative samples are only loaded where t_k < max_range (the ray stays inside the padded map there).
// (c, r) of the last consumed sample are recomputed from its t (kept in tp) when the loop leaves.
// Registers: as march_loop + t1 v28, t2 v30, t3 v32 (low halves of pairs, whose high halves the packed fma
// names but never reads: g v29, samples 1 and 2 in v31 / v33), positions / addresses v[34:39], tp v40 (pair),
// samples 0 and 3 in v42 / v43 — exactly the fixed registers of slots B and C of the several-rays-per-lane
// kernels, which are dead when this loop runs there (no register beyond theirs).
// ------------------------------------------------------------------------------
template <bool TILED, bool LIT = false, int CODE = 0>
__device__ __forceinline__ void march_drain4(float dx, float dy, float gx, float gy, float &t, int &c, int &r,
                                             float &d, const float *pdt, int stride, int nstride, uint32_t k4,
                                             float max_range)
{
    static_assert(TILED, "the speculative drain loop exists for the tiled step map only");
    // The loop also leaves when an iteration's FIRST prediction failed on every live lane (a ray along a
    // diagonal wall alternates between two steps and never repeats its predecessor): the caller then marches a
    // bounded stretch with the plain loop before the next attempt.
    unsigned long long save, ent, live, hit;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[ent], exec\n\t"
        "s_mov_b64 %[live], exec\n\t"
        "s_cbranch_execz L_drain_done_%=\n"
        "L_drain_%=:\n\t"
        "v_mov_b32_e32 v29, %[d]\n\t"                       // g
        "v_add_f32_e32 v28, v20, v29\n\t"                   // t1, t2, t3
        "v_add_f32_e32 v30, v28, v29\n\t"
        "v_add_f32_e32 v32, v30, v29\n\t"
        // sample 0 (every live lane)
        RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")
        "v_cvt_i32_f32_e32 v26, v26\n\t"
        "v_cvt_i32_f32_e32 v27, v27\n\t"
        "v_mad_i32_i24 v27, v27, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v27, %[nstride], v27\n\t"
        "v_lshl_add_u32 v26, v26, %[cs], v27\n\t"
        RM_LOAD("v42", "v26")
        // sample 1 where t1 is still inside the range window
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        RM_POS("v[34:35]", "v[22:23]", "v[28:29]", "v[24:25]")
        "v_cvt_i32_f32_e32 v34, v34\n\t"
        "v_cvt_i32_f32_e32 v35, v35\n\t"
        "v_mad_i32_i24 v35, v35, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v35, %[nstride], v35\n\t"
        "v_lshl_add_u32 v34, v34, %[cs], v35\n\t"
        RM_LOAD("v31", "v34")
        // sample 2
        "v_cmpx_gt_f32_e32 %[mx], v30\n\t"
        RM_POS("v[36:37]", "v[22:23]", "v[30:31]", "v[24:25]")
        "v_cvt_i32_f32_e32 v36, v36\n\t"
        "v_cvt_i32_f32_e32 v37, v37\n\t"
        "v_mad_i32_i24 v37, v37, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v37, %[nstride], v37\n\t"
        "v_lshl_add_u32 v36, v36, %[cs], v37\n\t"
        RM_LOAD("v33", "v36")
        // sample 3
        "v_cmpx_gt_f32_e32 %[mx], v32\n\t"
        RM_POS("v[38:39]", "v[22:23]", "v[32:33]", "v[24:25]")
        "v_cvt_i32_f32_e32 v38, v38\n\t"
        "v_cvt_i32_f32_e32 v39, v39\n\t"
        "v_mad_i32_i24 v39, v39, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v39, %[nstride], v39\n\t"
        "v_lshl_add_u32 v38, v38, %[cs], v39\n\t"
        RM_LOAD("v43", "v38")
        // stage 0: the sample at t is always real
        "s_mov_b64 exec, %[live]\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        RM_DECODE("v42")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v42\n\t"
        "v_add_f32_e32 v20, v20, v42\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"                  // still marching ...
        "v_cmpx_eq_f32_e32 v42, v29\n\t"                    // ... and the step was the predicted one
        "s_mov_b64 %[hit], exec\n\t"
        // stage 1: the march arrived at t1 exactly
        "s_waitcnt vmcnt(2)\n\t"
        RM_DECODE("v31")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v31\n\t"
        "v_add_f32_e32 v20, v20, v31\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v31, v29\n\t"
        // stage 2
        "s_waitcnt vmcnt(1)\n\t"
        RM_DECODE("v33")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v33\n\t"
        "v_add_f32_e32 v20, v20, v33\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v33, v29\n\t"
        // stage 3
        "s_waitcnt vmcnt(0)\n\t"
        RM_DECODE("v43")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v43\n\t"
        "v_add_f32_e32 v20, v20, v43\n\t"
        // who is still marching
        "s_mov_b64 exec, %[live]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[live], exec\n\t"
        "s_cbranch_execz L_drain_out_%=\n\t"
        "s_cmp_lg_u64 %[hit], 0\n\t"
        "s_cbranch_scc1 L_drain_%=\n"
        "L_drain_out_%=:\n\t"
        // cell of the last consumed sample of every ray that went through this loop
        "s_mov_b64 exec, %[ent]\n\t"
        RM_POS("v[26:27]", "v[22:23]", "v[40:41]", "v[24:25]")
        "v_cvt_i32_f32_e32 %[c], v26\n\t"
        "v_cvt_i32_f32_e32 %[r], v27\n"
        "L_drain_done_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [t] "+{v20}"(t), [c] "+v"(c), [r] "+v"(r), [d] "+v"(d), [save] "=&s"(save), [ent] "=&s"(ent),
          [live] "=&s"(live), [hit] "=&s"(hit)
        : [dy] "{v22}"(dy), [dx] "{v23}"(dx), [gx] "{v24}"(gx), [gy] "{v25}"(gy), [mx] "s"(max_range),
          [stride] "s"(stride), [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [lit] "n"(LIT ? 1 : 0), RM_CODE_OPERANDS
        : "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40",
          "v41", "v42", "v43", "vcc", "scc", "memory");
}


// ------------------------------------------------------------------------------
// The GROUP drain loop (round 5): 2 or 4 lanes per ray, 8 / 16 samples of one ray in flight.
// The last rays of a dry wave are single long chains whose steps repeat (rays of 48..80 samples: 80 % of the steps
// equal their predecessor, from 80 samples up 91 %: tools replay in profiles/r05/drain_depth_model.txt), and each
// dependent sample is a memory round trip.  march_drain4 speculates 4 deep because one lane has no registers for
// more; once a wave holds <= 32 / <= 16 live rays it has LANES to spare: the rays are laid out L = 2 / 4 lanes per ray
// (every lane of a group holds the ray's state), lane j of a group starts 4 j steps ahead — t advanced by 4 j
// SEQUENTIAL additions of the last step g, the same roundings the march itself would make — and runs the 4-deep
// speculation body of march_drain4 from there.  A lane's samples count only if every lane in front of it in the
// group found four repeats (its start is then bit for bit where the march arrived); the ray's new state is the state
// of the first lane whose chain broke (or of the last lane), broadcast to the group with ds_bpermute.  Same t
// sequence, same cells, same results as march_loop — up to 4 L samples per round trip.  Every iteration speculates
// (no plain stretch in between: the SIMD has nothing else to do, a failed prediction costs issue slots nobody wants).
// Leaves when at most `low_lanes` lanes are live.  lm1 = L - 1.
// Registers: exactly those of march_drain4 (the lane's place in its group is re-derived from the lane id in registers
// that are dead at that point: a single VGPR more in this block's footprint made the compiler spill ray state around the
// hot loops of the kernel, which sits at its 64-VGPR occupancy limit).
// ------------------------------------------------------------------------------
template <bool TILED, bool LIT = false, int CODE = 0>
__device__ __forceinline__ void march_drain_group(float dx, float dy, float gx, float gy, float &t, int &c, int &r,
                                                  float &d, const float *pdt, int stride, int nstride, uint32_t k4,
                                                  float max_range, uint32_t lm1, uint32_t low_lanes)
{
    static_assert(TILED, "the speculative drain loops exist for the tiled step map only");
    unsigned long long save, ent0, live, ent, am;
    uint32_t n;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[ent0], exec\n\t"
        "s_mov_b64 %[live], exec\n\t"
        "s_cbranch_execz L_grp_done_%=\n"
        "L_grp_%=:\n\t"
        "v_mov_b32_e32 v29, %[d]\n\t"                       // g
        // lane j of a group starts 4 j steps ahead: sequential additions, the march's own roundings
        "v_mbcnt_lo_u32_b32 v26, -1, 0\n\t"
        "v_mbcnt_hi_u32_b32 v26, -1, v26\n\t"
        "v_and_b32_e32 v27, %[lm1], v26\n\t"               // j = lane & (L - 1)
        "v_cmpx_lt_u32_e32 0, v27\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_cmpx_lt_u32_e32 1, v27\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_cmpx_lt_u32_e32 2, v27\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "v_add_f32_e32 v20, v20, v29\n\t"
        "s_mov_b64 exec, %[live]\n\t"
        // lanes whose start is still inside the range window take part (lane 0 of a live group always does)
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[ent], exec\n\t"
        "v_add_f32_e32 v28, v20, v29\n\t"                   // t1, t2, t3
        "v_add_f32_e32 v30, v28, v29\n\t"
        "v_add_f32_e32 v32, v30, v29\n\t"
        // sample 0
        RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")
        "v_cvt_i32_f32_e32 v26, v26\n\t"
        "v_cvt_i32_f32_e32 v27, v27\n\t"
        "v_mad_i32_i24 v27, v27, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v27, %[nstride], v27\n\t"
        "v_lshl_add_u32 v26, v26, %[cs], v27\n\t"
        RM_LOAD("v42", "v26")
        // sample 1 where t1 is still inside the range window
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        RM_POS("v[34:35]", "v[22:23]", "v[28:29]", "v[24:25]")
        "v_cvt_i32_f32_e32 v34, v34\n\t"
        "v_cvt_i32_f32_e32 v35, v35\n\t"
        "v_mad_i32_i24 v35, v35, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v35, %[nstride], v35\n\t"
        "v_lshl_add_u32 v34, v34, %[cs], v35\n\t"
        RM_LOAD("v31", "v34")
        // sample 2
        "v_cmpx_gt_f32_e32 %[mx], v30\n\t"
        RM_POS("v[36:37]", "v[22:23]", "v[30:31]", "v[24:25]")
        "v_cvt_i32_f32_e32 v36, v36\n\t"
        "v_cvt_i32_f32_e32 v37, v37\n\t"
        "v_mad_i32_i24 v37, v37, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v37, %[nstride], v37\n\t"
        "v_lshl_add_u32 v36, v36, %[cs], v37\n\t"
        RM_LOAD("v33", "v36")
        // sample 3
        "v_cmpx_gt_f32_e32 %[mx], v32\n\t"
        RM_POS("v[38:39]", "v[22:23]", "v[32:33]", "v[24:25]")
        "v_cvt_i32_f32_e32 v38, v38\n\t"
        "v_cvt_i32_f32_e32 v39, v39\n\t"
        "v_mad_i32_i24 v39, v39, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v39, %[nstride], v39\n\t"
        "v_lshl_add_u32 v38, v38, %[cs], v39\n\t"
        RM_LOAD("v43", "v38")
        // stage 0: the sample at the lane's start (real for lane 0; for lane j > 0 if the lanes in front all repeated)
        "s_mov_b64 exec, %[ent]\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        RM_DECODE("v42")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v42\n\t"
        "v_add_f32_e32 v20, v20, v42\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v42, v29\n\t"
        // stage 1
        "s_waitcnt vmcnt(2)\n\t"
        RM_DECODE("v31")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v31\n\t"
        "v_add_f32_e32 v20, v20, v31\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v31, v29\n\t"
        // stage 2
        "s_waitcnt vmcnt(1)\n\t"
        RM_DECODE("v33")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v33\n\t"
        "v_add_f32_e32 v20, v20, v33\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v33, v29\n\t"
        // stage 3
        "s_waitcnt vmcnt(0)\n\t"
        RM_DECODE("v43")
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v43\n\t"
        "v_add_f32_e32 v20, v20, v43\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v43, v29\n\t"
        "s_mov_b64 %[am], exec\n\t"                         // four repeats and still marching: the next lane's start is real
        // the group's new state: the first lane whose chain broke, or the last lane
        "s_mov_b64 exec, %[save]\n\t"
        "v_mbcnt_lo_u32_b32 v28, -1, 0\n\t"
        "v_mbcnt_hi_u32_b32 v28, -1, v28\n\t"
        "v_and_b32_e32 v30, %[lm1], v28\n\t"
        "v_sub_u32_e32 v28, v28, v30\n\t"                  // first lane of my group
        "v_lshrrev_b64 v[26:27], v28, %[am]\n\t"           // my group's all-match bits from bit 0 up
        "v_not_b32_e32 v26, v26\n\t"
        "v_ffbl_b32_e32 v26, v26\n\t"
        "v_min_u32_e32 v26, %[lm1], v26\n\t"               // the tail lane: first one whose chain broke, or the last
        "v_add_lshl_u32 v26, v26, v28, 2\n\t"
        "ds_bpermute_b32 v20, v26, v20\n\t"
        "ds_bpermute_b32 %[d], v26, %[d]\n\t"
        "ds_bpermute_b32 v40, v26, v40\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        // who is still marching
        "s_mov_b64 exec, %[live]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[live], exec\n\t"
        "s_bcnt1_i32_b64 %[n], exec\n\t"
        "s_cmp_gt_u32 %[n], %[low]\n\t"
        "s_cbranch_scc1 L_grp_%=\n\t"
        // cell of the last consumed sample of every ray that went through this loop
        "s_mov_b64 exec, %[ent0]\n\t"
        RM_POS("v[26:27]", "v[22:23]", "v[40:41]", "v[24:25]")
        "v_cvt_i32_f32_e32 %[c], v26\n\t"
        "v_cvt_i32_f32_e32 %[r], v27\n"
        "L_grp_done_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [t] "+{v20}"(t), [c] "+v"(c), [r] "+v"(r), [d] "+v"(d), [save] "=&s"(save), [ent0] "=&s"(ent0),
          [live] "=&s"(live), [ent] "=&s"(ent), [am] "=&s"(am), [n] "=&s"(n)
        : [dy] "{v22}"(dy), [dx] "{v23}"(dx), [gx] "{v24}"(gx), [gy] "{v25}"(gy), [mx] "s"(max_range),
          [stride] "s"(stride), [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [lit] "n"(LIT ? 1 : 0), RM_CODE_OPERANDS, [lm1] "s"(lm1),
          [low] "s"(low_lanes)
        : "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40",
          "v41", "v42", "v43", "vcc", "scc", "memory");
}

// Two rays per lane (SLOTS = 2 of the stream kernel): slot A and slot B of a lane are two independent
// rays with their own live masks.  The wave alternates EXEC between the masks — switching is scalar
// work — so the VALU count per sample stays 9 and a finished slot needs no predication, while BOTH
// slots' loads are in flight together: twice the memory-level parallelism of a wave that has at most 8
// siblings on its SIMD.  (Round 1 tried two slots with per-slot predication on the row-major layout:
// the extra VALU per sample made it 6 % slower.)  Registers are fixed as in march_loop: slot A
// t v20 / dir v[22:23] / origin v[24:25] / scratch v[26:27], slot B t v28 / v[30:31] / v[32:33] / v[34:35].
template <bool TILED, bool LIT = false, int CODE = 0>
__device__ __forceinline__ void march_loop2(float dxA, float dyA, float gxA, float gyA, float &tA, int &cA, int &rA,
                                            float &dA, float dxB, float dyB, float gxB, float gyB, float &tB,
                                            int &cB, int &rB, float &dB, const float *pdt, int stride, int nstride,
                                            uint32_t k4, float max_range, uint32_t low)
{
    unsigned long long save, mA, mB;
    uint32_t n, n2;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[mA], exec\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        "s_mov_b64 %[mB], exec\n"
        // (decoupled slots: a slot's next load is issued as soon as ITS previous one has returned — not after both
        //  have — so the two dependent chains only share the instruction stream, not each other's latency.
        //  CODE map: a form that issues a slot's palette look-up and runs the OTHER slot's arithmetic under it was
        //  built and is no faster — 186 vs 190 Grays/s at cfg2 / 300 steps, equal elsewhere: profiles/r06/ab_code_map.txt,
        //  tools/r06/rejected/code_pipe.patch — the other seven waves of the SIMD already cover the ~100 clocks)
        "s_mov_b64 exec, %[mA]\n\t"
        RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")
        "v_cvt_i32_f32_e32 %[cA], v26\n\t"
        "v_cvt_i32_f32_e32 %[rA], v27\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v26, %[rA], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v26, %[nstride], v26\n\t"
        "v_lshl_add_u32 v26, %[cA], %[cs], v26\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v26, %[rA], %[stride], %[cA]\n\t"
        "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"
        ".endif\n\t"
        RM_LOAD("%[dA]", "v26")
        "s_mov_b64 exec, %[mB]\n\t"
        RM_POS("v[34:35]", "v[30:31]", "v[28:29]", "v[32:33]")
        "v_cvt_i32_f32_e32 %[cB], v34\n\t"
        "v_cvt_i32_f32_e32 %[rB], v35\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v34, %[rB], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v34, %[nstride], v34\n\t"
        "v_lshl_add_u32 v34, %[cB], %[cs], v34\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v34, %[rB], %[stride], %[cB]\n\t"
        "v_lshl_add_u32 v34, v34, 2, %[k4]\n\t"
        ".endif\n\t"
        RM_LOAD("%[dB]", "v34")
        "L_march2_%=:\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        RM_DECODE("%[dA]")
        RM2_ADVANCE_A
        "s_mov_b64 exec, %[mB]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        RM_DECODE("%[dB]")
        RM2_ADVANCE_B
        "s_bcnt1_i32_b64 %[n], %[mA]\n\t"
        "s_bcnt1_i32_b64 %[n2], %[mB]\n\t"
        "s_add_u32 %[n], %[n], %[n2]\n\t"
        "s_cmp_gt_u32 %[n], %[low]\n\t"
        "s_cbranch_scc1 L_march2_%=\n\t"
        // the two samples in flight belong to live rays: consume them
        "s_mov_b64 exec, %[mA]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        RM_DECODE("%[dA]")
        "v_add_f32_e32 v20, v20, %[dA]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[mA], exec\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        RM_DECODE("%[dB]")
        "v_add_f32_e32 v28, v28, %[dB]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        "s_mov_b64 %[mB], exec\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [tA] "+{v20}"(tA), [cA] "+v"(cA), [rA] "+v"(rA), [dA] "+v"(dA), [tB] "+{v28}"(tB), [cB] "+v"(cB),
          [rB] "+v"(rB), [dB] "+v"(dB), [save] "=&s"(save), [mA] "=&s"(mA), [mB] "=&s"(mB), [n] "=&s"(n),
          [n2] "=&s"(n2)
        : [dyA] "{v22}"(dyA), [dxA] "{v23}"(dxA), [gxA] "{v24}"(gxA), [gyA] "{v25}"(gyA), [dyB] "{v30}"(dyB),
          [dxB] "{v31}"(dxB), [gxB] "{v32}"(gxB), [gyB] "{v33}"(gyB), [mx] "s"(max_range), [stride] "s"(stride),
          [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [lit] "n"(LIT ? 1 : 0), RM_CODE_OPERANDS, [low] "s"(low), [tiled] "n"(TILED ? 1 : 0)
        : "v26", "v27", "v34", "v35", "vcc", "scc", "memory");
}


// Three rays per lane: the same alternation over three live masks (slot C: t v36 / dir v[38:39] /
// origin v[40:41] / scratch v[42:43]).
// (Tried and dropped, no measurable change at cfg2 / 32 k poses: a drain-phase form that branches over
//  a slot whose rays have all finished instead of issuing its 9 VALU with EXEC = 0, and a 24-bit
//  multiply for the output index in the claim.)
template <bool TILED, bool LIT = false>
__device__ __forceinline__ void march_loop3(float dxA, float dyA, float gxA, float gyA, float &tA, int &cA, int &rA,
                                            float &dA, float dxB, float dyB, float gxB, float gyB, float &tB,
                                            int &cB, int &rB, float &dB, float dxC, float dyC, float gxC, float gyC,
                                            float &tC, int &cC, int &rC, float &dC, const float *pdt, int stride,
                                            int nstride, uint32_t k4, float max_range, uint32_t low)
{
    unsigned long long save, mA, mB, mC;
    uint32_t n, n2;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[mA], exec\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        "s_mov_b64 %[mB], exec\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v36\n\t"
        "s_mov_b64 %[mC], exec\n"
        // (decoupled slots, like march_loop2: a slot's next load goes out as soon as its own previous one is back)
        "s_mov_b64 exec, %[mA]\n\t"
        RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")
        "v_cvt_i32_f32_e32 %[cA], v26\n\t"
        "v_cvt_i32_f32_e32 %[rA], v27\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v26, %[rA], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v26, %[nstride], v26\n\t"
        "v_lshl_add_u32 v26, %[cA], 4, v26\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v26, %[rA], %[stride], %[cA]\n\t"
        "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[dA], v26, %[base]\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        RM_POS("v[34:35]", "v[30:31]", "v[28:29]", "v[32:33]")
        "v_cvt_i32_f32_e32 %[cB], v34\n\t"
        "v_cvt_i32_f32_e32 %[rB], v35\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v34, %[rB], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v34, %[nstride], v34\n\t"
        "v_lshl_add_u32 v34, %[cB], 4, v34\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v34, %[rB], %[stride], %[cB]\n\t"
        "v_lshl_add_u32 v34, v34, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[dB], v34, %[base]\n\t"
        "s_mov_b64 exec, %[mC]\n\t"
        RM_POS("v[42:43]", "v[38:39]", "v[36:37]", "v[40:41]")
        "v_cvt_i32_f32_e32 %[cC], v42\n\t"
        "v_cvt_i32_f32_e32 %[rC], v43\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v42, %[rC], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v42, %[nstride], v42\n\t"
        "v_lshl_add_u32 v42, %[cC], 4, v42\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v42, %[rC], %[stride], %[cC]\n\t"
        "v_lshl_add_u32 v42, v42, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[dC], v42, %[base]\n\t"
        "L_march3_%=:\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_add_f32_e32 v20, v20, %[dA]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[mA], exec\n\t"
        RM_POS("v[26:27]", "v[22:23]", "v[20:21]", "v[24:25]")
        "v_cvt_i32_f32_e32 %[cA], v26\n\t"
        "v_cvt_i32_f32_e32 %[rA], v27\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v26, %[rA], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v26, %[nstride], v26\n\t"
        "v_lshl_add_u32 v26, %[cA], 4, v26\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v26, %[rA], %[stride], %[cA]\n\t"
        "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[dA], v26, %[base]\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_add_f32_e32 v28, v28, %[dB]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        "s_mov_b64 %[mB], exec\n\t"
        RM_POS("v[34:35]", "v[30:31]", "v[28:29]", "v[32:33]")
        "v_cvt_i32_f32_e32 %[cB], v34\n\t"
        "v_cvt_i32_f32_e32 %[rB], v35\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v34, %[rB], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v34, %[nstride], v34\n\t"
        "v_lshl_add_u32 v34, %[cB], 4, v34\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v34, %[rB], %[stride], %[cB]\n\t"
        "v_lshl_add_u32 v34, v34, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[dB], v34, %[base]\n\t"
        "s_mov_b64 exec, %[mC]\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_add_f32_e32 v36, v36, %[dC]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v36\n\t"
        "s_mov_b64 %[mC], exec\n\t"
        RM_POS("v[42:43]", "v[38:39]", "v[36:37]", "v[40:41]")
        "v_cvt_i32_f32_e32 %[cC], v42\n\t"
        "v_cvt_i32_f32_e32 %[rC], v43\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v42, %[rC], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v42, %[nstride], v42\n\t"
        "v_lshl_add_u32 v42, %[cC], 4, v42\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v42, %[rC], %[stride], %[cC]\n\t"
        "v_lshl_add_u32 v42, v42, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[dC], v42, %[base]\n\t"
        "s_bcnt1_i32_b64 %[n], %[mA]\n\t"
        "s_bcnt1_i32_b64 %[n2], %[mB]\n\t"
        "s_add_u32 %[n], %[n], %[n2]\n\t"
        "s_bcnt1_i32_b64 %[n2], %[mC]\n\t"
        "s_add_u32 %[n], %[n], %[n2]\n\t"
        "s_cmp_gt_u32 %[n], %[low]\n\t"
        "s_cbranch_scc1 L_march3_%=\n\t"
        // the three samples in flight belong to live rays: consume them
        "s_mov_b64 exec, %[mA]\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_add_f32_e32 v20, v20, %[dA]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[mA], exec\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_add_f32_e32 v28, v28, %[dB]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        "s_mov_b64 %[mB], exec\n\t"
        "s_mov_b64 exec, %[mC]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_add_f32_e32 v36, v36, %[dC]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v36\n\t"
        "s_mov_b64 %[mC], exec\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [tA] "+{v20}"(tA), [cA] "+v"(cA), [rA] "+v"(rA), [dA] "+v"(dA), [tB] "+{v28}"(tB), [cB] "+v"(cB),
          [rB] "+v"(rB), [dB] "+v"(dB), [tC] "+{v36}"(tC), [cC] "+v"(cC), [rC] "+v"(rC), [dC] "+v"(dC),
          [save] "=&s"(save), [mA] "=&s"(mA), [mB] "=&s"(mB), [mC] "=&s"(mC), [n] "=&s"(n), [n2] "=&s"(n2)
        : [dyA] "{v22}"(dyA), [dxA] "{v23}"(dxA), [gxA] "{v24}"(gxA), [gyA] "{v25}"(gyA), [dyB] "{v30}"(dyB),
          [dxB] "{v31}"(dxB), [gxB] "{v32}"(gxB), [gyB] "{v33}"(gyB), [dyC] "{v38}"(dyC), [dxC] "{v39}"(dxC),
          [gxC] "{v40}"(gxC), [gyC] "{v41}"(gyC), [mx] "s"(max_range), [stride] "s"(stride),
          [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [lit] "n"(LIT ? 1 : 0), [low] "s"(low), [tiled] "n"(TILED ? 1 : 0)
        : "v26", "v27", "v34", "v35", "v42", "v43", "vcc", "scc", "memory");
}


struct PadMap {
    const float *pdt;        // padded step map (pad_dt_kernel / pad_dt_tiled_kernel); tiled: + (pad << 4) bytes,
                             //   the column bias of the address
    int stride, nstride, pad; // row-major: elements per row, 0; tiled: M = 4 + 2^(K-2), MASK = 0xC | (~0 << K)
    uint32_t k4;             // row-major: byte offset of map cell (0,0): (pad*stride + pad)*4; tiled: padM = (pad+4)*M
    FastDiv div_stride;
    float res;
};

// one ray in flight, as the hand-off passes it on: everything the march and the finish need (32 B, two 16-B stores)
struct __attribute__((aligned(16))) LeftoverRec {
    float gx, gy, dx, dy;    // grid origin, direction
    float t, d_last;         // where the march stands, the step that took it there
    uint32_t oidx;           // byte offset of the ray's range in `out`
    uint32_t pose;           // pose id (fused crash test)
};

struct StreamParams {
    const PoseRec *rec;      // sorted order
    const uint32_t *order;   // sorted slot -> pose index | POSE_INVALID
    const float *d0;         // sorted slot -> first step of the pose's rays (pose_first_step)
    const float2 *fan_tab;   // (cos, sin) of the num_rays beam angles, built once per (fov, num_rays)
    FastDiv div_B;           // division by num_rays
    int low_water;           // refill when <= low_water lanes are still marching
    int n_bands;
    const float *raw_poses;  // INLINE only: world poses (x, y, theta); every workgroup derives the
    const MapParams *map;    //   records of the chunks it owns itself (device copy of the map params)
    int k_max;               // INLINE only: LDS capacity in block records (BlockRec)
    uint32_t cpp;            // INLINE only: 64-ray blocks per pose, ceil(num_rays / 64) — blocks never straddle a pose
    FastDiv div_cpp;
    int drain_prio;          // raise wave priority once the workgroup's stream is exhausted
    int spec_drain;          // one ray per lane, stream exhausted: switch to the value-speculating loop (march_drain4)
                             //   once at most this many lanes are live (0 = never)
    int spec_stretch;        //   ... after this many samples of the plain loop, and again between two attempts
    int drain_cap;           // several rays per lane, stream dry: compact the wave's live rays into ONE slot once at most
                             //   this many are left (<= DRAIN_CAP)
    int drain_stretch;       //   ... and the plain stretch between two speculation attempts there
    int group_drain;         //   ... and from 32 / 16 live rays down 2 / 4 lanes per ray, 8 / 16 samples per round trip (march_drain_group)
    LiteralParams lit;       // LIT only (variant 3 in the stream kernel): rotation constant, sin / cos of the world angle
    int run_log2;            // a workgroup's stream interleaves RUNS of 2^run_log2 consecutive 64-ray blocks
    int stripe;              // INLINE only, where the band's pose ids come from: 0 = the caller's order
                             //   (band = index range), 1 = row stripes of the map compacted by every
                             //   workgroup itself (stripe_band_list), 2 = `order` (tile order from the
                             //   keys-only binning launch)
    int plain_store;         // 0: the ranges leave with non-temporal stores (range_store); 1: plain stores — for a caller
                             //   whose next kernel reads them back at once (FollowGap on the same stream: option nt_store 0)
    unsigned long long *dbg; // diagnostics (nullptr in production): 4 words per wave
    // several rays per lane, HAND-OFF (round 5): a wave whose workgroup's stream is dry does not drain its last long
    // rays in place — it writes them to `left_rec` (its own region of 2^left_cap_log2 records, no atomics), their
    // number to left_cnt[wave], and leaves; rm_leftover_kernel, the next launch on the stream, finishes them.  A
    // workgroup's slot is free ~2 us after its stream ran dry instead of ~14 us (its longest ray), so the slot goes
    // to the next workgroup — of the next launch in flight, or of this launch's next generation (grids beyond the
    // resident 2 workgroups per CU tile without a ragged end per generation).  nullptr: drain in place.
    LeftoverRec *left_rec;
    uint32_t *left_cnt;
    int left_cap_log2;
    // CODE launches: the map's palette (code_scan_kernel: exact float32 steps + the two stop codes), copied to LDS
    // behind the header by every workgroup
    const float *code_tab;
    int code_n;
    // TWO GENERATIONS of workgroups (round 6, lone whole-machine launches): the first tail_g1 workgroups of every band
    // — the resident generation — split the band's first (100 - tail_pct) % of runs as before; the workgroups behind
    // them split the rest and are dispatched as resident ones finish: the hardware's dispatcher is the work pool (a
    // claim costs nothing), the launch no longer ends with its most loaded resident workgroup.  0: one generation.
    int tail_g1, tail_pct;
};


// ------------------------------------------------------------------------------
// INLINE + stripe: XCD locality without a binning launch (512..8192 poses on maps larger than L2).
// Every workgroup ranks all P poses by (row bin of the pose, pose index) — 64 bins over the map's
// rows, an LDS histogram — and band b is ranks [P*b/nb, P*(b+1)/nb): a horizontal stripe of the
// map with exactly the pose count the contiguous split would give it (so the host's LDS sizing
// holds).  The workgroup then compacts the poses of ITS band, in pose-index order, into `list`:
// whole bins strictly inside the band, plus the first/last few poses of the two boundary bins
// (ordered counts by ballot + wave prefix).  All workgroups of a band compute the same list.
// Costs ~3 us per workgroup at 4096 poses instead of a ~9.5 us single-workgroup launch in front.
// ------------------------------------------------------------------------------
constexpr int STRIPE_BINS = 64;
constexpr int STRIPE_MAX_PER_LANE = 8;          // poses per lane of a 1024-thread workgroup: P <= 8192

__device__ __forceinline__ int stripe_row_bin(const MapParams &m, const float *__restrict__ poses, int p)
{
    float gx, gy, thg;
    world_to_grid(m, poses[3 * (size_t)p], poses[3 * (size_t)p + 1], 0.0f, gx, gy, thg);   // (heading not needed)
    const float u = gy * ((float)STRIPE_BINS / m.frows);
    return u >= 0.0f ? (u < (float)STRIPE_BINS ? (int)u : STRIPE_BINS - 1) : 0;     // NaN -> bin 0
}

// per-wave counts c[0..nw) in LDS (nw <= 64) -> sum of the waves before `wave`, and the total: one
// LDS read per lane and a shuffle scan instead of every lane walking the array
__device__ __forceinline__ void wave_counts_prefix(const int *c, int nw, int lane, int wave, int &pre, int &tot)
{
    const int v = lane < nw ? c[lane] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    tot = __shfl(incl, 63);
    pre = __shfl(incl - v, wave);
}

// scratch: STRIPE_BINS + 3*(NT/64) + 4 ints.  Returns the number of poses written to list
// (== hi_rank - lo_rank).  Ends with a __syncthreads().
template <int NT>
__device__ __forceinline__ uint32_t stripe_band_list(const MapParams &m, const float *__restrict__ poses,
                                                     int P, uint32_t lo_rank, uint32_t hi_rank,
                                                     uint32_t *__restrict__ list, int *__restrict__ scratch)
{
    constexpr int NW = NT / 64;
    int *hist = scratch, *wc = scratch + STRIPE_BINS, *meta = wc + 3 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < STRIPE_BINS) hist[tid] = 0;
    __syncthreads();
    // every pose is read once: up to STRIPE_MAX_PER_LANE row bins per lane stay in registers (all the
    // loads of a lane are in flight together — 512 workgroups read the same 48 KB at the same time)
    // (one byte per pose, 0xff = none: two registers, so that this prologue does not raise the
    //  kernel's VGPR count and cost the march its 8 waves per SIMD)
    unsigned long long packed = ~0ull;
#pragma unroll
    for (int u = 0; u < STRIPE_MAX_PER_LANE; ++u) {
        const int p = u * NT + tid;
        if (p < P) {
            const int b = stripe_row_bin(m, poses, p);
            packed = (packed & ~(0xffull << (8 * u))) | ((unsigned long long)b << (8 * u));
            atomicAdd(&hist[b], 1);
        }
    }
    __syncthreads();
    if (wave == 0) {                                   // 64 bins = one wave: inclusive scan by shuffles
        const int h = hist[lane];
        int incl = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        const int excl = incl - h;
        if (h > 0 && excl <= (int)lo_rank && (int)lo_rank < incl) { meta[0] = lane; meta[1] = (int)lo_rank - excl; }
        if (h > 0 && excl <= (int)hi_rank - 1 && (int)hi_rank - 1 < incl) { meta[2] = lane; meta[3] = (int)hi_rank - excl; }
    }
    __syncthreads();
    const int cl = meta[0], skip_lo = meta[1], ch = meta[2], take_hi = meta[3];
    int cnt_cl = 0, cnt_ch = 0;
    uint32_t npos = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 1
    for (int u = 0; u * NT < P; ++u) {
        const int p = u * NT + tid;
        const int byte = (int)((packed >> (8 * u)) & 0xffull);
        const int bin = byte == 0xff ? -1 : byte;
        const bool is_cl = bin == cl, is_ch = bin == ch && ch != cl;
        const unsigned long long b_cl = __ballot(is_cl), b_ch = __ballot(is_ch);
        if (lane == 0) wc[wave] = __popcll(b_cl) | (__popcll(b_ch) << 16);     // (both <= 64 per wave, sums <= 8192)
        __syncthreads();
        int pre, tot;
        wave_counts_prefix(wc, NW, lane, wave, pre, tot);
        const int pre_cl = pre & 0xffff, pre_ch = pre >> 16, tot_cl = tot & 0xffff, tot_ch = tot >> 16;
        const int idx_cl = cnt_cl + pre_cl + __popcll(b_cl & below);
        const int idx_ch = cnt_ch + pre_ch + __popcll(b_ch & below);
        const bool member = (bin > cl && bin < ch) ||
                            (is_cl && idx_cl >= skip_lo && (cl != ch || idx_cl < take_hi)) ||
                            (is_ch && idx_ch < take_hi);
        const unsigned long long b_m = __ballot(member);
        if (lane == 0) wc[2 * NW + wave] = __popcll(b_m);
        __syncthreads();
        int pre_m, tot_m;
        wave_counts_prefix(wc + 2 * NW, NW, lane, wave, pre_m, tot_m);
        if (member) list[npos + pre_m + __popcll(b_m & below)] = (uint32_t)p;
        cnt_cl += tot_cl;
        cnt_ch += tot_ch;
        npos += (uint32_t)tot_m;
        __syncthreads();
    }
    return npos;
}

// INLINE: everything a ray slot of a 64-ray block needs, in ONE 32-byte LDS record per owned block (two
// ds_read_b128).  Blocks of an INLINE launch never straddle a pose — a pose's beams are padded to a
// multiple of 64 (1081 beams: 7 idle slots in 1088, 0.65 %) — so block -> pose is one record, not the
// "which of two poses" decode of a dense ray stream (round 2: two 16-B records + pose id + first step +
// block word = 6 LDS reads and ~28 VALU per claim; now 3 reads and ~17).
struct __attribute__((aligned(32))) BlockRec {
    float gx, gy, ct, st;    // grid origin, cos / sin of the grid heading
    float d0;                // first step of the pose's rays (pose_first_step)
    uint32_t obase;          // BYTE offset of the block's first range in `out`: (pose * num_rays + j0) * 4
    uint32_t j0nv;           // first beam of the block | valid rays in it << 16
    uint32_t pose;           // pose id (fused crash test)
};

constexpr uint32_t NO_RAY = 0xffffffffu;       // output index of a slot that holds no ray

// Several rays per lane, stream dry: once at most DRAIN_CAP rays are live in a wave they are compacted into slot
// A (through DRAIN_FIELDS x DRAIN_CAP dwords of LDS per wave) and finished by the one-ray-per-lane drain loops
// (march_loop_capped / march_drain4).
constexpr int DRAIN_CAP = 64;                  // capacity; the threshold is StreamParams::drain_cap <= DRAIN_CAP
constexpr int DRAIN_FIELDS = 7;                // gx, gy, dx, dy, t, last step, output offset (+ 2 with the crash test)

template <bool AUX, bool CRASH, int NT, bool INLINE, bool TILED, int SLOTS = 1, bool LIT = false, int CODE = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8)))
void rm_fan_stream_kernel(PadMap pm, FanParams f, StreamParams sp, float *__restrict__ out,
                          int32_t *__restrict__ hits, uint16_t *__restrict__ steps, CrashParams cp)
{
    // CODE: pm describes the map of palette codes (pad_code_tiled_kernel), the palette sits in LDS behind the header;
    // every march loop of the two-rays-per-lane form decodes its samples through it (RM_DECODE)
    static_assert(CODE == 0 || (TILED && SLOTS == 2 && !AUX), "code map: tiled layout, two rays per lane, no diagnostics");
    // LIT: the upstream-literal arithmetic (variant 3) on this kernel's schedule — per-ray libm directions at claim time,
    // un-fused position (packed multiply + packed add in the march loops), un-fused hit range, records in (row, col)
    // naming: bit-identical to the checker's orc_rm_fan_libm.  Records come from the INLINE prologue only.
    static_assert(!LIT || (INLINE && TILED && !AUX), "literal stream form: INLINE records, tiled step map, no diagnostics");
    extern __shared__ __attribute__((aligned(32))) float lds_f[];
    const unsigned long long t_entry = sp.dbg ? wall_clock64() : 0ull;   // diagnostics
    uint32_t *q_next = reinterpret_cast<uint32_t *>(lds_f);     // shared slot counter
    // [2 .. STREAM_HDR): CRASH only — poses this workgroup has already reported as crashed
    // (direct-mapped): a pose scraping a wall crashes on hundreds of beams, all marched by this
    // workgroup, and only the first of them needs to touch the group's word in global memory
    uint32_t *crash_seen = reinterpret_cast<uint32_t *>(lds_f + 2);
    const size_t tabw = CODE ? ((size_t)sp.code_n + 1) & ~(size_t)1 : 0;     // palette words (kept even: float2 / double tables follow)
    float2 *fan_cs = reinterpret_cast<float2 *>(lds_f + STREAM_HDR + tabw);     // num_rays float2
    // CRASH: the car-outline table next to the fan table (read when a ray finishes: from LDS it does
    // not sit behind the range store in vmcnt — a global read there made every refill wait for the
    // store's acknowledgement and the kernel 2.5x slower)
    double *edge_l = reinterpret_cast<double *>(lds_f + STREAM_HDR + tabw + 2 * (size_t)f.num_rays);
    const size_t tables = STREAM_HDR + tabw + (CRASH ? 4 : 2) * (size_t)f.num_rays;
    // several rays per lane: per-wave compaction scratch of the drain phase (DRAIN_FIELDS x DRAIN_CAP dwords)
    constexpr int DRAIN_F = DRAIN_FIELDS + (CRASH ? 2 : 0);
    constexpr size_t DRAIN_WORDS = (SLOTS >= 2 && TILED) ? (size_t)(NT / 64) * DRAIN_F * DRAIN_CAP : 0;
    uint32_t *drain_scr = reinterpret_cast<uint32_t *>(lds_f + ((tables + 7) & ~(size_t)7)) +
                          (size_t)(threadIdx.x >> 6) * DRAIN_F * DRAIN_CAP;
    // INLINE: one BlockRec per owned block, filled below
    BlockRec *lrec = reinterpret_cast<BlockRec *>(lds_f + ((tables + 7) & ~(size_t)7) + DRAIN_WORDS);   // 32-B aligned
    if (threadIdx.x == 0) *q_next = 0;
    if (CRASH && threadIdx.x < STREAM_HDR - 2) crash_seen[threadIdx.x] = 0xffffffffu;
    // (the beam directions are the same for every workgroup of every launch with this fan: a table
    //  of the handle, fan_table_kernel — 1081 sincos per workgroup were 3 % of a cfg2 launch's VALU
    //  work and the first microsecond of every workgroup's life)
    for (int j = threadIdx.x; j < f.num_rays; j += NT) {
        fan_cs[j] = sp.fan_tab[j];
        if (CRASH) edge_l[j] = cp.edge[j];
    }
    if (CODE) {
        // (RM_DECODE names the table's LDS address as an immediate: the dynamic LDS block starts at 0 — this kernel has
        //  no static LDS —, anything else would be a build that silently reads the wrong table)
        if ((uint32_t)(uintptr_t)lds_f != 0u) __builtin_trap();
        for (int j = threadIdx.x; j < sp.code_n; j += NT) lds_f[STREAM_HDR + j] = sp.code_tab[j];
    }

    // ---- which band of the sorted pose list, and which workgroups share it
    const int nb = sp.n_bands;
    const int band = (int)(blockIdx.x % (unsigned)nb);
    const uint32_t g = blockIdx.x / (unsigned)nb;
    const uint32_t G = ((uint32_t)gridDim.x - (uint32_t)band + (uint32_t)nb - 1) / (uint32_t)nb;
    const uint32_t seg_lo = (uint32_t)(((long)f.n_poses * band) / nb);
    const uint32_t seg_hi = (uint32_t)(((long)f.n_poses * (band + 1)) / nb);
    // the band's rays in blocks of 64: this workgroup owns blocks g, g+G, ... (in runs) — K blocks, 64*K ray
    // slots.  Binned records: the band's rays pose-major, beam-minor, cut every 64 (any num_rays, no
    // padding lanes).  INLINE: cpp blocks per pose, the last one partly filled.
    const uint32_t seg_rays = (seg_hi - seg_lo) * (uint32_t)f.num_rays;
    const uint32_t seg_chunks = INLINE ? (seg_hi - seg_lo) * sp.cpp : (seg_rays + 63u) >> 6;
    const uint32_t rl = (uint32_t)sp.run_log2, rmask = (1u << rl) - 1u;
    const uint32_t seg_runs = (seg_chunks + rmask) >> rl;
    // (runs own_first, own_first + own_stride, ... below own_limit are this workgroup's)
    uint32_t own_first = g, own_stride = G, own_limit = seg_runs;
    if (sp.tail_g1 > 0 && G > (uint32_t)sp.tail_g1) {
        const uint32_t G1 = (uint32_t)sp.tail_g1, R1 = seg_runs - (seg_runs * (uint32_t)sp.tail_pct) / 100u;
        if (g < G1) {
            own_stride = G1;
            own_limit = R1;
        } else {
            own_first = R1 + (g - G1);
            own_stride = G - G1;
        }
    }
    const uint32_t K = (own_first < own_limit ? (own_limit - own_first + own_stride - 1) / own_stride : 0) << rl;
    const uint32_t total = K << 6;
    // i-th block of this workgroup's stream -> its index in the band / first ray of the block
    auto blkidx_of = [&](uint32_t i) { return ((own_first + (i >> rl) * own_stride) << rl) + (i & rmask); };
    auto blk_of = [&](uint32_t i) { return blkidx_of(i) << 6; };
    const unsigned lane = threadIdx.x & 63;
    if (INLINE) {
        // no binning launch (or a keys-only one) in front of the march — each workgroup turns the poses
        // of its own blocks into records (a few hundred, one per lane) and keeps them in LDS
        const MapParams mp = *sp.map;
        // stripe mode: this band's poses (a row stripe of the map) compacted here, in LDS
        uint32_t *list = reinterpret_cast<uint32_t *>(lrec + sp.k_max);
        if (sp.stripe == 1 && seg_hi > seg_lo)
            stripe_band_list<NT>(mp, sp.raw_poses, f.n_poses, seg_lo, seg_hi, list,
                                 reinterpret_cast<int *>(list + (seg_hi - seg_lo) + 1));
        for (uint32_t i = threadIdx.x; i < K; i += NT) {
            const uint32_t b = blkidx_of(i);
            BlockRec br{0.0f, 0.0f, 1.0f, 0.0f, PDT_NO_RAY, 0u, 0u, 0u};
            if (b < seg_chunks) {
                const uint32_t p0 = fast_div(b, sp.div_cpp);
                const uint32_t j0 = (b - p0 * sp.cpp) << 6;
                const uint32_t nvalid = min(64u, (uint32_t)f.num_rays - j0);
                const uint32_t pid = sp.stripe == 1 ? list[p0]
                                   : sp.stripe == 2 ? (sp.order[seg_lo + p0] & ~POSE_INVALID) : seg_lo + p0;
                PoseRec r;
                const uint32_t kf = LIT ? pose_record_lit(mp, sp.lit, sp.raw_poses, (int)pid, r)
                                        : pose_record(mp, sp.raw_poses, (int)pid, 0, 1, 1, r);
                br.gx = r.gx; br.gy = r.gy; br.ct = r.ct; br.st = r.st;
                br.d0 = pose_first_step(mp, r.gx, r.gy, kf, f.step_coeff);
                br.obase = (pid * (uint32_t)f.num_rays + j0) << 2;
                br.j0nv = j0 | (nvalid << 16);
                br.pose = pid;
            }
            lrec[i] = br;
        }
    }
    __syncthreads();
    const float INF = __builtin_inff();

    unsigned long long t_start = 0, t_drain = 0;   // diagnostics (sp.dbg): launch / stream-exhausted stamps
    uint32_t n_serv = 0, ns_drain = 0, drain_samples = 0;
    if (sp.dbg) t_start = wall_clock64();

    // one ray slot of a lane.  oidx (byte offset of the range in `out`) == NO_RAY: the slot holds no ray
    // (nothing to store when it is "finished")
    struct Slot {
        float gx, gy, dx, dy, t, d_last;
        int pc, pr;
        uint32_t oidx;
        uint32_t pose;         // CRASH only
        int jbeam;             // CRASH only
    };
    // ray slot q of this workgroup's stream -> the lane's slot state; false: a padding slot (no ray).
    // s.oidx is the BYTE offset of the ray's range in `out` (the store needs no shift).
    auto claim = [&](Slot &s, uint32_t q) -> bool {
        if (INLINE) {
            // everything is read before validity is known (one LDS round trip, not two): a padding slot of
            // a pose's last block becomes a slot without a ray — t past max_range, oidx NO_RAY — whose other
            // fields are never looked at (its beam index may point past the fan table: LDS reads are harmless)
            const uint4 *rp = reinterpret_cast<const uint4 *>(lrec + (q >> 6));
            const uint4 ra = rp[0], rb = rp[1];
            const uint32_t l = q & 63u;
            const bool valid = l < (rb.z >> 16);
            const uint32_t j = (rb.z & 0xffffu) + l;
            s.gx = __builtin_bit_cast(float, ra.x);
            s.gy = __builtin_bit_cast(float, ra.y);
            if (LIT) {
                // beam j of the pose at theta_p + (-fov/2 + j * (fov / B)), the product and both sums rounded to float32;
                // theta' = -theta + (-world_angle - 3 pi / 2); calc_range(y, x, theta') marches rows along cosf, columns
                // along sinf (glibc's algorithm, literal_math.h)
                const float aj = (float)j * f.inc;
                const float th = __builtin_bit_cast(float, ra.z) + (f.amin + aj);
                const float thp = -th + sp.lit.rotation_const;
                s.dy = lit_cosf(thp);
                s.dx = lit_sinf(thp);
            } else {
                const float2 cs = fan_cs[j];
                const float ct = __builtin_bit_cast(float, ra.z), st = __builtin_bit_cast(float, ra.w);
                s.dx = __builtin_fmaf(ct, cs.x, -(st * cs.y));
                s.dy = __builtin_fmaf(st, cs.x, ct * cs.y);
            }
            s.oidx = valid ? rb.y + (l << 2) : NO_RAY;
            if (CRASH) {
                s.pose = rb.w;
                s.jbeam = (int)j;
            }
            // the sample at t = 0 was taken with the pose record (pose_first_step)
            s.d_last = __builtin_bit_cast(float, rb.x);
            s.t = valid ? s.d_last : __builtin_inff();
            return valid;
        }
        const uint32_t ray = blk_of(q >> 6) + (q & 63);
        if (ray >= seg_rays) return false;
        const uint32_t spose = fast_div(ray, sp.div_B);
        const int j = (int)(ray - spose * (uint32_t)f.num_rays);
        // SGPR base + 32-bit lane offset (global_load ... s[base]) instead of 64-bit per-lane pointers
        const uint32_t si = seg_lo + spose;
        const uint32_t po = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(sp.order) + (si << 2));
        const PoseRec pr_ = *reinterpret_cast<const PoseRec *>(reinterpret_cast<const char *>(sp.rec) + (si << 4));
        const float2 cs = fan_cs[j];
        s.gx = pr_.gx;
        s.gy = pr_.gy;
        s.dx = __builtin_fmaf(pr_.ct, cs.x, -(pr_.st * cs.y));
        s.dy = __builtin_fmaf(pr_.st, cs.x, pr_.ct * cs.y);
        s.oidx = ((po & ~POSE_INVALID) * (uint32_t)f.num_rays + (uint32_t)j) << 2;
        if (CRASH) {
            s.pose = po & ~POSE_INVALID;
            s.jbeam = j;
        }
        s.t = s.d_last = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sp.d0) + (si << 2));
        return true;
    };
    auto crash_test = [&](const Slot &s, float r) {
        if (((double)r - edge_l[s.jbeam]) < cp.thresh) {
            uint32_t *seen = &crash_seen[s.pose & (STREAM_HDR - 3)];
            if (*seen != s.pose) {            // (a race only costs a redundant atomic)
                *seen = s.pose;
                crash_note(cp, s.pose);
            }
        }
    };

    if constexpr (SLOTS >= 2) {
        // ---------------- two (three) rays per lane (ranges, optionally the fused crash test; no diagnostics; tiled step map)
        static_assert(!(SLOTS >= 2) || !AUX, "multi-slot form: ranges (+ crash test), no diagnostics");
        Slot sa{0.f, 0.f, 0.f, 0.f, INF, 1.0f, 0, 0, NO_RAY, 0u, 0}, sb = sa, sc = sa;
        bool exhausted = total == 0;
        uint32_t left_n = 0;              // rays this wave hands to rm_leftover_kernel
        auto finish = [&](Slot &s) {
            float r = f.max_range;
            if (s.d_last == PDT_HIT) {
                const float xd = (float)s.pc - s.gx, yd = (float)s.pr - s.gy;
                if (LIT) {                       // sqrtf(xd * xd + yd * yd): every product and the sum its own rounding
                    const float xx = xd * xd, yy = yd * yd;
                    r = hit_sqrtf(yy + xx);
                } else {
                    r = hit_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                }
            }
            r *= pm.res;
            if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (s.oidx >> 2));
            if (out) range_store(out, s.oidx, r, sp.plain_store);
            if (CRASH) crash_test(s, r);
            s.oidx = NO_RAY;
        };
        for (;;) {
            const unsigned long long idle_a = __ballot(!(sa.t < f.max_range));
            const unsigned long long idle_b = __ballot(!(sb.t < f.max_range));
            const unsigned long long idle_c = SLOTS == 3 ? __ballot(!(sc.t < f.max_range)) : 0ull;
            if (idle_a | idle_b | idle_c) {
                const bool mine_a = !(sa.t < f.max_range), mine_b = !(sb.t < f.max_range);
                const bool mine_c = SLOTS == 3 && !(sc.t < f.max_range);
                if (mine_a && sa.oidx != NO_RAY) finish(sa);
                if (mine_b && sb.oidx != NO_RAY) finish(sb);
                if (SLOTS == 3 && mine_c && sc.oidx != NO_RAY) finish(sc);
                if (!exhausted) {                     // wave-uniform
                    const uint32_t cnt_a = (uint32_t)__popcll(idle_a), cnt_b = (uint32_t)__popcll(idle_b);
                    const uint32_t cnt = cnt_a + cnt_b + (uint32_t)__popcll(idle_c);
                    uint32_t qb = 0;
                    if (lane == 0) qb = atomicAdd(q_next, cnt);
                    qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
                    exhausted = qb + cnt >= total;
                    // slot-A lanes take the first cnt_a slots in lane order, then slot B's, then slot C's
                    const uint32_t qa = qb + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_a >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)idle_a, 0u));
                    const uint32_t qbb = qb + cnt_a + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_b >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)idle_b, 0u));
                    if (mine_a && qa < total) claim(sa, qa);
                    if (mine_b && qbb < total) claim(sb, qbb);
                    if (SLOTS == 3) {
                        const uint32_t qc = qb + cnt_a + cnt_b + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_c >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)idle_c, 0u));
                        if (mine_c && qc < total) claim(sc, qc);
                    }
                }
            }
            if (exhausted && !__ballot(sa.t < f.max_range) && !__ballot(sb.t < f.max_range) &&
                !__ballot(sa.oidx != NO_RAY) && !__ballot(sb.oidx != NO_RAY) &&
                (SLOTS < 3 || (!__ballot(sc.t < f.max_range) && !__ballot(sc.oidx != NO_RAY))))
                break;
            if (sp.dbg && exhausted && !t_drain) t_drain = wall_clock64();
            if constexpr (TILED) {
                if (exhausted && sp.spec_drain > 0) {
                    // drain phase.  (Every idle slot has been finished by the service above: what is live below is
                    // all this wave still owes.)
                    const unsigned long long la = __ballot(sa.t < f.max_range), lb = __ballot(sb.t < f.max_range);
                    const unsigned long long lc = SLOTS == 3 ? __ballot(sc.t < f.max_range) : 0ull;
                    const uint32_t na = (uint32_t)__popcll(la), nb2 = (uint32_t)__popcll(lb), nc = (uint32_t)__popcll(lc);
                    const uint32_t nlive = na + nb2 + nc;
                    const uint32_t cap = (uint32_t)__builtin_amdgcn_readfirstlane(
                        sp.left_rec ? min(max(sp.drain_cap, 1), 1 << sp.left_cap_log2) : min(max(sp.drain_cap, 1), DRAIN_CAP));
                    if (nlive > cap) {
                        // the plain loop until few rays are left
                        if (SLOTS == 3)
                            march_loop3<TILED, LIT>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx,
                                        sb.gy, sb.t, sb.pc, sb.pr, sb.d_last, sc.dx, sc.dy, sc.gx, sc.gy, sc.t, sc.pc, sc.pr,
                                        sc.d_last, pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range, cap);
                        else
                            march_loop2<TILED, LIT, CODE>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx,
                                        sb.gy, sb.t, sb.pc, sb.pr, sb.d_last, pm.pdt, pm.stride, pm.nstride, pm.k4,
                                        f.max_range, cap);
                        continue;
                    }
                    if (nlive > 0) {
                        // rays claimed a moment ago that were born finished (pose outside the map: a miss without a
                        // sample) still wait for their store: do it before their slots are recycled
                        if (!(sa.t < f.max_range) && sa.oidx != NO_RAY) finish(sa);
                        if (!(sb.t < f.max_range) && sb.oidx != NO_RAY) finish(sb);
                        if (SLOTS == 3 && !(sc.t < f.max_range) && sc.oidx != NO_RAY) finish(sc);
                        if (sp.left_rec) {
                            // HAND-OFF: the live rays of every slot go to this wave's region of the leftover list
                            // (ranks 0 .. nlive-1) and the wave leaves; rm_leftover_kernel finishes them
                            const size_t gwv = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
                            LeftoverRec *dst = sp.left_rec + (gwv << sp.left_cap_log2);
                            auto hand = [&](const Slot &s, uint32_t r) {
                                uint4 *q = reinterpret_cast<uint4 *>(dst + r);
                                q[0] = make_uint4(__builtin_bit_cast(uint32_t, s.gx), __builtin_bit_cast(uint32_t, s.gy),
                                                  __builtin_bit_cast(uint32_t, s.dx), __builtin_bit_cast(uint32_t, s.dy));
                                q[1] = make_uint4(__builtin_bit_cast(uint32_t, s.t), __builtin_bit_cast(uint32_t, s.d_last),
                                                  s.oidx, CRASH ? s.pose : 0u);
                            };
                            if (sa.t < f.max_range)
                                hand(sa, __builtin_amdgcn_mbcnt_hi((uint32_t)(la >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)la, 0u)));
                            if (sb.t < f.max_range)
                                hand(sb, na + __builtin_amdgcn_mbcnt_hi((uint32_t)(lb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lb, 0u)));
                            if (SLOTS == 3 && sc.t < f.max_range)
                                hand(sc, na + nb2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(lc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lc, 0u)));
                            left_n = nlive;
                            break;
                        }
                        // compact the live rays of every slot into slot A, lanes 0 .. nlive-1, through LDS
                        auto put = [&](const Slot &s, uint32_t r) {
                            drain_scr[0 * DRAIN_CAP + r] = __builtin_bit_cast(uint32_t, s.gx);
                            drain_scr[1 * DRAIN_CAP + r] = __builtin_bit_cast(uint32_t, s.gy);
                            drain_scr[2 * DRAIN_CAP + r] = __builtin_bit_cast(uint32_t, s.dx);
                            drain_scr[3 * DRAIN_CAP + r] = __builtin_bit_cast(uint32_t, s.dy);
                            drain_scr[4 * DRAIN_CAP + r] = __builtin_bit_cast(uint32_t, s.t);
                            drain_scr[5 * DRAIN_CAP + r] = __builtin_bit_cast(uint32_t, s.d_last);
                            drain_scr[6 * DRAIN_CAP + r] = s.oidx;
                            if (CRASH) {
                                drain_scr[7 * DRAIN_CAP + r] = s.pose;
                                drain_scr[8 * DRAIN_CAP + r] = (uint32_t)s.jbeam;
                            }
                        };
                        if (sa.t < f.max_range)
                            put(sa, __builtin_amdgcn_mbcnt_hi((uint32_t)(la >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)la, 0u)));
                        if (sb.t < f.max_range)
                            put(sb, na + __builtin_amdgcn_mbcnt_hi((uint32_t)(lb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lb, 0u)));
                        if (SLOTS == 3 && sc.t < f.max_range)
                            put(sc, na + nb2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(lc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lc, 0u)));
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: LDS operations complete in order)
                        sa.t = sb.t = INF;
                        sa.oidx = sb.oidx = NO_RAY;
                        if (SLOTS == 3) { sc.t = INF; sc.oidx = NO_RAY; }
                        if (lane < nlive) {
                            sa.gx = __builtin_bit_cast(float, drain_scr[0 * DRAIN_CAP + lane]);
                            sa.gy = __builtin_bit_cast(float, drain_scr[1 * DRAIN_CAP + lane]);
                            sa.dx = __builtin_bit_cast(float, drain_scr[2 * DRAIN_CAP + lane]);
                            sa.dy = __builtin_bit_cast(float, drain_scr[3 * DRAIN_CAP + lane]);
                            sa.t = __builtin_bit_cast(float, drain_scr[4 * DRAIN_CAP + lane]);
                            sa.d_last = __builtin_bit_cast(float, drain_scr[5 * DRAIN_CAP + lane]);
                            sa.oidx = drain_scr[6 * DRAIN_CAP + lane];
                            if (CRASH) {
                                sa.pose = drain_scr[7 * DRAIN_CAP + lane];
                                sa.jbeam = (int)drain_scr[8 * DRAIN_CAP + lane];
                            }
                        }
                        // ... and finish them with the one-ray-per-lane drain loops (value speculation on the step).
                        // Nothing of slots B / C is needed any more: the wave leaves from here (the drain loops use
                        // the registers of those slots as scratch).
                        // From 32 / 16 live rays down the wave has lanes to spare: the rays are laid out 2 / 4 lanes per ray
                        // and marched 8 / 16 samples per round trip (march_drain_group) — the launch ends with its longest
                        // chain, and a chain of 100 samples is ~39 round trips 4 deep, ~14 at 16 deep.
                        uint32_t L = 1;                       // lanes per ray (every lane of a group holds the ray's state;
                                                              //  only the first carries the output offset)
                        for (;;) {
                            const unsigned long long lv = __ballot(sa.t < f.max_range);
                            if (!lv) break;
                            const uint32_t nl = (uint32_t)__popcll(lv) >> (L >> 1);        // (L = 1, 2, 4: a shift, not a division)
                            // (group_drain = N: 4 lanes per ray from N live rays down, 2 lanes per ray from 2 N; N <= 16)
                            const uint32_t gdn = (uint32_t)sp.group_drain;
                            const uint32_t want = gdn ? (nl <= gdn ? 4u : (nl <= 2u * gdn ? 2u : 1u)) : 1u;
                            if (want > L) {
                                // rays that have finished leave first; the live ones are re-ranked through LDS
                                if (!(sa.t < f.max_range) && sa.oidx != NO_RAY) finish(sa);
                                const bool lead = sa.t < f.max_range && (lane & (L - 1u)) == 0u;
                                const unsigned long long lb2 = __ballot(lead);
                                if (lead)
                                    put(sa, __builtin_amdgcn_mbcnt_hi((uint32_t)(lb2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lb2, 0u)));
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                L = want;
                                const uint32_t ri = lane >> (L >> 1);
                                sa.t = INF;
                                sa.oidx = NO_RAY;
                                if (ri < nl) {
                                    sa.gx = __builtin_bit_cast(float, drain_scr[0 * DRAIN_CAP + ri]);
                                    sa.gy = __builtin_bit_cast(float, drain_scr[1 * DRAIN_CAP + ri]);
                                    sa.dx = __builtin_bit_cast(float, drain_scr[2 * DRAIN_CAP + ri]);
                                    sa.dy = __builtin_bit_cast(float, drain_scr[3 * DRAIN_CAP + ri]);
                                    sa.t = __builtin_bit_cast(float, drain_scr[4 * DRAIN_CAP + ri]);
                                    sa.d_last = __builtin_bit_cast(float, drain_scr[5 * DRAIN_CAP + ri]);
                                    if ((lane & (L - 1u)) == 0u) sa.oidx = drain_scr[6 * DRAIN_CAP + ri];
                                    if (CRASH) {
                                        sa.pose = drain_scr[7 * DRAIN_CAP + ri];
                                        sa.jbeam = (int)drain_scr[8 * DRAIN_CAP + ri];
                                    }
                                }
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (reads done before the next re-ranking writes)
                                continue;
                            }
                            if (L == 1u) {
                                march_loop_capped<TILED, LIT, CODE>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, pm.pdt,
                                                         pm.stride, pm.nstride, pm.k4, f.max_range, (uint32_t)sp.drain_stretch);
                                march_drain4<TILED, LIT, CODE>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, pm.pdt, pm.stride,
                                                    pm.nstride, pm.k4, f.max_range);
                            } else {
                                // (2 lanes per ray: until 16 rays are left — 32 lanes —, then 4 lanes per ray to the end)
                                march_drain_group<TILED, LIT, CODE>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, pm.pdt,
                                                         pm.stride, pm.nstride, pm.k4, f.max_range, L - 1u,
                                                         L == 2u ? 2u * gdn : 0u);
                            }
                        }
                        if (sa.oidx != NO_RAY) finish(sa);
                        break;
                    }
                }
            }
            if (SLOTS == 3)
                march_loop3<TILED, LIT>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx, sb.gy,
                            sb.t, sb.pc, sb.pr, sb.d_last, sc.dx, sc.dy, sc.gx, sc.gy, sc.t, sc.pc, sc.pr, sc.d_last,
                            pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range,
                            exhausted ? 0u : 3u * (uint32_t)sp.low_water);
            else
                march_loop2<TILED, LIT, CODE>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx, sb.gy,
                            sb.t, sb.pc, sb.pr, sb.d_last, pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range,
                            exhausted ? 0u : 2u * (uint32_t)sp.low_water);
        }
        // (every wave of the grid writes its count, 0 included: the list needs no clearing between launches)
        if (sp.left_cnt && lane == 0) sp.left_cnt[(size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)] = left_n;
        if (sp.dbg && lane == 0) {
            // diagnostics of the several-rays-per-lane form: absolute stamps {kernel entry, wave end, prologue done,
            // stream dry (0: never marched after exhaustion)} — tools/gpu_stamps_pipe.py
            const size_t gw = ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 4;
            sp.dbg[gw] = t_entry;
            sp.dbg[gw + 1] = wall_clock64();
            sp.dbg[gw + 2] = t_start;
            sp.dbg[gw + 3] = t_drain;
        }
        return;
    }

    bool exhausted = total == 0;
    Slot s1{0.f, 0.f, 0.f, 0.f, INF, 1.0f, 0, 0, NO_RAY, 0u, 0};
    // (s1.t < max_range  <=>  the lane is marching; d_last: PDT_HIT, PDT_OUTSIDE, or the free cell's step)
    uint32_t nstep = 0;

    for (;;) {
        // ---------------- service: finish pending rays, claim new slots
        const unsigned long long idle = __ballot(!(s1.t < f.max_range));
        if (idle) {
            if (sp.dbg) ++n_serv;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const bool mine = !(s1.t < f.max_range);
            if (mine && s1.oidx != NO_RAY) {
                const uint32_t oidx = s1.oidx >> 2;
                float r = f.max_range;
                int hc = -1, hr = -1;
                if (s1.d_last == PDT_HIT) {
                    hc = s1.pc;
                    hr = s1.pr;
                    const float xd = (float)hc - s1.gx, yd = (float)hr - s1.gy;
                    if (LIT) {
                        const float xx = xd * xd, yy = yd * yd;
                        r = hit_sqrtf(yy + xx);
                    } else {
                        r = hit_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                    }
                }
                r *= pm.res;
                if (f.noise_std > 0.0f)
                    r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + oidx);
                if (out) range_store(out, s1.oidx, r, sp.plain_store);
                if (AUX) {
                    if (sp.dbg && t_drain && nstep - ns_drain > drain_samples) drain_samples = nstep - ns_drain;
                    if (hits) { hits[2 * (size_t)oidx] = hc; hits[2 * (size_t)oidx + 1] = hr; }
                    // the read that found the border is not a map sample (the CPU statement
                    // leaves the loop before reading)
                    if (s1.d_last == PDT_OUTSIDE) --nstep;
                    if (steps) steps[oidx] = (uint16_t)(nstep > 65535u ? 65535u : nstep);
                }
                if (CRASH) crash_test(s1, r);
                s1.oidx = NO_RAY;
            }
            if (!exhausted) {                         // wave-uniform
                const uint32_t cnt = (uint32_t)__popcll(idle);
                uint32_t qb = 0;
                if (lane == 0) qb = atomicAdd(q_next, cnt);
                qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
                exhausted = qb + cnt >= total;
                const uint32_t q = qb + rank;
                if (mine && q < total) {
                    const bool got = claim(s1, q);
                    // (branch-free: a branch on `got` would split the claim's LDS reads into dependent trips)
                    if (AUX) nstep = got ? ((s1.t > 0.0f && s1.t < PDT_NO_RAY) ? 1u : 0u) : nstep;
                }
            }
        }
        // (no live lane and nothing left: done.  No live lane but slots left — every claimed ray was
        //  born finished, e.g. poses outside the map — falls through: the march loop below leaves at
        //  once when EXEC is empty, and keeping it unconditional keeps the ray state in place: a branch
        //  around the asm block made the compiler copy t / cell / step registers in and out of it,
        //  15 v_mov per service round)
        if (exhausted && !__ballot(s1.t < f.max_range) && !__ballot(s1.oidx != NO_RAY)) break;
        // ---------------- march while enough lanes are live (or nothing is left to claim)
        // a wave that can no longer refill is on the launch's critical path (its longest ray
        // decides when the kernel ends): let it win issue arbitration against refilling waves
        if (exhausted && sp.drain_prio) __builtin_amdgcn_s_setprio(3);
        if (sp.dbg && exhausted && !t_drain) {              // drain phase starts: samples so far per lane
            t_drain = wall_clock64();
            ns_drain = nstep;
        }
        if constexpr (!AUX && TILED) {
            // drain phase: the plain loop while more than a handful of lanes are live, then the
            // value-speculating loop for the last long rays (march_drain4)
            if (exhausted && sp.spec_drain > 0) {
                march_loop<AUX, TILED, LIT>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, nstep, pm.pdt,
                                       pm.stride, pm.nstride, pm.k4, f.max_range, (uint32_t)sp.spec_drain);
                // what is still marching after a stretch of the plain loop is a long chain: speculate on it
                // while that pays, fall back to the plain loop for a stretch when it does not
                while (__ballot(s1.t < f.max_range)) {
                    march_loop_capped<TILED, LIT>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, pm.pdt,
                                             pm.stride, pm.nstride, pm.k4, f.max_range, (uint32_t)sp.spec_stretch);
                    march_drain4<TILED, LIT>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, pm.pdt, pm.stride,
                                        pm.nstride, pm.k4, f.max_range);
                }
                continue;
            }
        }
        march_loop<AUX, TILED, LIT>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, nstep, pm.pdt, pm.stride,
                               pm.nstride, pm.k4, f.max_range, exhausted ? 0u : (uint32_t)sp.low_water);
    }
    uint32_t ds_max = 0;
    if (AUX && sp.dbg) {                                   // longest chain of samples marched after exhaustion
        ds_max = drain_samples;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ds_max = max(ds_max, (uint32_t)__shfl_xor((int)ds_max, off));
    }
    if (sp.dbg && lane == 0) {
        const size_t gw = ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 4;
        sp.dbg[gw] = t_start;
        sp.dbg[gw + 1] = wall_clock64();
        sp.dbg[gw + 2] = ((unsigned long long)n_serv << 32) | ds_max;
        sp.dbg[gw + 3] = ((unsigned long long)(uint32_t)(t_drain ? t_drain - t_start : 0) << 32) |
                         ((unsigned long long)(K & 0xffffffu) << 8) | (uint32_t)(band & 0xff);
    }
}

// ------------------------------------------------------------------------------
// rm_leftover_kernel: the second launch of a hand-off march (StreamParams::left_rec).  Leftover wave w gathers the
// records of 64 >> cap_log2 consecutive source waves (2^cap_log2 record slots each, left_cnt[] of them filled) —
// one ray per lane — and finishes them with the one-ray-per-lane drain loops: a bounded stretch of the plain loop,
// then the value-speculating loop (march_drain4), alternating.  Same arithmetic, same results: the t sequence of a
// ray does not depend on which wave adds its steps.  No LDS, no workgroup-level state: 64..256-lane workgroups that
// fit between the main kernels of the other streams.
// ------------------------------------------------------------------------------
template <bool CRASH>
__global__ __launch_bounds__(256) void rm_leftover_kernel(PadMap pm, FanParams f, const LeftoverRec *__restrict__ rec,
                                                          const uint32_t *__restrict__ cnt, int n_src, int cap_log2,
                                                          int stretch, int plain_store, float *__restrict__ out,
                                                          CrashParams cp)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t per_log2 = 6u - (uint32_t)cap_log2;
    const uint32_t src = (w << per_log2) + (lane >> cap_log2), e = lane & ((1u << cap_log2) - 1u);
    const bool has = src < (uint32_t)n_src && e < cnt[src];
    float gx = 0.f, gy = 0.f, dx = 0.f, dy = 0.f, t = __builtin_inff(), d_last = 1.0f;
    uint32_t oidx = NO_RAY, pose = 0;
    int pc = 0, pr = 0;
    if (has) {
        const uint4 *q = reinterpret_cast<const uint4 *>(rec + (((size_t)src << cap_log2) + e));
        const uint4 a = q[0], b = q[1];
        gx = __builtin_bit_cast(float, a.x); gy = __builtin_bit_cast(float, a.y);
        dx = __builtin_bit_cast(float, a.z); dy = __builtin_bit_cast(float, a.w);
        t = __builtin_bit_cast(float, b.x); d_last = __builtin_bit_cast(float, b.y);
        oidx = b.z; pose = b.w;
    }
    if (!__ballot(has)) return;
    while (__ballot(t < f.max_range)) {
        march_loop_capped<true>(dx, dy, gx, gy, t, pc, pr, d_last, pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range,
                                (uint32_t)stretch);
        march_drain4<true>(dx, dy, gx, gy, t, pc, pr, d_last, pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range);
    }
    if (has) {
        float r = f.max_range;
        if (d_last == PDT_HIT) {
            const float xd = (float)pc - gx, yd = (float)pr - gy;
            r = hit_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
        }
        r *= pm.res;
        if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (oidx >> 2));
        if (out) range_store(out, oidx, r, plain_store);
        if (CRASH) {
            // Car::isCrashed racecar/src/racecar.cpp:320 on this ray (beam index from the output offset)
            const uint32_t jbeam = (oidx >> 2) - pose * (uint32_t)f.num_rays;
            if (((double)r - cp.edge[jbeam]) < cp.thresh) crash_note(cp, pose);
        }
    }
}

}  // namespace scan
