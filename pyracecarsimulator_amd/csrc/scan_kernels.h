// scan_kernels.h — gfx950 kernels of the lidar scan path.
//
//   K0  edt_*            exact Euclidean distance transform of the occupancy grid
//                        (range_libc DistanceTransform)
//   K1  rm_fan_kernel    fan-expanding sphere tracing on the float32 EDT
//                        (RayMarching / RayMarchingGPU, rows a8-a11); bit-exact
//       rm_rays_kernel   one (x,y,theta) row per ray (upstream 2-arg API)
//
// Work decomposition of K1 (wave64-first, not a warp-shaped port of kernels.cu's
// thread-per-ray 1024x256 grid): the unit of work is a CHUNK = 64 consecutive
// beams of one pose, owned by one wavefront, so the 64 lanes of a wave march 64
// neighbouring beams (angular spacing fov/num_rays ~ 0.25 deg): their samples
// fall in the same few EDT cache lines, their step counts are similar (less
// divergence), and the wave leaves the march loop as soon as every lane has hit
// or left the map (exec-mask early termination).  The per-beam (cos a_j, sin a_j)
// fan table is computed once per workgroup and kept in LDS; per-pose constants are
// wave-uniform.  Waves are persistent and take chunks round-robin, so a launch
// has 256 CUs x 8 workgroups regardless of the batch size.
#pragma once
#include "scan_device.h"

namespace scan {

constexpr int WG = 256;                 // 4 waves
constexpr int WAVES_PER_WG = WG / 64;
constexpr int GINF = 30000;             // "no obstacle in this column" (maps <= 16384 per side)
constexpr uint32_t GINF2 = (uint32_t)GINF * (uint32_t)GINF;

// ------------------------------------------------------------------------------
// K0: exact EDT.  Pass 1: per column, distance to the nearest occupied cell of the
// column.  A workgroup owns 64 adjacent columns (one lane each, so every row access is
// a coalesced 64-byte read) and splits the rows into 16 segments, one per wave; the
// waves exchange "last occupied row below / first occupied row above my segment"
// through LDS, so a column is swept by 16 waves in parallel instead of one lane
// walking all rows (1.5 ms -> ~0.1 ms at 2049^2: the table rebuild must keep up with
// per-tick map changes, scripts/two_player/rcs_two_player.py:110-121).
// ------------------------------------------------------------------------------
constexpr int EDT_SEGS = 16;

__global__ __launch_bounds__(1024) void edt_cols_kernel(const uint8_t *__restrict__ occ, int rows,
                                                        int cols, int *__restrict__ g)
{
    __shared__ int s_last[EDT_SEGS][64], s_first[EDT_SEGS][64];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < cols;
    const int seg_len = (rows + EDT_SEGS - 1) / EDT_SEGS;
    const int r0 = seg * seg_len, r1 = min(rows, r0 + seg_len);
    // (a) last / first occupied row inside my segment
    int last = -GINF, first = 4 * GINF;
    if (ok) {
#pragma unroll 8
        for (int r = r0; r < r1; ++r) {
            if (occ[(size_t)r * cols + c]) {
                last = r;
                first = min(first, r);
            }
        }
    }
    s_last[seg][lane] = last;
    s_first[seg][lane] = first;
    __syncthreads();
    if (!ok) return;
    // (b) carries from the segments below / above
    int below = -GINF, above = 4 * GINF;
    for (int k = 0; k < seg; ++k) below = max(below, s_last[k][lane]);
    for (int k = seg + 1; k < EDT_SEGS; ++k) above = min(above, s_first[k][lane]);
    // (c) down sweep then up sweep over my segment
    last = below;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        if (occ[(size_t)r * cols + c]) last = r;
        const int d = r - last;
        g[(size_t)r * cols + c] = d > GINF ? GINF : d;
    }
    int nxt = above;
#pragma unroll 8
    for (int r = r1 - 1; r >= r0; --r) {
        if (occ[(size_t)r * cols + c]) nxt = r;
        const int d = nxt - r;
        const int old = g[(size_t)r * cols + c];
        g[(size_t)r * cols + c] = d < old ? d : old;
    }
}

// Pass 2: one workgroup per row, the row of column distances staged in LDS; each
// cell widens its search k = 1,2,.. while k^2 can still beat the best d^2 found, so
// the work per cell is O(distance), not O(cols).  d^2 is an exact integer; the
// result is sqrtf((float)d2), correctly rounded == the CPU statement.
__global__ __launch_bounds__(256) void edt_rows_kernel(const int *__restrict__ g, int rows,
                                                       int cols, float *__restrict__ dt)
{
    extern __shared__ int grow[];
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) grow[c] = g[(size_t)r * cols + c];
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        uint32_t v0 = (uint32_t)grow[c];
        uint32_t best = v0 * v0;
        for (int k = 1; (uint32_t)k * (uint32_t)k < best; ++k) {
            const bool l_ok = c - k >= 0, r_ok = c + k < cols;
            if (!l_ok && !r_ok) break;
            const uint32_t kk = (uint32_t)k * (uint32_t)k;
            if (l_ok) {
                uint32_t v = (uint32_t)grow[c - k];
                uint32_t cand = kk + v * v;
                best = cand < best ? cand : best;
            }
            if (r_ok) {
                uint32_t v = (uint32_t)grow[c + k];
                uint32_t cand = kk + v * v;
                best = cand < best ? cand : best;
            }
        }
        dt[(size_t)r * cols + c] = best >= GINF2 ? 1e10f : sqrtf((float)best);
    }
}

// bit-packed occupancy rows (bit c&31 of word c>>5), for the LDS-tiled kernels
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *__restrict__ occ, int rows,
                                                        int cols, int stride,
                                                        uint32_t *__restrict__ bits)
{
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (w >= stride || r >= rows) return;
    uint32_t word = 0;
    int c0 = w * 32;
#pragma unroll 4
    for (int b = 0; b < 32; ++b) {
        int c = c0 + b;
        if (c < cols && occ[(size_t)r * cols + c]) word |= 1u << b;
    }
    bits[(size_t)r * stride + w] = word;
}

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
    return (uint32_t)(((int)r.gy >> tile_shift) * tiles_x + ((int)r.gx >> tile_shift));
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
                kf[u] = inb ? (uint32_t)(((int)gy >> tile_shift) * tiles_x + ((int)gx >> tile_shift))
                            : (uint32_t)n_tiles - 1;
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
constexpr int POSES_PER_WG = 512;      // (2048 while one workgroup scanned all the counters; with the per-tile
                                       //  scan 256..1024 are equally good and 4..13 % ahead of that)

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
// workgroup PER TILE in two small launches — (a) inside the tile's own run of n_wg counters (coalesced
// 256-wide pieces, running carry) + the tile's total, (b) add the totals of the tiles in front — instead
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

__global__ __launch_bounds__(256) void tile_scan_b_kernel(uint32_t *__restrict__ hist_all, int n_wg,
                                                          const uint32_t *__restrict__ tile_total)
{
    __shared__ uint32_t part[4];
    const int t = blockIdx.x;
    uint32_t mine = 0;
    for (int k = threadIdx.x; k < t; k += 256) mine += tile_total[k];
    uint32_t base;
    (void)wg256_excl_scan(mine, part, base);           // base = poses in the tiles in front of this one
    uint32_t *row = hist_all + (size_t)t * n_wg;
    for (int i = threadIdx.x; i < n_wg; i += 256) row[i] += base;
}

__global__ __launch_bounds__(256) void pose_scatter_kernel(int n, const PoseRec *__restrict__ rec,
                                                           const uint32_t *__restrict__ keys,
                                                           const uint32_t *__restrict__ base_all,
                                                           int n_wg, int n_tiles,
                                                           PoseRec *__restrict__ rec_sorted,
                                                           uint32_t *__restrict__ order, int poses_per_wg,
                                                           MapParams m, float *__restrict__ d0, float coeff)
{
    extern __shared__ uint32_t cursor[];           // n_tiles
    const int w = blockIdx.x;
    for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) cursor[i] = base_all[(size_t)i * n_wg + w];
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
template <bool AUX, bool TILED>
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
        "v_pk_fma_f32 v[26:27], v[22:23], v[20:21], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_i32_f32_e32 %[c], v26\n\t"
        "v_cvt_i32_f32_e32 %[r], v27\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v26, %[nstride], v26\n\t"
        "v_lshl_add_u32 v26, %[c], 4, v26\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[c]\n\t"
        "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[d], v26, %[base]\n\t"
        ".if %[aux]\n\t"
        "v_add_u32_e32 %[ns], 1, %[ns]\n\t"
        ".endif\n\t"
        "s_waitcnt vmcnt(0)\n\t"
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
          [base] "s"(pdt), [low] "s"(low), [aux] "n"(AUX ? 1 : 0), [tiled] "n"(TILED ? 1 : 0)
        : "v26", "v27", "vcc", "scc", "memory");
}


// march_loop with an iteration cap (drain phase: a bounded stretch of the plain loop between two attempts of
// the speculating loop).  Leaves when no lane is live or after `iters` samples per lane.
template <bool TILED>
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
        "v_pk_fma_f32 v[26:27], v[22:23], v[20:21], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_i32_f32_e32 %[c], v26\n\t"
        "v_cvt_i32_f32_e32 %[r], v27\n\t"
        ".if %[tiled]\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[k4]\n\t"
        "v_and_b32_e32 v26, %[nstride], v26\n\t"
        "v_lshl_add_u32 v26, %[c], 4, v26\n\t"
        ".else\n\t"
        "v_mad_i32_i24 v26, %[r], %[stride], %[c]\n\t"
        "v_lshl_add_u32 v26, v26, 2, %[k4]\n\t"
        ".endif\n\t"
        "global_load_dword %[d], v26, %[base]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_add_f32_e32 v20, v20, %[d]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_cbranch_execz L_cap_done_%=\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 L_cap_%=\n"
        "L_cap_done_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [t] "+{v20}"(t), [c] "+v"(c), [r] "+v"(r), [d] "+v"(d), [save] "=&s"(save), [n] "+s"(n)
        : [dy] "{v22}"(dy), [dx] "{v23}"(dx), [gx] "{v24}"(gx), [gy] "{v25}"(gy), [mx] "s"(max_range),
          [stride] "s"(stride), [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [tiled] "n"(TILED ? 1 : 0)
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
template <bool TILED>
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
        "v_pk_fma_f32 v[26:27], v[22:23], v[20:21], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_i32_f32_e32 v26, v26\n\t"
        "v_cvt_i32_f32_e32 v27, v27\n\t"
        "v_mad_i32_i24 v27, v27, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v27, %[nstride], v27\n\t"
        "v_lshl_add_u32 v26, v26, 4, v27\n\t"
        "global_load_dword v42, v26, %[base]\n\t"
        // sample 1 where t1 is still inside the range window
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        "v_pk_fma_f32 v[34:35], v[22:23], v[28:29], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_i32_f32_e32 v34, v34\n\t"
        "v_cvt_i32_f32_e32 v35, v35\n\t"
        "v_mad_i32_i24 v35, v35, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v35, %[nstride], v35\n\t"
        "v_lshl_add_u32 v34, v34, 4, v35\n\t"
        "global_load_dword v31, v34, %[base]\n\t"
        // sample 2
        "v_cmpx_gt_f32_e32 %[mx], v30\n\t"
        "v_pk_fma_f32 v[36:37], v[22:23], v[30:31], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_i32_f32_e32 v36, v36\n\t"
        "v_cvt_i32_f32_e32 v37, v37\n\t"
        "v_mad_i32_i24 v37, v37, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v37, %[nstride], v37\n\t"
        "v_lshl_add_u32 v36, v36, 4, v37\n\t"
        "global_load_dword v33, v36, %[base]\n\t"
        // sample 3
        "v_cmpx_gt_f32_e32 %[mx], v32\n\t"
        "v_pk_fma_f32 v[38:39], v[22:23], v[32:33], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_i32_f32_e32 v38, v38\n\t"
        "v_cvt_i32_f32_e32 v39, v39\n\t"
        "v_mad_i32_i24 v39, v39, %[stride], %[k4]\n\t"
        "v_and_b32_e32 v39, %[nstride], v39\n\t"
        "v_lshl_add_u32 v38, v38, 4, v39\n\t"
        "global_load_dword v43, v38, %[base]\n\t"
        // stage 0: the sample at t is always real
        "s_mov_b64 exec, %[live]\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v42\n\t"
        "v_add_f32_e32 v20, v20, v42\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"                  // still marching ...
        "v_cmpx_eq_f32_e32 v42, v29\n\t"                    // ... and the step was the predicted one
        "s_mov_b64 %[hit], exec\n\t"
        // stage 1: the march arrived at t1 exactly
        "s_waitcnt vmcnt(2)\n\t"
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v31\n\t"
        "v_add_f32_e32 v20, v20, v31\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v31, v29\n\t"
        // stage 2
        "s_waitcnt vmcnt(1)\n\t"
        "v_mov_b32_e32 v40, v20\n\t"
        "v_mov_b32_e32 %[d], v33\n\t"
        "v_add_f32_e32 v20, v20, v33\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "v_cmpx_eq_f32_e32 v33, v29\n\t"
        // stage 3
        "s_waitcnt vmcnt(0)\n\t"
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
        "v_pk_fma_f32 v[26:27], v[22:23], v[40:41], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_i32_f32_e32 %[c], v26\n\t"
        "v_cvt_i32_f32_e32 %[r], v27\n"
        "L_drain_done_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [t] "+{v20}"(t), [c] "+v"(c), [r] "+v"(r), [d] "+v"(d), [save] "=&s"(save), [ent] "=&s"(ent),
          [live] "=&s"(live), [hit] "=&s"(hit)
        : [dy] "{v22}"(dy), [dx] "{v23}"(dx), [gx] "{v24}"(gx), [gy] "{v25}"(gy), [mx] "s"(max_range),
          [stride] "s"(stride), [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt)
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
template <bool TILED>
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
        "L_march2_%=:\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "v_pk_fma_f32 v[26:27], v[22:23], v[20:21], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
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
        "v_pk_fma_f32 v[34:35], v[30:31], v[28:29], v[32:33] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
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
        "s_mov_b64 exec, %[mA]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_add_f32_e32 v20, v20, %[dA]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v20\n\t"
        "s_mov_b64 %[mA], exec\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_add_f32_e32 v28, v28, %[dB]\n\t"
        "v_cmpx_gt_f32_e32 %[mx], v28\n\t"
        "s_mov_b64 %[mB], exec\n\t"
        "s_bcnt1_i32_b64 %[n], %[mA]\n\t"
        "s_bcnt1_i32_b64 %[n2], exec\n\t"
        "s_add_u32 %[n], %[n], %[n2]\n\t"
        "s_cmp_gt_u32 %[n], %[low]\n\t"
        "s_cbranch_scc1 L_march2_%=\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [tA] "+{v20}"(tA), [cA] "+v"(cA), [rA] "+v"(rA), [dA] "+v"(dA), [tB] "+{v28}"(tB), [cB] "+v"(cB),
          [rB] "+v"(rB), [dB] "+v"(dB), [save] "=&s"(save), [mA] "=&s"(mA), [mB] "=&s"(mB), [n] "=&s"(n),
          [n2] "=&s"(n2)
        : [dyA] "{v22}"(dyA), [dxA] "{v23}"(dxA), [gxA] "{v24}"(gxA), [gyA] "{v25}"(gyA), [dyB] "{v30}"(dyB),
          [dxB] "{v31}"(dxB), [gxB] "{v32}"(gxB), [gyB] "{v33}"(gyB), [mx] "s"(max_range), [stride] "s"(stride),
          [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [low] "s"(low), [tiled] "n"(TILED ? 1 : 0)
        : "v26", "v27", "v34", "v35", "vcc", "scc", "memory");
}


// Three rays per lane: the same alternation over three live masks (slot C: t v36 / dir v[38:39] /
// origin v[40:41] / scratch v[42:43]).
// (Tried and dropped, no measurable change at cfg2 / 32 k poses: a drain-phase form that branches over
//  a slot whose rays have all finished instead of issuing its 9 VALU with EXEC = 0, and a 24-bit
//  multiply for the output index in the claim.)
template <bool TILED>
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
        "L_march3_%=:\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "v_pk_fma_f32 v[26:27], v[22:23], v[20:21], v[24:25] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
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
        "v_pk_fma_f32 v[34:35], v[30:31], v[28:29], v[32:33] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
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
        "v_pk_fma_f32 v[42:43], v[38:39], v[36:37], v[40:41] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
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
        "s_bcnt1_i32_b64 %[n], %[mA]\n\t"
        "s_bcnt1_i32_b64 %[n2], %[mB]\n\t"
        "s_add_u32 %[n], %[n], %[n2]\n\t"
        "s_bcnt1_i32_b64 %[n2], exec\n\t"
        "s_add_u32 %[n], %[n], %[n2]\n\t"
        "s_cmp_gt_u32 %[n], %[low]\n\t"
        "s_cbranch_scc1 L_march3_%=\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [tA] "+{v20}"(tA), [cA] "+v"(cA), [rA] "+v"(rA), [dA] "+v"(dA), [tB] "+{v28}"(tB), [cB] "+v"(cB),
          [rB] "+v"(rB), [dB] "+v"(dB), [tC] "+{v36}"(tC), [cC] "+v"(cC), [rC] "+v"(rC), [dC] "+v"(dC),
          [save] "=&s"(save), [mA] "=&s"(mA), [mB] "=&s"(mB), [mC] "=&s"(mC), [n] "=&s"(n), [n2] "=&s"(n2)
        : [dyA] "{v22}"(dyA), [dxA] "{v23}"(dxA), [gxA] "{v24}"(gxA), [gyA] "{v25}"(gyA), [dyB] "{v30}"(dyB),
          [dxB] "{v31}"(dxB), [gxB] "{v32}"(gxB), [gyB] "{v33}"(gyB), [dyC] "{v38}"(dyC), [dxC] "{v39}"(dxC),
          [gxC] "{v40}"(gxC), [gyC] "{v41}"(gyC), [mx] "s"(max_range), [stride] "s"(stride),
          [nstride] "s"(nstride), [k4] "v"(k4), [base] "s"(pdt), [low] "s"(low), [tiled] "n"(TILED ? 1 : 0)
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
    int run_log2;            // a workgroup's stream interleaves RUNS of 2^run_log2 consecutive 64-ray blocks
    int stripe;              // INLINE only, where the band's pose ids come from: 0 = the caller's order
                             //   (band = index range), 1 = row stripes of the map compacted by every
                             //   workgroup itself (stripe_band_list), 2 = `order` (tile order from the
                             //   keys-only binning launch)
    unsigned long long *dbg; // diagnostics (nullptr in production): 4 words per wave
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

// LDS header of the stream kernels in floats: [0] slot counter, [1] spare, [2..66) crash_seen
constexpr int STREAM_HDR = 66;

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

template <bool AUX, bool CRASH, int NT, bool INLINE, bool TILED, int SLOTS = 1>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8)))
void rm_fan_stream_kernel(PadMap pm, FanParams f, StreamParams sp, float *__restrict__ out,
                          int32_t *__restrict__ hits, uint16_t *__restrict__ steps, CrashParams cp)
{
    extern __shared__ __attribute__((aligned(32))) float lds_f[];
    const unsigned long long t_entry = sp.dbg ? wall_clock64() : 0ull;   // diagnostics
    uint32_t *q_next = reinterpret_cast<uint32_t *>(lds_f);     // shared slot counter
    // [2 .. STREAM_HDR): CRASH only — poses this workgroup has already reported as crashed
    // (direct-mapped): a pose scraping a wall crashes on hundreds of beams, all marched by this
    // workgroup, and only the first of them needs to touch the group's word in global memory
    uint32_t *crash_seen = reinterpret_cast<uint32_t *>(lds_f + 2);
    float2 *fan_cs = reinterpret_cast<float2 *>(lds_f + STREAM_HDR);     // num_rays float2
    // CRASH: the car-outline table next to the fan table (read when a ray finishes: from LDS it does
    // not sit behind the range store in vmcnt — a global read there made every refill wait for the
    // store's acknowledgement and the kernel 2.5x slower)
    double *edge_l = reinterpret_cast<double *>(lds_f + STREAM_HDR + 2 * (size_t)f.num_rays);
    const size_t tables = STREAM_HDR + (CRASH ? 4 : 2) * (size_t)f.num_rays;
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
    const uint32_t K = (g < seg_runs ? (seg_runs - g + G - 1) / G : 0) << rl;
    const uint32_t total = K << 6;
    // i-th block of this workgroup's stream -> its index in the band / first ray of the block
    auto blkidx_of = [&](uint32_t i) { return ((g + (i >> rl) * G) << rl) + (i & rmask); };
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
                const uint32_t kf = pose_record(mp, sp.raw_poses, (int)pid, 0, 1, 1, r);
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
            const float2 cs = fan_cs[j];
            const float ct = __builtin_bit_cast(float, ra.z), st = __builtin_bit_cast(float, ra.w);
            s.gx = __builtin_bit_cast(float, ra.x);
            s.gy = __builtin_bit_cast(float, ra.y);
            s.dx = __builtin_fmaf(ct, cs.x, -(st * cs.y));
            s.dy = __builtin_fmaf(st, cs.x, ct * cs.y);
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
        auto finish = [&](Slot &s) {
            float r = f.max_range;
            if (s.d_last == PDT_HIT) {
                const float xd = (float)s.pc - s.gx, yd = (float)s.pr - s.gy;
                r = hit_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
            }
            r *= pm.res;
            if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (s.oidx >> 2));
            if (out) *reinterpret_cast<float *>(reinterpret_cast<char *>(out) + s.oidx) = r;
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
                    const uint32_t cap = (uint32_t)__builtin_amdgcn_readfirstlane(min(max(sp.drain_cap, 1), DRAIN_CAP));
                    if (nlive > cap) {
                        // the plain loop until few rays are left
                        if (SLOTS == 3)
                            march_loop3<TILED>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx,
                                        sb.gy, sb.t, sb.pc, sb.pr, sb.d_last, sc.dx, sc.dy, sc.gx, sc.gy, sc.t, sc.pc, sc.pr,
                                        sc.d_last, pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range, cap);
                        else
                            march_loop2<TILED>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx,
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
                        while (__ballot(sa.t < f.max_range)) {
                            march_loop_capped<TILED>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, pm.pdt,
                                                     pm.stride, pm.nstride, pm.k4, f.max_range, (uint32_t)sp.drain_stretch);
                            march_drain4<TILED>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, pm.pdt, pm.stride,
                                                pm.nstride, pm.k4, f.max_range);
                        }
                        if (sa.oidx != NO_RAY) finish(sa);
                        break;
                    }
                }
            }
            if (SLOTS == 3)
                march_loop3<TILED>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx, sb.gy,
                            sb.t, sb.pc, sb.pr, sb.d_last, sc.dx, sc.dy, sc.gx, sc.gy, sc.t, sc.pc, sc.pr, sc.d_last,
                            pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range,
                            exhausted ? 0u : 3u * (uint32_t)sp.low_water);
            else
                march_loop2<TILED>(sa.dx, sa.dy, sa.gx, sa.gy, sa.t, sa.pc, sa.pr, sa.d_last, sb.dx, sb.dy, sb.gx, sb.gy,
                            sb.t, sb.pc, sb.pr, sb.d_last, pm.pdt, pm.stride, pm.nstride, pm.k4, f.max_range,
                            exhausted ? 0u : 2u * (uint32_t)sp.low_water);
        }
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
                    r = hit_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                }
                r *= pm.res;
                if (f.noise_std > 0.0f)
                    r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + oidx);
                if (out) *reinterpret_cast<float *>(reinterpret_cast<char *>(out) + s1.oidx) = r;
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
                march_loop<AUX, TILED>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, nstep, pm.pdt,
                                       pm.stride, pm.nstride, pm.k4, f.max_range, (uint32_t)sp.spec_drain);
                // what is still marching after a stretch of the plain loop is a long chain: speculate on it
                // while that pays, fall back to the plain loop for a stretch when it does not
                while (__ballot(s1.t < f.max_range)) {
                    march_loop_capped<TILED>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, pm.pdt,
                                             pm.stride, pm.nstride, pm.k4, f.max_range, (uint32_t)sp.spec_stretch);
                    march_drain4<TILED>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, pm.pdt, pm.stride,
                                        pm.nstride, pm.k4, f.max_range);
                }
                continue;
            }
        }
        march_loop<AUX, TILED>(s1.dx, s1.dy, s1.gx, s1.gy, s1.t, s1.pc, s1.pr, s1.d_last, nstep, pm.pdt, pm.stride,
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

}  // namespace scan

// ==============================================================================
// K3: GiantLUTCast — the bandwidth-bound variant.
// Table: uint16 lut[row][col][theta_bin], so the fan of one pose is ONE contiguous
// run of ~num_rays entries (bin spacing ~ beam spacing when theta_disc ~ 2pi*B/fov):
// a query streams ~2 B/ray in and 4 B/ray out, nothing else.  2000^2 x 1442 bins =
// 11.5 GB of the 288 GB HBM.  Built on the device with the K1 march from every cell
// corner (range_libc seeds its table with RayMarching the same way).
// ==============================================================================
namespace scan {

struct LutParams {
    uint16_t *lut;
    int theta_disc;
    float bins_per_rad;      // theta_disc / 2pi (float)
    float bin_width;         // 2pi / theta_disc
    float quant, dequant;    // 65535/max_range, max_range/65535
    int debug;               // diagnostics only: bit0 skip table loads, bit1 skip range stores
};

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
            if (inb && idx < D && need) regs[n] = *reinterpret_cast<const uint4 *>(row + idx);
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
struct BlPad {
    const uint32_t *bits;       // both padded copies in one buffer
    uint32_t k_n, k_t;          // byte offset of the word holding cell (0, 0): normal / transposed copy
    int stride_n, stride_t;     // words per padded row
    float near;                 // origins with -near < g < dim + near are covered by the padding
};

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
                out[oidx] = r;
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

// ==============================================================================
// K3b: CDDTCast (scripts/two_player/scan.py:46).
// Table: for every theta bin in [0, pi) the edge cells of the map are projected into
// the bin's rotated frame and bucketed by their rotated row; each bucket holds the
// sorted rotated x of the cells it covers (CSR: offsets[] + xs[]).  A query rotates
// the ray origin into the frame of the bin nearest to -heading and binary-searches
// ONE bucket for the next stored x ahead of (or, for the flipped half turn, behind)
// the origin.  Built entirely on the device: edge list -> count -> scan -> fill ->
// segmented sort.
// ==============================================================================
namespace scan {

constexpr float CDDT_EPS = 1e-5f;

struct CddtParams {
    int theta_disc, n_bins;
    const float *cosv, *sinv, *trans;   // per bin
    const int *width;                   // per bin: buckets
    const uint32_t *bucket_off;         // per bin: first bucket (n_bins + 1)
    uint32_t *offsets;                  // per bucket: [start, end) in xs (n_buckets + 1)   (build intermediate)
    float *xs;                          // CSR values as projected, unsorted                 (build intermediate)
    // what the queries read: the blocked table.  A bucket of n values owns a run of 128-B lines starting at
    // line hdr[b].x: its values in LEAVES of 32 (sorted, the last one padded with +inf), and — more than one
    // leaf — in front of them the SEPARATORS, the first value of every leaf, 32 per line (padded with +inf).
    // A query reads the header, one separator line and one leaf line: two table lines instead of the 3.6 a
    // bisection over the packed CSR run touched, three dependent loads instead of eight.
    uint2 *hdr;                         // per bucket: {first line, n}
    float *tab;
    float bins_per_rad;
    int debug;                          // diagnostics only: bit0 skip the bucket searches, bit1 skip the range stores
};

constexpr int EDGE_ROWS_PER_WG = 8;

__global__ __launch_bounds__(256) void cddt_edges_kernel(const uint8_t *__restrict__ occ, int rows,
                                                         int cols, uint32_t *__restrict__ n_edges,
                                                         uint32_t *__restrict__ edges /* r<<16|c */)
{
    // a workgroup owns 256 columns x EDGE_ROWS_PER_WG rows; ONE global atomic per workgroup reserves
    // its run of the list (same-word atomics retire ~10 per us: per-cell or per-wave atomics would
    // dominate a 2049^2 map).  The order of the list is irrelevant: every bucket is sorted afterwards.
    __shared__ uint32_t s_cnt[4 * EDGE_ROWS_PER_WG + 1];
    const int c = blockIdx.x * 256 + threadIdx.x;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long bal[EDGE_ROWS_PER_WG];
    uint32_t edge_bits = 0;
#pragma unroll
    for (int k = 0; k < EDGE_ROWS_PER_WG; ++k) {
        const int r = blockIdx.y * EDGE_ROWS_PER_WG + k;
        bool edge = false;
        if (c < cols && r < rows && occ[(size_t)r * cols + c]) {
            // occupied cell with a free 4-neighbour; border cells count as edges
            edge = r == 0 || c == 0 || r == rows - 1 || c == cols - 1;
            if (!edge)
                edge = !occ[(size_t)(r - 1) * cols + c] || !occ[(size_t)(r + 1) * cols + c] ||
                       !occ[(size_t)r * cols + c - 1] || !occ[(size_t)r * cols + c + 1];
        }
        bal[k] = __ballot(edge);
        edge_bits |= (edge ? 1u : 0u) << k;
        if (lane == 0) s_cnt[k * 4 + wave] = (uint32_t)__popcll(bal[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int i = 0; i < 4 * EDGE_ROWS_PER_WG; ++i) { const uint32_t v = s_cnt[i]; s_cnt[i] = tot; tot += v; }
        s_cnt[4 * EDGE_ROWS_PER_WG] = tot ? atomicAdd(n_edges, tot) : 0u;
    }
    __syncthreads();
    const uint32_t base = s_cnt[4 * EDGE_ROWS_PER_WG];
#pragma unroll
    for (int k = 0; k < EDGE_ROWS_PER_WG; ++k)
        if ((edge_bits >> k) & 1u) {
            const int r = blockIdx.y * EDGE_ROWS_PER_WG + k;
            edges[base + s_cnt[k * 4 + wave] + (uint32_t)__popcll(bal[k] & ((1ull << lane) - 1ull))] =
                ((uint32_t)r << 16) | (uint32_t)c;
        }
}

// Projection of the edge cells into the buckets of every theta bin, in two passes around an exclusive
// scan: COUNT (bucket sizes into counts[]) and FILL (values into xs at the CSR offsets).
// A straight wall parallel to a bin's direction lands in ONE bucket, so a lane-per-(cell, bin) kernel
// with global atomics serialises hundreds of same-word atomics (~10 per us: 34 us on the 435x350
// colombia map).  Here a workgroup owns (a chunk of CDDT_CHUNK edge cells) x (ONE theta bin) and
// histograms its chunk in LDS first; only one global atomic per touched bucket leaves the workgroup.
// FILL reserves each bucket's run with atomicSub on counts[] — the counts return to zero, so the next
// rebuild needs no memset and the scan's input is consumed in place — then hands out positions from
// LDS cursors in a second sweep.  The order inside a bucket is irrelevant (sorted afterwards).
constexpr int CDDT_CHUNK = 2048;

template <bool FILL>
__global__ __launch_bounds__(256) void cddt_project_kernel(CddtParams cp, const uint32_t *__restrict__ edges,
                                                           const uint32_t *__restrict__ n_edges,
                                                           uint32_t *__restrict__ counts)
{
    extern __shared__ uint32_t lh[];              // width[a] local counters (+ width[a] bases when FILL)
    const int a = blockIdx.y;
    const uint32_t ne = *n_edges;
    const uint32_t e0 = blockIdx.x * (uint32_t)CDDT_CHUNK;
    if (e0 >= ne) return;
    const uint32_t e1 = min(ne, e0 + (uint32_t)CDDT_CHUNK);
    const int wdt = cp.width[a];
    const float cs = cp.cosv[a], sn = cp.sinv[a], tr = cp.trans[a];
    const float half = (fabsf(sn) + fabsf(cs)) * 0.5f;
    const uint32_t b0 = cp.bucket_off[a];
    for (int i = threadIdx.x; i < wdt; i += 256) lh[i] = 0;
    __syncthreads();
    auto span = [&](uint32_t e, float &lx, int &lower, int &upper) {
        const float px = (float)(e & 0xFFFFu) + 0.5f, py = (float)(e >> 16) + 0.5f;
        lx = __builtin_fmaf(px, cs, -(py * sn));
        const float ly = __builtin_fmaf(px, sn, py * cs) + tr;
        upper = (int)((ly + half) - CDDT_EPS);
        lower = (int)((ly - half) + CDDT_EPS);
        if (lower < 0) lower = 0;
        if (upper >= wdt) upper = wdt - 1;
    };
    for (uint32_t ei = e0 + threadIdx.x; ei < e1; ei += 256) {
        float lx;
        int lower, upper;
        span(edges[ei], lx, lower, upper);
        for (int k = lower; k <= upper; ++k) atomicAdd(&lh[k], 1u);
    }
    __syncthreads();
    if (!FILL) {
        for (int i = threadIdx.x; i < wdt; i += 256) {
            const uint32_t c = lh[i];
            if (c) atomicAdd(&counts[b0 + (uint32_t)i], c);
        }
        return;
    }
    uint32_t *base = lh + wdt;
    for (int i = threadIdx.x; i < wdt; i += 256) {
        const uint32_t c = lh[i];
        if (c) base[i] = cp.offsets[b0 + (uint32_t)i] + atomicSub(&counts[b0 + (uint32_t)i], c) - c;
        lh[i] = 0;
    }
    __syncthreads();
    for (uint32_t ei = e0 + threadIdx.x; ei < e1; ei += 256) {
        float lx;
        int lower, upper;
        span(edges[ei], lx, lower, upper);
        for (int k = lower; k <= upper; ++k) cp.xs[base[k] + atomicAdd(&lh[k], 1u)] = lx;
    }
}

// Exclusive scan of the bucket counters -> CSR offsets, out of place (the counts stay: FILL consumes
// them), in two small launches with one workgroup per theta bin: (1) scan inside the bin's own run of
// buckets (coalesced 256-wide tiles, running carry) and publish the bin's total, (2) add the totals
// of the bins in front.  (One workgroup walking all ~30 000 counters with a lane-strided pattern
// took 42 us on colombia.)  Pass 2 also queues the buckets too large for the one-wave sort.
// lines of the blocked table a bucket of n values owns: its leaves of 32 + (more than one leaf) the separator lines
__device__ __forceinline__ uint32_t cddt_bucket_lines(uint32_t n)
{
    const uint32_t nleaf = (n + 31u) >> 5;
    return nleaf + (nleaf > 1u ? (nleaf + 31u) >> 5 : 0u);
}

__global__ __launch_bounds__(256) void cddt_scan_bins_kernel(CddtParams cp, const uint32_t *__restrict__ counts,
                                                             uint32_t *__restrict__ bin_total,
                                                             uint32_t *__restrict__ bin_lines)
{
    __shared__ uint32_t part[4], part2[4];
    const int a = blockIdx.x;
    const int wdt = cp.width[a];
    const uint32_t b0 = cp.bucket_off[a];
    uint32_t carry = 0, carry2 = 0;
    for (int i0 = 0; i0 < wdt; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        const uint32_t v = i < wdt ? counts[b0 + (uint32_t)i] : 0u;
        uint32_t tot, tot2;
        const uint32_t ex = wg256_excl_scan(v, part, tot);
        const uint32_t ex2 = wg256_excl_scan(cddt_bucket_lines(v), part2, tot2);
        if (i < wdt) {
            cp.offsets[b0 + (uint32_t)i] = carry + ex;
            cp.hdr[b0 + (uint32_t)i] = make_uint2(carry2 + ex2, v);
        }
        carry += tot;
        carry2 += tot2;
    }
    if (threadIdx.x == 0) {
        bin_total[a] = carry;
        bin_lines[a] = carry2;
    }
}

__global__ __launch_bounds__(256) void cddt_scan_add_kernel(CddtParams cp, const uint32_t *__restrict__ counts,
                                                            const uint32_t *__restrict__ bin_total,
                                                            const uint32_t *__restrict__ bin_lines,
                                                            uint32_t *__restrict__ big_list,
                                                            uint32_t *__restrict__ big_count)
{
    __shared__ uint32_t part[4], part2[4];
    const int a = blockIdx.x;
    uint32_t mine = 0, mine2 = 0;
    for (int k = threadIdx.x; k < a; k += 256) {
        mine += bin_total[k];
        mine2 += bin_lines[k];
    }
    uint32_t base, base2;
    (void)wg256_excl_scan(mine, part, base);           // base = sum of the totals of bins 0 .. a-1
    (void)wg256_excl_scan(mine2, part2, base2);
    const int wdt = cp.width[a];
    const uint32_t b0 = cp.bucket_off[a];
    for (int i = threadIdx.x; i < wdt; i += 256) {
        cp.offsets[b0 + (uint32_t)i] += base;
        cp.hdr[b0 + (uint32_t)i].x += base2;
        if (counts[b0 + (uint32_t)i] > 64u) big_list[atomicAdd(big_count, 1u)] = b0 + (uint32_t)i;
    }
    if (a == cp.n_bins - 1 && threadIdx.x == 0) cp.offsets[b0 + (uint32_t)wdt] = base + bin_total[a];
}

// Sort of every bucket, CSR run -> its lines of the blocked table, ONE launch.  No library call: hipcub's
// segmented sort reads segment statistics back to the host, and the two-player tick must stay a pure
// enqueue.
//  * workgroups >= n_big_wg: buckets of up to 64 values — nearly all of them — one wave each: a lane
//    holds one value and finds its rank among the others with a loop of lane broadcasts (ties broken
//    by position, so the ranks are a permutation);
//  * workgroups < n_big_wg: the queued buckets of more than 64 values (long straight walls parallel to
//    a bin's direction), one workgroup each: bitonic sort in LDS up to lds_cap values, beyond that a
//    rank sort straight from global memory (quadratic, but such a bucket needs a wall of > 5000 cells).
constexpr uint32_t CDDT_LDS_SORT = 16384;

// where rank r of a bucket goes in the blocked table, and the padding the ranks leave free
struct CddtRun {
    float *sep, *leaves;
    uint32_t n, nleaf, nsl;
};
__device__ __forceinline__ CddtRun cddt_run(const uint2 *__restrict__ hdr, float *__restrict__ tab, uint32_t b)
{
    const uint2 hd = hdr[b];
    CddtRun r;
    r.n = hd.y;
    r.nleaf = (r.n + 31u) >> 5;
    r.nsl = r.nleaf > 1u ? (r.nleaf + 31u) >> 5 : 0u;
    r.sep = tab + (size_t)hd.x * 32;
    r.leaves = r.sep + (size_t)r.nsl * 32;
    return r;
}
__device__ __forceinline__ void cddt_put(const CddtRun &r, uint32_t rank, float x)
{
    r.leaves[rank] = x;
    if ((rank & 31u) == 0u && r.nsl) r.sep[rank >> 5] = x;
}
__device__ __forceinline__ void cddt_pad(const CddtRun &r, uint32_t tid, uint32_t nt)
{
    for (uint32_t i = r.n + tid; i < r.nleaf * 32u; i += nt) r.leaves[i] = __builtin_inff();
    for (uint32_t k = r.nleaf + tid; k < r.nsl * 32u; k += nt) r.sep[k] = __builtin_inff();
}

__global__ __launch_bounds__(256) void cddt_sort_kernel(const uint32_t *__restrict__ offsets, uint32_t n_buckets,
                                                        const float *__restrict__ src,
                                                        const uint2 *__restrict__ hdr, float *__restrict__ tab,
                                                        const uint32_t *__restrict__ big_list,
                                                        const uint32_t *__restrict__ big_count,
                                                        uint32_t n_big_wg, uint32_t lds_cap)
{
    extern __shared__ float sv[];                      // lds_cap (<= CDDT_LDS_SORT) floats
    if (blockIdx.x >= n_big_wg) {
        const int lane = threadIdx.x & 63;
        const uint32_t wave = (blockIdx.x - n_big_wg) * 4u + (threadIdx.x >> 6);
        const uint32_t n_waves = (gridDim.x - n_big_wg) * 4u;
        for (uint32_t b = wave; b < n_buckets; b += n_waves) {
            const uint32_t lo = offsets[b], n = offsets[b + 1] - lo;
            if (n == 0 || n > 64u) continue;
            const CddtRun run = cddt_run(hdr, tab, b);
            const float x = (uint32_t)lane < n ? src[lo + lane] : __builtin_inff();
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const float xj = __shfl(x, (int)j);
                rank += (xj < x || (xj == x && j < (uint32_t)lane)) ? 1u : 0u;
            }
            if ((uint32_t)lane < n) cddt_put(run, rank, x);
            cddt_pad(run, (uint32_t)lane, 64u);
        }
        return;
    }
    const uint32_t nbig = *big_count;
    for (uint32_t q = blockIdx.x; q < nbig; q += n_big_wg) {
        const uint32_t b = big_list[q];
        const uint32_t lo = offsets[b], n = offsets[b + 1] - lo;
        const CddtRun run = cddt_run(hdr, tab, b);
        if (n <= lds_cap) {
            uint32_t m2 = 128;
            while (m2 < n) m2 <<= 1;
            for (uint32_t i = threadIdx.x; i < m2; i += 256) sv[i] = i < n ? src[lo + i] : __builtin_inff();
            __syncthreads();
            for (uint32_t k = 2; k <= m2; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = threadIdx.x; i < m2; i += 256) {
                        const uint32_t p = i ^ j;
                        if (p > i) {
                            const float x = sv[i], y = sv[p];
                            const bool up = (i & k) == 0;
                            if ((x > y) == up) { sv[i] = y; sv[p] = x; }
                        }
                    }
                    __syncthreads();
                }
            for (uint32_t i = threadIdx.x; i < n; i += 256) cddt_put(run, i, sv[i]);
            __syncthreads();
        } else {
            for (uint32_t i = threadIdx.x; i < n; i += 256) {
                const float x = src[lo + i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < n; ++j) {
                    const float xj = src[lo + j];
                    rank += (xj < x || (xj == x && j < i)) ? 1u : 0u;
                }
                cddt_put(run, rank, x);
            }
        }
        cddt_pad(run, threadIdx.x, 256u);
    }
}

// BOTH directions of one table bin from a grid origin: raw bin t (the ray runs along +x of the bin's frame:
// first stored x >= origin) and raw bin t + theta_disc/2 (half a turn away: last stored x <= origin) project
// the origin with the same rotation into the same bucket.  On the blocked table: count the separators <= lx
// (the leaf whose first value is the last one <= lx holds the backward answer and, unless all of it is below
// lx, the forward one — otherwise that is the next separator), then one pass of min / max over the leaf's 32
// values.  No bisection, no sortedness needed inside a line; the padding (+inf) never wins.  out_f / out_b
// are the two ranges in pixels (max_range when nothing is stored on that side).
__device__ __forceinline__ void cddt_query_pair(const CddtParams &cp, float max_range, float gx, float gy, int t,
                                                float &out_f, float &out_b)
{
    const float cs = cp.cosv[t], sn = cp.sinv[t];
    const float lx = __builtin_fmaf(gx, cs, -(gy * sn));
    const float ly = __builtin_fmaf(gx, sn, gy * cs) + cp.trans[t];
    out_f = max_range;
    out_b = max_range;
    if (ly >= 0.0f && ly < (float)cp.width[t]) {
        const uint2 hd = cp.hdr[cp.bucket_off[t] + (uint32_t)(int)ly];
        const uint32_t n = hd.y;
        if (n) {
            const float INF = __builtin_inff();
            const uint32_t nleaf = (n + 31u) >> 5;
            const float4 *lines = reinterpret_cast<const float4 *>(cp.tab) + (size_t)hd.x * 8;
            float fwd = INF, bwd = -INF;
            uint32_t leaf = 0;
            bool have_leaf = true;
            if (nleaf > 1u) {
                const uint32_t nsl = (nleaf + 31u) >> 5;
                uint32_t L = 0;
                if (nsl > 1u) {                        // > 1024 values: the separator line whose first value is the last <= lx
                    uint32_t a = 0, z = nsl;
                    while (z - a > 1u) {
                        const uint32_t mid = (a + z) >> 1;
                        if (lines[(size_t)mid * 8].x <= lx) a = mid; else z = mid;
                    }
                    L = a;
                    if (L + 1u < nsl) fwd = lines[(size_t)(L + 1u) * 8].x;      // (> lx: line z was probed)
                }
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 q = lines[(size_t)L * 8 + i];
                    c += (q.x <= lx ? 1u : 0u) + (q.y <= lx ? 1u : 0u) + (q.z <= lx ? 1u : 0u) + (q.w <= lx ? 1u : 0u);
                    fwd = __builtin_fminf(fwd, q.x > lx ? q.x : INF);
                    fwd = __builtin_fminf(fwd, q.y > lx ? q.y : INF);
                    fwd = __builtin_fminf(fwd, q.z > lx ? q.z : INF);
                    fwd = __builtin_fminf(fwd, q.w > lx ? q.w : INF);
                }
                have_leaf = L * 32u + c > 0u;
                leaf = L * 32u + c - 1u;
                lines += (size_t)nsl * 8;
            }
            if (have_leaf) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 q = lines[(size_t)leaf * 8 + i];
                    fwd = __builtin_fminf(fwd, q.x >= lx ? q.x : INF);
                    fwd = __builtin_fminf(fwd, q.y >= lx ? q.y : INF);
                    fwd = __builtin_fminf(fwd, q.z >= lx ? q.z : INF);
                    fwd = __builtin_fminf(fwd, q.w >= lx ? q.w : INF);
                    bwd = __builtin_fmaxf(bwd, q.x <= lx ? q.x : -INF);
                    bwd = __builtin_fmaxf(bwd, q.y <= lx ? q.y : -INF);
                    bwd = __builtin_fmaxf(bwd, q.z <= lx ? q.z : -INF);
                    bwd = __builtin_fmaxf(bwd, q.w <= lx ? q.w : -INF);
                }
            }
            out_f = __builtin_fminf(fwd - lx, max_range);
            out_b = __builtin_fminf(lx - bwd, max_range);
        }
    }
}

// ---- reductions inside an aligned group of 8 lanes (the theta-major search kernel does a look-up with 8 lanes:
// lane c reads float4 #c of the separator line and of the leaf line — a 128-B line is ONE coalesced access of
// two quads instead of eight 16-B loads per lane to 64 different lines): three DPP steps, quad_perm xor 1,
// xor 2, row_half_mirror; every lane of the group ends with the group's result
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    // (bound_ctrl: lets the compiler fold the move into the consuming VALU instruction's DPP operand; every
    // source lane of these three patterns exists)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float min8(float v)
{
    v = __builtin_fminf(v, dpp_f<0xB1>(v));            // quad_perm [1,0,3,2]
    v = __builtin_fminf(v, dpp_f<0x4E>(v));            // quad_perm [2,3,0,1]
    return __builtin_fminf(v, dpp_f<0x141>(v));        // row_half_mirror: lane i <-> 7 - i
}
__device__ __forceinline__ float max8(float v)
{
    v = __builtin_fmaxf(v, dpp_f<0xB1>(v));
    v = __builtin_fmaxf(v, dpp_f<0x4E>(v));
    return __builtin_fmaxf(v, dpp_f<0x141>(v));
}
__device__ __forceinline__ uint32_t sum8(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
    return v + (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);
}
// a lane's four values are ascending (ranks 4c .. 4c+3 of a sorted run, +inf padding last): the first one
// >= x / > x and the last one <= x by select chains
__device__ __forceinline__ float first_ge(const float4 &q, float x, float none)
{
    float r = q.w >= x ? q.w : none;
    r = q.z >= x ? q.z : r;
    r = q.y >= x ? q.y : r;
    return q.x >= x ? q.x : r;
}
__device__ __forceinline__ float first_gt(const float4 &q, float x, float none)
{
    float r = q.w > x ? q.w : none;
    r = q.z > x ? q.z : r;
    r = q.y > x ? q.y : r;
    return q.x > x ? q.x : r;
}
__device__ __forceinline__ float last_le(const float4 &q, float x, float none)
{
    float r = q.x <= x ? q.x : none;
    r = q.y <= x ? q.y : r;
    r = q.z <= x ? q.z : r;
    return q.w <= x ? q.w : r;
}

// one ray: the nearest bin of -heading; bins of the second half turn use the table bin half a turn away,
// searching backwards (CDDTCast::calc_range)
__device__ __forceinline__ float cddt_query(const MapParams &m, const CddtParams &cp, float max_range,
                                            float gx, float gy, float th)
{
    LutParams lp{};
    lp.theta_disc = cp.theta_disc;
    lp.bins_per_rad = cp.bins_per_rad;
    int b = lut_bin(-th, lp);                 // nearest bin of -heading in [0, theta_disc)
    bool flipped = false;
    if (b >= cp.n_bins) { b -= cp.theta_disc / 2; flipped = true; }
    if (b >= cp.n_bins) b = cp.n_bins - 1;
    float rf, rb;
    cddt_query_pair(cp, max_range, gx, gy, b, rf, rb);
    return (flipped ? rb : rf) * m.res;
}

// the raw theta bins a fan can touch: beam angles grow with j, so the bins of -(heading + alpha_j) are the
// circular run from the last beam's bin up to the first beam's — `cnt` bins from `first`, one bin of margin on
// either side (all theta_disc bins for fans close to a full turn, negative increments and huge headings)
__device__ __forceinline__ void cddt_fan_run(const FanParams &f, const CddtParams &cp, const LutParams &lp, float thg,
                                             float td_f, float inv_td, int &first, int &cnt)
{
    const float u0 = __builtin_rintf(-(thg + fan_alpha(f, 0)) * cp.bins_per_rad);
    const float u1 = __builtin_rintf(-(thg + fan_alpha(f, f.num_rays - 1)) * cp.bins_per_rad);
    const float spanf = u0 - u1;
    const bool all = !(spanf >= 0.0f && spanf < td_f - 4.0f) || !(__builtin_fabsf(u0) < 8388608.0f) ||
                     !(__builtin_fabsf(u1) < 8388608.0f);
    first = 0;
    cnt = cp.theta_disc;
    if (!all) {
        first = lut_bin_fast(-(thg + fan_alpha(f, f.num_rays - 1)), lp, td_f, inv_td) - 1;
        if (first < 0) first += cp.theta_disc;
        cnt = (int)spanf + 3;
    }
}

// The fan form.  A CDDT answer depends on the ray's ORIGIN and its theta BIN only, so every beam of a
// pose whose heading falls into one bin gets the same range (theta_disc 108 over a 4.71-rad fan of
// 1081 beams: ~13 beams per bin), and the two raw bins half a turn apart share one bucket search
// (cddt_query_pair).  A workgroup takes PP poses at a time, one lane per (pose, TABLE bin): only the
// table bins the fan touches in either direction are searched (fov 4.71: all 54 of theta_disc 108, for 81-82
// raw bins — a third fewer searches and table lines than one per raw bin, half of one per bin of the full
// turn), all of them in flight at once; the results are parked in LDS and the beams only look their bin up —
// the kernel turns from a latency-bound search per ray into a stream of range stores.  Bit-identical to the
// per-ray statement (same bin index arithmetic, same insertion points).
// `order` (optional): the poses in map-tile order (the keys-only binning launch of the ray-marching
// path); the sorted list is cut into n_bands bands, band x walked by the workgroups with
// blockIdx % n_bands == x — one XCD under round-robin dispatch —, so the workgroups of an XCD query
// neighbouring origins at the same time: for every theta bin they land in neighbouring buckets, and the
// table lines one pose fetched are L2 hits for the next (the table is ~10x an XCD's L2).
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(88), amdgpu_waves_per_eu(8, 8))) void cddt_fan_bins_kernel(MapParams m, FanParams f, CddtParams cp,
                                                             const float *__restrict__ poses,
                                                             float *__restrict__ out,
                                                             const uint32_t *__restrict__ order, int n_bands,
                                                             int lanes_per_pose, int pp)
{
    extern __shared__ float bin_range[];                 // pp x theta_disc floats (this workgroup's poses)
    const int nt = (int)blockDim.x, td = cp.theta_disc, half = td / 2;
    LutParams lp{};
    lp.theta_disc = td;
    lp.bins_per_rad = cp.bins_per_rad;
    const float td_f = (float)td, inv_td = 1.0f / (float)td;
    const int band = (int)(blockIdx.x % (unsigned)n_bands), g = (int)(blockIdx.x / (unsigned)n_bands);
    const int G = ((int)gridDim.x - band + n_bands - 1) / n_bands;
    const int lo = (int)(((long)f.n_poses * band) / n_bands), hi = (int)(((long)f.n_poses * (band + 1)) / n_bands);
    const int q = (int)threadIdx.x / lanes_per_pose, t0 = (int)threadIdx.x % lanes_per_pose;
    for (int s0 = lo + g * pp; s0 < hi; s0 += G * pp) {
        if (q < pp && s0 + q < hi) {
            const int pose = order ? (int)(order[s0 + q] & ~POSE_INVALID) : s0 + q;
            float gx, gy, thg;
            world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                          poses[3 * (size_t)pose + 2], gx, gy, thg);
            int first, cnt;
            cddt_fan_run(f, cp, lp, thg, td_f, inv_td, first, cnt);
            float *br = bin_range + (size_t)q * td;
            for (int t = t0; t < cp.n_bins; t += lanes_per_pose) {
                // raw bin t searches forward in table bin t; raw bin t + td/2 (if it maps here: >= n_bins)
                // backward
                const int rb = t + half;
                const bool has_b = rb >= cp.n_bins && rb < td;
                int df = t - first, db = rb - first;
                df += df < 0 ? td : 0;
                db += db < 0 ? td : 0;
                const bool need_f = df < cnt, need_b = has_b && db < cnt;
                if (need_f || need_b) {
                    float rf = 1.0f, rbk = 1.0f;
                    if (!(cp.debug & 1)) cddt_query_pair(cp, f.max_range, gx, gy, t, rf, rbk);
                    if (need_f) br[t] = rf * m.res;
                    if (need_b) br[rb] = rbk * m.res;
                }
            }
        }
        __syncthreads();
        for (int qq = 0; qq < pp && s0 + qq < hi; ++qq) {
            const int pose = order ? (int)(order[s0 + qq] & ~POSE_INVALID) : s0 + qq;
            const float thg = poses[3 * (size_t)pose + 2] + m.wa;
            const float *br = bin_range + (size_t)qq * td;
            float *dst = out + (size_t)pose * f.num_rays;
            for (int j = threadIdx.x; j < f.num_rays; j += nt) {
                float r = br[lut_bin_fast(-(thg + fan_alpha(f, j)), lp, td_f, inv_td)];
                if (f.noise_std > 0.0f)
                    r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (size_t)pose * f.num_rays + j);
                if (!(cp.debug & 2) || r == 123.456f) dst[j] = r;
            }
        }
        __syncthreads();
    }
}

// ---- theta-major form (large batches) ------------------------------------------------------------------
// The pose-major kernel above misses L2 on nearly every table line (75 MB of table, 4 MB of L2 per XCD).  Here
// ALL poses go against ONE table bin at a time: a unit of work is (table bin, block of 256 poses), lane = pose;
// the units are laid out bin-major and cut into n_xcd equal runs, run x walked in order by the workgroups
// with blockIdx % n_xcd == x — one XCD under round-robin dispatch — so an XCD works on one bin (two at a
// run's ends) at any time and that bin's slice of the blocked table (1.4 MB at cfg3) stays in its L2.  Both
// ranges of a look-up go to an intermediate R[raw bin][pose] (coalesced along the poses); the second kernel
// turns `ppb` poses x theta_disc bins into fans — the store phase of the pose-major kernel.  Same look-up,
// same bin arithmetic: bit-identical.
constexpr int CDDT_TK = 4;            // look-ups an 8-lane group keeps in flight (theta-major search kernel)

// per pose, once per launch: grid origin and the run of raw bins its fan touches {gx, gy, first, cnt} — the
// search kernel visits every pose once per table bin
__global__ __launch_bounds__(256) void cddt_theta_prep_kernel(MapParams m, FanParams f, CddtParams cp,
                                                              const float *__restrict__ poses, float4 *__restrict__ prep)
{
    LutParams lp{};
    lp.theta_disc = cp.theta_disc;
    lp.bins_per_rad = cp.bins_per_rad;
    const float td_f = (float)cp.theta_disc, inv_td = 1.0f / (float)cp.theta_disc;
    for (int pose = blockIdx.x * 256 + threadIdx.x; pose < f.n_poses; pose += gridDim.x * 256) {
        float gx, gy, thg;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1], poses[3 * (size_t)pose + 2], gx, gy, thg);
        int first, cnt;
        cddt_fan_run(f, cp, lp, thg, td_f, inv_td, first, cnt);
        prep[pose] = make_float4(gx, gy, __builtin_bit_cast(float, first), __builtin_bit_cast(float, cnt));
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(88), amdgpu_waves_per_eu(8, 8)))
void cddt_theta_search_kernel(MapParams m, FanParams f, CddtParams cp, const float *__restrict__ poses,
                              const float4 *__restrict__ prep, float *__restrict__ R, int n_xcd)
{
    const int td = cp.theta_disc, half = td / 2;
    const float INF = __builtin_inff();
    // 8 lanes per look-up (cddt_query_pair8's scheme), CDDT_TK look-ups of consecutive poses per group in flight
    // at once — branch-free: a look-up that needs no line reads line 0 and drops it — so the three dependent
    // loads of a look-up overlap with its neighbours'.  A unit = (table bin, block of 32 x CDDT_TK poses); XCD
    // x owns the WHOLE bins [n_bins * x / n_xcd, n_bins * (x + 1) / n_xcd) and walks them bin by bin.
    constexpr int PB = 32 * CDDT_TK;
    const int n_pb = (f.n_poses + PB - 1) / PB;
    const int x = (int)(blockIdx.x % (unsigned)n_xcd), g = (int)(blockIdx.x / (unsigned)n_xcd);
    const int G = ((int)gridDim.x - x + n_xcd - 1) / n_xcd;
    const int t_lo = (int)((long)cp.n_bins * x / n_xcd), t_hi = (int)((long)cp.n_bins * (x + 1) / n_xcd);
    const long u1 = (long)(t_hi - t_lo) * n_pb;
    const int c = (int)threadIdx.x & 7, grp = (int)threadIdx.x >> 3;
    const float4 *tab4 = reinterpret_cast<const float4 *>(cp.tab);
    for (long u = g; u < u1; u += G) {
        const int t = t_lo + (int)(u / n_pb), p0 = (int)(u % n_pb) * PB + grp * CDDT_TK;
        const float cs = cp.cosv[t], sn = cp.sinv[t], tr = cp.trans[t], wdt = (float)cp.width[t];
        const uint32_t boff = cp.bucket_off[t];
        const int rb = t + half;
        const bool has_b = rb >= cp.n_bins && rb < td;
        float lx[CDDT_TK];
        bool need_f[CDDT_TK], need_b[CDDT_TK], slow[CDDT_TK];
        uint2 hd[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            const float4 pr = prep[min(p0 + k, f.n_poses - 1)];
            const float gx = pr.x, gy = pr.y;
            const int first = __builtin_bit_cast(int, pr.z), cnt = __builtin_bit_cast(int, pr.w);
            int df = t - first, db = rb - first;
            df += df < 0 ? td : 0;
            db += db < 0 ? td : 0;
            const bool live = p0 + k < f.n_poses && !(cp.debug & 1);
            need_f[k] = live && df < cnt;
            need_b[k] = live && has_b && db < cnt;
            lx[k] = __builtin_fmaf(gx, cs, -(gy * sn));
            const float ly = __builtin_fmaf(gx, sn, gy * cs) + tr;
            const bool inside = (need_f[k] || need_b[k]) && ly >= 0.0f && ly < wdt;
            hd[k] = cp.hdr[inside ? boff + (uint32_t)(int)ly : 0u];
            if (!inside) hd[k].y = 0u;                       // (nothing stored: both ranges stay max_range)
        }
        float4 sq[CDDT_TK];
        uint32_t nleaf[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            nleaf[k] = (hd[k].y + 31u) >> 5;
            slow[k] = nleaf[k] > 32u;                        // several separator lines: the general look-up below
            sq[k] = tab4[(nleaf[k] > 1u && !slow[k]) ? (size_t)hd[k].x * 8 + c : (size_t)c];
        }
        float fwd[CDDT_TK];
        float4 lq[CDDT_TK];
        bool have_leaf[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            const bool seps = nleaf[k] > 1u && !slow[k];
            const float x_ = lx[k];
            const float4 q = sq[k];
            const uint32_t cnt = sum8((q.x <= x_ ? 1u : 0u) + (q.y <= x_ ? 1u : 0u) + (q.z <= x_ ? 1u : 0u) +
                                      (q.w <= x_ ? 1u : 0u));
            const float fs = min8(first_gt(q, x_, INF));
            fwd[k] = seps ? fs : INF;
            // one leaf: it is the leaf; separators: the leaf whose first value is the last one <= x (none: cnt 0)
            have_leaf[k] = hd[k].y != 0u && !slow[k] && (!seps || cnt > 0u);
            const uint32_t leaf = seps ? cnt - 1u : 0u;
            lq[k] = tab4[have_leaf[k] ? ((size_t)hd[k].x + (seps ? 1u : 0u) + leaf) * 8 + c : (size_t)c];
        }
        float my_f = 0.0f, my_b = 0.0f;
        bool my_nf = false, my_nb = false;
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            const float x_ = lx[k];
            const float4 q = lq[k];
            const float f4 = min8(first_ge(q, x_, INF));
            const float b4 = max8(last_le(q, x_, -INF));
            const float rf = __builtin_fminf((have_leaf[k] ? __builtin_fminf(fwd[k], f4) : fwd[k]) - x_, f.max_range);
            const float rbk = __builtin_fminf(x_ - (have_leaf[k] ? b4 : -INF), f.max_range);
            // lane k of the group stores look-up k: a wave's 8 groups x CDDT_TK consecutive poses are one line
            if (c == k) {
                my_f = rf;
                my_b = rbk;
                my_nf = need_f[k] && !slow[k];
                my_nb = need_b[k] && !slow[k];
            }
        }
        if (my_nf) R[(size_t)t * f.n_poses + p0 + c] = my_f * m.res;
        if (my_nb) R[(size_t)rb * f.n_poses + p0 + c] = my_b * m.res;
        // buckets beyond 1024 values (several separator lines; a long straight wall along the bin's direction):
        // the general one-lane look-up, rolled
        uint32_t slow_mask = 0;
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) slow_mask |= slow[k] ? (((need_f[k] ? 1u : 0u) | (need_b[k] ? 2u : 0u)) << (2 * k)) : 0u;
#pragma unroll 1
        for (int k = 0; slow_mask >> (2 * k); ++k) {
            const uint32_t need = (slow_mask >> (2 * k)) & 3u;
            if (!need || c != 0) continue;
            const int pose = p0 + k;
            float gx, gy, thg, rf, rbk;
            world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1], poses[3 * (size_t)pose + 2], gx, gy, thg);
            cddt_query_pair(cp, f.max_range, gx, gy, t, rf, rbk);
            if (need & 1u) R[(size_t)t * f.n_poses + pose] = rf * m.res;
            if (need & 2u) R[(size_t)rb * f.n_poses + pose] = rbk * m.res;
        }
    }
}

__global__ __launch_bounds__(256) void cddt_theta_fan_kernel(MapParams m, FanParams f, CddtParams cp,
                                                             const float *__restrict__ poses,
                                                             const float *__restrict__ R, float *__restrict__ out,
                                                             int ppb_log2, int stride)
{
    extern __shared__ float bin_range[];                 // ppb x stride floats (stride: theta_disc made odd)
    const int td = cp.theta_disc, ppb = 1 << ppb_log2;
    LutParams lp{};
    lp.theta_disc = td;
    lp.bins_per_rad = cp.bins_per_rad;
    const float td_f = (float)td, inv_td = 1.0f / (float)td;
    const int n_grp = (f.n_poses + ppb - 1) >> ppb_log2;
    for (int grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int p0 = grp << ppb_log2, np = min(ppb, f.n_poses - p0);
        // (bins the fan does not touch were never written: read as they are, never looked up)
        for (int i = threadIdx.x; i < (td << ppb_log2); i += 256) {
            const int bin = i >> ppb_log2, q = i & (ppb - 1);
            if (q < np) bin_range[q * stride + bin] = R[(size_t)bin * f.n_poses + p0 + q];
        }
        __syncthreads();
        for (int q = 0; q < np; ++q) {
            const int pose = p0 + q;
            const float thg = poses[3 * (size_t)pose + 2] + m.wa;
            const float *br = bin_range + (size_t)q * stride;
            float *dst = out + (size_t)pose * f.num_rays;
            for (int j = threadIdx.x; j < f.num_rays; j += 256) {
                float r = br[lut_bin_fast(-(thg + fan_alpha(f, j)), lp, td_f, inv_td)];
                if (f.noise_std > 0.0f)
                    r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + (size_t)pose * f.num_rays + j);
                if (!(cp.debug & 2) || r == 123.456f) dst[j] = r;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void cddt_fan_kernel(MapParams m, FanParams f, CddtParams cp,
                                                       const float *__restrict__ poses,
                                                       float *__restrict__ out)
{
    for (int pose = blockIdx.x; pose < f.n_poses; pose += gridDim.x) {
        float gx, gy, thg;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                      poses[3 * (size_t)pose + 2], gx, gy, thg);
        for (int j = threadIdx.x; j < f.num_rays; j += blockDim.x) {
            const size_t i = (size_t)pose * f.num_rays + j;
            float r = cddt_query(m, cp, f.max_range, gx, gy, thg + fan_alpha(f, j));
            if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
            out[i] = r;
        }
    }
}

__global__ __launch_bounds__(256) void cddt_rays_kernel(MapParams m, FanParams f, CddtParams cp,
                                                        const float *__restrict__ ins, long n,
                                                        float *__restrict__ out)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gx, gy, thg;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], gx, gy, thg);
        float r = cddt_query(m, cp, f.max_range, gx, gy, thg);
        if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
        out[i] = r;
    }
}

}  // namespace scan
