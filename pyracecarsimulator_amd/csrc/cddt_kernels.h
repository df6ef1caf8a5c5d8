// cddt_kernels.h — K3b: CDDTCast: device table build (blocked layout), the pose-major fan
// kernel, the theta-major kernels for large batches, the per-ray kernels.  Part of scan_kernels.h.
#pragma once
#include "scan_device.h"
#include "scan_params.h"
#include "rm_kernels.h"
#include "lut_kernels.h"

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

// Eight independent 8-lane reductions, hand-interleaved (the theta-major search kernel reduces two values of
// each of its four look-ups per stage): the DPP permutation rides on the min / max / add itself, and step s of
// a chain sits eight instructions behind step s-1 — no DPP read-after-write no-ops, no canonicalising copies.
#define CDDT_DPP3(OPA, OPB, CTRL)                                                                     \
    OPA " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OPB " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    OPA " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OPB " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    OPA " %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OPB " %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    OPA " %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OPB " %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
// (count, min) x 4
__device__ __forceinline__ void red8_add_min_x4(uint32_t &n0, float &f0, uint32_t &n1, float &f1, uint32_t &n2, float &f2,
                                                uint32_t &n3, float &f3)
{
    asm volatile("s_nop 1\n\t"
                 CDDT_DPP3("v_add_u32_dpp", "v_min_f32_dpp", "quad_perm:[1,0,3,2]")
                 CDDT_DPP3("v_add_u32_dpp", "v_min_f32_dpp", "quad_perm:[2,3,0,1]")
                 CDDT_DPP3("v_add_u32_dpp", "v_min_f32_dpp", "row_half_mirror")
                 : "+v"(n0), "+v"(f0), "+v"(n1), "+v"(f1), "+v"(n2), "+v"(f2), "+v"(n3), "+v"(f3));
}
// (min, max) x 4
__device__ __forceinline__ void red8_min_max_x4(float &a0, float &b0, float &a1, float &b1, float &a2, float &b2, float &a3,
                                                float &b3)
{
    asm volatile("s_nop 1\n\t"
                 CDDT_DPP3("v_min_f32_dpp", "v_max_f32_dpp", "quad_perm:[1,0,3,2]")
                 CDDT_DPP3("v_min_f32_dpp", "v_max_f32_dpp", "quad_perm:[2,3,0,1]")
                 CDDT_DPP3("v_min_f32_dpp", "v_max_f32_dpp", "row_half_mirror")
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
}
#undef CDDT_DPP3
// min / max of two finite-or-infinite floats without the canonicalising copies fminf / fmaxf carry
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// ... the same two written on the ONE set of compares (q_i <= x) that count_le / last_le use (round 6: the search kernel
// issued 64 v_cmp per round of 32 look-ups, half of them the mirror image of the other half): the values are finite or
// +inf and x is finite, so "not <=" is ">"
__device__ __forceinline__ float first_gt_le(const float4 &q, float x, float none)
{
    float r = !(q.w <= x) ? q.w : none;
    r = !(q.z <= x) ? q.z : r;
    r = !(q.y <= x) ? q.y : r;
    return !(q.x <= x) ? q.x : r;
}
// how many of a lane's four ascending values are <= x
__device__ __forceinline__ uint32_t count_le(const float4 &q, float x)
{
    uint32_t n = q.x <= x ? 1u : 0u;
    n = q.y <= x ? 2u : n;
    n = q.z <= x ? 3u : n;
    return q.w <= x ? 4u : n;
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
    // 8 lanes per look-up (lane c reads float4 #c of a line), CDDT_TK look-ups of consecutive poses per group in flight
    // at once — branch-free: a look-up that needs no line reads line 0 and drops it — so the three dependent
    // loads of a look-up overlap with its neighbours'.  A unit = (table bin, block of 32 x CDDT_TK poses); XCD
    // x owns the WHOLE bins [n_bins * x / n_xcd, n_bins * (x + 1) / n_xcd) and walks them bin by bin.
    constexpr int PB = 32 * CDDT_TK;
    const int n_pb = (f.n_poses + PB - 1) / PB;
    const int x = (int)(blockIdx.x % (unsigned)n_xcd), g = (int)(blockIdx.x / (unsigned)n_xcd);
    const int G = ((int)gridDim.x - x + n_xcd - 1) / n_xcd;
    const int t_lo = (int)((long)cp.n_bins * x / n_xcd), t_hi = (int)((long)cp.n_bins * (x + 1) / n_xcd);
    const uint32_t u1 = (uint32_t)(t_hi - t_lo) * (uint32_t)n_pb;      // (< 2^31: the planner bounds poses x theta_disc)
    const int c = (int)threadIdx.x & 7, grp = (int)threadIdx.x >> 3;
    const float4 *tab4 = reinterpret_cast<const float4 *>(cp.tab);
    for (uint32_t u = (uint32_t)g; u < u1; u += (uint32_t)G) {
        const uint32_t ut = u / (uint32_t)n_pb;
        const int t = t_lo + (int)ut, p0 = (int)(u - ut * (uint32_t)n_pb) * PB + grp * CDDT_TK;
        const float cs = cp.cosv[t], sn = cp.sinv[t], tr = cp.trans[t], wdt = (float)cp.width[t];
        const uint32_t boff = cp.bucket_off[t];
        const int rb = t + half;
        const bool has_b = rb >= cp.n_bins && rb < td;
        static_assert(CDDT_TK == 4, "the reductions below are written for four look-ups per group");
        float lx[CDDT_TK];
        bool need_f[CDDT_TK], need_b[CDDT_TK], slow[CDDT_TK];
        uint2 hd[CDDT_TK];
        const bool dbg_off = !(cp.debug & 1);
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            const float4 pr = prep[min(p0 + k, f.n_poses - 1)];
            const float gx = pr.x, gy = pr.y;
            const int first = __builtin_bit_cast(int, pr.z), cnt = __builtin_bit_cast(int, pr.w);
            int df = t - first, db = rb - first;
            df += df < 0 ? td : 0;
            db += db < 0 ? td : 0;
            const bool live = (p0 + k < f.n_poses) & dbg_off;
            need_f[k] = live & (df < cnt);
            need_b[k] = live & has_b & (db < cnt);
            lx[k] = __builtin_fmaf(gx, cs, -(gy * sn));
            const float ly = __builtin_fmaf(gx, sn, gy * cs) + tr;
            const bool inside = (need_f[k] | need_b[k]) & (ly >= 0.0f) & (ly < wdt);
            hd[k] = cp.hdr[inside ? boff + (uint32_t)(int)ly : 0u];
            hd[k].y = inside ? hd[k].y : 0u;                 // (nothing stored: both ranges stay max_range)
        }
        float4 sq[CDDT_TK];
        uint32_t nleaf[CDDT_TK];
        bool seps[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            nleaf[k] = (hd[k].y + 31u) >> 5;
            slow[k] = nleaf[k] > 32u;                        // several separator lines: the general look-up below
            seps[k] = (nleaf[k] > 1u) & !slow[k];
            sq[k] = tab4[seps[k] ? (size_t)hd[k].x * 8 + c : (size_t)c];
        }
        uint32_t cn[CDDT_TK];
        float fs[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            cn[k] = count_le(sq[k], lx[k]);
            fs[k] = first_gt(sq[k], lx[k], INF);
        }
        red8_add_min_x4(cn[0], fs[0], cn[1], fs[1], cn[2], fs[2], cn[3], fs[3]);
        float4 lq[CDDT_TK];
        bool have_leaf[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            fs[k] = seps[k] ? fs[k] : INF;
            // one leaf: it is the leaf; separators: the leaf whose first value is the last one <= x (none: count 0)
            have_leaf[k] = (hd[k].y != 0u) & !slow[k] & (!seps[k] | (cn[k] > 0u));
            // (a NaN origin counts the +inf padding as well: stay inside the bucket's lines)
            const uint32_t leaf = seps[k] ? min(cn[k], nleaf[k]) - 1u : 0u;
            lq[k] = tab4[have_leaf[k] ? ((size_t)hd[k].x + (seps[k] ? 1u : 0u) + leaf) * 8 + c : (size_t)c];
        }
        float f4[CDDT_TK], b4[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            f4[k] = first_ge(lq[k], lx[k], INF);
            b4[k] = last_le(lq[k], lx[k], -INF);
        }
        red8_min_max_x4(f4[0], b4[0], f4[1], b4[1], f4[2], b4[2], f4[3], b4[3]);
        float my_f = 0.0f, my_b = 0.0f;
        bool my_nf = false, my_nb = false;
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            const float x_ = lx[k];
            const float rf = vmin((have_leaf[k] ? vmin(fs[k], f4[k]) : fs[k]) - x_, f.max_range);
            const float rbk = vmin(x_ - (have_leaf[k] ? b4[k] : -INF), f.max_range);
            // lane k of the group stores look-up k: a wave's 8 groups x CDDT_TK consecutive poses are one line
            const bool mine = c == k;
            my_f = mine ? rf : my_f;
            my_b = mine ? rbk : my_b;
            my_nf = mine ? (need_f[k] & !slow[k]) : my_nf;
            my_nb = mine ? (need_b[k] & !slow[k]) : my_nb;
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

// Round 5: the same look-ups with the per-look-up PREPARATION done once instead of eight times.  The kernel above is
// VALU-bound (~18 wave instructions per look-up, 66 M per cfg3 launch = 107 us of issue in a 118-us kernel,
// profiles/r04/kt_cfg3_cddt): all 8 lanes of a group — and every lane for all four look-ups of its group — repeat the
// projection, the fan-run test, the bucket header fetch and its decoding.  Here lane i of a wave prepares look-up i
// (pose p0 + i: one coalesced 1-KiB record read per wave, one projection, ONE header gather per look-up), and the
// 8-lane groups pick their look-up's three words {lx, first line, count | flags} from the preparing lane with
// ds_bpermute: 64 look-ups per wave iteration in two rounds of 8 groups x CDDT_TK.  Same arithmetic, same table
// lines, same insertion points: bit-identical.  A unit = (table bin, 256 poses).
// One table bin t for the 64 poses p0 .. p0 + 63 (lane i prepares pose p0 + i; `pr` = its {gx, gy, first raw bin, count}
// from cddt_theta_prep_kernel): both raw bins the look-up answers go to sink(raw bin, pose, range in metres).
template <class Sink>
__device__ __forceinline__ void cddt_theta_search_bin(const MapParams &m, const FanParams &f, const CddtParams &cp,
                                                      const float4 pr, int p0, int t, Sink &&sink)
{
    const int td = cp.theta_disc, half = td / 2;
    const float INF = __builtin_inff();
    const int lane = (int)threadIdx.x & 63;
    const int c = lane & 7, grp = lane >> 3;
    const float4 *tab4 = reinterpret_cast<const float4 *>(cp.tab);
    const bool dbg_off = !(cp.debug & 1);
    static_assert(CDDT_TK == 4, "the reductions below are written for four look-ups per group");
    const float cs = cp.cosv[t], sn = cp.sinv[t], tr = cp.trans[t], wdt = (float)cp.width[t];
    const uint32_t boff = cp.bucket_off[t];
    const int rb = t + half;
    const bool has_b = rb >= cp.n_bins && rb < td;
    // ---- preparation: lane i <-> pose p0 + i
    const int pose = p0 + lane;
    const int first = __builtin_bit_cast(int, pr.z), cnt = __builtin_bit_cast(int, pr.w);
    int df = t - first, db = rb - first;
    df += df < 0 ? td : 0;
    db += db < 0 ? td : 0;
    const bool live = (pose < f.n_poses) & dbg_off;
    const bool nf = live & (df < cnt), nbk = live & has_b & (db < cnt);
    const float my_lx = __builtin_fmaf(pr.x, cs, -(pr.y * sn));
    const float ly = __builtin_fmaf(pr.x, sn, pr.y * cs) + tr;
    const bool inside = (nf | nbk) & (ly >= 0.0f) & (ly < wdt);
    uint2 hd = cp.hdr[inside ? boff + (uint32_t)(int)ly : 0u];
    hd.y = inside ? hd.y : 0u;                             // (nothing stored: both ranges stay max_range)
    const bool my_slow = ((hd.y + 31u) >> 5) > 32u;        // several separator lines: the general look-up below
    // the three words a group needs of its look-up: lx, first line, count | need flags (counts stay below 2^29)
    const int w_lx = __builtin_bit_cast(int, my_lx), w_line = (int)hd.x;
    const int w_cnt = (int)((my_slow ? 0u : hd.y) | (nf ? 0x80000000u : 0u) | (nbk ? 0x40000000u : 0u) |
                            (my_slow ? 0x20000000u : 0u));
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        float lx[CDDT_TK];
        uint32_t line0[CDDT_TK], nval[CDDT_TK], flg[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            const int src = (r * 32 + k * 8 + grp) << 2;   // the lane that prepared this group's look-up k
            lx[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, w_lx));
            line0[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, w_line);
            const uint32_t wc = (uint32_t)__builtin_amdgcn_ds_bpermute(src, w_cnt);
            nval[k] = wc & 0x1fffffffu;
            flg[k] = wc >> 29;                             // bit 2 need_f, bit 1 need_b, bit 0 slow
        }
        float4 sq[CDDT_TK];
        uint32_t nleaf[CDDT_TK];
        bool seps[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            nleaf[k] = (nval[k] + 31u) >> 5;
            seps[k] = nleaf[k] > 1u;
            sq[k] = tab4[seps[k] ? (size_t)line0[k] * 8 + c : (size_t)c];
        }
        uint32_t cn[CDDT_TK];
        float fs[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            cn[k] = count_le(sq[k], lx[k]);
            fs[k] = first_gt_le(sq[k], lx[k], INF);
        }
        red8_add_min_x4(cn[0], fs[0], cn[1], fs[1], cn[2], fs[2], cn[3], fs[3]);
        float4 lq[CDDT_TK];
        bool have_leaf[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            fs[k] = seps[k] ? fs[k] : INF;
            have_leaf[k] = (nval[k] != 0u) & (!seps[k] | (cn[k] > 0u));
            const uint32_t leaf = seps[k] ? min(cn[k], nleaf[k]) - 1u : 0u;
            lq[k] = tab4[have_leaf[k] ? ((size_t)line0[k] + (seps[k] ? 1u : 0u) + leaf) * 8 + c : (size_t)c];
        }
        float f4[CDDT_TK], b4[CDDT_TK];
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            // (first value > x and last value <= x on the same four compares; the first value >= x follows behind the
            //  reduction: it is x itself exactly when the last value <= x is x)
            f4[k] = first_gt_le(lq[k], lx[k], INF);
            b4[k] = last_le(lq[k], lx[k], -INF);
        }
        red8_min_max_x4(f4[0], b4[0], f4[1], b4[1], f4[2], b4[2], f4[3], b4[3]);
        // lane k of the group stores look-up k: its five words are picked first, the ranges computed once
        float m_fs = INF, m_f4 = INF, m_b4 = -INF, m_x = 0.0f;
        uint32_t my_flg = 0;
        bool m_leaf = false;
#pragma unroll
        for (int k = 0; k < CDDT_TK; ++k) {
            const bool mine = c == k;
            m_fs = mine ? fs[k] : m_fs;
            m_f4 = mine ? f4[k] : m_f4;
            m_b4 = mine ? b4[k] : m_b4;
            m_x = mine ? lx[k] : m_x;
            m_leaf = mine ? have_leaf[k] : m_leaf;
            my_flg = mine ? flg[k] : my_flg;
        }
        m_f4 = m_b4 == m_x ? m_x : m_f4;                   // first value >= x of the leaf
        const float my_f = vmin((m_leaf ? vmin(m_fs, m_f4) : m_fs) - m_x, f.max_range);
        const float my_b = vmin(m_x - (m_leaf ? m_b4 : -INF), f.max_range);
        // the round's 32 look-ups are the poses p0 + r * 32 + k * 8 + grp: lanes (grp, c = k < 4) write one 128-B line
        const int pq = p0 + r * 32 + c * 8 + grp;
        if (c < CDDT_TK && (my_flg & 1u) == 0u) {
            if (my_flg & 4u) sink(t, pq, my_f * m.res);
            if (my_flg & 2u) sink(rb, pq, my_b * m.res);
        }
    }
    // buckets beyond 1024 values (several separator lines; a long straight wall along the bin's direction): the
    // general one-lane look-up, by the lane that prepared the pose
    if (my_slow && (nf | nbk)) {
        float rf, rbk;
        cddt_query_pair(cp, f.max_range, pr.x, pr.y, t, rf, rbk);
        if (nf) sink(t, pose, rf * m.res);
        if (nbk) sink(rb, pose, rbk * m.res);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(88), amdgpu_waves_per_eu(8, 8)))
void cddt_theta_search2_kernel(MapParams m, FanParams f, CddtParams cp, const float *__restrict__ poses,
                               const float4 *__restrict__ prep, float *__restrict__ R, int n_xcd)
{
    constexpr int PB = 256;
    const int n_pb = (f.n_poses + PB - 1) / PB;
    const int x = (int)(blockIdx.x % (unsigned)n_xcd), g = (int)(blockIdx.x / (unsigned)n_xcd);
    const int G = ((int)gridDim.x - x + n_xcd - 1) / n_xcd;
    const int t_lo = (int)((long)cp.n_bins * x / n_xcd), t_hi = (int)((long)cp.n_bins * (x + 1) / n_xcd);
    const uint32_t u1 = (uint32_t)(t_hi - t_lo) * (uint32_t)n_pb;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    for (uint32_t u = (uint32_t)g; u < u1; u += (uint32_t)G) {
        const uint32_t ut = u / (uint32_t)n_pb;
        const int t = t_lo + (int)ut, p0 = (int)(u - ut * (uint32_t)n_pb) * PB + wave * 64;
        if (p0 >= f.n_poses) continue;                         // (wave-uniform: this wave's 64 poses do not exist)
        const float4 pr = prep[min(p0 + lane, f.n_poses - 1)];
        cddt_theta_search_bin(m, f, cp, pr, p0, t,
                              [&](int bin, int pose, float v) { R[(size_t)bin * f.n_poses + pose] = v; });
    }
}

// The ranges of the poses p0 .. p0 + np - 1 from their per-bin results in LDS (bin_range[q * stride + raw bin], headings
// thg_l[q]): every beam looks its bin up.
__device__ __forceinline__ void cddt_theta_fan_group(const FanParams &f, const CddtParams &cp, const LutParams &lp, float td_f,
                                                     float inv_td, const float *bin_range, const float *thg_l, int stride,
                                                     int p0, int np, float *__restrict__ out)
{
    // the group's ranges are ONE contiguous run of np x num_rays floats (pose-major output): written 16 B per
    // lane — 1 KiB per wave instruction — when the run starts on a 16-B boundary (always for the planner's
    // groups of >= 4 poses on an aligned buffer), else beam by beam.  (Round 4: the dword row stores of the
    // first form ran at 4.7 TB/s, profiles/r04/write_probe.txt.)
    const size_t run0 = (size_t)p0 * f.num_rays;
    const uint32_t B = (uint32_t)f.num_rays, total = (uint32_t)np * B;
    float *run = out + run0;
    auto lookup = [&](uint32_t q, uint32_t j, uint32_t e) {
        float r = bin_range[(size_t)q * stride + lut_bin_fast(-(thg_l[q] + fan_alpha(f, (int)j)), lp, td_f, inv_td)];
        if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + run0 + e);
        return r;
    };
    if ((reinterpret_cast<uintptr_t>(run) & 15) == 0 && !(cp.debug & 2)) {
        const float inv_b = 1.0f / (float)B;
        for (uint32_t c = threadIdx.x; c < (total >> 2); c += 256) {
            const uint32_t e = c << 2;
            uint32_t q = (uint32_t)(((float)e + 0.5f) * inv_b);          // e / B (e < 2^24: one correction step)
            q -= (q * B > e) ? 1u : 0u;
            q += ((q + 1u) * B <= e) ? 1u : 0u;
            uint32_t j = e - q * B;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j == B) { j = 0; ++q; }
                v[k] = lookup(q, j, e + (uint32_t)k);
                ++j;
            }
            *reinterpret_cast<float4 *>(run + e) = make_float4(v[0], v[1], v[2], v[3]);
        }
        const uint32_t e = (total & ~3u) + threadIdx.x;                     // the run's last 0..3 floats
        if (threadIdx.x < (total & 3u)) {
            const uint32_t q = e / B;
            run[e] = lookup(q, e - q * B, e);
        }
    } else {
        for (uint32_t q = 0; q < (uint32_t)np; ++q)
            for (uint32_t j = threadIdx.x; j < B; j += 256) {
                const float r = lookup(q, j, q * B + j);
                if (!(cp.debug & 2) || r == 123.456f) run[q * B + j] = r;
            }
    }
}

__global__ __launch_bounds__(256) void cddt_theta_fan_kernel(MapParams m, FanParams f, CddtParams cp,
                                                             const float *__restrict__ poses,
                                                             const float *__restrict__ R, float *__restrict__ out,
                                                             int ppb_log2, int stride)
{
    extern __shared__ float bin_range[];                 // ppb x stride floats (stride: theta_disc made odd), then ppb headings
    const int td = cp.theta_disc, ppb = 1 << ppb_log2;
    float *thg_l = bin_range + ((size_t)stride << ppb_log2);
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
        if ((int)threadIdx.x < np) thg_l[threadIdx.x] = poses[3 * (size_t)(p0 + (int)threadIdx.x) + 2] + m.wa;
        __syncthreads();
        cddt_theta_fan_group(f, cp, lp, td_f, inv_td, bin_range, thg_l, stride, p0, np, out);
        __syncthreads();
    }
}


// Search and fan of a tile of 64 poses in ONE workgroup (round 5): the four waves share the tile's table bins (wave w:
// bins w, w + 4, ...), every look-up's two raw-bin results go straight into the LDS array the fan stage reads — R[bin][pose]
// never exists in memory (29 MB written and read back per cfg3 step by the two-kernel form) and the fan's stores of one
// tile overlap the searches of the tiles next to it on the same CU (29 KB of LDS per tile: five tiles per CU).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8)))
void cddt_theta_fused_kernel(MapParams m, FanParams f, CddtParams cp, const float *__restrict__ poses,
                             const float4 *__restrict__ prep, float *__restrict__ out, int stride)
{
    extern __shared__ float bin_range[];                 // 64 x stride floats (stride: theta_disc made odd), then 64 headings
    constexpr int TP = 64;
    float *thg_l = bin_range + (size_t)stride * TP;
    const int td = cp.theta_disc;
    LutParams lp{};
    lp.theta_disc = td;
    lp.bins_per_rad = cp.bins_per_rad;
    const float td_f = (float)td, inv_td = 1.0f / (float)td;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int n_tiles = (f.n_poses + TP - 1) / TP;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int p0 = tile * TP, np = min(TP, f.n_poses - p0);
        const float4 pr = prep[min(p0 + lane, f.n_poses - 1)];
        if (wave == 0 && lane < np) thg_l[lane] = poses[3 * (size_t)(p0 + lane) + 2] + m.wa;
        for (int t = wave; t < cp.n_bins; t += 4)
            cddt_theta_search_bin(m, f, cp, pr, p0, t,
                                  [&](int bin, int pose, float v) { bin_range[(pose - p0) * stride + bin] = v; });
        __syncthreads();
        cddt_theta_fan_group(f, cp, lp, td_f, inv_td, bin_range, thg_l, stride, p0, np, out);
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
