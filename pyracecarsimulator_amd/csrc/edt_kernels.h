// edt_kernels.h — K0: exact Euclidean distance transform of the occupancy grid (range_libc DistanceTransform,
// SURVEY.md row a7).  Part of scan_kernels.h.
#pragma once
#include "scan_device.h"

namespace scan {

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
// Edge cells of the map (occupied with a free 4-neighbour): the input of every CDDT table of this map, built with
// the other map tables (abi_map.hip: map_build_tables).
// ------------------------------------------------------------------------------
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

// rl_map_stamp_cells: occupied cells laid over the map (the other car's outline of the two-player tick,
// scripts/two_player/rcs_two_player.py:110-116): indices outside the grid are skipped, as the reference's guard skips them
// ONE workgroup: the previous stamp's cells go back to the base map's values, then the new cells are set, and the new
// list is kept (device side) as the next call's "previous" — a tick costs one launch, not a copy of the grid
__global__ __launch_bounds__(1024) void stamp_swap_kernel(uint8_t *__restrict__ occ, const uint8_t *__restrict__ base,
                                                          size_t n_cells, int32_t *__restrict__ prev, int n_prev,
                                                          const int32_t *__restrict__ idx, int n, uint8_t value)
{
    for (int i = threadIdx.x; i < n_prev; i += 1024) {
        const int32_t c = prev[i];
        if (c >= 0 && (size_t)c < n_cells) occ[c] = base[c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int32_t c = idx[i];
        prev[i] = c;
        if (c >= 0 && (size_t)c < n_cells) occ[c] = value;
    }
}

}  // namespace scan
