// consumer_kernels.h — per-scan consumers of a batch of ranges.
//
// followgap_kernel: FollowGap::eval (/root/reference/followgap/followgap.hpp:104-129) for a batch
// of scans, ONE WAVE PER SCAN, the scan held in LDS.  The reference is four serial passes over the
// beams; here
//   preprocessLidar (:18-27)  — clamp fused into the load (beams [0, size-10));
//   min_point (:112-119)      — a running strict minimum over the non-zero beams seeded with beam
//                               0 == the lexicographic (value, index) minimum over
//                               {0} ∪ {i : v[i] != 0}: per-lane partial + 6 xor-shuffles;
//   safetyBubble (:67-79)     — 11 lanes zero their beam;
//   findMaxGap (:29-65)       — first longest run of beams > 1.75: every lane owns a contiguous
//                               chunk, a suffix-min over lanes gives the next beam <= 1.75 after the
//                               chunk, a backward walk yields the run length at every run start,
//                               then a (length desc, start asc) wave reduction;
//   getSteerAng (:81-97)      — lane 0, same float/double expression.
// Undefined corners of the reference are given a definition (include/scanlib.h): size < 10 is
// rejected by the host; best == size (gap = the single last beam) reads beam size-1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scan {

struct FollowGapParams {
    float max_distance, max_angle, angle_inc;
    int size;                // beams per scan (>= 10)
};

__global__ __launch_bounds__(64) void followgap_kernel(const float *__restrict__ scans, int n_scans,
                                                       FollowGapParams p, float *__restrict__ angles)
{
    extern __shared__ float v[];                       // size beams of this wave's scan
    const int lane = threadIdx.x;
    const int size = p.size;
    for (int s = blockIdx.x; s < n_scans; s += gridDim.x) {
        const float *lidar = scans + (size_t)s * size;
        // ---- load + clamp; partial (value, index) minimum over the candidates
        const float v0 = lidar[0] > p.max_distance && size > 10 ? p.max_distance : lidar[0];
        float best_v = v0;
        int best_i = 0;
        // (the beams of a lane are loaded eight at a time before any of them is looked at: one memory round trip per
        //  eight instead of one per beam; twenty at a time was slower)
        constexpr int LU = 8;
        for (int i0 = lane; i0 < size; i0 += 64 * LU) {
            float xs[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + 64 * u;
                xs[u] = i < size ? lidar[i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + 64 * u;
                if (i >= size) break;
                float x = xs[u];
                if (i < size - 10 && x > p.max_distance) x = p.max_distance;
                v[i] = x;
                // running rule `v[i] != 0 && v[i] < v[min_point]` (NaN never passes)
                if (x != 0.0f && (x < best_v || (x == best_v && i < best_i))) { best_v = x; best_i = i; }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best_v, off);
            const int oi = __shfl_xor(best_i, off);
            // a NaN seed (v[0]) loses to nothing: every lane carries the same seed, so x < NaN is
            // false everywhere and index 0 survives
            if (ov < best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
        }
        __syncthreads();
        // ---- safety bubble, radius 5
        if (lane < 11) {
            const int idx = best_i + lane - 5;         // lane 10 -> the centre itself
            if (lane == 10) v[best_i] = 0.0f;
            else if (idx > 0 && idx < size - 1) v[idx] = 0.0f;
        }
        __syncthreads();
        // ---- first longest run of beams > 1.75
        const int C = (size + 63) >> 6;
        const int lo = lane * C, hi = min(lo + C, size);
        int first_zero = 0x7fffffff;
        int run_len = 0, run_start = 0;
        constexpr int CM = 20;                             // beams per lane held in registers (scans up to 1280 beams)
        if (C <= CM) {
            // the lane's chunk (and the beam in front of it) in ONE round of independent LDS reads; the two walks below
            // are register arithmetic (they were ~3 C dependent LDS round trips per lane)
            float w[CM];
#pragma unroll
            for (int u = 0; u < CM; ++u) w[u] = lo + u < hi ? v[lo + u] : 0.0f;
            const float before = lo > 0 && lo < size ? v[lo - 1] : 0.0f;
#pragma unroll
            for (int u = CM - 1; u >= 0; --u)
                if (lo + u < hi && !(w[u] > 1.75f)) first_zero = lo + u;
            int nz = first_zero;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(nz, off);
                if (lane + off < 64) nz = min(nz, o);
            }
            int next_zero = __shfl_down(nz, 1);
            if (lane == 63) next_zero = 0x7fffffff;
            next_zero = min(next_zero, size);
#pragma unroll
            for (int u = CM - 1; u >= 0; --u) {
                const int i = lo + u;
                if (i < hi) {
                    if (!(w[u] > 1.75f)) next_zero = i;
                    else if (i == 0 || !((u > 0 ? w[u > 0 ? u - 1 : 0] : before) > 1.75f)) {   // a run starts here
                        const int len = next_zero - i;
                        if (len >= run_len) { run_len = len; run_start = i; }   // walking backwards: ties -> smaller start
                    }
                }
            }
        } else {
            for (int i = lo; i < hi; ++i)
                if (!(v[i] > 1.75f)) { first_zero = i; break; }
            // exclusive suffix-min over lanes: next beam <= 1.75 after this lane's chunk
            int nz = first_zero;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(nz, off);
                if (lane + off < 64) nz = min(nz, o);
            }
            int next_zero = __shfl_down(nz, 1);
            if (lane == 63) next_zero = 0x7fffffff;
            next_zero = min(next_zero, size);
            for (int i = hi - 1; i >= lo; --i) {
                if (!(v[i] > 1.75f)) { next_zero = i; continue; }
                if (i == 0 || !(v[i - 1] > 1.75f)) {       // a run starts here
                    const int len = next_zero - i;
                    if (len >= run_len) { run_len = len; run_start = i; }   // walking backwards: ties -> smaller start
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int ol = __shfl_xor(run_len, off);
            const int os = __shfl_xor(run_start, off);
            if (ol > run_len || (ol == run_len && os < run_start)) { run_len = ol; run_start = os; }
        }
        if (lane == 0) {
            if (run_len == 0) run_start = 0;
            const int best = (run_start + run_start + run_len + 1) / 2;    // findBestPoint(start, start+len+1)
            const float d = lidar[best < size ? best : size - 1];
            float angle;
            if (best > size / 2) angle = (float)(-p.angle_inc * ((size / 2.0) - best));
            else angle = (float)(p.angle_inc * (best - (size / 2.0)));
            angle = 2 * (angle / d);
            const float lo_a = -p.max_angle;
            const float a1 = (angle < lo_a) ? lo_a : angle;
            angles[s] = (p.max_angle < a1) ? p.max_angle : a1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------
// followgap_bits_kernel: the same four passes for scans of up to 64 * FG_ROWS beams, the beams of a scan held in
// registers (beam 64 u + lane in row u of every lane: coalesced loads, all issued before the first is looked at) and
// the gap search done on ONE BIT PER BEAM.  followgap_kernel above spends ~1700 wave instructions per 1081-beam scan
// (two unrolled per-beam walks under lane-divergent branches, LDS round trips, 30 ds_bpermute shuffles), which made
// `--gather steer` 0.70 of the plain scan: with several scans in flight the consumer competes with the march for
// issue slots.  Here:
//   preprocessLidar + min_point — per row: clamp (select), running strict minimum (rows ascend, so within a lane the
//                                 index order is the beam order); one (value, index) butterfly over the wave;
//   findMaxGap                  — `v > 1.75` per row is a v_cmp whose 64-bit result IS 64 consecutive bits of the
//                                 scan's bit string; the rows go through LDS (17 words) and every lane takes a chunk
//                                 of ceil(size / 64) consecutive bits plus the bit in front of it;
//   safetyBubble                — the zeroed beams (best-5 ... best+4) are one contiguous range: cleared in the lane's bits;
//                                 the next zero behind a chunk: a ballot of the lanes that hold one, a 64-bit shift
//                                 and one shuffle; the runs that START in a chunk are walked with ctz (a scan has
//                                 few, the loop runs to the wave's maximum);
//                                 (length desc, start asc) as one packed key, a max-butterfly;
//   getSteerAng                 — the raw beam read back from the registers (v_readlane), lane 0.
// ------------------------------------------------------------------------------
constexpr int FG_ROWS = 20;                                // scans of up to 1280 beams

// v_writelane_b32 (clang has no builtin for it; as the intrinsic, not as inline asm: the compiler must see it to keep the
// two wait states gfx950 needs between a VALU write of an SGPR and a VALU read of it)
extern "C" __device__ int fg_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int fg_dpp_quad1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ int fg_dpp_quad2(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false); }
__device__ __forceinline__ int fg_dpp_half_mirror(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false); }
__device__ __forceinline__ int fg_dpp_mirror(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false); }

// ROWS = ceil(size / 64): the rows below ROWS - 2 hold 64 beams that all exist and are all clamped (size - 10 > 64 (ROWS - 2))
template <int ROWS>
__global__ __launch_bounds__(64) void followgap_bits_kernel(const float *__restrict__ scans, int n_scans,
                                                            FollowGapParams p, float *__restrict__ angles)
{
    static_assert(ROWS >= 1 && ROWS <= FG_ROWS, "one instantiation per row count");
    __shared__ uint32_t bits[2 * ROWS + 4];                // dword 0 = 0 (the bit in front of beam 0), beam i = bit 32 + i
    const int lane = threadIdx.x;
    const int size = p.size;                               // 64 (ROWS - 1) < size <= 64 ROWS (the host picks ROWS)
    constexpr int cl = ROWS;                               // bits per lane: 64 * cl >= size
    for (int s = blockIdx.x; s < n_scans; s += gridDim.x) {
        const float *lidar = scans + (size_t)s * size;
        float raw[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            if (u < ROWS - 1) raw[u] = lidar[64 * u + lane];
            else raw[u] = 64 * u + lane < size ? lidar[64 * u + lane] : 0.0f;
        }
        const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, raw[0])));
        float best_v = x0 > p.max_distance && size > 10 ? p.max_distance : x0;
        int best_u = -1;                                   // row of the lane's minimum (-1: the seed, beam 0)
        uint32_t wlo = 0, whi = 0;                         // lane u: bits of row u
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            float x = raw[u];
            if (u < ROWS - 2) x = x > p.max_distance ? p.max_distance : x;
            else x = (64 * u + lane < size - 10 && x > p.max_distance) ? p.max_distance : x;
            // running rule `v[i] != 0 && v[i] < v[min_point]` (NaN never passes; beams past the end are 0)
            const bool take = x != 0.0f && x < best_v;
            best_v = take ? x : best_v;
            best_u = take ? u : best_u;
            const unsigned long long w = __ballot(x > 1.75f);
            wlo = (uint32_t)fg_writelane((int)(uint32_t)w, u, (int)wlo);
            whi = (uint32_t)fg_writelane((int)(uint32_t)(w >> 32), u, (int)whi);
        }
        int best_i = best_u < 0 ? 0 : 64 * best_u + lane;
        if (lane < ROWS) { bits[1 + 2 * lane] = wlo; bits[2 + 2 * lane] = whi; }
        if (lane == 0) { bits[0] = 0u; bits[1 + 2 * ROWS] = 0u; bits[2 + 2 * ROWS] = 0u; }
        // ---- (value, index) minimum over the wave; a NaN seed is the same in every lane and index 0 survives
        {
            auto step = [&](float ov, int oi) {
                const bool take = ov < best_v || (ov == best_v && oi < best_i);
                best_v = take ? ov : best_v;
                best_i = take ? oi : best_i;
            };
            step(__builtin_bit_cast(float, fg_dpp_quad1(__builtin_bit_cast(int, best_v))), fg_dpp_quad1(best_i));
            step(__builtin_bit_cast(float, fg_dpp_quad2(__builtin_bit_cast(int, best_v))), fg_dpp_quad2(best_i));
            step(__builtin_bit_cast(float, fg_dpp_half_mirror(__builtin_bit_cast(int, best_v))), fg_dpp_half_mirror(best_i));
            step(__builtin_bit_cast(float, fg_dpp_mirror(__builtin_bit_cast(int, best_v))), fg_dpp_mirror(best_i));
            step(__shfl_xor(best_v, 16), __shfl_xor(best_i, 16));
            step(__shfl_xor(best_v, 32), __shfl_xor(best_i, 32));
        }
        __syncthreads();
        // ---- the lane's chunk: beams [a, a + cl), and the beam in front of it as bit 0 of `e`
        const int a = cl * lane;
        const int pos = 31 + a;
        const uint32_t d0 = bits[pos >> 5], d1 = bits[(pos >> 5) + 1];
        uint32_t e = (uint32_t)((((unsigned long long)d1 << 32) | d0) >> (pos & 31)) & ((2u << cl) - 1u);
        // safety bubble (:67-79, `for i = -5; i < 5`): beams best-5 ... best+4 inside (0, size - 1) and the centre itself —
        // one contiguous range [lo_b, hi_b]
        {
            const int lo_b = min(best_i, max(best_i - 5, 1));
            const int hi_b = max(best_i, min(best_i + 4, size - 2));
            const int b0 = max(lo_b - (a - 1), 0), b1 = min(hi_b - (a - 1), cl);
            if (b0 <= b1) e &= ~(((2u << (b1 - b0)) - 1u) << b0);
        }
        const uint32_t full = (1u << cl) - 1u;
        const uint32_t m = (e >> 1) & full;
        // the next beam <= 1.75 behind the chunk: the first lane after this one that holds a zero, its first zero
        const int fz_abs = a + __builtin_ctz(~m);          // (== a + cl when the chunk is all ones; not used then)
        const unsigned long long zl = __ballot(m != full);
        const unsigned long long zs = (zl >> lane) >> 1;
        const int nl = lane + 1 + (zs ? __builtin_ctzll(zs) : 0);
        int next_zero = __shfl(fz_abs, nl & 63);
        next_zero = zs ? min(next_zero, size) : size;
        // ---- the runs that start in this chunk
        uint32_t starts = m & ~(e & full);                 // bit u: beam a+u > 1.75 and the beam in front of it is not
        uint32_t key = 0;
        while (__ballot(starts != 0)) {
            if (starts) {
                const int st = __builtin_ctz(starts);
                starts &= starts - 1;
                const int z = __builtin_ctz(~(m >> st));   // bits >= cl of m are 0: z <= cl - st
                const int len = st + z < cl ? z : next_zero - (a + st);
                const uint32_t k = ((uint32_t)len << 11) | (uint32_t)(2047 - (a + st));
                key = k > key ? k : key;
            }
        }
        {
            auto mx = [&](int o) { key = (uint32_t)o > key ? (uint32_t)o : key; };
            mx(fg_dpp_quad1((int)key));
            mx(fg_dpp_quad2((int)key));
            mx(fg_dpp_half_mirror((int)key));
            mx(fg_dpp_mirror((int)key));
            mx(__shfl_xor((int)key, 16));
            mx(__shfl_xor((int)key, 32));
        }
        key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);         // every lane holds the maximum: scalar from here
        const int run_len = (int)(key >> 11);
        const int run_start = run_len ? 2047 - (int)(key & 2047u) : 0;
        const int best = (run_start + run_start + run_len + 1) / 2;        // findBestPoint(start, start+len+1), uniform
        const int bi = best < size ? best : size - 1;
        float d = 0.0f;
#pragma unroll
        for (int u = 0; u < ROWS; ++u)
            if (u == (bi >> 6)) d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, raw[u]), bi & 63));
        if (lane == 0) {
            float angle;
            if (best > size / 2) angle = (float)(-p.angle_inc * ((size / 2.0) - best));
            else angle = (float)(p.angle_inc * (best - (size / 2.0)));
            angle = 2 * (angle / d);
            const float lo_a = -p.max_angle;
            const float a1 = (angle < lo_a) ? lo_a : angle;
            angles[s] = (p.max_angle < a1) ? p.max_angle : a1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------
// Ranges as 16-bit fixed point for the multi-GPU exchange (opt-in, LOSSY, labelled wherever it is used):
// q = rint(clamp(r, 0, max) * 65535 / max), r' = q * max / 65535 — half the bytes of the all-gather of
// ranges over xGMI (BASELINE.json north_star) at max/65535/2 = 0.11 mm of error for the reference's 15 m
// lidar (params.yaml:39), three orders of magnitude inside north_star's one-cell (50 mm) tolerance.
// HBM-bound streaming passes: 8 ranges per lane and trip (two 16-B loads, one 16-B store).
// ------------------------------------------------------------------------------
// `head` leading elements (and everything, when vec == 0) go one by one: slices of a larger buffer start
// wherever a pose block starts, and the 16-B forms need both pointers aligned at the same element
__global__ __launch_bounds__(256) void ranges_to_u16_kernel(const float *__restrict__ r, size_t n, float max_m,
                                                            float scale, uint16_t *__restrict__ q, size_t head, int vec)
{
    auto cv = [&](float v) { return (uint32_t)__builtin_rintf(__builtin_fminf(__builtin_fmaxf(v, 0.0f), max_m) * scale); };
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if (!vec) head = n;
    for (size_t k = tid; k < head; k += nthr) q[k] = (uint16_t)cv(r[k]);
    for (size_t i = head + tid * 8; i < n; i += nthr * 8) {
        if (i + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4 *>(r + i);
            const float4 b = *reinterpret_cast<const float4 *>(r + i + 4);
            uint4 o;
            o.x = cv(a.x) | (cv(a.y) << 16);
            o.y = cv(a.z) | (cv(a.w) << 16);
            o.z = cv(b.x) | (cv(b.y) << 16);
            o.w = cv(b.z) | (cv(b.w) << 16);
            *reinterpret_cast<uint4 *>(q + i) = o;
        } else {
            for (size_t k = i; k < n; ++k) q[k] = (uint16_t)cv(r[k]);
        }
    }
}

__global__ __launch_bounds__(256) void ranges_from_u16_kernel(const uint16_t *__restrict__ q, size_t n, float inv_scale,
                                                              float *__restrict__ r, size_t head, int vec)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if (!vec) head = n;
    for (size_t k = tid; k < head; k += nthr) r[k] = (float)q[k] * inv_scale;
    for (size_t i = head + tid * 8; i < n; i += nthr * 8) {
        if (i + 8 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(q + i);
            float4 a, b;
            a.x = (float)(v.x & 0xffffu) * inv_scale; a.y = (float)(v.x >> 16) * inv_scale;
            a.z = (float)(v.y & 0xffffu) * inv_scale; a.w = (float)(v.y >> 16) * inv_scale;
            b.x = (float)(v.z & 0xffffu) * inv_scale; b.y = (float)(v.z >> 16) * inv_scale;
            b.z = (float)(v.w & 0xffffu) * inv_scale; b.w = (float)(v.w >> 16) * inv_scale;
            *reinterpret_cast<float4 *>(r + i) = a;
            *reinterpret_cast<float4 *>(r + i + 4) = b;
        } else {
            for (size_t k = i; k < n; ++k) r[k] = (float)q[k] * inv_scale;
        }
    }
}

}  // namespace scan
