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
