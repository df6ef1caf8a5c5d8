// probe_kernels.h — diagnostics: the CU's scattered-gather rate (rl_probe_gather_rate).
//
// The ray-marching kernels are bound by how fast a CU retires a wave-wide global_load_dword whose 64
// lanes read 64 unrelated cells of a cache-resident table.  This is that
// instruction in isolation: every wave of a full machine (2 workgroups of 1024 per CU) gathers random
// cells of a 32x32 window of a small tiled table, 8 independent loads in flight, `active` lanes live.
// bench.py runs it in its untimed section and reports the march kernel's samples/s against it.
#pragma once
#include <hip/hip_runtime.h>
#include "literal_math.h"

namespace scan {

__global__ __launch_bounds__(1024) void gather_probe_kernel(const float *__restrict__ tab,
                                                            const int *__restrict__ lane_off,
                                                            unsigned long long mask, int iters,
                                                            float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const int off = lane_off[lane];
    float acc = 0.f;
    int rot = (threadIdx.x >> 6) & 3;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += tab[off + ((rot + u) & 3) * 2048];
            rot = (rot + 1) & 3;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// ------------------------------------------------------------------------------
// HBM stream probe (rl_probe_hbm): what a hand-written 16-B-per-lane grid-stride kernel moves on this box —
// the practical ceiling the bandwidth-bound kernels (GiantLUT) are held against.  tools/hbm_probe.py measured
// torch's elementwise kernels (copy 4.7, fill 6.8, sum 4.0 TB/s); MI355X_MICROARCH.md quotes 6.29 TB/s for a
// float4 copy: these kernels are that copy (mode 0), a read-only sweep (1), a write-only fill (2), and the
// copy / fill with non-temporal stores (3, 4); rl_probe_hbm_nt: the read-only sweep, the copy and the copy with
// non-temporal stores, all three with NON-TEMPORAL loads (5, 6, 7) — what GiantLUT's row fetch uses since round 4.
// Persistent grid (8 workgroups of 256 per CU), 4 x 16 B in flight per lane.
// ------------------------------------------------------------------------------
typedef unsigned int probe_v4u __attribute__((ext_vector_type(4)));   // (the non-temporal builtin wants a native vector)

__device__ __forceinline__ void nt_store16(uint4 *p, const uint4 &v)
{
    probe_v4u w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<probe_v4u *>(p));
}

__device__ __forceinline__ uint4 nt_load16(const uint4 *p)
{
    const probe_v4u w = __builtin_nontemporal_load(reinterpret_cast<const probe_v4u *>(p));
    return make_uint4(w.x, w.y, w.z, w.w);
}

__global__ __launch_bounds__(256) void hbm_probe_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                        int mode, uint32_t *__restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    const uint4 fillv = make_uint4(1u, 2u, 3u, 4u);
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = fillv, b = fillv, c = fillv, d = fillv;
        if (mode == 0 || mode == 1 || mode == 3) {
            a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride];
        } else if (mode >= 5) {       // 5: read-only, 6: copy, 7: copy with non-temporal stores — NON-TEMPORAL loads
            a = nt_load16(src + i); b = nt_load16(src + i + stride); c = nt_load16(src + i + 2 * stride);
            d = nt_load16(src + i + 3 * stride);
        }
        if (mode == 1 || mode == 5) {
            acc ^= a.x ^ b.y ^ c.z ^ d.w;
        } else if (mode == 3 || mode == 4 || mode == 7) {
            nt_store16(dst + i, a);
            nt_store16(dst + i + stride, b);
            nt_store16(dst + i + 2 * stride, c);
            nt_store16(dst + i + 3 * stride, d);
        } else {
            dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
        }
    }
    for (; i < n16; i += stride) {
        uint4 a = fillv;
        if (mode == 0 || mode == 1 || mode == 3 || mode >= 5) a = src[i];
        if (mode == 1 || mode == 5) acc ^= a.x;
        else dst[i] = a;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// diagnostics: lit_sinf / lit_cosf of an array (rl_probe_literal_sincosf: the GPU test holds it against the host's libm)
__global__ __launch_bounds__(256) void literal_sincosf_kernel(const float *__restrict__ x, long n, float *__restrict__ s,
                                                              float *__restrict__ c)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        s[i] = lit_sinf(x[i]);
        c[i] = lit_cosf(x[i]);
    }
}

}  // namespace scan
