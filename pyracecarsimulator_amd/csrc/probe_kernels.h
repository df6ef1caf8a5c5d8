// probe_kernels.h — diagnostics: the CU's scattered-gather rate (rl_probe_gather_rate).
//
// The ray-marching kernels are bound by how fast a CU retires a wave-wide global_load_dword whose 64
// lanes read 64 unrelated cells of a cache-resident table.  This is that
// instruction in isolation: every wave of a full machine (2 workgroups of 1024 per CU) gathers random
// cells of a 32x32 window of a small tiled table, 8 independent loads in flight, `active` lanes live.
// bench.py runs it in its untimed section and reports the march kernel's samples/s against it.
#pragma once
#include <hip/hip_runtime.h>

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

}  // namespace scan
