// L1 (TCP) gather microbenchmark #2: 64 lanes sample random cells of a W x W window of the 4-row
// interleaved step map (scan_kernels.h pad_dt_tiled_kernel).  Same multiset of cells, three lane
// orders: random, sorted by address, sorted within each 16-lane quarter.  Also row-major layout.
//   build: hipcc -O2 --offload-arch=gfx950 -o tcp_probe2 tcp_probe2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(1024) void probe(const float *__restrict__ tab, const int *__restrict__ lane_off,
                                              int iters, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const int off = lane_off[lane];
    float acc = 0.f;
    int rot = (threadIdx.x >> 6) & 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += tab[off + ((rot + u) & 3) * 2048];
        rot = (rot + 1) & 3;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    float *tab, *sink;
    int *d_off;
    hipMalloc(&tab, 4 * 2048 * sizeof(float));
    hipMemset(tab, 0, 4 * 2048 * sizeof(float));
    hipMalloc(&sink, 4);
    hipMalloc(&d_off, 64 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000, grid = n_cu * 2;
    printf("%-10s %-8s %3s %6s %8s\n", "layout", "order", "W", "lines", "clk/wave-load/CU (avg of 8 draws)");
    for (int layout = 0; layout < 2; ++layout)
        for (int W : {2, 4, 8, 12, 16, 24, 32})
            for (int order = 0; order < 3; ++order) {
                double sum = 0, lines_sum = 0;
                for (int draw = 0; draw < 8; ++draw) {
                    srand(100 + draw);
                    std::vector<int> off(64);
                    for (int l = 0; l < 64; ++l) {
                        const int r = rand() % W, c = rand() % W + 3;   // (unaligned window)
                        off[l] = layout == 0 ? (r >> 2) * 4 * 64 + 4 * c + (r & 3) : r * 64 + c;
                    }
                    if (order == 1) std::sort(off.begin(), off.end());
                    if (order == 2) for (int q = 0; q < 4; ++q) std::sort(off.begin() + 16 * q, off.begin() + 16 * q + 16);
                    std::vector<int> ln(off);
                    for (auto &x : ln) x >>= 5;
                    std::sort(ln.begin(), ln.end());
                    lines_sum += std::unique(ln.begin(), ln.end()) - ln.begin();
                    hipMemcpy(d_off, off.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
                    hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, tab, d_off, 10, sink);
                    hipDeviceSynchronize();
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, tab, d_off, iters, sink);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    sum += ms * 1e-3 * clk / (2.0 * 16 * iters * 8);
                }
                printf("%-10s %-8s %3d %6.1f %8.2f\n", layout ? "row-major" : "tiled4x8",
                       order == 0 ? "random" : order == 1 ? "sorted" : "sorted16", W, lines_sum / 8, sum / 8);
            }
    return 0;
}
