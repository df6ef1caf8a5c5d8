// L1 (TCP) gather microbenchmark: how many cycles does one wave-wide global_load_dword cost when its
// 64 lanes touch L distinct 128-B lines, and does it matter WHICH lanes share a line?
//   build: hipcc -O2 --offload-arch=gfx950 -o tcp_probe tcp_probe.hip      run: ./tcp_probe
// pattern 0: blocked     lane -> line = lane / (64/L)      (neighbouring lanes share a line)
// pattern 1: interleaved lane -> line = lane % L
// pattern 2: random      lane -> line = perm(lane) / (64/L)
// The table (64 lines = 8 KiB per wave slot, 32 KiB total) stays L1/L2 resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(1024) void probe(const float *__restrict__ tab, const int *__restrict__ lane_off,
                                              int iters, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    // per-lane element offset (line*32 + word); rotate by a wave-uniform amount each iteration so
    // the compiler cannot hoist the loads; the table is 4 x 64 lines so rotations stay in range
    const int off = lane_off[lane];
    float acc = 0.f;
    int rot = (threadIdx.x >> 6) & 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc += tab[off + ((rot + u) & 3) * 2048];
        }
        rot = (rot + 1) & 3;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;   // Hz
    float *tab, *sink;
    int *d_off;
    hipMalloc(&tab, 4 * 2048 * sizeof(float));
    hipMemset(tab, 0, 4 * 2048 * sizeof(float));
    hipMalloc(&sink, 4);
    hipMalloc(&d_off, 64 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000, grid = n_cu * 2;   // 2 x 1024 threads per CU = 32 waves per CU
    printf("CUs %d  clock %.0f MHz\n", n_cu, clk / 1e6);
    printf("%-12s %4s  %10s %14s\n", "pattern", "L", "ms", "clk/wave-load/CU");
    const char *names[] = {"blocked", "interleaved", "random", "blocked+word"};
    for (int pat = 0; pat < 4; ++pat) {
        for (int L = 1; L <= 64; L *= 2) {
            std::vector<int> off(64);
            std::vector<int> perm(64);
            for (int i = 0; i < 64; ++i) perm[i] = i;
            srand(7);
            for (int i = 63; i > 0; --i) { int j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
            for (int lane = 0; lane < 64; ++lane) {
                int line;
                if (pat == 0 || pat == 3) line = lane / (64 / L);
                else if (pat == 1) line = lane % L;
                else line = perm[lane] / (64 / L);
                int word = (pat == 3) ? lane % 32 : 0;   // pattern 3: distinct words within the line
                off[lane] = line * 32 + word;
            }
            hipMemcpy(d_off, off.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, tab, d_off, 10, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, tab, d_off, iters, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double wave_loads_per_cu = 2.0 * 16 * iters * 8;
            printf("%-12s %4d  %10.3f %14.2f\n", names[pat], L, ms, ms * 1e-3 * clk / wave_loads_per_cu);
        }
    }
    return 0;
}
