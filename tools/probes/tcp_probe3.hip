// L1 (TCP/TA) gather microbenchmark #3: cost of a scattered wave-wide global_load_dword as a function
// of WHICH lanes are active (EXEC).  Scattered = random cells of a 32x32 window (~30 lines).
//   build: hipcc -O2 --offload-arch=gfx950 -o tcp_probe3 tcp_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(1024) void probe(const float *__restrict__ tab, const int *__restrict__ lane_off,
                                              unsigned long long mask, int iters, float *__restrict__ sink)
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
    srand(5);
    std::vector<int> off(64);
    for (int l = 0; l < 64; ++l) {
        const int r = rand() % 32, c = rand() % 32 + 3;
        off[l] = (r >> 2) * 4 * 64 + 4 * c + (r & 3);
    }
    hipMemcpy(d_off, off.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    struct { const char *name; unsigned long long m; } pats[] = {
        {"all 64", ~0ull},
        {"lanes 0-31", 0xffffffffull},
        {"lanes 0-15", 0xffffull},
        {"even lanes (32)", 0x5555555555555555ull},
        {"1 per quad (16)", 0x1111111111111111ull},
        {"2 per quad (32)", 0x3333333333333333ull},
        {"3 per quad (48)", 0x7777777777777777ull},
        {"1 per 16 (4)", 0x0001000100010001ull},
        {"random 46", 0},
        {"random 32", 0},
        {"random 16", 0},
    };
    for (auto &p : pats) {
        unsigned long long m = p.m;
        if (!m) {
            int want = atoi(p.name + 7);
            while (__builtin_popcountll(m) < want) m |= 1ull << (rand() % 64);
        }
        hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, tab, d_off, m, 10, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, tab, d_off, m, iters, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double c = ms * 1e-3 * clk / (2.0 * 16 * iters * 8);
        printf("%-18s active %2d   %6.2f clk/wave-load/CU   %5.2f active lanes/clk\n", p.name,
               __builtin_popcountll(m), c, __builtin_popcountll(m) / c);
    }
    return 0;
}
