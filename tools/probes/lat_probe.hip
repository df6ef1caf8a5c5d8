// Dependent-load latency seen by ONE wave (pointer chase, global_load_dword, 1 lane and 64 lanes)
// for footprints that sit in L1 (16 KiB), L2 (2 MiB), Infinity Cache (64 MiB) and HBM (4 GiB).
//   build: hipcc -O2 --offload-arch=gfx950 -o lat_probe lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>

__global__ void chase(const uint32_t *__restrict__ next, int steps, int lanes, uint32_t *out, long long *clk)
{
    if ((int)threadIdx.x >= lanes) return;
    uint32_t p = threadIdx.x * 32;         // every lane its own chain start (a different line)
    // warm: one pass
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}

int main()
{
    uint32_t *d_next, *d_out;
    long long *d_clk;
    const size_t max_elems = (size_t)1 << 30;        // 4 GiB of uint32
    hipMalloc(&d_next, max_elems * 4);
    hipMalloc(&d_out, 64 * 4);
    hipMalloc(&d_clk, 8);
    printf("%-10s %6s %12s\n", "footprint", "lanes", "ns per dependent load (wall_clock64 = 100 MHz)");
    for (size_t bytes : {(size_t)16 << 10, (size_t)256 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)4 << 30}) {
        const size_t lines = bytes / 128;
        // random cyclic permutation over lines; element index = line*32
        std::vector<uint32_t> perm(lines);
        std::iota(perm.begin(), perm.end(), 0u);
        srand(3);
        for (size_t i = lines - 1; i > 0; --i) { size_t j = ((size_t)rand() * RAND_MAX + rand()) % (i + 1); std::swap(perm[i], perm[j]); }
        std::vector<uint32_t> h(lines * 32, 0);
        for (size_t i = 0; i < lines; ++i) {
            const uint32_t nxt = perm[(i + 1) % lines] * 32;
            for (int w = 0; w < 32; ++w) h[(size_t)perm[i] * 32 + w] = nxt;     // any word of the line leads on
        }
        hipMemcpy(d_next, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int lanes : {1, 64}) {
            const int steps = (int)std::min<size_t>(lines * 4, 20000);
            long long c = 0;
            for (int rep = 0; rep < 3; ++rep) {       // 3rd run: caches as warm as the footprint allows
                hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d_next, steps, lanes, d_out, d_clk);
                hipDeviceSynchronize();
            }
            hipMemcpy(&c, d_clk, 8, hipMemcpyDeviceToHost);
            printf("%7zu KiB %6d %12.1f\n", bytes >> 10, lanes, c * 10.0 / steps);
        }
    }
    return 0;
}
