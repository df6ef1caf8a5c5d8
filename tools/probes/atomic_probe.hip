// atomic_probe.hip — what a device-scope returning atomic costs on MI355X, the design input of the band-shared
// work pool of rm_fan_stream_kernel (round 5): (1) latency of one dependent chain from one lane on an idle machine,
// (2) throughput when W workgroups (one lane each, the shape of a workgroup-level claim) hammer ONE word, 8 words
// (one per band), 32 words (4 sub-pools per band), workgroup b on word b % words.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void chain_kernel(unsigned *ctr, int iters, unsigned *sink, long long *clk)
{
    if (threadIdx.x != 0) return;
    unsigned v = 0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) v += __hip_atomic_fetch_add(ctr + (v & 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t1 = wall_clock64();
    sink[blockIdx.x] = v;
    clk[blockIdx.x] = t1 - t0;
}

__global__ void hammer_kernel(unsigned *ctr, int words, int stride_words, int iters, unsigned *sink)
{
    if (threadIdx.x != 0) return;
    unsigned *w = ctr + (size_t)(blockIdx.x % words) * stride_words;
    unsigned v = 0;
    for (int i = 0; i < iters; ++i) v += __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sink[blockIdx.x] = v;
}

int main()
{
    unsigned *ctr, *sink;
    long long *clk;
    CHK(hipMalloc(&ctr, 1 << 20));
    CHK(hipMalloc(&sink, 1 << 16));
    CHK(hipMalloc(&clk, 1 << 16));
    CHK(hipMemset(ctr, 0, 1 << 20));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    // (1) dependent chain, idle machine
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, ctr, 2000, sink, clk);
        CHK(hipDeviceSynchronize());
    }
    long long c;
    CHK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    printf("dependent device-scope returning atomic, idle machine: %.1f ns each (100 MHz wall clock)\n", c * 10.0 / 2000);
    // (2) throughput
    const int iters = 400;
    for (int wgs : {64, 512, 2048}) {
        for (int words : {1, 8, 32, 512}) {
            if (words > wgs) continue;
            for (int stride : {1, 32}) {           // adjacent words (one line) vs one 128-B line per word
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHK(hipEventRecord(e0));
                    hipLaunchKernelGGL(hammer_kernel, dim3(wgs), dim3(64), 0, 0, ctr, words, stride, iters, sink);
                    CHK(hipEventRecord(e1));
                    CHK(hipEventSynchronize(e1));
                    float ms;
                    CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("workgroups %4d  words %3d  word pitch %3d B: %.2f ms for %d atomics = %.1f per us in all, %.2f per us per word\n",
                       wgs, words, stride * 4, best, wgs * iters, wgs * iters / (best * 1e3), wgs * iters / (best * 1e3) / words);
            }
        }
    }
    return 0;
}
