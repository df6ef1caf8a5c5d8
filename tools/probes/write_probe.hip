// What limits a write-only stream on this machine?  torch's fill_ reaches 6.85 TB/s where rl_probe_hbm's write-only
// sweep gets 4.85: the same 2 GiB written with different shapes.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/write_probe tools/probes/write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st16(v4u *p, v4u v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void st4(float *p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// A: persistent grid, 4 x 16 B per lane in flight, the four a whole grid apart (rl_probe_hbm)
template <bool NT> __global__ __launch_bounds__(256) void wA(v4u *dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const v4u f = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n16; i += 4 * stride) {
        st16<NT>(dst + i, f); st16<NT>(dst + i + stride, f); st16<NT>(dst + i + 2 * stride, f); st16<NT>(dst + i + 3 * stride, f);
    }
}
// B: one workgroup per 16 KiB, contiguous (torch's vectorised elementwise shape)
template <bool NT> __global__ __launch_bounds__(256) void wB(v4u *dst, size_t n16)
{
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const v4u f = {1, 2, 3, 4};
    if (b + 768 < n16) { st16<NT>(dst + b, f); st16<NT>(dst + b + 256, f); st16<NT>(dst + b + 512, f); st16<NT>(dst + b + 768, f); }
}
// C: persistent grid walking contiguous 16 KiB blocks
template <bool NT> __global__ __launch_bounds__(256) void wC(v4u *dst, size_t n16)
{
    const v4u f = {1, 2, 3, 4};
    for (size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x; b + 768 < n16; b += (size_t)gridDim.x * 1024) {
        st16<NT>(dst + b, f); st16<NT>(dst + b + 256, f); st16<NT>(dst + b + 512, f); st16<NT>(dst + b + 768, f);
    }
}
// D: one wave per 1081-float row, 17 dword stores per lane (GiantLUT's output shape), persistent
template <bool NT> __global__ __launch_bounds__(256) void wD(float *dst, size_t rows)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    for (size_t r = wave; r < rows; r += nw) {
        float *d = dst + r * 1081;
#pragma unroll
        for (int k = 0; k < 17; ++k) { const int j = k * 64 + lane; if (j < 1081) st4<NT>(d + j, 1.0f); }
    }
}
// E: D with each WORKGROUP (4 waves) writing 4 consecutive rows as one contiguous 17 296-B run, 16 B per lane
template <bool NT> __global__ __launch_bounds__(256) void wE(float *dst, size_t rows)
{
    const v4u f = {1, 2, 3, 4};
    for (size_t g = blockIdx.x; g * 4 + 3 < rows; g += gridDim.x) {
        v4u *d = reinterpret_cast<v4u *>(dst + g * 4 * 1081);          // 4 rows = 4324 floats = 1081 x 16 B
        for (int i = threadIdx.x; i < 1081; i += 256) st16<NT>(d + i, f);
    }
}
int main()
{
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16, rows = bytes / (1081 * 4);
    void *d; CHK(hipMalloc(&d, bytes));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        launch(); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-78s %6.0f GB/s\n", name, bytes * 10.0 / (ms * 1e-3) / 1e9);
    };
    for (int gm : {4, 8, 16}) {
        char nm[128];
        snprintf(nm, sizeof nm, "A persistent x%d/CU, 4 stores a grid apart", gm); timeit(nm, [&] { wA<false><<<ncu * gm, 256>>>((v4u *)d, n16); });
        snprintf(nm, sizeof nm, "A ... non-temporal"); timeit(nm, [&] { wA<true><<<ncu * gm, 256>>>((v4u *)d, n16); });
        snprintf(nm, sizeof nm, "C persistent x%d/CU, contiguous 16 KiB blocks", gm); timeit(nm, [&] { wC<false><<<ncu * gm, 256>>>((v4u *)d, n16); });
        snprintf(nm, sizeof nm, "C ... non-temporal"); timeit(nm, [&] { wC<true><<<ncu * gm, 256>>>((v4u *)d, n16); });
        snprintf(nm, sizeof nm, "D persistent x%d/CU, wave per 1081-float row, dword stores", gm); timeit(nm, [&] { wD<false><<<ncu * gm, 256>>>((float *)d, rows); });
        snprintf(nm, sizeof nm, "D ... non-temporal"); timeit(nm, [&] { wD<true><<<ncu * gm, 256>>>((float *)d, rows); });
        snprintf(nm, sizeof nm, "E persistent x%d/CU, workgroup per 4 rows, 16-B stores", gm); timeit(nm, [&] { wE<false><<<ncu * gm, 256>>>((float *)d, rows); });
        snprintf(nm, sizeof nm, "E ... non-temporal"); timeit(nm, [&] { wE<true><<<ncu * gm, 256>>>((float *)d, rows); });
    }
    timeit("B one workgroup per 16 KiB (torch's shape)", [&] { wB<false><<<(unsigned)(n16 / 1024), 256>>>((v4u *)d, n16); });
    timeit("B ... non-temporal", [&] { wB<true><<<(unsigned)(n16 / 1024), 256>>>((v4u *)d, n16); });
    return 0;
}
