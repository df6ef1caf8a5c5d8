// Calibration of rocprofv3's FETCH_SIZE for SCATTERED 4-byte reads on gfx950 (MI355X_MICROARCH.md, section HBM:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  The ray-marching and CDDT kernels
// gather single dwords; this probe reads KNOWN line counts from an 8 GiB table (32x the Infinity Cache, every
// line touched once) in four patterns, one kernel each, so that a --pmc FETCH_SIZE pass gives KiB per pattern:
//   gather1   one dword per lane, every lane its own random 128-B line               N lines
//   gather2   two dwords per lane, 64 B apart in the lane's own random line          N lines (2N 64-B halves)
//   quad1     one dword per lane, 4 consecutive lanes share a random line            N/4 lines
//   stream16  16 B per lane, coalesced, the guide's reference pattern                N*16 bytes
//   build: hipcc -O2 --offload-arch=gfx950 -o fetch_probe fetch_probe.hip ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// a bijection of [0, 2^26) (lines of the table): odd multiplier + xorshift inside 26 bits
__device__ __forceinline__ uint32_t perm26(uint32_t i)
{
    uint32_t x = (i * 0x2545F491u + 0x1234567u) & 0x3FFFFFFu;
    x ^= x >> 13; x = (x * 0x9E3779B1u) & 0x3FFFFFFu; x ^= x >> 11;
    return x & 0x3FFFFFFu;
}

__global__ void gather1(const uint32_t *__restrict__ tab, uint32_t n, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = tab[(size_t)perm26(i) * 32 + (mix(i) & 31)];
    if (v == 0x12345u) *sink = v;
}

__global__ void gather2(const uint32_t *__restrict__ tab, uint32_t n, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t base = (size_t)perm26(i) * 32 + (mix(i) & 15);
    const uint32_t v = tab[base] ^ tab[base + 16];
    if (v == 0x12345u) *sink = v;
}

__global__ void quad1(const uint32_t *__restrict__ tab, uint32_t n, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = tab[(size_t)perm26(i >> 2) * 32 + (mix(i) & 31)];
    if (v == 0x12345u) *sink = v;
}

__global__ void stream16(const uint4 *__restrict__ tab, uint32_t n, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 v = tab[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = v.x;
}

int main()
{
    const size_t bytes = (size_t)8 << 30;                 // 2^26 lines of 128 B
    uint32_t *tab, *sink;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(tab, 0, bytes);
    (void)hipDeviceSynchronize();
    const uint32_t n = 1u << 24;                          // 16 Mi lanes per kernel
    const dim3 grid(n / 256), block(256);
    hipLaunchKernelGGL(gather1, grid, block, 0, 0, tab, n, sink);
    hipLaunchKernelGGL(gather2, grid, block, 0, 0, tab, n, sink);
    hipLaunchKernelGGL(quad1, grid, block, 0, 0, tab, n, sink);
    hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const uint4 *)tab, n, sink);
    (void)hipDeviceSynchronize();
    printf("lanes per kernel %u: gather1 touches %u lines (%.0f KiB at 128 B), gather2 the same lines in two halves, "
           "quad1 %u lines (%.0f KiB), stream16 %.0f KiB\n", n, n, n * 128.0 / 1024, n / 4, n / 4 * 128.0 / 1024,
           n * 16.0 / 1024);
    return 0;
}
