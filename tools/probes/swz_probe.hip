// Swizzled-buffer probe: does buffer_load_dword ... idxen offen with SWIZZLE_ENABLE compute
//   byte = (idx / IS) * stride * IS + (off / ES) * IS * ES + (idx % IS) * ES + off % ES
// on gfx950 (the address unit does a tiled 2D address for free), how does its out-of-range test
// behave, and does a scattered gather through it run at the rate of global_load_dword?
//   build: hipcc -O2 --offload-arch=gfx950 -o swz_probe swz_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_rsrc(const void *base, uint32_t stride, uint32_t num_records, uint32_t w3,
                                           bool swizzle)
{
    const uint64_t a = (uint64_t)base;
    u32x4 r;
    r.x = (uint32_t)a;
    r.y = (uint32_t)(a >> 32) | (stride << 16) | (swizzle ? 0x80000000u : 0u);
    r.z = num_records;
    r.w = w3;
    r.x = __builtin_amdgcn_readfirstlane(r.x);
    r.y = __builtin_amdgcn_readfirstlane(r.y);
    r.z = __builtin_amdgcn_readfirstlane(r.z);
    r.w = __builtin_amdgcn_readfirstlane(r.w);
    return r;
}

// out[i] = word loaded for (idx[i], off[i])
__global__ void semantic(const uint32_t *buf, uint32_t stride, uint32_t num_records, uint32_t w3, int swizzle,
                         const uint32_t *idx, const uint32_t *off, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 rs = make_rsrc(buf, stride, num_records, w3, swizzle != 0);
    uint32_t v;
    const uint32_t a = idx[i], b = off[i];
    asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\t"
                 "buffer_load_dword %0, v[10:11], %3, 0 idxen offen\n\ts_waitcnt vmcnt(0)"
                 : "=v"(v)
                 : "v"(a), "v"(b), "s"(rs)
                 : "v10", "v11", "memory");
    out[i] = v;
}

// rate: every lane walks a pseudo-random path of cells inside a window (dependent loads, 8 waves/SIMD)
template <int MODE>   // 0 global_load row-major, 1 buffer swizzled, 2 buffer idxen (row-major), 3 global software-tiled (4 rows)
__global__ __launch_bounds__(1024) void rate(const uint32_t *buf, uint32_t stride, uint32_t num_records, uint32_t w3,
                                             int rows, int cols, int iters, uint32_t *sink)
{
    const u32x4 rs = make_rsrc(buf, stride, num_records, w3, MODE == 1);
    uint32_t r = (blockIdx.x * 37 + (threadIdx.x >> 6) * 11) % (rows - 64) + (threadIdx.x & 7);
    uint32_t c = (blockIdx.x * 101 + (threadIdx.x >> 6) * 29) % (cols - 64) + ((threadIdx.x >> 3) & 7);
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t v;
        if (MODE == 0) {
            const uint32_t o = (r * (stride >> 2) + c) << 2;
            asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(o), "s"(buf) : "memory");
        } else if (MODE == 3) {
            const uint32_t o = (c << 4) + r * stride - (r & 3) * (stride - 4);
            asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(o), "s"(buf) : "memory");
        } else {
            const uint32_t c4 = c << 2;
            asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\t"
                         "buffer_load_dword %0, v[10:11], %3, 0 idxen offen\n\ts_waitcnt vmcnt(0)"
                         : "=v"(v)
                         : "v"(r), "v"(c4), "s"(rs)
                         : "v10", "v11", "memory");
        }
        acc += v;
        // the loaded word (a hash) steers the next cell: a short hop like a march step
        r = r + (v & 3) - 1;
        c = c + ((v >> 2) & 7) - 2;
        r = min(max(r, 8u), (uint32_t)rows - 9);
        c = min(max(c, 8u), (uint32_t)cols - 9);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const int rows = 2048, cols = 2048;           // 16 MiB of words
    const uint32_t stride = cols * 4;             // 8192 B < 16384
    const size_t n_words = (size_t)rows * cols;
    std::vector<uint32_t> h(n_words);
    for (size_t i = 0; i < n_words; ++i) h[i] = (uint32_t)i;          // word index
    uint32_t *buf, *d_idx, *d_off, *d_out, *sink;
    hipMalloc(&buf, n_words * 4 + 4096);
    hipMemcpy(buf, h.data(), n_words * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, 4);
    const int n = 4096;
    std::vector<uint32_t> idx(n), off(n), out(n);
    for (int i = 0; i < n; ++i) {
        idx[i] = hash32(i * 2 + 1) % rows;
        off[i] = (hash32(i * 2 + 2) % cols) * 4;
    }
    // a few probes of the range test: idx == rows (one past), off == stride, off == stride + 4
    idx[0] = rows; off[0] = 0;
    idx[1] = 5; off[1] = stride;
    idx[2] = 5; off[2] = stride + 4;
    idx[3] = rows - 1; off[3] = stride - 4;
    hipMalloc(&d_idx, n * 4); hipMalloc(&d_off, n * 4); hipMalloc(&d_out, n * 4);
    hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_off, off.data(), n * 4, hipMemcpyHostToDevice);

    const uint32_t W3_BASE = 0x00020000u;          // DATA_FORMAT 32 (gfx9 raw buffers)
    for (int es = 0; es < 4; ++es)
        for (int is = 0; is < 4; ++is) {
            const uint32_t ES = 2u << es, IS = 8u << is;
            const uint32_t w3 = W3_BASE | ((uint32_t)es << 19) | ((uint32_t)is << 21);
            hipMemset(d_out, 0xff, n * 4);
            semantic<<<n / 256, 256>>>(buf, stride, rows, w3, 1, d_idx, d_off, d_out, n);
            hipMemcpy(out.data(), d_out, n * 4, hipMemcpyDeviceToHost);
            int ok = 0, lin = 0;
            for (int i = 4; i < n; ++i) {
                const uint64_t b = (uint64_t)(idx[i] / IS) * stride * IS + (uint64_t)(off[i] / ES) * IS * ES +
                                   (idx[i] % IS) * ES + off[i] % ES;
                ok += out[i] == (uint32_t)(b / 4);
                lin += out[i] == idx[i] * (stride / 4) + off[i] / 4;
            }
            printf("ES %2u IS %2u: formula %4d / %d, linear %4d | idx=rows -> %08x  off=stride -> %08x  off=stride+4 -> %08x  last -> %08x\n",
                   ES, IS, ok, n - 4, lin, out[0], out[1], out[2], out[3]);
        }
    {
        hipMemset(d_out, 0xff, n * 4);
        semantic<<<n / 256, 256>>>(buf, stride, rows, W3_BASE, 0, d_idx, d_off, d_out, n);
        hipMemcpy(out.data(), d_out, n * 4, hipMemcpyDeviceToHost);
        int lin = 0;
        for (int i = 4; i < n; ++i) lin += out[i] == idx[i] * (stride / 4) + off[i] / 4;
        printf("no swizzle: linear %4d / %d | idx=rows -> %08x  off=stride -> %08x  off=stride+4 -> %08x\n", lin, n - 4,
               out[0], out[1], out[2]);
    }

    // rate
    for (size_t i = 0; i < n_words; ++i) h[i] = hash32((uint32_t)i);
    hipMemcpy(buf, h.data(), n_words * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, grid = n_cu * 2;
    const uint32_t w3s = W3_BASE | (3u << 19) | (0u << 21);       // ES 16, IS 8: a 128-B line = 4 columns x 8 rows
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<grid, 1024>>>(buf, stride, rows, W3_BASE, rows, cols, iters, sink);
            if (mode == 1) rate<1><<<grid, 1024>>>(buf, stride, rows, w3s, rows, cols, iters, sink);
            if (mode == 2) rate<2><<<grid, 1024>>>(buf, stride, rows, W3_BASE, rows, cols, iters, sink);
            if (mode == 3) rate<3><<<grid, 1024>>>(buf, stride, rows, W3_BASE, rows, cols, iters, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double loads = (double)grid * 1024 * iters;
            if (rep)
                printf("mode %d (%s): %.3f ms, %.1f G lane-loads/s\n", mode,
                       mode == 0 ? "global row-major" : mode == 1 ? "buffer swizzled ES16 IS8"
                                 : mode == 2 ? "buffer idxen row-major" : "global software-tiled 4 rows",
                       ms, loads / ms * 1e-6);
        }
    return 0;
}
