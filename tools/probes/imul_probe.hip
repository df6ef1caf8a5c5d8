// Integer-multiply throughput on gfx950: which of v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32 and the
// 24-bit forms are full rate?  (Philox-2x32 needs the 64-bit product of two 32-bit words per round.)
// 8 waves per SIMD, 8 independent chains per lane, wave-instructions per clock per SIMD.
//   build: hipcc -O2 --offload-arch=gfx950 -o imul_probe imul_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(512) void probe(uint32_t seed, int iters, uint32_t *sink)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u;
    uint32_t a4 = a0 * 11u + 4u, a5 = a0 * 13u + 5u, a6 = a0 * 17u + 6u, a7 = a0 * 19u + 7u;
    const uint32_t m = 0xD256D193u;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                                                           \
    if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "s"(m));                              \
    if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "s"(m));                              \
    if (OP == 2) { uint64_t p_; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p_) : "v"(x), "s"(m) : "vcc"); x = (uint32_t)(p_ >> 32) ^ (uint32_t)p_; } \
    if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "s"(m));                             \
    if (OP == 4) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "s"(m));                          \
    if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "s"(m));                                 \
    if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "s"(m));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    const uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345u) sink[0] = r;
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    uint32_t *sink;
    (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000, grid = n_cu * 4;          // 4 x 512 threads = 32 waves per CU = 8 per SIMD
    const char *names[] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32 (+1 xor)", "v_mul_u32_u24", "v_mul_hi_u32_u24",
                           "v_xor_b32", "v_mad_u32_u24"};
    for (int rep = 0; rep < 2; ++rep)
        for (int op = 0; op < 7; ++op) {
            (void)hipEventRecord(e0);
            switch (op) {
            case 0: probe<0><<<grid, 512>>>(rep, iters, sink); break;
            case 1: probe<1><<<grid, 512>>>(rep, iters, sink); break;
            case 2: probe<2><<<grid, 512>>>(rep, iters, sink); break;
            case 3: probe<3><<<grid, 512>>>(rep, iters, sink); break;
            case 4: probe<4><<<grid, 512>>>(rep, iters, sink); break;
            case 5: probe<5><<<grid, 512>>>(rep, iters, sink); break;
            default: probe<6><<<grid, 512>>>(rep, iters, sink); break;
            }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double wave_insts = (double)grid * 8 * iters * 8;       // 8 waves per block, 8 ops per iteration
            const double clk_per_inst_per_simd = ms * 1e-3 * clk * (n_cu * 4) / wave_insts;
            if (rep) printf("%-26s %.2f clk per wave instruction per SIMD\n", names[op], clk_per_inst_per_simd);
        }
    return 0;
}
