// global_load_lds_dword on gfx950: where does lane l's dword land?  (round 5: the LDS-landing form of lead (c))
// every wave of a 256-thread workgroup loads p[tid] with M0 = its own 256-B row of a dynamic LDS array placed behind
// `pad` bytes; the kernel then reads the row back with ds_read at row + 4 * lane and at a few other places.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_load_probe lds_load_probe.hip ; run: /tmp/lds_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *p, float *out, int pad_words, int masked)
{
    extern __shared__ float lds[];
    float *row = lds + pad_words + (threadIdx.x >> 6) * 64;
    for (int i = threadIdx.x; i < pad_words + 256; i += 256) lds[i] = -1.0f;
    __syncthreads();
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) float *)row);
    const unsigned voff = threadIdx.x * 4;
    unsigned long long m = masked ? 0x00ff00ff00ff00ffull : ~0ull, save;
    unsigned t;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_and_b64 exec, exec, %[m]\n\t"
        "s_mov_b32 %[t], m0\n\t"
        "s_mov_b32 m0, %[lb]\n\t"
        "global_load_lds_dword %[vo], %[base]\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        : [t] "=&s"(t), [save] "=&s"(save) : [lb] "s"(lds_addr), [vo] "v"(voff), [base] "s"(p), [m] "s"(m) : "memory");
    __syncthreads();
    out[threadIdx.x] = row[threadIdx.x & 63];
    out[256 + threadIdx.x] = lds[threadIdx.x];          // the first 256 words of the array (pad region when pad > 0)
    if (threadIdx.x == 0) out[512] = (float)lds_addr;
}
int main()
{
    float *p, *out;
    hipMalloc(&p, 1024); hipMalloc(&out, 4096);
    std::vector<float> h(256); for (int i = 0; i < 256; ++i) h[i] = 1000.0f + i;
    hipMemcpy(p, h.data(), 1024, hipMemcpyHostToDevice);
    for (int pad : {0, 1000, 20000}) for (int masked : {0, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), (pad + 256) * 4, 0, p, out, pad, masked);
        std::vector<float> o(513); hipMemcpy(o.data(), out, 513 * 4, hipMemcpyDeviceToHost);
        int ok = 0, untouched = 0; for (int i = 0; i < 256; ++i) { ok += o[i] == 1000.0f + i; untouched += o[i] == -1.0f; }
        printf("pad %5d words masked %d: lds_addr %.0f | lanes with their own dword at row + 4*lane: %d / 256, untouched %d | row[0..3] %.0f %.0f %.0f %.0f | wave1 row[0] %.0f | lds[0..1] %.0f %.0f\n",
               pad, masked, o[512], ok, untouched, o[0], o[1], o[2], o[3], o[64], o[256], o[257]);
    }
    return 0;
}
