// Does an XCD's L2 keep read-only data ACROSS kernel launches on this machine?
// One wave per XCD (blockIdx % 8) chases a pointer cycle of 4096 128-B lines (512 KiB, inside its own slice) twice
// per launch and reports ns per dependent load of each pass.  A first pass at L2 latency (~90-120 ns) in the second
// launch = retained; at Infinity-Cache / HBM latency (~300-500 ns) = the launch boundary invalidated the L2.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/l2_retention_probe tools/probes/l2_retention_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int LINES = 4096, LINE_WORDS = 32, SLICE_WORDS = LINES * LINE_WORDS;

__global__ void chase(const uint32_t *buf, unsigned long long *out, int launch)
{
    const int x = blockIdx.x;                    // one workgroup per XCD (round-robin dispatch)
    if (threadIdx.x != 0) return;
    const uint32_t *s = buf + (size_t)x * SLICE_WORDS;
    uint32_t p = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned long long t0 = wall_clock64();
        for (int i = 0; i < LINES; ++i) p = s[p];
        const unsigned long long t1 = wall_clock64();
        out[(launch * 8 + x) * 2 + pass] = t1 - t0 + (p == 0xffffffffu);
    }
}
__global__ void other(float *q) { q[threadIdx.x + blockIdx.x * blockDim.x] += 1.0f; }

int main()
{
    std::vector<uint32_t> h(8 * (size_t)SLICE_WORDS, 0);
    std::mt19937 rng(1);
    for (int x = 0; x < 8; ++x) {
        std::vector<int> perm(LINES);
        std::iota(perm.begin(), perm.end(), 0);
        std::shuffle(perm.begin() + 1, perm.end(), rng);
        for (int i = 0; i < LINES; ++i)          // line perm[i] points to line perm[i+1]
            h[(size_t)x * SLICE_WORDS + (size_t)perm[i] * LINE_WORDS] = (uint32_t)perm[(i + 1) % LINES] * LINE_WORDS;
    }
    uint32_t *d; unsigned long long *o; float *q;
    CHK(hipMalloc(&d, h.size() * 4)); CHK(hipMalloc(&o, 64 * 16 * 8)); CHK(hipMalloc(&q, 1 << 20));
    CHK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(q, 0, 1 << 20));
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const char *names[] = {"same stream, back to back", "a small kernel on ANOTHER stream between the launches",
                           "a small kernel on the SAME stream between the launches", "host synchronisation between the launches"};
    for (int mode = 0; mode < 4; ++mode) {
        CHK(hipDeviceSynchronize());
        for (int l = 0; l < 4; ++l) {
            hipLaunchKernelGGL(chase, dim3(8), dim3(64), 0, s1, d, o, l);
            if (mode == 1) hipLaunchKernelGGL(other, dim3(64), dim3(256), 0, s2, q);
            if (mode == 2) hipLaunchKernelGGL(other, dim3(64), dim3(256), 0, s1, q);
            if (mode == 3) CHK(hipStreamSynchronize(s1));
        }
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> r(64);
        CHK(hipMemcpy(r.data(), o, 64 * 8, hipMemcpyDeviceToHost));
        printf("%s\n", names[mode]);
        for (int l = 0; l < 4; ++l) {
            double a = 0, b = 0;
            for (int x = 0; x < 8; ++x) { a += r[(l * 8 + x) * 2]; b += r[(l * 8 + x) * 2 + 1]; }
            printf("  launch %d: first pass %6.1f ns per load, second pass %6.1f ns   (mean of 8 XCDs; wall_clock64 = 100 MHz)\n",
                   l, a / 8 * 10.0 / LINES, b / 8 * 10.0 / LINES);
        }
    }
    return 0;
}
