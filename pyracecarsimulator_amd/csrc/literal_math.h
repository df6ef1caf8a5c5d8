// literal_math.h — glibc's sinf / cosf stated for the device, and the per-map constants of range_libc's RangeMethod:
// the arithmetic the upstream-literal mode (option variant 3) shares between its one-lane-per-ray diagnostics kernel
// (literal_kernels.h) and the production stream kernel (rm_kernels.h, template argument LIT).
// lit_sinf / lit_cosf are the algorithm of glibc >= 2.28 (ARM optimized routines' sinf.c / cosf.c / sincosf.h:
// double-precision reduction and polynomials, one rounding to float at the end) with the contractions of the x86-64
// FMA build written as explicit fma; the checker walks the same statement against the host's libm over EVERY finite
// float (orc_libm_restatement_check: 0 mismatches on glibc 2.35).
#pragma once
#include "scan_device.h"

namespace scan {

struct LiteralParams {
    float rotation_const;     // (float)(-world_angle - 3pi/2), double arithmetic on the host
    float wsin, wcos;         // (float)sin / cos of world_angle, the host's libm in double
};

namespace lit {
constexpr double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0, PI63 = 0x1.921FB54442D18p-62;
constexpr double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16, S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
                 S3 = -0x1.994eb3774cf24p-13;

__device__ __forceinline__ uint32_t abstop12(float x) { return (__builtin_bit_cast(uint32_t, x) >> 20) & 0x7ffu; }

// sinf_poly: n even -> sine polynomial of x (|x| <= pi/4), n odd -> cosine; `neg`: the table with negated cosine
// coefficients (quadrants 2, 3)
__device__ __forceinline__ float poly(double x, double x2, bool neg, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = __builtin_fma(x2, S3, S2);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, S1, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    const double sg = neg ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, sg * C4, sg * C3);
    const double c1 = __builtin_fma(x2, sg * C1, sg * C0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, sg * C2, c1);
    return (float)__builtin_fma(x6, c2, c);
}

__device__ __forceinline__ double reduce_fast(double x, int &n)
{
    const double r = x * HPI_INV;
    n = ((int32_t)r + 0x800000) >> 24;
    return __builtin_fma(-(double)n, HPI, x);
}

// |x| >= 120: 4/pi to 192 bits, 32-bit windows a byte apart
__device__ __forceinline__ double reduce_large(uint32_t xi, int &np)
{
    const uint32_t inv_pio4[24] = {0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u,
                                   0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u,
                                   0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u,
                                   0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (int)((xi >> 23) & 7);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    np = (int)n;
    return (double)(int64_t)res0 * PI63;
}

__device__ __forceinline__ bool flip(int q) { return ((q + 1) & 2) != 0; }      // sign[] = {1, -1, -1, 1}
}  // namespace lit

__device__ __forceinline__ float lit_sinf(float y)
{
    using namespace lit;
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return poly(x, x * x, false, 0);
    }
    if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, n);
        const double s = flip(n & 3) ? -1.0 : 1.0;
        return poly(x * s, x * x, (n & 2) != 0, n);
    }
    if (abstop12(y) < abstop12(__builtin_inff())) {
        const uint32_t xi = __builtin_bit_cast(uint32_t, y);
        const int sign = (int)(xi >> 31);
        x = reduce_large(xi, n);
        const double s = flip((n + sign) & 3) ? -1.0 : 1.0;
        return poly(x * s, x * x, ((n + sign) & 2) != 0, n);
    }
    return y - y;
}

__device__ __forceinline__ float lit_cosf(float y)
{
    using namespace lit;
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return poly(x, x * x, false, 1);
    }
    if (abstop12(y) < abstop12(120.0f))
        x = reduce_fast(x, n);
    else if (abstop12(y) < abstop12(__builtin_inff()))
        x = reduce_large(__builtin_bit_cast(uint32_t, y), n);
    else
        return y - y;
    const double s = flip((n + 1) & 3) ? -1.0 : 1.0;
    return poly(x * s, x * x, ((n + 1) & 2) != 0, n ^ 1);
}

}  // namespace scan
