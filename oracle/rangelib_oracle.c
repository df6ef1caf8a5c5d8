/*
 * rangelib_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See rangelib_oracle.h for scope, the "PARITY UNPINNED" statement and the
 * list of reference call sites this restatement is anchored on.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * Every float32 operation whose result feeds a truncation is written as an
 * explicit single IEEE operation or an explicit fmaf(), so that a GPU
 * restatement can be bit-identical.
 */
#define _GNU_SOURCE
#include "rangelib_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* deterministic sin/cos: SURVEY.md Appendix A "dir: canonical form"          */
/* ------------------------------------------------------------------------ */
void orc_sincosf(float x, float *s, float *c)
{
    /* k = nearest multiple of pi/2 (round-half-even), 3-term Cody-Waite */
    const float TWO_OVER_PI = 0x1.45f306p-1f;          /* 0.63661975 */
    const float P1 = 0x1.921fb6p+0f;                   /* float(pi/2)            */
    const float P2 = -0x1.777a5cp-25f;                 /* float(pi/2 - P1)       */
    const float P3 = -0x1.ee59dap-50f;                 /* float(pi/2 - P1 - P2)  */
    float k = rintf(x * TWO_OVER_PI);
    float r = fmaf(-k, P1, x);
    r = fmaf(-k, P2, r);
    r = fmaf(-k, P3, r);
    float z = r * r;
    /* sin(r), |r| <= pi/4 */
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    float sr = fmaf(r * z, ps, r);
    /* cos(r) */
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    float ss = (q & 1) ? cr : sr;
    float cc = (q & 1) ? sr : cr;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    *s = ss;
    *c = cc;
}

/* ------------------------------------------------------------------------ */
/* OMap + world->grid  (rows a6, a9)                                          */
/* ------------------------------------------------------------------------ */
void orc_map_init(orc_map *m, const uint8_t *occ, int rows, int cols,
                  float res, float ox, float oy, float oyaw)
{
    m->rows = rows;
    m->cols = cols;
    m->occ = occ;
    m->res = res;
    m->ox = ox;
    m->oy = oy;
    m->wa = -oyaw;                       /* PyOMap: world_angle = -1.0*yaw */
    orc_sincosf(m->wa, &m->wa_sin, &m->wa_cos);
    m->inv_res = (float)(1.0 / (double)res);
}

/* world pose -> grid position (col,row units) and grid heading */
static inline void world_to_grid(const orc_map *m, float xw, float yw, float thw,
                                 float *gx, float *gy, float *thg)
{
    float x = (xw - m->ox) * m->inv_res;
    float y = (yw - m->oy) * m->inv_res;
    *gx = fmaf(m->wa_cos, x, -(m->wa_sin * y));
    *gy = fmaf(m->wa_sin, x, m->wa_cos * y);
    *thg = thw + m->wa;
}

/* beam j of a fan: alpha_j = -fov/2 + j*(fov/num_rays)
 * (scripts/ros_interface.py:342-344, scripts/racecar_simulator_v2.py:47-50,
 *  scripts/two_player/scan.py:57-62) */
static inline float fan_alpha(float fov, int num_rays, int j)
{
    float amin = -0.5f * fov;
    float inc = fov / (float)num_rays;
    return fmaf((float)j, inc, amin);
}

static inline void fan_dir(float ct, float st, float fov, int num_rays, int j,
                           float *dx, float *dy)
{
    float sa, ca;
    orc_sincosf(fan_alpha(fov, num_rays, j), &sa, &ca);
    *dx = fmaf(ct, ca, -(st * sa));
    *dy = fmaf(st, ca, ct * sa);
}

/* ------------------------------------------------------------------------ */
/* exact EDT (row a7): column pass + Felzenszwalb lower envelope per row      */
/* ------------------------------------------------------------------------ */
#define EDT_INF ((int64_t)1 << 40)

void orc_edt_sq(const uint8_t *occ, int rows, int cols, uint32_t *d2)
{
    size_t n = (size_t)rows * (size_t)cols;
    int64_t *g = (int64_t *)malloc(n * sizeof(int64_t));
    /* pass 1: per column, squared vertical distance to nearest occupied cell */
    for (int c = 0; c < cols; ++c) {
        int64_t last = -EDT_INF;
        for (int r = 0; r < rows; ++r) {
            if (occ[(size_t)r * cols + c]) last = r;
            int64_t d = (last < 0 && last == -EDT_INF) ? EDT_INF : (r - last);
            g[(size_t)r * cols + c] = d;
        }
        last = EDT_INF;
        for (int r = rows - 1; r >= 0; --r) {
            if (occ[(size_t)r * cols + c]) last = r;
            int64_t d = (last == EDT_INF) ? EDT_INF : (last - r);
            if (d < g[(size_t)r * cols + c]) g[(size_t)r * cols + c] = d;
        }
    }
    /* pass 2: per row, min over c' of (c-c')^2 + g[r][c']^2 by lower envelope */
    int *v = (int *)malloc((size_t)cols * sizeof(int));
    double *z = (double *)malloc(((size_t)cols + 1) * sizeof(double));
    int64_t *f = (int64_t *)malloc((size_t)cols * sizeof(int64_t));
    for (int r = 0; r < rows; ++r) {
        int m = 0;
        for (int c = 0; c < cols; ++c) {
            int64_t gv = g[(size_t)r * cols + c];
            if (gv >= EDT_INF) continue;     /* parabola at infinity never wins */
            f[c] = gv * gv;
            /* insert parabola rooted at c */
            while (m > 0) {
                int p = v[m - 1];
                /* intersection of parabolas p and c */
                double s = ((double)(f[c] + (int64_t)c * c) - (double)(f[p] + (int64_t)p * p)) /
                           (2.0 * (double)(c - p));
                if (s <= z[m - 1]) { --m; continue; }
                z[m] = s;
                break;
            }
            if (m == 0) z[0] = -1e300;
            v[m] = c;
            ++m;
        }
        if (m == 0) {
            for (int c = 0; c < cols; ++c) d2[(size_t)r * cols + c] = UINT32_MAX;
            continue;
        }
        int k = 0;
        for (int c = 0; c < cols; ++c) {
            while (k + 1 < m && z[k + 1] < (double)c) ++k;
            /* z boundaries are real-valued; check both neighbours to stay exact at ties */
            int64_t best = (int64_t)(c - v[k]) * (c - v[k]) + f[v[k]];
            if (k + 1 < m) {
                int64_t alt = (int64_t)(c - v[k + 1]) * (c - v[k + 1]) + f[v[k + 1]];
                if (alt < best) best = alt;
            }
            if (k > 0) {
                int64_t alt = (int64_t)(c - v[k - 1]) * (c - v[k - 1]) + f[v[k - 1]];
                if (alt < best) best = alt;
            }
            d2[(size_t)r * cols + c] = best >= (int64_t)UINT32_MAX ? UINT32_MAX - 1 : (uint32_t)best;
        }
    }
    free(f);
    free(z);
    free(v);
    free(g);
}

void orc_edt(const uint8_t *occ, int rows, int cols, float *dt)
{
    size_t n = (size_t)rows * (size_t)cols;
    uint32_t *d2 = (uint32_t *)malloc(n * sizeof(uint32_t));
    orc_edt_sq(occ, rows, cols, d2);
    for (size_t i = 0; i < n; ++i)
        dt[i] = d2[i] == UINT32_MAX ? 1e10f : sqrtf((float)d2[i]);
    free(d2);
}

/* ------------------------------------------------------------------------ */
/* RayMarching::calc_range (row a8) / cuda_ray_marching (row a11)             */
/* ------------------------------------------------------------------------ */
static inline float rm_cast(const orc_map *m, const float *dt, float max_range, float step_coeff,
                            float gx, float gy, float dx, float dy,
                            int32_t *hit, uint16_t *steps)
{
    const float fcols = (float)m->cols, frows = (float)m->rows;
    float t = 0.0f;
    unsigned n = 0;
    float out = max_range;
    int hc = -1, hr = -1;
    while (t < max_range) {
        float fx = fmaf(dx, t, gx);
        float fy = fmaf(dy, t, gy);
        /* same set as (int)fx in [0,cols) && (int)fy in [0,rows) for every
         * float inside int range (trunc(-0.3)=0 stays in-map, as upstream);
         * NaN / huge values miss deterministically                             */
        if (!(fx > -1.0f && fx < fcols && fy > -1.0f && fy < frows)) break;
        int pc = (int)fx, pr = (int)fy;
        float d = dt[(size_t)pr * m->cols + pc];
        ++n;
        if (d <= 0.0f) {
            float xd = (float)pc - gx;
            float yd = (float)pr - gy;
            out = sqrtf(fmaf(xd, xd, yd * yd));
            hc = pc;
            hr = pr;
            break;
        }
        t += fmaxf(d * step_coeff, 1.0f);
    }
    if (hit) { hit[0] = hc; hit[1] = hr; }
    if (steps) *steps = (uint16_t)(n > 65535u ? 65535u : n);
    return out * m->res;
}

void orc_rm_fan(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                const float *poses, int n_poses, float fov, int num_rays,
                float *ranges, int32_t *hits, uint16_t *steps, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
    for (int p = 0; p < n_poses; ++p) {
        float gx, gy, thg, st, ct;
        world_to_grid(m, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], &gx, &gy, &thg);
        orc_sincosf(thg, &st, &ct);
        for (int j = 0; j < num_rays; ++j) {
            float dx, dy;
            fan_dir(ct, st, fov, num_rays, j, &dx, &dy);
            size_t i = (size_t)p * num_rays + j;
            ranges[i] = rm_cast(m, dt, max_range_px, step_coeff, gx, gy, dx, dy,
                                hits ? hits + 2 * i : NULL, steps ? steps + i : NULL);
        }
    }
}

void orc_rm_rays(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                 const float *ins, int n, float *ranges, int32_t *hits, uint16_t *steps,
                 int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads > 1 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        float gx, gy, thg, dx, dy;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], &gx, &gy, &thg);
        orc_sincosf(thg, &dy, &dx);
        ranges[i] = rm_cast(m, dt, max_range_px, step_coeff, gx, gy, dx, dy,
                            hits ? hits + 2 * (size_t)i : NULL, steps ? steps + i : NULL);
    }
}

/* Upstream-literal form: RangeMethod::numpy_calc_range + RayMarching::calc_range
 * (rows a8/a9): theta' = -theta_w + (-world_angle - 3pi/2); calc_range(y, x, theta')
 * marches (first=row, second=col) along (cosf, sinf) of theta' with libm trig and
 * un-fused multiply-add.  Used only to show the canonical form is the same
 * geometry (tests compare within one cell).                                    */
/* one upstream-literal cast from world (xw, yw, theta_w); hit = (col, row) or (-1, -1) */
static float rm_cast_libm(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                          float rotation_const, float wsin, float wcos, float xw, float yw, float thw,
                          int32_t *hit, uint16_t *steps)
{
    float theta = -thw + rotation_const;
    float x = (xw - m->ox) * m->inv_res;
    float y = (yw - m->oy) * m->inv_res;
    float temp = x;
    x = wcos * x - wsin * y;
    y = wsin * temp + wcos * y;
    /* calc_range(y, x, theta): first coordinate indexes rows */
    float x0 = y, y0 = x;
    float rdx = cosf(theta), rdy = sinf(theta);
    float t = 0.0f, out = max_range_px;
    int hc = -1, hr = -1;
    unsigned n = 0;
    while (t < max_range_px) {
        int px = (int)(x0 + rdx * t);
        int py = (int)(y0 + rdy * t);
        if (px >= m->rows || px < 0 || py < 0 || py >= m->cols) break;
        float d = dt[(size_t)px * m->cols + py];
        ++n;
        if (d <= 0.0f) {
            float xd = (float)px - x0, yd = (float)py - y0;
            out = sqrtf(xd * xd + yd * yd);
            hc = py;
            hr = px;
            break;
        }
        float st = d * step_coeff;
        t += st > 1.0f ? st : 1.0f;
    }
    if (hit) { hit[0] = hc; hit[1] = hr; }
    if (steps) *steps = (uint16_t)(n > 65535u ? 65535u : n);
    return out * m->res;
}

void orc_rm_rays_libm(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                      const float *ins, int n, float *ranges, int32_t *hits, uint16_t *steps)
{
    float rotation_const = (float)(-1.0 * (double)m->wa - 3.0 * M_PI / 2.0);
    float wsin = (float)sin((double)m->wa), wcos = (float)cos((double)m->wa);
    for (int i = 0; i < n; ++i)
        ranges[i] = rm_cast_libm(m, dt, max_range_px, step_coeff, rotation_const, wsin, wcos, ins[3 * i],
                                 ins[3 * i + 1], ins[3 * i + 2], hits ? hits + 2 * (size_t)i : NULL,
                                 steps ? steps + i : NULL);
}

/* the fork's 4-arg form stated literally: beam j of pose p is one libm cast at
 * theta_p + (-fov/2 + j * (fov / num_rays)), every operation a separate float32 rounding (no fma,
 * no shared per-pose sincos, no angle-addition formula) — scripts/scan_simulator.py:103-106,
 * fan convention scripts/ros_interface.py:342-344.                                           */
void orc_rm_fan_libm(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                     const float *poses, int n_poses, float fov, int num_rays, float *ranges,
                     int32_t *hits, uint16_t *steps)
{
    float rotation_const = (float)(-1.0 * (double)m->wa - 3.0 * M_PI / 2.0);
    float wsin = (float)sin((double)m->wa), wcos = (float)cos((double)m->wa);
    const float amin = -0.5f * fov, inc = fov / (float)num_rays;
    for (int p = 0; p < n_poses; ++p)
        for (int j = 0; j < num_rays; ++j) {
            const size_t i = (size_t)p * num_rays + j;
            volatile float aj = (float)j * inc;      /* (volatile: keep the two roundings apart) */
            const float th = poses[3 * p + 2] + (amin + aj);
            ranges[i] = rm_cast_libm(m, dt, max_range_px, step_coeff, rotation_const, wsin, wcos,
                                     poses[3 * p], poses[3 * p + 1], th, hits ? hits + 2 * i : NULL,
                                     steps ? steps + i : NULL);
        }
}

/* ------------------------------------------------------------------------ */
/* The host libm's sinf / cosf, in bulk, and the statement of its algorithm that the product's AUDIT mode runs on
 * the device (pyracecarsimulator_amd/csrc/literal_kernels.h: lit_sinf / lit_cosf).  glibc >= 2.28 (sysdeps/ieee754/
 * flt-32/s_sinf.c, s_cosf.c, sincosf.h — ARM optimized routines): double-precision range reduction and
 * polynomials, one rounding to float32 at the end; the x86-64 builds with FMA contract every multiply-add, written
 * out as fma() here.  orc_libm_restatement_check walks float bit patterns and counts the inputs on which the
 * statement and THIS host's libm differ: 0 over every finite float on glibc 2.35 / x86-64 with FMA — on such a host
 * the device's audit mode is bit-identical to orc_rm_fan_libm / orc_rm_rays_libm above.                        */
/* ------------------------------------------------------------------------ */
void orc_libm_sincosf(const float *x, long n, float *s_out, float *c_out)
{
    for (long i = 0; i < n; ++i) {
        s_out[i] = sinf(x[i]);
        c_out[i] = cosf(x[i]);
    }
}

static const double LIT_HPI_INV = 0x1.45F306DC9C883p+23, LIT_HPI = 0x1.921FB54442D18p0, LIT_PI63 = 0x1.921FB54442D18p-62;
static const double LIT_C[5] = {0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16};
static const double LIT_S[3] = {-0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13};
static const uint32_t LIT_INV_PIO4[24] = {
    0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u,
    0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u,
    0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};

static inline uint32_t lit_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t lit_top12(float x) { return (lit_bits(x) >> 20) & 0x7ffu; }
static inline int lit_flip(int q) { return ((q + 1) & 2) != 0; }               /* sign[] = {1, -1, -1, 1} */

static inline float lit_poly(double x, double x2, int neg, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = fma(x2, LIT_S[2], LIT_S[1]);
        const double x7 = x3 * x2;
        const double s = fma(x3, LIT_S[0], x);
        return (float)fma(x7, s1, s);
    }
    const double sg = neg ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2 = fma(x2, sg * LIT_C[4], sg * LIT_C[3]);
    const double c1 = fma(x2, sg * LIT_C[1], sg * LIT_C[0]);
    const double x6 = x4 * x2;
    const double c = fma(x4, sg * LIT_C[2], c1);
    return (float)fma(x6, c2, c);
}

static inline double lit_reduce_fast(double x, int *np)
{
    const double r = x * LIT_HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, LIT_HPI, x);
}

static inline double lit_reduce_large(uint32_t xi, int *np)
{
    const uint32_t *arr = &LIT_INV_PIO4[(xi >> 26) & 15];
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
    *np = (int)n;
    return (double)(int64_t)res0 * LIT_PI63;
}

static float lit_sinf(float y)
{
    double x = y;
    int n;
    if (lit_top12(y) < lit_top12(0x1.921FB6p-1f)) {
        if (lit_top12(y) < lit_top12(0x1p-12f)) return y;
        return lit_poly(x, x * x, 0, 0);
    }
    if (lit_top12(y) < lit_top12(120.0f)) {
        x = lit_reduce_fast(x, &n);
        return lit_poly(x * (lit_flip(n & 3) ? -1.0 : 1.0), x * x, (n & 2) != 0, n);
    }
    if (lit_top12(y) < lit_top12(INFINITY)) {
        const uint32_t xi = lit_bits(y);
        const int sign = (int)(xi >> 31);
        x = lit_reduce_large(xi, &n);
        return lit_poly(x * (lit_flip((n + sign) & 3) ? -1.0 : 1.0), x * x, ((n + sign) & 2) != 0, n);
    }
    return y - y;
}

static float lit_cosf(float y)
{
    double x = y;
    int n;
    if (lit_top12(y) < lit_top12(0x1.921FB6p-1f)) {
        if (lit_top12(y) < lit_top12(0x1p-12f)) return 1.0f;
        return lit_poly(x, x * x, 0, 1);
    }
    if (lit_top12(y) < lit_top12(120.0f))
        x = lit_reduce_fast(x, &n);
    else if (lit_top12(y) < lit_top12(INFINITY))
        x = lit_reduce_large(lit_bits(y), &n);
    else
        return y - y;
    return lit_poly(x * (lit_flip((n + 1) & 3) ? -1.0 : 1.0), x * x, ((n + 1) & 2) != 0, n ^ 1);
}

/* the statement above on an array (the GPU test compares the device's lit_sinf / lit_cosf with it as well) */
void orc_lit_sincosf(const float *x, long n, float *s_out, float *c_out)
{
    for (long i = 0; i < n; ++i) {
        s_out[i] = lit_sinf(x[i]);
        c_out[i] = lit_cosf(x[i]);
    }
}

/* float bit patterns first, first + step, ... below +inf, both signs: inputs on which the statement and the host's
 * libm differ (sinf mismatches returned, cosf mismatches in *bad_cos).  step 1 = every finite float (~20 s on 8 cores). */
long orc_libm_restatement_check(uint32_t first, uint32_t step, long *bad_cos, int nthreads)
{
    long bs = 0, bc = 0;
    if (step == 0) step = 1;
    const long count = ((long)0x7f800000u - (long)first + (long)step - 1) / (long)step;
    (void)nthreads;
#pragma omp parallel for reduction(+ : bs, bc) schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
    for (long k = 0; k < count; ++k) {
        const uint32_t u = first + (uint32_t)k * step;
        for (uint32_t sg = 0; sg < 2; ++sg) {
            const uint32_t b = u | (sg << 31);
            float x;
            memcpy(&x, &b, 4);
            if (lit_bits(sinf(x)) != lit_bits(lit_sinf(x))) ++bs;
            if (lit_bits(cosf(x)) != lit_bits(lit_cosf(x))) ++bc;
        }
    }
    if (bad_cos) *bad_cos = bc;
    return bs;
}

/* ------------------------------------------------------------------------ */
/* BresenhamsLine::calc_range (row a12)                    */
/* ------------------------------------------------------------------------ */
static inline float bl_cast(const orc_map *m, float max_range,
                            float gx, float gy, float dx, float dy,
                            int32_t *hit, uint16_t *steps)
{
    const float fcols = (float)m->cols, frows = (float)m->rows;
    int hc = -1, hr = -1;
    unsigned n = 0;
    float out = max_range;
    /* poses that cannot index the grid (non-finite, absurdly far) miss without walking:
     * upstream's (int) conversions are undefined there */
    const int sane = fabsf(gx) < 1e9f && fabsf(gy) < 1e9f && (dx - dx) + (dy - dy) == 0.0f;
    /* start cell occupied -> 0 */
    if (!sane) {
        /* miss, no steps */
    } else if (gx > -1.0f && gx < fcols && gy > -1.0f && gy < frows &&
        m->occ[(size_t)(int)gy * m->cols + (int)gx]) {
        out = 0.0f;
        hc = (int)gx;
        hr = (int)gy;
    } else {
        float x0 = gx, y0 = gy;
        float x1 = fmaf(max_range, dx, gx);
        float y1 = fmaf(max_range, dy, gy);
        int steep = fabsf(y1 - y0) > fabsf(x1 - x0);
        if (steep) {
            float tmp = x0; x0 = y0; y0 = tmp;
            tmp = x1; x1 = y1; y1 = tmp;
        }
        /* major axis = x (after the swap); lim_major/minor are the map extents there */
        const float lim_major = steep ? frows : fcols;
        const float lim_minor = steep ? fcols : frows;
        float deltax = fabsf(x1 - x0), deltay = fabsf(y1 - y0);
        float error = 0.0f;
        float _x = x0, _y = y0;
        float xstep = x0 < x1 ? 1.0f : -1.0f;
        float ystep = y0 < y1 ? 1.0f : -1.0f;
        int end = (int)(x1 + xstep);
        int cap = (int)max_range + 3;        /* guard only; never binds for finite inputs */
        while ((int)_x != end && cap-- > 0) {
            _x += xstep;
            error += deltay;
            if (error * 2.0f >= deltax) {
                _y += ystep;
                error -= deltax;
            }
            ++n;
            if (_x >= 0.0f && _x < lim_major && _y >= 0.0f && _y < lim_minor) {
                int col = steep ? (int)_y : (int)_x;
                int row = steep ? (int)_x : (int)_y;
                if (m->occ[(size_t)row * m->cols + col]) {
                    float xd = _x - x0, yd = _y - y0;
                    out = sqrtf(fmaf(xd, xd, yd * yd));
                    hc = col;
                    hr = row;
                    break;
                }
            }
        }
    }
    if (hit) { hit[0] = hc; hit[1] = hr; }
    if (steps) *steps = (uint16_t)n;
    return out * m->res;
}

void orc_bl_fan(const orc_map *m, float max_range_px,
                const float *poses, int n_poses, float fov, int num_rays,
                float *ranges, int32_t *hits, uint16_t *steps, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
    for (int p = 0; p < n_poses; ++p) {
        float gx, gy, thg, st, ct;
        world_to_grid(m, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], &gx, &gy, &thg);
        orc_sincosf(thg, &st, &ct);
        for (int j = 0; j < num_rays; ++j) {
            float dx, dy;
            fan_dir(ct, st, fov, num_rays, j, &dx, &dy);
            size_t i = (size_t)p * num_rays + j;
            ranges[i] = bl_cast(m, max_range_px, gx, gy, dx, dy,
                                hits ? hits + 2 * i : NULL, steps ? steps + i : NULL);
        }
    }
}

void orc_bl_rays(const orc_map *m, float max_range_px,
                 const float *ins, int n, float *ranges, int32_t *hits, uint16_t *steps,
                 int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads > 1 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        float gx, gy, thg, dx, dy;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], &gx, &gy, &thg);
        orc_sincosf(thg, &dy, &dx);
        ranges[i] = bl_cast(m, max_range_px, gx, gy, dx, dy,
                            hits ? hits + 2 * (size_t)i : NULL, steps ? steps + i : NULL);
    }
}

/* Upstream-literal form of BresenhamsLine (RangeMethod::numpy_calc_range + BresenhamsLine::calc_range, row a12 /
 * Appendix A), with the conventions of rm_cast_libm above: theta' = -theta_w + (-world_angle - 3pi/2),
 * calc_range(y, x, theta') — the first coordinate indexes rows —, end point x0 + max_range * cosf(theta'),
 * y0 + max_range * sinf(theta') with libm trig, every product and sum its own float32 rounding, hit distance
 * sqrtf(xd * xd + yd * yd).  PARITY UNPINNED (range_libc is absent): the closest available statement of upstream's
 * arithmetic; the canonical orc_bl_* forms are gated against it (tests/golden/table_libm_forms.npz,
 * tests/test_gpu_parity.py::test_device_tables_vs_upstream_literal_libm_forms).  hit = (col, row) or (-1, -1).      */
static float bl_cast_libm(const orc_map *m, float max_range, float rotation_const, float wsin, float wcos,
                          float xw, float yw, float thw, int32_t *hit, uint16_t *steps)
{
    float theta = -thw + rotation_const;
    float x = (xw - m->ox) * m->inv_res;
    float y = (yw - m->oy) * m->inv_res;
    float temp = x;
    x = wcos * x - wsin * y;
    y = wsin * temp + wcos * y;
    /* calc_range(y, x, theta): first coordinate indexes rows */
    const float r0 = y, c0 = x;
    const float frows = (float)m->rows, fcols = (float)m->cols;
    int hc = -1, hr = -1;
    unsigned n = 0;
    float out = max_range;
    const int sane = fabsf(r0) < 1e9f && fabsf(c0) < 1e9f && (theta - theta) == 0.0f;
    if (!sane) {
        /* miss, no steps (upstream's (int) conversions are undefined there) */
    } else if (r0 > -1.0f && r0 < frows && c0 > -1.0f && c0 < fcols && m->occ[(size_t)(int)r0 * m->cols + (int)c0]) {
        out = 0.0f;
        hc = (int)c0;
        hr = (int)r0;
    } else {
        volatile float mx = max_range * cosf(theta), my = max_range * sinf(theta);   /* (volatile: product and sum apart) */
        float x0 = r0, y0 = c0;                  /* upstream's names: x walks rows, y walks columns */
        float x1 = x0 + mx, y1 = y0 + my;
        const int steep = fabsf(y1 - y0) > fabsf(x1 - x0);
        if (steep) {
            float tmp = x0; x0 = y0; y0 = tmp;
            tmp = x1; x1 = y1; y1 = tmp;
        }
        const float lim_major = steep ? fcols : frows;
        const float lim_minor = steep ? frows : fcols;
        const float deltax = fabsf(x1 - x0), deltay = fabsf(y1 - y0);
        float error = 0.0f;
        float _x = x0, _y = y0;
        const float xstep = x0 < x1 ? 1.0f : -1.0f;
        const float ystep = y0 < y1 ? 1.0f : -1.0f;
        const int end = (int)(x1 + xstep);
        int cap = (int)max_range + 3;
        while ((int)_x != end && cap-- > 0) {
            _x += xstep;
            error += deltay;
            if (error * 2.0f >= deltax) {
                _y += ystep;
                error -= deltax;
            }
            ++n;
            if (_x >= 0.0f && _x < lim_major && _y >= 0.0f && _y < lim_minor) {
                const int row = steep ? (int)_y : (int)_x;
                const int col = steep ? (int)_x : (int)_y;
                if (m->occ[(size_t)row * m->cols + col]) {
                    volatile float xx = (_x - x0) * (_x - x0), yy = (_y - y0) * (_y - y0);
                    out = sqrtf(xx + yy);
                    hc = col;
                    hr = row;
                    break;
                }
            }
        }
    }
    if (hit) { hit[0] = hc; hit[1] = hr; }
    if (steps) *steps = (uint16_t)n;
    return out * m->res;
}

void orc_bl_rays_libm(const orc_map *m, float max_range_px, const float *ins, int n, float *ranges, int32_t *hits,
                      uint16_t *steps)
{
    float rotation_const = (float)(-1.0 * (double)m->wa - 3.0 * M_PI / 2.0);
    float wsin = (float)sin((double)m->wa), wcos = (float)cos((double)m->wa);
    for (int i = 0; i < n; ++i)
        ranges[i] = bl_cast_libm(m, max_range_px, rotation_const, wsin, wcos, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2],
                                 hits ? hits + 2 * (size_t)i : NULL, steps ? steps + i : NULL);
}

/* the 4-argument fan form, beam j of pose p at theta_p + (-fov/2 + j * (fov / num_rays)) rounded to float32 (as
 * orc_rm_fan_libm) */
void orc_bl_fan_libm(const orc_map *m, float max_range_px, const float *poses, int n_poses, float fov, int num_rays,
                     float *ranges, int32_t *hits, uint16_t *steps)
{
    float rotation_const = (float)(-1.0 * (double)m->wa - 3.0 * M_PI / 2.0);
    float wsin = (float)sin((double)m->wa), wcos = (float)cos((double)m->wa);
    const float amin = -0.5f * fov, inc = fov / (float)num_rays;
    for (int p = 0; p < n_poses; ++p)
        for (int j = 0; j < num_rays; ++j) {
            const size_t i = (size_t)p * num_rays + j;
            volatile float aj = (float)j * inc;
            const float th = poses[3 * p + 2] + (amin + aj);
            ranges[i] = bl_cast_libm(m, max_range_px, rotation_const, wsin, wcos, poses[3 * p], poses[3 * p + 1], th,
                                     hits ? hits + 2 * i : NULL, steps ? steps + i : NULL);
        }
}

/* ------------------------------------------------------------------------ */
/* GiantLUTCast (row a14)                                                     */
/* ------------------------------------------------------------------------ */
static inline float lut_bins_per_rad(int theta_disc)
{
    return (float)theta_disc * 0.15915494309189535f;   /* theta_disc / 2pi */
}

static inline int lut_bin(float th, int theta_disc)
{
    /* GiantLUTCast::discretize_theta: nearest bin, wrapped into [0,theta_disc) */
    float u = rintf(th * lut_bins_per_rad(theta_disc));
    /* |u| stays far inside int range for any sane heading; clamp keeps it defined */
    if (!(u > -1e9f && u < 1e9f)) u = 0.0f;
    int b = (int)u % theta_disc;
    return b < 0 ? b + theta_disc : b;
}

/* upstream's discretize_theta stated literally (as recalled, [UPSTREAM-RECALL]): fmod into [0, 2pi), then
 * roundf(theta * theta_disc / 2pi) (halves away from zero), wrapped */
static inline int lut_bin_libm(float th, int theta_disc)
{
    const float two_pi = 6.283185307179586f;
    if (!(th > -1e9f && th < 1e9f)) return 0;
    float t = fmodf(th, two_pi);
    if (t < 0.0f) t += two_pi;
    volatile float scaled = t * (float)theta_disc;
    int b = (int)roundf(scaled / two_pi) % theta_disc;
    return b < 0 ? b + theta_disc : b;
}

static inline uint16_t lut_quant(float r_px, float max_range)
{
    float q = rintf(fminf(r_px, max_range) * (65535.0f / max_range));
    return (uint16_t)q;
}

static inline float lut_dequant(uint16_t q, float max_range)
{
    return (float)q * (max_range / 65535.0f);
}

void orc_lut_build(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                   int theta_disc, int r0, int r1, uint16_t *lut, int nthreads)
{
    (void)nthreads;
    orc_map unit = *m;
    unit.res = 1.0f;                           /* ranges in cells */
    float bin_w = 6.283185307179586f / (float)theta_disc;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 1 ? nthreads : 1)
    for (int r = r0; r < r1; ++r) {
        for (int c = 0; c < m->cols; ++c) {
            uint16_t *row = lut + ((size_t)(r - r0) * m->cols + c) * theta_disc;
            for (int b = 0; b < theta_disc; ++b) {
                float dx, dy;
                orc_sincosf((float)b * bin_w, &dy, &dx);
                float rr = rm_cast(&unit, dt, max_range_px, step_coeff, (float)c, (float)r,
                                   dx, dy, NULL, NULL);
                row[b] = lut_quant(rr, max_range_px);
            }
        }
    }
}

static inline float lut_query(const orc_map *m, const uint16_t *lut, int theta_disc,
                              float max_range, float gx, float gy, float th)
{
    const float fcols = (float)m->cols, frows = (float)m->rows;
    /* GiantLUTCast::calc_range: outside the map -> max_range */
    if (!(gx >= 0.0f && gx < fcols && gy >= 0.0f && gy < frows)) return max_range * m->res;
    size_t cell = (size_t)(int)gy * m->cols + (int)gx;
    return lut_dequant(lut[cell * theta_disc + lut_bin(th, theta_disc)], max_range) * m->res;
}

void orc_lut_fan(const orc_map *m, const uint16_t *lut, int theta_disc, float max_range_px,
                 const float *poses, int n_poses, float fov, int num_rays,
                 float *ranges, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
    for (int p = 0; p < n_poses; ++p) {
        float gx, gy, thg;
        world_to_grid(m, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], &gx, &gy, &thg);
        for (int j = 0; j < num_rays; ++j)
            ranges[(size_t)p * num_rays + j] =
                lut_query(m, lut, theta_disc, max_range_px, gx, gy,
                          thg + fan_alpha(fov, num_rays, j));
    }
}

/* The same fan query with the table handed over as ONE theta row per pose (the row of the
 * pose's own cell; ignored for poses outside the map): lets a test check fan queries on a
 * table too large for the host (cfg3: 2000^2 x 1442 bins = 11.5 GB) by fetching only the rows
 * of the sampled poses.  rows_out[p] <- row index (int)gy, cols_out[p] <- (int)gx, or -1. */
void orc_lut_pose_cells(const orc_map *m, const float *poses, int n_poses, int *rows_out, int *cols_out)
{
    const float fcols = (float)m->cols, frows = (float)m->rows;
    for (int p = 0; p < n_poses; ++p) {
        float gx, gy, thg;
        world_to_grid(m, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], &gx, &gy, &thg);
        const int in = gx >= 0.0f && gx < fcols && gy >= 0.0f && gy < frows;
        rows_out[p] = in ? (int)gy : -1;
        cols_out[p] = in ? (int)gx : -1;
    }
}

void orc_lut_fan_rows(const orc_map *m, const uint16_t *pose_rows /* n_poses * theta_disc */,
                      int theta_disc, float max_range_px, const float *poses, int n_poses,
                      float fov, int num_rays, float *ranges)
{
    const float fcols = (float)m->cols, frows = (float)m->rows;
    for (int p = 0; p < n_poses; ++p) {
        float gx, gy, thg;
        world_to_grid(m, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], &gx, &gy, &thg);
        const int in = gx >= 0.0f && gx < fcols && gy >= 0.0f && gy < frows;
        const uint16_t *row = pose_rows + (size_t)p * theta_disc;
        for (int j = 0; j < num_rays; ++j) {
            const float th = thg + fan_alpha(fov, num_rays, j);
            ranges[(size_t)p * num_rays + j] =
                in ? lut_dequant(row[lut_bin(th, theta_disc)], max_range_px) * m->res
                   : max_range_px * m->res;
        }
    }
}

void orc_lut_rays(const orc_map *m, const uint16_t *lut, int theta_disc, float max_range_px,
                  const float *ins, int n, float *ranges, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        float gx, gy, thg;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], &gx, &gy, &thg);
        ranges[i] = lut_query(m, lut, theta_disc, max_range_px, gx, gy, thg);
    }
}

/* ------------------------------------------------------------------------ */
/* CDDTCast (row a13)                                                         */
/* ------------------------------------------------------------------------ */
#define CDDT_EPS 1e-5f

static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* per-bin geometry shared by build and query.
 * lit (the *_libm statements): upstream's literal arithmetic as recalled (SURVEY row a13,
 * [UPSTREAM-RECALL]) — the bin angle from a double-precision product rounded once (`M_2PI * i / theta_discretization`
 * with a double constant), libm cosf / sinf, every product and sum its own float32 rounding (no fma).  The canonical
 * form (lit = 0) is what the device builds bit for bit; the literal form measures how far that is from a libm build. */
static void cddt_bin_geometry(const orc_map *m, int theta_disc, int a, int lit,
                              float *cosv, float *sinv, int *width, float *translation)
{
    float s, c;
    if (lit) {
        float ang = (float)(6.283185307179586 * (double)a / (double)theta_disc);
        c = cosf(ang);
        s = sinf(ang);
    } else {
        float ang = (float)a * (6.283185307179586f / (float)theta_disc);
        orc_sincosf(ang, &s, &c);
    }
    *cosv = c;
    *sinv = s;
    float W = (float)m->cols, H = (float)m->rows;
    /* height of the rotated map's bounding box = number of buckets */
    float rotated_height = fabsf(W * s) + fabsf(H * c);
    *width = (int)ceilf(rotated_height - CDDT_EPS) + 1;
    /* lowest rotated corner -> translation making every bucket index >= 0 */
    volatile float ws = W * s, hc = H * c;                 /* (volatile: the literal sum stays un-fused) */
    float lt = H * c, rt = lit ? ws + hc : fmaf(W, s, H * c), rb = W * s;
    float mn = fminf(lt, fminf(rt, rb));
    *translation = fmaxf(0.0f, -mn - CDDT_EPS);
}

static inline int is_edge(const orc_map *m, int r, int c)
{
    /* OMap::make_edge_map: occupied cell with at least one free 4-neighbour
     * (cells on the map border count as edges) */
    if (!m->occ[(size_t)r * m->cols + c]) return 0;
    if (r == 0 || c == 0 || r == m->rows - 1 || c == m->cols - 1) return 1;
    return !m->occ[(size_t)(r - 1) * m->cols + c] || !m->occ[(size_t)(r + 1) * m->cols + c] ||
           !m->occ[(size_t)r * m->cols + c - 1] || !m->occ[(size_t)r * m->cols + c + 1];
}

/* lut-space projection of a point for bin a (lit: un-fused products, as upstream's x*cos - y*sin) */
static inline void cddt_project(int lit, float c, float s, float tr, float x, float y, float *lx, float *ly)
{
    if (lit) {
        volatile float xc = x * c, ys = y * s, xs = x * s, yc = y * c;
        *lx = xc - ys;
        *ly = (xs + yc) + tr;
        return;
    }
    *lx = fmaf(x, c, -(y * s));
    *ly = fmaf(x, s, y * c) + tr;
}

static orc_cddt *cddt_build(const orc_map *m, int theta_disc, int lit);
orc_cddt *orc_cddt_build(const orc_map *m, int theta_disc) { return cddt_build(m, theta_disc, 0); }
orc_cddt *orc_cddt_build_libm(const orc_map *m, int theta_disc) { return cddt_build(m, theta_disc, 1); }

static orc_cddt *cddt_build(const orc_map *m, int theta_disc, int lit)
{
    orc_cddt *cd = (orc_cddt *)calloc(1, sizeof(orc_cddt));
    cd->literal = lit;
    int nb = (theta_disc + 1) / 2;
    cd->theta_disc = theta_disc;
    cd->n_bins = nb;
    cd->lut_width = (int *)malloc(nb * sizeof(int));
    cd->lut_translation = (float *)malloc(nb * sizeof(float));
    cd->cosv = (float *)malloc(nb * sizeof(float));
    cd->sinv = (float *)malloc(nb * sizeof(float));
    cd->bucket_off = (int64_t *)malloc((nb + 1) * sizeof(int64_t));
    int64_t nbk = 0;
    for (int a = 0; a < nb; ++a) {
        cddt_bin_geometry(m, theta_disc, a, lit, &cd->cosv[a], &cd->sinv[a], &cd->lut_width[a],
                          &cd->lut_translation[a]);
        cd->bucket_off[a] = nbk;
        nbk += cd->lut_width[a];
    }
    cd->bucket_off[nb] = nbk;
    cd->n_buckets = nbk;
    /* pass 1: count, pass 2: fill, then sort+unique each bucket */
    int64_t *cnt = (int64_t *)calloc((size_t)nbk + 1, sizeof(int64_t));
    for (int pass = 0; pass < 2; ++pass) {
        int64_t *cur = NULL;
        if (pass == 1) {
            int64_t acc = 0;
            for (int64_t b = 0; b <= nbk; ++b) { int64_t t = cnt[b]; cnt[b] = acc; acc += t; }
            cd->xs = (float *)malloc((size_t)(acc > 0 ? acc : 1) * sizeof(float));
            cur = (int64_t *)malloc((size_t)nbk * sizeof(int64_t));
            memcpy(cur, cnt, (size_t)nbk * sizeof(int64_t));
        }
        for (int r = 0; r < m->rows; ++r)
            for (int c = 0; c < m->cols; ++c) {
                if (!is_edge(m, r, c)) continue;
                float px = (float)c + 0.5f, py = (float)r + 0.5f;
                for (int a = 0; a < nb; ++a) {
                    float cs = cd->cosv[a], sn = cd->sinv[a];
                    float half = (fabsf(sn) + fabsf(cs)) * 0.5f;
                    float lx, ly;
                    cddt_project(lit, cs, sn, cd->lut_translation[a], px, py, &lx, &ly);
                    int upper = (int)((ly + half) - CDDT_EPS);
                    int lower = (int)((ly - half) + CDDT_EPS);
                    if (lower < 0) lower = 0;
                    if (upper >= cd->lut_width[a]) upper = cd->lut_width[a] - 1;
                    for (int i = lower; i <= upper; ++i) {
                        int64_t b = cd->bucket_off[a] + i;
                        if (pass == 0) cnt[b]++;
                        else cd->xs[cur[b]++] = lx;
                    }
                }
            }
        if (pass == 1) free(cur);
    }
    /* sort + unique, compact in place */
    cd->offsets = (int64_t *)malloc(((size_t)nbk + 1) * sizeof(int64_t));
    int64_t w = 0;
    for (int64_t b = 0; b < nbk; ++b) {
        int64_t s = cnt[b], e = cnt[b + 1];
        qsort(cd->xs + s, (size_t)(e - s), sizeof(float), cmp_float);
        cd->offsets[b] = w;
        for (int64_t i = s; i < e; ++i)
            if (i == s || cd->xs[i] != cd->xs[i - 1]) cd->xs[w++] = cd->xs[i];
    }
    cd->offsets[nbk] = w;
    cd->n_xs = w;
    free(cnt);
    return cd;
}

void orc_cddt_free(orc_cddt *c)
{
    if (!c) return;
    free(c->lut_width); free(c->lut_translation); free(c->cosv); free(c->sinv);
    free(c->bucket_off); free(c->offsets); free(c->xs); free(c);
}

static inline float cddt_query(const orc_map *m, const orc_cddt *cd, float max_range,
                               float gx, float gy, float th)
{
    /* CDDTCast::discretize_theta(-heading): nearest bin of -th in [0, theta_disc);
     * bins >= theta_disc/2 use the bin half a turn away, searching backwards      */
    int td = cd->theta_disc;
    int b = cd->literal ? lut_bin_libm(-th, td) : lut_bin(-th, td);
    int flipped = 0;
    if (b >= cd->n_bins) { b -= td / 2; flipped = 1; }
    if (b >= cd->n_bins) b = cd->n_bins - 1;           /* odd theta_disc guard */
    float lx, ly;
    cddt_project(cd->literal, cd->cosv[b], cd->sinv[b], cd->lut_translation[b], gx, gy, &lx, &ly);
    float out = max_range;
    if (ly >= 0.0f && ly < (float)cd->lut_width[b]) {
        int64_t bk = cd->bucket_off[b] + (int)ly;
        const float *xs = cd->xs;
        int64_t lo = cd->offsets[bk], hi = cd->offsets[bk + 1];
        if (!flipped) {
            /* ray runs along +x in lut space: first stored x >= lx */
            int64_t a = lo, z = hi;
            while (a < z) { int64_t mid = (a + z) >> 1; if (xs[mid] < lx) a = mid + 1; else z = mid; }
            if (a < hi) out = fminf(xs[a] - lx, max_range);
        } else {
            /* ray runs along -x: last stored x <= lx */
            int64_t a = lo, z = hi;
            while (a < z) { int64_t mid = (a + z) >> 1; if (xs[mid] <= lx) a = mid + 1; else z = mid; }
            if (a > lo) out = fminf(lx - xs[a - 1], max_range);
        }
    }
    return out * m->res;
}

void orc_cddt_fan(const orc_map *m, const orc_cddt *c, float max_range_px,
                  const float *poses, int n_poses, float fov, int num_rays,
                  float *ranges, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
    for (int p = 0; p < n_poses; ++p) {
        float gx, gy, thg;
        world_to_grid(m, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], &gx, &gy, &thg);
        for (int j = 0; j < num_rays; ++j)
            ranges[(size_t)p * num_rays + j] =
                cddt_query(m, c, max_range_px, gx, gy, thg + fan_alpha(fov, num_rays, j));
    }
}

/* world pose -> grid as RangeMethod::numpy_calc_range states it (rm_cast_libm above): un-fused rotation with the
 * double-precision sin / cos of the world angle rounded once */
static inline void world_to_grid_libm(const orc_map *m, float xw, float yw, float thw, float *gx, float *gy, float *thg)
{
    const float wsin = (float)sin((double)m->wa), wcos = (float)cos((double)m->wa);
    float x = (xw - m->ox) * m->inv_res;
    float y = (yw - m->oy) * m->inv_res;
    volatile float a = wcos * x, b = wsin * y, c = wsin * x, d = wcos * y;
    *gx = a - b;
    *gy = c + d;
    *thg = thw + m->wa;
}

/* the 2-argument per-ray form on a literal table (what scripts/two_player/scan.py:57-70 feeds: per-ray float32 thetas) */
void orc_cddt_rays_libm(const orc_map *m, const orc_cddt *c, float max_range_px, const float *ins, int n, float *ranges)
{
    for (int i = 0; i < n; ++i) {
        float gx, gy, thg;
        world_to_grid_libm(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], &gx, &gy, &thg);
        ranges[i] = cddt_query(m, c, max_range_px, gx, gy, thg);
    }
}

/* GiantLUT, literal statement: one libm ray-marching cast per (cell, bin) from the cell's integer corner along
 * (cosf, sinf) of the bin angle (double product rounded once), un-fused march (rm_cast_libm's arithmetic in grid
 * coordinates), the same quantisation; rows [r0, r1) */
void orc_lut_build_libm(const orc_map *m, const float *dt, float max_range_px, float step_coeff, int theta_disc,
                        int r0, int r1, uint16_t *lut, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 1 ? nthreads : 1)
    for (int r = r0; r < r1; ++r)
        for (int c = 0; c < m->cols; ++c) {
            uint16_t *row = lut + ((size_t)(r - r0) * m->cols + c) * theta_disc;
            for (int b = 0; b < theta_disc; ++b) {
                const float ang = (float)(6.283185307179586 * (double)b / (double)theta_disc);
                const float dx = cosf(ang), dy = sinf(ang);
                const float x0 = (float)c, y0 = (float)r;
                float t = 0.0f, out = max_range_px;
                while (t < max_range_px) {
                    volatile float ax = dx * t, ay = dy * t;
                    const int px = (int)(x0 + ax), py = (int)(y0 + ay);
                    if (px < 0 || py < 0 || px >= m->cols || py >= m->rows) break;
                    const float d = dt[(size_t)py * m->cols + px];
                    if (d <= 0.0f) {
                        volatile float xd = (float)px - x0, yd = (float)py - y0;
                        volatile float xx = xd * xd, yy = yd * yd;
                        out = sqrtf(xx + yy);
                        break;
                    }
                    const float st = d * step_coeff;
                    t += st > 1.0f ? st : 1.0f;
                }
                row[b] = lut_quant(out, max_range_px);
            }
        }
}

/* fan query on literal rows (one theta row per pose, as orc_lut_fan_rows): literal world->grid and bin rule */
void orc_lut_fan_rows_libm(const orc_map *m, const uint16_t *pose_rows, int theta_disc, float max_range_px,
                           const float *poses, int n_poses, float fov, int num_rays, float *ranges)
{
    const float fcols = (float)m->cols, frows = (float)m->rows;
    const float amin = -0.5f * fov, inc = fov / (float)num_rays;
    for (int p = 0; p < n_poses; ++p) {
        float gx, gy, thg;
        world_to_grid_libm(m, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], &gx, &gy, &thg);
        const int in = gx >= 0.0f && gx < fcols && gy >= 0.0f && gy < frows;
        const uint16_t *row = pose_rows + (size_t)p * theta_disc;
        for (int j = 0; j < num_rays; ++j) {
            volatile float aj = (float)j * inc;
            const float th = thg + (amin + aj);
            ranges[(size_t)p * num_rays + j] = in ? lut_dequant(row[lut_bin_libm(th, theta_disc)], max_range_px) * m->res
                                                  : max_range_px * m->res;
        }
    }
}

void orc_cddt_rays(const orc_map *m, const orc_cddt *c, float max_range_px,
                   const float *ins, int n, float *ranges, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        float gx, gy, thg;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], &gx, &gy, &thg);
        ranges[i] = cddt_query(m, c, max_range_px, gx, gy, thg);
    }
}

/* ------------------------------------------------------------------------ */
/* consumers: racecar/src/racecar.cpp:239-292 and :305-328                    */
/* ------------------------------------------------------------------------ */
void orc_edge_distances(int num_rays, double min_ang, double inc, double scan_dist_to_base,
                        double width, double wheelbase, double *edge)
{
    const double PI_REF = 3.145;                 /* racecar/include/racecar.hpp:117 */
    double side = width / 2.0;
    double front = wheelbase - scan_dist_to_base;
    double back = scan_dist_to_base;
    double ang = min_ang;
    for (int i = 0; i < num_rays; ++i) {
        ang += inc;                              /* incremented BEFORE use (:256) */
        double a, d1, d2;
        if (ang > 0.0) {
            if (ang < PI_REF / 2.0) { a = ang; d2 = front / cos(a); }
            else { a = ang - PI_REF / 2.0; d2 = back / cos(a); }
        } else {
            if (ang == 0.0) ang += 0.0001;
            if (ang > -PI_REF / 2.0) { a = -ang; d2 = front / cos(a); }
            else { a = -ang - PI_REF / 2.0; d2 = back / cos(a); }
        }
        d1 = side / sin(a);
        edge[i] = d1 < d2 ? d1 : d2;
    }
}

int orc_is_crashed(const float *rays, int num_rays, int poses, const double *edge,
                   double crash_thresh)
{
    int i;
    for (i = 1; i < poses + 1; ++i)
        for (int j = 0; j < num_rays; ++j)
            if (((double)rays[(size_t)(i - 1) * num_rays + j] - edge[j]) < crash_thresh)
                return i - 1;
    return -i;
}


/* ------------------------------------------------------------------------------
 * FollowGap::eval — restated from followgap/followgap.hpp:104-129 with its
 * helpers preprocessLidar (:18-27), safetyBubble (:67-79), findMaxGap (:29-65), findBestPoint
 * (:99-102) and getSteerAng (:81-97).  Pinned against the reference header itself, compiled in
 * place into oracle/_ref/libfollowgap_ref.so (tests/golden/followgap_ref.npz).
 * Two corners of the reference are undefined and are given a definition here:
 *   size < 10   : `lidar.size()-10` wraps (size_t) and the loop runs off the vector -> rejected (NaN);
 *   best == size: happens when the only/first longest gap is the single last beam; the reference
 *                 reads lidar[size] (one past the end) -> this statement reads lidar[size-1].
 * ------------------------------------------------------------------------------ */
float orc_followgap_eval(const float *lidar, int size, float max_distance, float max_angle,
                         float angle_inc)
{
    if (size < 10) return NAN;
    float *v = (float *)malloc((size_t)size * sizeof(float));
    for (int i = 0; i < size; ++i) v[i] = lidar[i];
    for (int i = 0; i < size - 10; ++i)                    /* preprocessLidar :18-27 */
        if (v[i] > max_distance) v[i] = max_distance;
    int min_point = 0;                                     /* eval :112-119 */
    for (int i = 0; i < size; ++i)
        if (v[i] != 0 && v[i] < v[min_point]) min_point = i;
    v[min_point] = 0.0f;                                   /* safetyBubble(v, min_point, 5) :67-79 */
    for (int i = -5; i < 5; ++i)
        if (min_point + i > 0 && min_point + i < size - 1) v[min_point + i] = 0.0f;
    int max_start = 0, max_size = 0, c = 0;                /* findMaxGap :29-65 */
    int cur_start = 0, cur_size = 0;
    while (c < size) {
        cur_start = c;
        cur_size = 0;
        while (c < size && v[c] > 1.75) { ++cur_size; ++c; }
        if (cur_size > max_size) { max_start = cur_start; max_size = cur_size; cur_size = 0; }
        ++c;
    }
    int g0, g1;
    if (cur_size > max_size) { g0 = cur_start; g1 = cur_start + cur_size + 1; }
    else { g0 = max_start; g1 = max_start + max_size + 1; }
    free(v);
    const int best = (g0 + g1) / 2;                        /* findBestPoint :99-102 */
    const float d = lidar[best < size ? best : size - 1];
    float angle;                                           /* getSteerAng :81-97 */
    if (best > size / 2) angle = (float)(-angle_inc * ((size / 2.0) - best));
    else angle = (float)(angle_inc * (best - (size / 2.0)));
    angle = 2 * (angle / d);
    const float lo = -max_angle;
    const float a1 = (angle < lo) ? lo : angle;            /* std::max(angle, -max_angle) */
    return (max_angle < a1) ? max_angle : a1;              /* std::min(.., max_angle)    */
}
