// scan_kernels.h — gfx950 kernels of the lidar scan path.
//
//   K0  edt_*            exact Euclidean distance transform of the occupancy grid
//                        (range_libc DistanceTransform)
//   K1  rm_fan_kernel    fan-expanding sphere tracing on the float32 EDT
//                        (RayMarching / RayMarchingGPU, rows a8-a11); bit-exact
//       rm_rays_kernel   one (x,y,theta) row per ray (upstream 2-arg API)
//
// Work decomposition of K1 (wave64-first, not a warp-shaped port of kernels.cu's
// thread-per-ray 1024x256 grid): the unit of work is a CHUNK = 64 consecutive
// beams of one pose, owned by one wavefront, so the 64 lanes of a wave march 64
// neighbouring beams (angular spacing fov/num_rays ~ 0.25 deg): their samples
// fall in the same few EDT cache lines, their step counts are similar (less
// divergence), and the wave leaves the march loop as soon as every lane has hit
// or left the map (exec-mask early termination).  The per-beam (cos a_j, sin a_j)
// fan table is computed once per workgroup and kept in LDS; per-pose constants are
// wave-uniform.  Waves are persistent and take chunks round-robin, so a launch
// has 256 CUs x 8 workgroups regardless of the batch size.
#pragma once
#include "scan_device.h"

namespace scan {

constexpr int WG = 256;                 // 4 waves
constexpr int WAVES_PER_WG = WG / 64;
constexpr int GINF = 30000;             // "no obstacle in this column" (maps <= 16384 per side)
constexpr uint32_t GINF2 = (uint32_t)GINF * (uint32_t)GINF;

// ------------------------------------------------------------------------------
// K0: exact EDT.  Pass 1: per column, distance to the nearest occupied cell of the
// column (down then up sweep, one lane per column, coalesced rows).
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edt_cols_kernel(const uint8_t *__restrict__ occ, int rows,
                                                       int cols, int *__restrict__ g)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    int last = -GINF;
    for (int r = 0; r < rows; ++r) {
        if (occ[(size_t)r * cols + c]) last = r;
        int d = r - last;
        g[(size_t)r * cols + c] = d > GINF ? GINF : d;
    }
    last = 4 * GINF;
    for (int r = rows - 1; r >= 0; --r) {
        if (occ[(size_t)r * cols + c]) last = r;
        int d = last - r;
        int old = g[(size_t)r * cols + c];
        g[(size_t)r * cols + c] = d < old ? d : old;
    }
}

// Pass 2: one workgroup per row, the row of column distances staged in LDS; each
// cell widens its search k = 1,2,.. while k^2 can still beat the best d^2 found, so
// the work per cell is O(distance), not O(cols).  d^2 is an exact integer; the
// result is sqrtf((float)d2), correctly rounded == the CPU statement.
__global__ __launch_bounds__(256) void edt_rows_kernel(const int *__restrict__ g, int rows,
                                                       int cols, float *__restrict__ dt)
{
    extern __shared__ int grow[];
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) grow[c] = g[(size_t)r * cols + c];
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        uint32_t v0 = (uint32_t)grow[c];
        uint32_t best = v0 * v0;
        for (int k = 1; (uint32_t)k * (uint32_t)k < best; ++k) {
            const bool l_ok = c - k >= 0, r_ok = c + k < cols;
            if (!l_ok && !r_ok) break;
            const uint32_t kk = (uint32_t)k * (uint32_t)k;
            if (l_ok) {
                uint32_t v = (uint32_t)grow[c - k];
                uint32_t cand = kk + v * v;
                best = cand < best ? cand : best;
            }
            if (r_ok) {
                uint32_t v = (uint32_t)grow[c + k];
                uint32_t cand = kk + v * v;
                best = cand < best ? cand : best;
            }
        }
        dt[(size_t)r * cols + c] = best >= GINF2 ? 1e10f : sqrtf((float)best);
    }
}

// bit-packed occupancy rows (bit c&31 of word c>>5), for the LDS-tiled kernels
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *__restrict__ occ, int rows,
                                                        int cols, int stride,
                                                        uint32_t *__restrict__ bits)
{
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (w >= stride || r >= rows) return;
    uint32_t word = 0;
    int c0 = w * 32;
#pragma unroll 4
    for (int b = 0; b < 32; ++b) {
        int c = c0 + b;
        if (c < cols && occ[(size_t)r * cols + c]) word |= 1u << b;
    }
    bits[(size_t)r * stride + w] = word;
}

// ------------------------------------------------------------------------------
// K1: fan-expanding ray marching.  out[pose*num_rays + j] in metres.
// ------------------------------------------------------------------------------
struct CrashParams {
    const double *edge;      // num_rays doubles (Car::setCarEdgeDistances) or nullptr
    double thresh;
    int *first_crashed;      // atomicMin target, initialised to INT_MAX
};

template <bool AUX, bool CRASH>
__global__ __launch_bounds__(WG) void rm_fan_kernel(MapParams m, FanParams f,
                                                    const float *__restrict__ poses,
                                                    float *__restrict__ out,
                                                    int32_t *__restrict__ hits,
                                                    uint16_t *__restrict__ steps, CrashParams cp)
{
    extern __shared__ float2 fan_cs[];   // per-beam (cos a_j, sin a_j), staged once per WG
    for (int j = threadIdx.x; j < f.num_rays; j += WG) {
        float s, c;
        det_sincosf(fan_alpha(f, j), s, c);
        fan_cs[j] = make_float2(c, s);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * WAVES_PER_WG;
    const int cpp = (f.num_rays + 63) >> 6;                 // chunks per pose
    const long n_chunks = (long)f.n_poses * cpp;

    for (long ch = (long)blockIdx.x * WAVES_PER_WG + wave_in_wg; ch < n_chunks; ch += n_waves) {
        const int pose = (int)(ch / cpp);
        const int j = ((int)(ch - (long)pose * cpp) << 6) + lane;
        // wave-uniform pose constants
        float gx, gy, thg, st, ct;
        world_to_grid(m, poses[3 * (size_t)pose], poses[3 * (size_t)pose + 1],
                      poses[3 * (size_t)pose + 2], gx, gy, thg);
        det_sincosf(thg, st, ct);
        if (j < f.num_rays) {
            const float2 cs = fan_cs[j];
            const float dx = __builtin_fmaf(ct, cs.x, -(st * cs.y));
            const float dy = __builtin_fmaf(st, cs.x, ct * cs.y);
            RayResult rr = rm_march(m, f.max_range, f.step_coeff, gx, gy, dx, dy);
            const size_t i = (size_t)pose * f.num_rays + j;
            float r = rr.range_px * m.res;
            if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
            if (out) out[i] = r;
            if (AUX) {
                if (hits) { hits[2 * i] = rr.hit_c; hits[2 * i + 1] = rr.hit_r; }
                if (steps) steps[i] = (uint16_t)(rr.steps > 65535u ? 65535u : rr.steps);
            }
            if (CRASH) {
                // Car::isCrashed racecar/src/racecar.cpp:320: (rays - edge[j]) < CRASH_THRESH
                const bool crashed = ((double)r - cp.edge[j]) < cp.thresh;
                if (__ballot(crashed)) {
                    if (lane == __ffsll((long long)__ballot(crashed)) - 1)
                        atomicMin(cp.first_crashed, pose);
                }
            }
        }
    }
}

// one world (x, y, theta) row per ray: upstream calc_range_many(ins, outs)
__global__ __launch_bounds__(WG) void rm_rays_kernel(MapParams m, FanParams f,
                                                     const float *__restrict__ ins, long n,
                                                     float *__restrict__ out,
                                                     int32_t *__restrict__ hits,
                                                     uint16_t *__restrict__ steps)
{
    const long stride = (long)gridDim.x * WG;
    for (long i = (long)blockIdx.x * WG + threadIdx.x; i < n; i += stride) {
        float gx, gy, thg, dx, dy;
        world_to_grid(m, ins[3 * i], ins[3 * i + 1], ins[3 * i + 2], gx, gy, thg);
        det_sincosf(thg, dy, dx);
        RayResult rr = rm_march(m, f.max_range, f.step_coeff, gx, gy, dx, dy);
        float r = rr.range_px * m.res;
        if (f.noise_std > 0.0f) r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + i);
        out[i] = r;
        if (hits) { hits[2 * i] = rr.hit_c; hits[2 * i + 1] = rr.hit_r; }
        if (steps) steps[i] = (uint16_t)(rr.steps > 65535u ? 65535u : rr.steps);
    }
}

}  // namespace scan
