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


// ==============================================================================
// K1b: the MI355X-shaped ray-marching path (variant 1, default).
//
//  (0) pad_dt_kernel — per method: the float32 EDT copied into an array with a border of
//      ceil(max_range)+2 cells holding -1 ("outside the map").  A ray whose origin is
//      inside the map stays within max_range of it while it is live, so the march loop
//      needs no bounds test and no address clamp: leaving the map reads -1 and stops
//      like a hit does.  (Origins outside the map are misses before the first sample —
//      decided once per pose.)
//  (1) pose_bin_kernel — one 1024-lane workgroup turns the pose list into per-pose
//      records (gx, gy, cos th, sin th) ordered by the map tile the pose stands in
//      (LDS histogram -> scan -> scatter).  Costs O(P); the march costs O(P*B*samples).
//  (2) rm_fan_stream_kernel.  The tile-ordered pose list is cut into 8 contiguous BANDS,
//      band x marched only by workgroups with blockIdx % 8 == x — one XCD under
//      round-robin dispatch (speed only, never correctness) — so each XCD's 4 MiB L2
//      holds one band of the map.  Inside a band, workgroup g owns the 64-beam chunks
//      g, g+G, g+2G, ... (interleaved: every workgroup sees the band's average cost).
//      A workgroup's 16 waves share ONE stream of ray slots through an LDS counter: a
//      wave marches while more than `low_water` of its lanes are live, then every
//      finished lane stores its range and claims the next slot (ballot + mbcnt ranks,
//      one LDS atomic per wave).  Lanes stay busy although samples-per-ray is ~7 on
//      average and ~25 at the wave maximum.  (A global work counter per band was tried
//      first and rejected: returning atomics on one contended word retire at ~10/us on
//      MI355X, which made the launch atomic-bound.)
//      rocprofv3 showed this kernel is instruction-issue bound (L1 hit 76 %, L2 latency
//      ~130 cycles, ~60 VALU+SALU per sample in the first version), so the march loop is
//      written predicated — every lane executes every instruction, a finished lane has
//      t = +inf and re-reads its origin cell — with no EXEC-mask traffic:
//      14 VALU + 1 load + ~5 SALU per sample.
// Results are bit-identical to K1 (same arithmetic; only the schedule differs).
// ==============================================================================
struct PoseRec {
    float gx, gy, ct, st;
};

constexpr uint32_t POSE_INVALID = 0x80000000u;   // order[] flag: origin outside the map / non-finite

__global__ __launch_bounds__(256) void pad_dt_kernel(const float *__restrict__ dt, int rows, int cols,
                                                     float *__restrict__ pdt, int pad, int stride)
{
    const int pr = blockIdx.y;                      // padded row
    const int r = pr - pad;
    for (int pc = blockIdx.x * blockDim.x + threadIdx.x; pc < stride; pc += gridDim.x * blockDim.x) {
        const int c = pc - pad;
        float v = -1.0f;
        if (r >= 0 && r < rows && c >= 0 && c < cols) v = dt[(size_t)r * cols + c];
        pdt[(size_t)pr * stride + pc] = v;
    }
}

__global__ __launch_bounds__(1024) void pose_bin_kernel(MapParams m, const float *__restrict__ poses,
                                                        int n, PoseRec *__restrict__ rec,
                                                        PoseRec *__restrict__ rec_sorted,
                                                        uint32_t *__restrict__ order,
                                                        uint32_t *__restrict__ keys, int tile_shift,
                                                        int tiles_x, int n_tiles, int do_sort)
{
    extern __shared__ uint32_t hist[];          // n_tiles counters, then 1024 scan partials
    uint32_t *part = hist + n_tiles;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_tiles; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int p = tid; p < n; p += 1024) {
        PoseRec r;
        float thg;
        world_to_grid(m, poses[3 * (size_t)p], poses[3 * (size_t)p + 1], poses[3 * (size_t)p + 2],
                      r.gx, r.gy, thg);
        det_sincosf(thg, r.st, r.ct);
        // the t = 0 sample of every beam is the origin: outside the map (or non-finite) means
        // every beam of the pose misses without sampling
        const bool fin = (r.ct - r.ct) + (r.st - r.st) == 0.0f;
        const bool inb = r.gx > -1.0f && r.gx < m.fcols && r.gy > -1.0f && r.gy < m.frows;
        const uint32_t flag = (fin && inb) ? 0u : POSE_INVALID;
        if (!(fin && inb)) { r.gx = 0.0f; r.gy = 0.0f; r.ct = 1.0f; r.st = 0.0f; }
        uint32_t key = (uint32_t)n_tiles - 1;       // invalid poses go last; they cost nothing
        if (!flag)
            key = (uint32_t)(((int)r.gy >> tile_shift) * tiles_x + ((int)r.gx >> tile_shift));
        if (do_sort) {
            rec[p] = r;
            keys[p] = key | flag;
            atomicAdd(&hist[key], 1u);
        } else {
            rec_sorted[p] = r;
            order[p] = (uint32_t)p | flag;
        }
    }
    if (!do_sort) return;
    __syncthreads();
    // exclusive scan of hist[0..n_tiles): each lane owns E consecutive counters
    const int E = (n_tiles + 1023) / 1024;
    uint32_t local = 0;
    for (int e = 0; e < E; ++e) {
        int i = tid * E + e;
        if (i < n_tiles) local += hist[i];
    }
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t base = part[tid] - local;
    for (int e = 0; e < E; ++e) {
        int i = tid * E + e;
        if (i < n_tiles) {
            uint32_t c = hist[i];
            hist[i] = base;
            base += c;
        }
    }
    __syncthreads();
    for (int p = tid; p < n; p += 1024) {
        const uint32_t kf = keys[p];
        const uint32_t slot = atomicAdd(&hist[kf & ~POSE_INVALID], 1u);
        order[slot] = (uint32_t)p | (kf & POSE_INVALID);
        rec_sorted[slot] = rec[p];
    }
}

// unsigned division by a launch-time constant (round-up method, any 32-bit dividend)
struct FastDiv {
    uint32_t mul, sh1, sh2, d;
};
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv &f)
{
    uint32_t t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// ------------------------------------------------------------------------------
// The march loop of K1b, hand-scheduled for gfx950.  EXEC holds the live lanes
// (v_cmpx drops a lane the moment its t reaches max_range, hits, or leaves the map),
// so finished lanes cost nothing but their slot and keep (c, r, d) of their last
// sample; the loop leaves when at most `low` lanes are still live.
// Per sample: 12 VALU + 1 global load + 4 SALU (unit step coefficient).
//   fx = fma(dx,t,gx); fy = fma(dy,t,gy); c = (int)fx; r = (int)fy      (Appendix A "march")
//   d  = pdt[(r*stride + c)*4 + k4]          border cells read -1 => stop (left the map)
//   t  = d > 0 ? t + max(d*coeff, 1) : +inf   (d == 0: hit)
// ------------------------------------------------------------------------------
template <bool UNIT, bool AUX>
__device__ __forceinline__ void march_loop(float dx, float dy, float gx, float gy, float &t, int &c,
                                           int &r, float &d, uint32_t &nstep, const float *pdt,
                                           int stride, uint32_t k4, float max_range, float coeff,
                                           uint32_t low)
{
    float a, b;
    unsigned long long save;
    uint32_t n;
    const float inf = __builtin_inff();
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_gt_f32_e32 %[mx], %[t]\n"
        "L_march_%=:\n\t"
        "v_fma_f32 %[a], %[dx], %[t], %[gx]\n\t"
        "v_fma_f32 %[b], %[dy], %[t], %[gy]\n\t"
        "v_cvt_i32_f32_e32 %[c], %[a]\n\t"
        "v_cvt_i32_f32_e32 %[r], %[b]\n\t"
        "v_mad_i32_i24 %[a], %[r], %[stride], %[c]\n\t"
        "v_lshl_add_u32 %[a], %[a], 2, %[k4]\n\t"
        "global_load_dword %[d], %[a], %[base]\n\t"
        ".if %[aux]\n\t"
        "v_add_u32_e32 %[ns], 1, %[ns]\n\t"
        ".endif\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        ".if %[unit]\n\t"
        "v_max_i32_e32 %[a], 1.0, %[d]\n\t"
        ".else\n\t"
        "v_mul_f32_e32 %[a], %[co], %[d]\n\t"
        "v_max_f32_e32 %[a], 1.0, %[a]\n\t"
        ".endif\n\t"
        "v_add_f32_e32 %[a], %[t], %[a]\n\t"
        "v_cmp_lt_f32_e32 vcc, 0, %[d]\n\t"
        "v_cndmask_b32_e32 %[t], %[inf], %[a], vcc\n\t"
        "v_cmpx_gt_f32_e32 %[mx], %[t]\n\t"
        "s_bcnt1_i32_b64 %[n], exec\n\t"
        "s_cmp_gt_u32 %[n], %[low]\n\t"
        "s_cbranch_scc1 L_march_%=\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [t] "+v"(t), [c] "+v"(c), [r] "+v"(r), [d] "+v"(d), [ns] "+v"(nstep), [a] "=&v"(a),
          [b] "=&v"(b), [save] "=&s"(save), [n] "=&s"(n)
        : [dx] "v"(dx), [dy] "v"(dy), [gx] "v"(gx), [gy] "v"(gy), [inf] "v"(inf),
          [mx] "s"(max_range), [stride] "s"(stride), [k4] "s"(k4), [base] "s"(pdt),
          [co] "s"(coeff), [low] "s"(low), [unit] "n"(UNIT ? 1 : 0), [aux] "n"(AUX ? 1 : 0)
        : "vcc", "scc", "memory");
}

struct PadMap {
    const float *pdt;        // padded EDT, (rows+2*pad) x stride, border = -1
    int stride, pad;
    uint32_t k4;             // byte offset of map cell (0,0): (pad*stride + pad)*4
    FastDiv div_stride;
    float res;
};

struct StreamParams {
    const PoseRec *rec;      // sorted order
    const uint32_t *order;   // sorted slot -> pose index | POSE_INVALID
    FastDiv div_cpp;         // chunks per pose
    int cpp;
    int low_water;           // refill when <= low_water lanes are still marching
    int n_bands;
    int drain_prio;          // raise wave priority once the workgroup's stream is exhausted
    unsigned long long *dbg; // diagnostics (nullptr in production): 4 words per wave
};

template <bool AUX, bool CRASH, bool UNIT, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80)))
void rm_fan_stream_kernel(PadMap pm, FanParams f, StreamParams sp, float *__restrict__ out,
                          int32_t *__restrict__ hits, uint16_t *__restrict__ steps, CrashParams cp)
{
    extern __shared__ float lds_f[];
    uint32_t *q_next = reinterpret_cast<uint32_t *>(lds_f);     // shared slot counter
    float2 *fan_cs = reinterpret_cast<float2 *>(lds_f + 2);     // num_rays float2
    if (threadIdx.x == 0) *q_next = 0;
    for (int j = threadIdx.x; j < f.num_rays; j += NT) {
        float s, c;
        det_sincosf(fan_alpha(f, j), s, c);
        fan_cs[j] = make_float2(c, s);
    }
    __syncthreads();

    // ---- which band of the sorted pose list, and which workgroups share it
    const int nb = sp.n_bands;
    const int band = (int)(blockIdx.x % (unsigned)nb);
    const uint32_t g = blockIdx.x / (unsigned)nb;
    const uint32_t G = ((uint32_t)gridDim.x - (uint32_t)band + (uint32_t)nb - 1) / (uint32_t)nb;
    const uint32_t seg_lo = (uint32_t)(((long)f.n_poses * band) / nb);
    const uint32_t seg_hi = (uint32_t)(((long)f.n_poses * (band + 1)) / nb);
    const uint32_t seg_chunks = (seg_hi - seg_lo) * (uint32_t)sp.cpp;
    // this workgroup owns chunks g, g+G, ... of the band: K chunks, 64*K ray slots
    const uint32_t K = g < seg_chunks ? (seg_chunks - g + G - 1) / G : 0;
    const uint32_t total = K << 6;
    const unsigned lane = threadIdx.x & 63;
    const float INF = __builtin_inff();

    unsigned long long t_start = 0;
    uint32_t n_serv = 0;
    if (sp.dbg) t_start = wall_clock64();

    bool exhausted = total == 0;
    bool has_ray = false;
    float gx = 0, gy = 0, dx = 0, dy = 0;
    float t = INF;                 // t < max_range  <=>  the lane is marching
    float d_last = 1.0f;           // last sample: 0 = hit, -1 = left the map, > 0 = free
    int pc = 0, pr = 0;            // cell of the last sample
    uint32_t oidx = 0, nstep = 0, pose = 0;
    int jbeam = 0;

    for (;;) {
        // ---------------- service: finish pending rays, claim new slots
        const unsigned long long idle = __ballot(!(t < f.max_range));
        if (idle) {
            if (sp.dbg) ++n_serv;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const bool mine = !(t < f.max_range);
            if (mine && has_ray) {
                float r = f.max_range;
                int hc = -1, hr = -1;
                if (d_last == 0.0f) {
                    hc = pc;
                    hr = pr;
                    const float xd = (float)hc - gx, yd = (float)hr - gy;
                    r = __builtin_sqrtf(__builtin_fmaf(xd, xd, yd * yd));
                }
                r *= pm.res;
                if (f.noise_std > 0.0f)
                    r += f.noise_std * gauss_noise(f.noise_seed, f.ray_offset + oidx);
                if (out) out[oidx] = r;
                if (AUX) {
                    if (hits) { hits[2 * (size_t)oidx] = hc; hits[2 * (size_t)oidx + 1] = hr; }
                    if (steps) steps[oidx] = (uint16_t)(nstep > 65535u ? 65535u : nstep);
                }
                if (CRASH) {
                    if (((double)r - cp.edge[jbeam]) < cp.thresh) atomicMin(cp.first_crashed, (int)pose);
                }
                has_ray = false;
            }
            if (!exhausted) {                         // wave-uniform
                const uint32_t cnt = (uint32_t)__popcll(idle);
                uint32_t qb = 0;
                if (lane == 0) qb = atomicAdd(q_next, cnt);
                qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
                exhausted = qb + cnt >= total;
                const uint32_t q = qb + rank;
                if (mine && q < total) {
                    const uint32_t chunk = g + (q >> 6) * G;
                    const uint32_t spose = fast_div(chunk, sp.div_cpp);
                    const int j = (int)((chunk - spose * (uint32_t)sp.cpp) << 6) + (int)(q & 63);
                    if (j < f.num_rays) {
                        const uint32_t po = sp.order[seg_lo + spose];
                        const PoseRec pr_ = sp.rec[seg_lo + spose];
                        const float2 cs = fan_cs[j];
                        pose = po & ~POSE_INVALID;
                        gx = pr_.gx;
                        gy = pr_.gy;
                        dx = __builtin_fmaf(pr_.ct, cs.x, -(pr_.st * cs.y));
                        dy = __builtin_fmaf(pr_.st, cs.x, pr_.ct * cs.y);
                        d_last = 1.0f;
                        nstep = 0;
                        jbeam = j;
                        oidx = pose * (uint32_t)f.num_rays + (uint32_t)j;
                        has_ray = true;
                        t = (po & POSE_INVALID) ? INF : 0.0f;
                    }
                }
            }
        }
        const unsigned long long act = __ballot(t < f.max_range);
        if (!act) {
            if (exhausted && !__ballot(has_ray)) break;
            continue;
        }
        // ---------------- march while enough lanes are live (or nothing is left to claim)
        // a wave that can no longer refill is on the launch's critical path (its longest ray
        // decides when the kernel ends): let it win issue arbitration against refilling waves
        if (exhausted && sp.drain_prio) __builtin_amdgcn_s_setprio(3);
        march_loop<UNIT, AUX>(dx, dy, gx, gy, t, pc, pr, d_last, nstep, pm.pdt, pm.stride, pm.k4,
                              f.max_range, f.step_coeff, exhausted ? 0u : (uint32_t)sp.low_water);
    }
    if (sp.dbg && lane == 0) {
        const size_t gw = ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 4;
        sp.dbg[gw] = t_start;
        sp.dbg[gw + 1] = wall_clock64();
        sp.dbg[gw + 2] = ((unsigned long long)n_serv << 32);
        sp.dbg[gw + 3] = ((unsigned long long)K << 32) | (uint32_t)band;
    }
}

}  // namespace scan
