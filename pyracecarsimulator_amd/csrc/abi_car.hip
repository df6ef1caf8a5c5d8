// abi_car.hip — what sits in front of and behind the scan path in libscan_amd.so (C ABI: include/scanlib.h; SURVEY.md
// section 8f): the MCTS roll-out generator, FollowGap, 16-bit ranges for the xGMI exchange, diagnostics probes, the
// car-outline table and Car::isCrashed on the host.
#include "abi_internal.h"
#include <array>
#include <utility>
#include "car_kernels.h"
#include "consumer_kernels.h"
#include "probe_kernels.h"

struct rl_car {
    std::vector<rl_car *> reps;          // multi-device (rl_car_create_multi): one ordinary handle per device
    std::unique_ptr<MultiPool> pool;
    int device = 0;
    CarParams P{};
    hipStream_t stream = nullptr;
    DevBuf states, actions, poses, states_out, vel, ranges, edge, first;
    std::mutex mu;
};

extern "C" int rl_car_create(int device, const double *p, rl_car **out)
{
    if (!p || !out) return fail(RL_ERR_INVALID, "rl_car_create: null pointer");
    int ndev = rl_device_count();
    if (ndev <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    rl_car *c = new (std::nothrow) rl_car();
    if (!c) return fail(RL_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->P = CarParams{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12],
                     p[13], p[14], p[15], p[16]};
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(RL_ERR_HIP, "stream creation failed");
    }
    *out = c;
    return RL_OK;
}

extern "C" int rl_car_create_multi(const int *devices, int n_devices, const double *p, rl_car **out)
{
    if (!p || !out || !devices) return fail(RL_ERR_INVALID, "rl_car_create_multi: null pointer");
    if (n_devices < 1 || n_devices > 64) return fail(RL_ERR_INVALID, "rl_car_create_multi: 1..64 devices (got %d)", n_devices);
    rl_car *c = new (std::nothrow) rl_car();
    if (!c) return fail(RL_ERR_NOMEM, "out of host memory");
    std::vector<int> devs;
    for (int i = 0; i < n_devices; ++i) {
        rl_car *r = nullptr;
        const int rc = rl_car_create(devices[i], p, &r);
        if (rc) {
            const std::string keep = last_error();
            rl_car_destroy(c);
            set_last_error(keep);
            return rc;
        }
        c->reps.push_back(r);
        devs.push_back(devices[i]);
    }
    c->device = devices[0];
    c->P = c->reps[0]->P;
    c->pool = std::make_unique<MultiPool>();
    c->pool->start(devs);
    *out = c;
    return RL_OK;
}

extern "C" void rl_car_destroy(rl_car *c)
{
    if (!c) return;
    if (!c->reps.empty() || c->pool) {
        c->pool.reset();
        for (rl_car *r : c->reps) rl_car_destroy(r);
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->states, &c->actions, &c->poses, &c->states_out, &c->vel, &c->ranges, &c->edge, &c->first})
        b->release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int car_rollout_device(rl_car *c, const double *states_in, const double *actions, int R,
                              int n_steps, int every, double dt, bool want_states, bool want_vel)
{
    if (R < 0 || n_steps <= 0 || every <= 0) return fail(RL_ERR_INVALID, "n_rollouts >= 0, n_steps > 0, action_every > 0 required");
    if ((long)R * n_steps > INT_MAX / 4) return fail(RL_ERR_INVALID, "too many roll-out poses");
    HIPCHK(hipSetDevice(c->device));
    if (R == 0) return RL_OK;
    const int n_act = (n_steps + every - 1) / every;
    int rc;
    if ((rc = c->states.ensure((size_t)R * 11 * 8)) || (rc = c->actions.ensure((size_t)R * n_act * 16)) ||
        (rc = c->poses.ensure((size_t)R * n_steps * 12)) || (rc = c->states_out.ensure((size_t)R * 11 * 8)) ||
        (rc = c->vel.ensure((size_t)R * n_steps * 8)))
        return rc;
    HIPCHK(hipMemcpyAsync(c->states.p, states_in, (size_t)R * 11 * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->actions.p, actions, (size_t)R * n_act * 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(rollout_kernel, dim3((R + 63) / 64), dim3(64), 0, c->stream, c->P,
                       (const double *)c->states.p, (const double *)c->actions.p, R, n_steps, every, dt,
                       (float *)c->poses.p, want_states ? (double *)c->states_out.p : nullptr,
                       want_vel ? (double *)c->vel.p : nullptr);
    HIPCHK(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_car_rollout(rl_car *c, const double *states_in, const double *actions, int R,
                              int n_steps, int every, double dt, float *poses_out, double *states_out,
                              double *vel_out)
{
    if (!c || (R > 0 && (!states_in || !actions || !poses_out))) return fail(RL_ERR_INVALID, "rl_car_rollout: null pointer");
    if (!c->reps.empty()) {
        // roll-outs are independent: contiguous blocks of them, one per device
        if (R < 0 || n_steps <= 0 || every <= 0) return fail(RL_ERR_INVALID, "n_rollouts >= 0, n_steps > 0, action_every > 0 required");
        std::lock_guard<std::mutex> lk(c->mu);
        const int k = (int)std::max<long>(1, std::min<long>((long)c->reps.size(), (long)R / 64));
        const size_t n_act = (size_t)(n_steps + every - 1) / every;
        std::vector<std::function<int()>> jobs;
        for (int i = 0; i < k; ++i) {
            long lo, hi;
            block_of(R, i, k, lo, hi);
            rl_car *r = c->reps[i];
            jobs.push_back([=]() {
                return rl_car_rollout(r, states_in + 11 * lo, actions + 2 * n_act * lo, (int)(hi - lo), n_steps, every, dt,
                                      poses_out + (size_t)3 * n_steps * lo, states_out ? states_out + 11 * lo : nullptr,
                                      vel_out ? vel_out + (size_t)n_steps * lo : nullptr);
            });
        }
        return c->pool->run(jobs);
    }
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = car_rollout_device(c, states_in, actions, R, n_steps, every, dt, states_out != nullptr, vel_out != nullptr);
    if (rc || R == 0) return rc;
    HIPCHK(hipMemcpyAsync(poses_out, c->poses.p, (size_t)R * n_steps * 12, hipMemcpyDeviceToHost, c->stream));
    if (states_out) HIPCHK(hipMemcpyAsync(states_out, c->states_out.p, (size_t)R * 11 * 8, hipMemcpyDeviceToHost, c->stream));
    if (vel_out) HIPCHK(hipMemcpyAsync(vel_out, c->vel.p, (size_t)R * n_steps * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RL_OK;
}

extern "C" int rl_car_rollout_check(rl_car *c, rl_method *h, const double *states_in, const double *actions,
                                    int R, int n_steps, int every, double dt, float fov, int num_rays,
                                    const double *edge, double crash_thresh, int *first_crashed,
                                    double *states_out, double *vel_out)
{
    if (!c || !h || (R > 0 && (!states_in || !actions || !edge || !first_crashed)))
        return fail(RL_ERR_INVALID, "rl_car_rollout_check: null pointer");
    if (c->reps.empty() != h->reps.empty() || c->reps.size() != h->reps.size())
        return fail(RL_ERR_INVALID, "car and range method must both be single-device or span the same devices");
    if (!c->reps.empty()) {
        // MCTS.rollout + checkCollisionMany for R roll-outs over several devices: contiguous blocks of roll-outs,
        // each device integrates, scans and tests its own (nothing but the crash indices comes back)
        if (R < 0 || n_steps <= 0 || every <= 0) return fail(RL_ERR_INVALID, "n_rollouts >= 0, n_steps > 0, action_every > 0 required");
        for (size_t i = 0; i < c->reps.size(); ++i)
            if (c->reps[i]->device != h->reps[i]->map->device)
                return fail(RL_ERR_INVALID, "car and range method replicas live on different devices");
        std::scoped_lock lk(c->mu, h->mu);
        std::shared_lock<std::shared_mutex> ml(h->map->multi_mu);
        if (h->map->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
        const int k = (int)std::max<long>(1, std::min<long>(multi_parts(h, (long)R * n_steps), std::max(R, 1)));
        const size_t n_act = (size_t)(n_steps + every - 1) / every;
        const float nstd = h->noise_std;
        const uint64_t seed = h->noise_seed, off = h->ray_offset;
        std::vector<std::function<int()>> jobs;
        for (int i = 0; i < k; ++i) {
            long lo, hi;
            block_of(R, i, k, lo, hi);
            rl_car *cr = c->reps[i];
            rl_method *hr = h->reps[i];
            jobs.push_back([=]() {
                if (hi <= lo) return (int)RL_OK;
                int rc = rl_set_noise(hr, nstd, seed, off + (uint64_t)lo * n_steps * num_rays);
                if (rc) return rc;
                return rl_car_rollout_check(cr, hr, states_in + 11 * lo, actions + 2 * n_act * lo, (int)(hi - lo), n_steps,
                                            every, dt, fov, num_rays, edge, crash_thresh, first_crashed + lo,
                                            states_out ? states_out + 11 * lo : nullptr,
                                            vel_out ? vel_out + (size_t)n_steps * lo : nullptr);
            });
        }
        return c->pool->run(jobs);
    }
    if (c->device != h->map->device) return fail(RL_ERR_INVALID, "car and range method live on different devices");
    std::scoped_lock lk(c->mu, h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    int rc = car_rollout_device(c, states_in, actions, R, n_steps, every, dt, states_out != nullptr, vel_out != nullptr);
    if (rc || R == 0) return rc;
    if ((rc = check_fan_args(h, R * n_steps, fov, num_rays))) return rc;
    const size_t n_rays = (size_t)R * n_steps * num_rays;
    if ((rc = c->ranges.ensure(n_rays * 4)) || (rc = c->edge.ensure((size_t)num_rays * 8)) ||
        (rc = c->first.ensure((size_t)R * 4)))
        return rc;
    HIPCHK(hipMemcpyAsync(c->edge.p, edge, (size_t)num_rays * 8, hipMemcpyHostToDevice, c->stream));
    rc = crash_groups_device(h, (const float *)c->poses.p, R, n_steps, fov, num_rays,
                             (const double *)c->edge.p, crash_thresh, (int *)c->first.p,
                             (float *)c->ranges.p, true, c->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(first_crashed, c->first.p, (size_t)R * 4, hipMemcpyDeviceToHost, c->stream));
    if (states_out) HIPCHK(hipMemcpyAsync(states_out, c->states_out.p, (size_t)R * 11 * 8, hipMemcpyDeviceToHost, c->stream));
    if (vel_out) HIPCHK(hipMemcpyAsync(vel_out, c->vel.p, (size_t)R * n_steps * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RL_OK;
}


// ---------------------------------------------------------------- FollowGap
struct rl_followgap {
    int device = 0;
    FollowGapParams P{};
    int window_size = 0;           // kept for the caller; FollowGap::eval never reads it
    int n_cu = 256;                // (queried once: hipGetDeviceProperties costs the host tens of microseconds per call)
    bool walk_kernel = false;      // diagnostics (environment RL_FOLLOWGAP_WALK=1 at create): followgap_kernel at every size
    hipStream_t stream = nullptr;
    DevBuf scans, angles;
    std::mutex mu;
};

extern "C" int rl_followgap_create(int device, int window_size, float max_distance, float max_angle,
                                   float angle_inc, rl_followgap **out)
{
    if (!out) return fail(RL_ERR_INVALID, "rl_followgap_create: null pointer");
    int ndev = rl_device_count();
    if (ndev <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    rl_followgap *g = new (std::nothrow) rl_followgap();
    if (!g) return fail(RL_ERR_NOMEM, "out of host memory");
    g->device = device;
    g->window_size = window_size;
    g->P.max_distance = max_distance;
    g->P.max_angle = max_angle;
    g->P.angle_inc = angle_inc;
    const char *walk = getenv("RL_FOLLOWGAP_WALK");
    g->walk_kernel = walk && walk[0] == '1';
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
        delete g;
        return fail(RL_ERR_HIP, "stream creation failed");
    }
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) g->n_cu = n_cu;
    *out = g;
    return RL_OK;
}

extern "C" void rl_followgap_destroy(rl_followgap *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    g->scans.release();
    g->angles.release();
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

// followgap_bits_kernel<ROWS>, ROWS = 1 ... FG_ROWS
typedef void (*fg_bits_fn)(const float *, int, FollowGapParams, float *);
template <int... R>
static constexpr std::array<fg_bits_fn, sizeof...(R)> fg_bits_make(std::integer_sequence<int, R...>)
{
    return {{followgap_bits_kernel<R + 1>...}};
}
static const std::array<fg_bits_fn, FG_ROWS> fg_bits_table = fg_bits_make(std::make_integer_sequence<int, FG_ROWS>());

static int followgap_launch(rl_followgap *g, const float *d_scans, int n_scans, int size,
                            float *d_angles, hipStream_t stream)
{
    if (n_scans < 0) return fail(RL_ERR_INVALID, "n_scans must be >= 0");
    // (the reference's preprocessLidar runs off its vector below 10 beams, followgap.hpp:21)
    if (size < 10) return fail(RL_ERR_INVALID, "FollowGap needs at least 10 beams per scan (got %d)", size);
    if (size > 12288) return fail(RL_ERR_UNSUPPORTED, "at most 12288 beams per scan (got %d)", size);
    if (n_scans == 0) return RL_OK;
    FollowGapParams p = g->P;
    p.size = size;
    const int grid = std::min(n_scans, g->n_cu * 32);
    if (size <= 64 * FG_ROWS && !g->walk_kernel)
        fg_bits_table[(size + 63) / 64 - 1]<<<dim3(grid), dim3(64), 0, stream>>>(d_scans, n_scans, p, d_angles);
    else
        hipLaunchKernelGGL(followgap_kernel, dim3(grid), dim3(64), (size_t)size * sizeof(float), stream,
                           d_scans, n_scans, p, d_angles);
    HIPCHK(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_followgap_eval(rl_followgap *g, const float *scans, int n_scans, int size,
                                 float *angles)
{
    if (!g || !scans || !angles) return fail(RL_ERR_INVALID, "rl_followgap_eval: null pointer");
    std::lock_guard<std::mutex> lk(g->mu);
    HIPCHK(hipSetDevice(g->device));
    if (n_scans < 0 || size < 10)
        return followgap_launch(g, nullptr, n_scans, size, nullptr, g->stream);   // (argument errors)
    if (n_scans == 0) return RL_OK;
    const size_t bytes = (size_t)n_scans * size * sizeof(float);
    int rc;
    if ((rc = g->scans.ensure(bytes)) || (rc = g->angles.ensure((size_t)n_scans * sizeof(float)))) return rc;
    HIPCHK(hipMemcpyAsync(g->scans.p, scans, bytes, hipMemcpyHostToDevice, g->stream));
    if ((rc = followgap_launch(g, (const float *)g->scans.p, n_scans, size, (float *)g->angles.p, g->stream)))
        return rc;
    HIPCHK(hipMemcpyAsync(angles, g->angles.p, (size_t)n_scans * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return RL_OK;
}

extern "C" int rl_followgap_eval_device(rl_followgap *g, const float *d_scans, int n_scans, int size,
                                        float *d_angles, void *hip_stream)
{
    if (!g || (n_scans > 0 && (!d_scans || !d_angles)))
        return fail(RL_ERR_INVALID, "rl_followgap_eval_device: null pointer");
    std::lock_guard<std::mutex> lk(g->mu);
    HIPCHK(hipSetDevice(g->device));
    return followgap_launch(g, d_scans, n_scans, size, d_angles, (hipStream_t)hip_stream);
}

// ---------------------------------------------------------------- diagnostics: HBM stream probe
static int probe_hbm_modes(int device, size_t bytes, double *gbs_out, int mode_lo, int mode_hi);

extern "C" int rl_probe_hbm(int device, size_t bytes, double *gbs_out5)
{
    return probe_hbm_modes(device, bytes, gbs_out5, 0, 5);
}

extern "C" int rl_probe_hbm_nt(int device, size_t bytes, double *gbs_out3)
{
    return probe_hbm_modes(device, bytes, gbs_out3, 5, 8);
}

extern "C" int rl_probe_literal_sincosf(int device, const float *x, size_t n, float *sin_out, float *cos_out)
{
    if (!x || !sin_out || !cos_out) return fail(RL_ERR_INVALID, "rl_probe_literal_sincosf: null pointer");
    int ndev = rl_device_count();
    if (ndev <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    if (n == 0) return RL_OK;
    HIPCHK(hipSetDevice(device));
    float *d = nullptr;
    if (hipMalloc((void **)&d, 3 * n * sizeof(float)) != hipSuccess) return fail(RL_ERR_NOMEM, "rl_probe_literal_sincosf: %zu floats", 3 * n);
    int rc = RL_OK;
    if (hipMemcpy(d, x, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = fail(RL_ERR_HIP, "upload failed");
    if (rc == RL_OK) {
        const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(literal_sincosf_kernel, dim3(grid), dim3(256), 0, nullptr, d, (long)n, d + n, d + 2 * n);
        if (hipMemcpy(sin_out, d + n, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(cos_out, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(RL_ERR_HIP, "rl_probe_literal_sincosf: kernel or download failed");
    }
    (void)hipFree(d);
    return rc;
}

static int probe_hbm_modes(int device, size_t bytes, double *gbs_out5, int mode_lo, int mode_hi)
{
    if (!gbs_out5) return fail(RL_ERR_INVALID, "rl_probe_hbm: null pointer");
    if (bytes < ((size_t)1 << 20)) return fail(RL_ERR_INVALID, "rl_probe_hbm: at least 1 MiB per buffer");
    int ndev = rl_device_count();
    if (ndev <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    uint4 *a = nullptr, *b = nullptr;
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    int rc = RL_OK;
    const size_t n16 = bytes / 16;
    if (hipMalloc((void **)&a, n16 * 16) != hipSuccess || hipMalloc((void **)&b, n16 * 16) != hipSuccess ||
        hipMalloc((void **)&sink, 4) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMemsetAsync(a, 1, n16 * 16, st) != hipSuccess ||
        hipMemsetAsync(b, 2, n16 * 16, st) != hipSuccess) {
        rc = fail(RL_ERR_NOMEM, "rl_probe_hbm: setup failed (2 x %zu bytes)", n16 * 16);
    } else {
        const int grid = prop.multiProcessorCount * 8, reps = 10;
        for (int mode = mode_lo; mode < mode_hi && rc == RL_OK; ++mode) {
            hipLaunchKernelGGL(hbm_probe_kernel, dim3(grid), dim3(256), 0, st, a, b, n16, mode, sink);     // warm
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r)
                hipLaunchKernelGGL(hbm_probe_kernel, dim3(grid), dim3(256), 0, st, a, b, n16, mode, sink);
            (void)hipEventRecord(e1, st);
            float ms = 0.f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) {
                rc = fail(RL_ERR_HIP, "rl_probe_hbm: launch failed");
                break;
            }
            const double moved = (double)n16 * 16.0 * ((mode == 0 || mode == 3 || mode == 6 || mode == 7) ? 2.0 : 1.0);
            gbs_out5[mode - mode_lo] = moved * reps / ((double)ms * 1e-3) / 1e9;
        }
    }
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

// ---------------------------------------------------------------- 16-bit ranges for the xGMI exchange (opt-in, lossy)
static int u16_args(int device, size_t n, float max_range_m, const void *a, const void *b)
{
    if (!(max_range_m > 0.0f)) return fail(RL_ERR_INVALID, "max_range_m must be > 0");
    if (n > 0 && (!a || !b)) return fail(RL_ERR_INVALID, "null device pointer");
    int ndev = rl_device_count();
    if (ndev <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    return RL_OK;
}

// leading elements until the u16 pointer is 16-B aligned, and whether the f32 pointer is aligned there too
static void u16_split(const void *f32, const void *u16, size_t n, size_t &head, int &vec)
{
    head = ((16 - ((uintptr_t)u16 & 15)) & 15) / 2;
    if (head > n) head = n;
    vec = (((uintptr_t)f32 + 4 * head) & 15) == 0 && ((uintptr_t)u16 & 1) == 0 && ((uintptr_t)f32 & 3) == 0;
}

extern "C" int rl_ranges_to_u16_device(int device, const float *d_ranges, size_t n, float max_range_m,
                                       uint16_t *d_out, void *hip_stream)
{
    int rc = u16_args(device, n, max_range_m, d_ranges, d_out);
    if (rc || n == 0) return rc;
    HIPCHK(hipSetDevice(device));
    size_t head;
    int vec;
    u16_split(d_ranges, d_out, n, head, vec);
    const int grid = (int)std::min<size_t>((n / 8 + 255) / 256 + 1, 256 * 16);
    hipLaunchKernelGGL(ranges_to_u16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, d_ranges, n,
                       max_range_m, 65535.0f / max_range_m, d_out, head, vec);
    HIPCHK(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_ranges_from_u16_device(int device, const uint16_t *d_in, size_t n, float max_range_m,
                                         float *d_ranges, void *hip_stream)
{
    int rc = u16_args(device, n, max_range_m, d_in, d_ranges);
    if (rc || n == 0) return rc;
    HIPCHK(hipSetDevice(device));
    size_t head;
    int vec;
    u16_split(d_ranges, d_in, n, head, vec);
    const int grid = (int)std::min<size_t>((n / 8 + 255) / 256 + 1, 256 * 16);
    hipLaunchKernelGGL(ranges_from_u16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, d_in, n,
                       max_range_m / 65535.0f, d_ranges, head, vec);
    HIPCHK(hipGetLastError());
    return RL_OK;
}

// ---------------------------------------------------------------- diagnostics: gather-rate probe
extern "C" int rl_probe_gather_rate(int device, int active_lanes, double *lanes_per_clk_per_cu,
                                    double *clock_hz, int *n_cu_out)
{
    if (!lanes_per_clk_per_cu) return fail(RL_ERR_INVALID, "rl_probe_gather_rate: null pointer");
    if (active_lanes < 1 || active_lanes > 64) return fail(RL_ERR_INVALID, "active_lanes must be in [1,64]");
    int ndev = rl_device_count();
    if (ndev <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount;
    const double clk = (double)prop.clockRate * 1e3;
    float *tab = nullptr, *sink = nullptr;
    int *d_off = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    int rc = RL_OK;
    auto cleanup = [&]() {
        if (tab) (void)hipFree(tab);
        if (sink) (void)hipFree(sink);
        if (d_off) (void)hipFree(d_off);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (st) (void)hipStreamDestroy(st);
    };
    // random cells of a 32x32 window in the 4-row-interleaved layout of the step map, fixed seed
    int off[64];
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) & 0xffffu; };
    for (int l = 0; l < 64; ++l) {
        const int r = (int)(rnd() % 32), c = (int)(rnd() % 32) + 3;
        off[l] = (r >> 2) * 4 * 64 + 4 * c + (r & 3);
    }
    unsigned long long mask = 0;
    while (__builtin_popcountll(mask) < active_lanes) mask |= 1ull << (rnd() % 64);
    const int iters = 2000, grid = n_cu * 2;
    float ms = 0.f;
    if (hipMalloc((void **)&tab, 4 * 2048 * sizeof(float)) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess ||
        hipMalloc((void **)&d_off, sizeof off) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipMemsetAsync(tab, 0, 4 * 2048 * sizeof(float), st) != hipSuccess ||
        hipMemcpyAsync(d_off, off, sizeof off, hipMemcpyHostToDevice, st) != hipSuccess) {
        rc = fail(RL_ERR_HIP, "gather probe: setup failed");
    } else {
        hipLaunchKernelGGL(gather_probe_kernel, dim3(grid), dim3(1024), 0, st, tab, d_off, mask, 10, sink);
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(gather_probe_kernel, dim3(grid), dim3(1024), 0, st, tab, d_off, mask, iters, sink);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f))
            rc = fail(RL_ERR_HIP, "gather probe: launch failed");
    }
    cleanup();
    if (rc) return rc;
    // 2 workgroups x 16 waves per CU, each iters x 8 wave-loads
    const double clk_per_wave_load = (double)ms * 1e-3 * clk / (2.0 * 16 * iters * 8);
    *lanes_per_clk_per_cu = (double)active_lanes / clk_per_wave_load;
    if (clock_hz) *clock_hz = clk;
    if (n_cu_out) *n_cu_out = n_cu;
    return RL_OK;
}

// ---------------------------------------------------------------- Car outline table and crash test (host)
// Car::setCarEdgeDistances (racecar/src/racecar.cpp:239-292): for every beam, how far from the lidar
// the car's own outline lies.  A one-off table per configuration, so it is host C++ (the crash test
// over scanned batches is fused into the march kernels, see CrashParams).  The reference's quirks are
// part of the contract (the crash codes scripts/mcts.py acts on depend on them): the beam angle is
// advanced BEFORE it is used (the table is shifted by one increment, :256), pi is 3.145
// (racecar.hpp:117), and a beam at exactly 0 rad is nudged to +1e-4 rad while still being treated as a
// non-positive angle, so its side distance is width/2 / sin(-1e-4): about -1016 m, and that beam
// reports a crash for any range (:277-283).  The nudge stays in the running angle.
extern "C" int rl_car_edge_distances(int num_rays, double min_ang, double ang_inc, double scan_dist_to_base,
                                     double width, double wheelbase, double *edge_out)
{
    if (num_rays < 0 || (num_rays > 0 && !edge_out))
        return fail(RL_ERR_INVALID, "rl_car_edge_distances: bad arguments");
    const double quarter_turn = 3.145 / 2.0;
    const double to_side = width / 2.0, to_front = wheelbase - scan_dist_to_base, to_back = scan_dist_to_base;
    double beam = min_ang;
    for (int j = 0; j < num_rays; ++j) {
        beam += ang_inc;
        const bool left = beam > 0.0;                           // decided before the nudge
        if (!left && beam == 0.0) beam += 0.0001;
        const double turned = left ? beam : -beam;              // angle away from straight ahead
        const bool ahead = turned < quarter_turn;               // hits the front edge, else the rear edge
        const double off_axis = ahead ? turned : turned - quarter_turn;
        const double along = (ahead ? to_front : to_back) / cos(off_axis);
        const double across = to_side / sin(off_axis);
        edge_out[j] = across < along ? across : along;
    }
    return RL_OK;
}

// Car::isCrashed (racecar/src/racecar.cpp:305-328) over host ranges: index of the first scan with a
// beam inside the car outline (+ threshold), else -(n_scans + 1).
extern "C" int rl_car_is_crashed(const float *ranges, int num_rays, int n_scans, const double *edge,
                                 double crash_thresh, int *first_crashed)
{
    if (!first_crashed || num_rays < 0 || n_scans < 0 || ((size_t)num_rays * n_scans > 0 && (!ranges || !edge)))
        return fail(RL_ERR_INVALID, "rl_car_is_crashed: bad arguments");
    *first_crashed = -(n_scans + 1);
    for (int k = 0; k < n_scans; ++k) {
        const float *scan = ranges + (size_t)k * num_rays;
        for (int j = 0; j < num_rays; ++j)
            if (((double)scan[j] - edge[j]) < crash_thresh) {
                *first_crashed = k;
                return RL_OK;
            }
    }
    return RL_OK;
}

