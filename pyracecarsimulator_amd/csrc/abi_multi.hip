// abi_multi.hip — the host-pointer entry points of libscan_amd.so (C ABI: include/scanlib.h) and their multi-device forms:
// one contiguous pose block per device of a handle made by rl_map_create_multi, a worker thread per device.
// (scripts/scan_simulator.py:113-135 scanMany, scripts/mcts.py:237 checkCollisionMany: ONE Python process hands over a batch.)
#include "abi_internal.h"

// ------------------------------------------------------------------------------
// multi-device forms of the host-pointer entry points: contiguous pose blocks, one per device, each
// device writing its block of the results straight into the caller's buffer (in a pinned block of
// rl_host_alloc the kernels write it directly: 4 B per ray over that device's own PCIe link).  Noise
// stays keyed by the GLOBAL ray id (the replica's ray offset is the parent's + the block's first ray),
// crash indices are global: the result is bit-identical to the single-device call.
// ------------------------------------------------------------------------------
static int multi_fan(rl_method *h, const float *poses, const float *rows3, int n_poses, float fov, int num_rays,
                     float *outs, int32_t *hits, uint16_t *steps)
{
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->multi_mu);
    if (h->map->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
    const int k = multi_parts(h, n_poses);
    const float nstd = h->noise_std;
    const uint64_t seed = h->noise_seed, off = h->ray_offset;
    std::vector<std::function<int()>> jobs;
    for (int i = 0; i < k; ++i) {
        long lo, hi;
        block_of(n_poses, i, k, lo, hi);
        rl_method *r = h->reps[i];
        const size_t r0 = (size_t)lo * num_rays;
        jobs.push_back([=]() {
            int rc = rl_set_noise(r, nstd, seed, off + r0);
            if (rc) return rc;
            if (rows3)          // the fork's sparse 4-argument layout: pose p in row p * num_rays
                return rl_calc_range_many_fan(r, rows3 + r0 * 3, outs + r0, (int)(hi - lo) * num_rays, fov, num_rays);
            return rl_calc_range_fan(r, poses + 3 * lo, (int)(hi - lo), fov, num_rays, outs + r0,
                                     hits ? hits + 2 * r0 : nullptr, steps ? steps + r0 : nullptr);
        });
    }
    return h->pool->run(jobs);
}

static int multi_rays(rl_method *h, const float *ins, float *outs, int n)
{
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->multi_mu);
    if (h->map->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
    const int k = (int)std::max<long>(1, std::min<long>((long)h->reps.size(), (long)n / (64L * std::max(h->multi_min_poses, 1) * 16)));
    const float nstd = h->noise_std;
    const uint64_t seed = h->noise_seed, off = h->ray_offset;
    std::vector<std::function<int()>> jobs;
    for (int i = 0; i < k; ++i) {
        long lo, hi;
        block_of(n, i, k, lo, hi);
        rl_method *r = h->reps[i];
        jobs.push_back([=]() {
            int rc = rl_set_noise(r, nstd, seed, off + (uint64_t)lo);
            if (rc) return rc;
            return rl_calc_range_many(r, ins + 3 * lo, outs + lo, (int)(hi - lo));
        });
    }
    return h->pool->run(jobs);
}

// groups of `group` poses (group == n_poses, n_groups == 1 with `single`: rl_check_collision_many's one index)
static int multi_crash(rl_method *h, const float *poses, int n_groups, int group, float fov, int num_rays,
                       const double *edge, double thresh, int *first_crashed, float *ranges, bool single)
{
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->multi_mu);
    if (h->map->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
    const long n_units = single ? group : n_groups;              // what is cut: poses of the one batch | roll-outs
    const long poses_per_unit = single ? 1 : group;
    const int k = (int)std::max<long>(1, std::min<long>(multi_parts(h, n_units * poses_per_unit), n_units));
    const float nstd = h->noise_std;
    const uint64_t seed = h->noise_seed, off = h->ray_offset;
    std::vector<int> part(k, 0);
    std::vector<long> los(k, 0), his(k, 0);
    std::vector<std::function<int()>> jobs;
    for (int i = 0; i < k; ++i) {
        long lo, hi;
        block_of(n_units, i, k, lo, hi);
        los[i] = lo;
        his[i] = hi;
        rl_method *r = h->reps[i];
        const size_t p0 = (size_t)lo * poses_per_unit, r0 = p0 * num_rays;
        int *res = single ? &part[i] : first_crashed + lo;
        jobs.push_back([=]() {
            if (hi <= lo) return (int)RL_OK;
            int rc = rl_set_noise(r, nstd, seed, off + r0);
            if (rc) return rc;
            if (single)
                return rl_check_collision_many(r, poses + 3 * p0, (int)(hi - lo), fov, num_rays, edge, thresh, res,
                                               ranges ? ranges + r0 : nullptr);
            return rl_check_collision_groups(r, poses + 3 * p0, (int)(hi - lo), group, fov, num_rays, edge, thresh, res,
                                             ranges ? ranges + r0 : nullptr);
        });
    }
    const int rc = h->pool->run(jobs);
    if (rc) return rc;
    if (single) {
        *first_crashed = -(group + 1);                            // Car::isCrashed: -(poses + 1) when none crashed
        for (int i = 0; i < k; ++i)
            if (his[i] > los[i] && part[i] >= 0) {
                *first_crashed = (int)los[i] + part[i];
                break;
            }
    }
    return RL_OK;
}

// ------------------------------------------------------------------------------
// DEVICE-RESIDENT exchange of the one-process multi-device handle (round 6).  The reference's caller is ONE process
// (scripts/mcts.py:237 -> scripts/racecar_simulator_v2.py:146-167): the only way it ever moves ranges over xGMI is a
// library that does so behind one call.  Every replica marches its contiguous pose block into its own HBM, chunk by
// chunk; behind every chunk its copy stream sends the chunk to the CONSUMER device with hipMemcpyPeerAsync (device to
// device over xGMI once peer access is enabled; the runtime stages through the host where it is not), so chunk k
// travels under chunk k + 1's march.  The consumer's own block is marched straight into the destination.  Worker
// threads as everywhere in this unit: nothing is forked or re-executed after GPU initialisation.  Results bit-identical
// to the single-device scan (noise keyed by the global ray id, crash indices global).
// ------------------------------------------------------------------------------
static int ensure_copy_stream(rl_method *r)
{
    if (!r->copy_stream) HIPCHK(hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t &e : r->slice_ev)
        if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return RL_OK;
}

// (device `from` is current) let its copies reach `to` directly; a refusal only means staged copies
static void try_peer(int from, int to)
{
    if (from == to) return;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(to, 0);
    (void)hipGetLastError();                            // ("already enabled" is not an error worth keeping)
}

static int replica_fan_to_consumer(rl_method *r, const float *poses_blk, int np, float fov, int num_rays, float nstd,
                                   uint64_t seed, uint64_t off0, bool is_consumer, int consumer_dev, float *d_dst, int chunks)
{
    if (np <= 0) return RL_OK;
    int rc = set_device(r->map);
    if (rc) return rc;
    const int dev = r->map->device;
    const size_t n_rays = (size_t)np * num_rays;
    float *d_poses = nullptr, *d_local = nullptr;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if ((rc = r->poses.ensure((size_t)np * 12))) return rc;
        if (!is_consumer && (rc = r->outs.ensure(n_rays * sizeof(float)))) return rc;
        if ((rc = ensure_copy_stream(r))) return rc;
        d_poses = (float *)r->poses.p;
        d_local = is_consumer ? d_dst : (float *)r->outs.p;
    }
    if (!is_consumer) try_peer(dev, consumer_dev);
    HIPCHK(hipMemcpyAsync(d_poses, poses_blk, (size_t)np * 12, hipMemcpyHostToDevice, r->stream));
    const int k = std::max(1, std::min(chunks, np));
    for (int c = 0; c < k; ++c) {
        long lo, hi;
        block_of(np, c, k, lo, hi);
        const size_t r0 = (size_t)lo * num_rays, nr = (size_t)(hi - lo) * num_rays;
        if ((rc = rl_set_noise(r, nstd, seed, off0 + r0))) return rc;
        if ((rc = rl_calc_range_fan_device(r, d_poses + 3 * lo, (int)(hi - lo), fov, num_rays, d_local + r0, nullptr, nullptr,
                                           (void *)r->stream)))
            return rc;
        if (!is_consumer) {
            hipEvent_t ev = r->slice_ev[c & 3];
            HIPCHK(hipEventRecord(ev, r->stream));
            HIPCHK(hipStreamWaitEvent(r->copy_stream, ev, 0));
            HIPCHK(hipMemcpyPeerAsync(d_dst + r0, consumer_dev, d_local + r0, dev, nr * sizeof(float), r->copy_stream));
        }
    }
    HIPCHK(hipStreamSynchronize(r->stream));
    if (!is_consumer) HIPCHK(hipStreamSynchronize(r->copy_stream));
    return RL_OK;
}

static int replica_crash_to_consumer(rl_method *r, const float *poses_blk, int n_groups, int group, float fov, int num_rays,
                                     const double *edge, double thresh, float nstd, uint64_t seed, uint64_t off0,
                                     bool is_consumer, int consumer_dev, int *d_dst)
{
    if (n_groups <= 0) return RL_OK;
    int rc = set_device(r->map);
    if (rc) return rc;
    const int dev = r->map->device;
    const size_t np = (size_t)n_groups * group;
    float *d_poses = nullptr;
    int *d_local = nullptr;
    const double *d_edge = nullptr;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if ((rc = r->poses.ensure(np * 12)) || (rc = upload_edge(r, edge, num_rays))) return rc;
        if (!is_consumer && (rc = r->flag.ensure((size_t)n_groups * sizeof(int)))) return rc;
        if ((rc = ensure_copy_stream(r))) return rc;
        d_poses = (float *)r->poses.p;
        d_edge = (const double *)r->edge.p;
        d_local = is_consumer ? d_dst : (int *)r->flag.p;
    }
    if (!is_consumer) try_peer(dev, consumer_dev);
    HIPCHK(hipMemcpyAsync(d_poses, poses_blk, np * 12, hipMemcpyHostToDevice, r->stream));
    if ((rc = rl_set_noise(r, nstd, seed, off0))) return rc;
    if ((rc = rl_check_collision_groups_device(r, d_poses, n_groups, group, fov, num_rays, d_edge, thresh, d_local, nullptr,
                                               (void *)r->stream)))
        return rc;
    if (!is_consumer)
        HIPCHK(hipMemcpyPeerAsync(d_dst, consumer_dev, d_local, dev, (size_t)n_groups * sizeof(int), r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    return RL_OK;
}

static int multi_device_check(rl_method *h, int consumer, const void *dst, const char *fn)
{
    if (!h) return fail(RL_ERR_INVALID, "%s: null method handle", fn);
    if (h->reps.empty()) return fail(RL_ERR_INVALID, "%s needs a method of a multi-device map (rl_map_create_multi)", fn);
    if (consumer < 0 || consumer >= (int)h->reps.size())
        return fail(RL_ERR_INVALID, "%s: consumer %d is not a replica index of this handle (0..%zu)", fn, consumer, h->reps.size() - 1);
    if (!dst) return fail(RL_ERR_INVALID, "%s: null destination", fn);
    return RL_OK;
}

extern "C" int rl_calc_range_fan_multi_device(rl_method *h, const float *poses, int n_poses, float fov, int num_rays,
                                              int consumer, float *d_outs_on_consumer, int chunks)
{
    int rc = check_fan_args(h, n_poses, fov, num_rays);
    if (rc) return rc;
    if (n_poses == 0) return RL_OK;
    if ((rc = multi_device_check(h, consumer, d_outs_on_consumer, "rl_calc_range_fan_multi_device"))) return rc;
    if (!poses) return fail(RL_ERR_INVALID, "rl_calc_range_fan_multi_device: null pose pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->multi_mu);
    if (h->map->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
    // (every device takes part from multi_min_poses poses per device up; the consumer always owns a block — block 0 is its own)
    const int k = multi_parts(h, n_poses);
    const int consumer_dev = h->reps[consumer]->map->device;
    const float nstd = h->noise_std;
    const uint64_t seed = h->noise_seed, off = h->ray_offset;
    // block i goes to replica order[i]: the consumer first, so that a batch too small for every device stays where it is wanted
    std::vector<int> order;
    order.push_back(consumer);
    for (int i = 0; i < (int)h->reps.size(); ++i)
        if (i != consumer) order.push_back(i);
    // (jobs[0] runs on the calling thread, jobs[i] on the worker of replica i: hand replica order[b]'s block to ITS thread)
    std::vector<std::function<int()>> jobs(h->reps.size(), []() { return (int)RL_OK; });
    for (int b = 0; b < k; ++b) {
        long lo, hi;
        block_of(n_poses, b, k, lo, hi);
        rl_method *r = h->reps[order[b]];
        const size_t r0 = (size_t)lo * num_rays;
        const bool is_c = order[b] == consumer;
        const int ch = chunks > 0 ? chunks : 4;
        jobs[order[b]] = [=]() {
            return replica_fan_to_consumer(r, poses + 3 * lo, (int)(hi - lo), fov, num_rays, nstd, seed, off + r0, is_c,
                                           consumer_dev, d_outs_on_consumer + r0, ch);
        };
    }
    while (jobs.size() > 1 && std::find(order.begin(), order.begin() + k, (int)jobs.size() - 1) == order.begin() + k) jobs.pop_back();
    return h->pool->run(jobs);
}

extern "C" int rl_check_collision_groups_multi_device(rl_method *h, const float *poses, int n_groups, int group, float fov,
                                                      int num_rays, const double *edge, double crash_thresh, int consumer,
                                                      int *d_first_on_consumer)
{
    if (n_groups < 0 || group <= 0) return fail(RL_ERR_INVALID, "n_groups >= 0 and group > 0 required");
    if ((long)n_groups * group > INT_MAX) return fail(RL_ERR_INVALID, "too many poses");
    int rc = check_fan_args(h, n_groups * group, fov, num_rays);
    if (rc) return rc;
    if (n_groups == 0) return RL_OK;
    if ((rc = multi_device_check(h, consumer, d_first_on_consumer, "rl_check_collision_groups_multi_device"))) return rc;
    if (!poses || !edge) return fail(RL_ERR_INVALID, "rl_check_collision_groups_multi_device: null pointer");
    if (h->kind != RL_RM && h->kind != RL_RM_GPU) return fail(RL_ERR_UNSUPPORTED, "fused crash test needs a ray-marching method");
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->multi_mu);
    if (h->map->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
    const int k = (int)std::max<long>(1, std::min<long>(multi_parts(h, (long)n_groups * group), n_groups));
    const int consumer_dev = h->reps[consumer]->map->device;
    const float nstd = h->noise_std;
    const uint64_t seed = h->noise_seed, off = h->ray_offset;
    std::vector<int> order;
    order.push_back(consumer);
    for (int i = 0; i < (int)h->reps.size(); ++i)
        if (i != consumer) order.push_back(i);
    std::vector<std::function<int()>> jobs(h->reps.size(), []() { return (int)RL_OK; });
    for (int b = 0; b < k; ++b) {
        long lo, hi;
        block_of(n_groups, b, k, lo, hi);
        rl_method *r = h->reps[order[b]];
        const size_t p0 = (size_t)lo * group;
        const bool is_c = order[b] == consumer;
        jobs[order[b]] = [=]() {
            return replica_crash_to_consumer(r, poses + 3 * p0, (int)(hi - lo), group, fov, num_rays, edge, crash_thresh, nstd, seed,
                                             off + p0 * num_rays, is_c, consumer_dev, d_first_on_consumer + lo);
        };
    }
    while (jobs.size() > 1 && std::find(order.begin(), order.begin() + k, (int)jobs.size() - 1) == order.begin() + k) jobs.pop_back();
    return h->pool->run(jobs);
}

extern "C" int rl_calc_range_fan(rl_method *h, const float *poses, int n_poses, float fov,
                                 int num_rays, float *outs, int32_t *hits, uint16_t *steps)
{
    int rc = check_fan_args(h, n_poses, fov, num_rays);
    if (rc) return rc;
    if (n_poses > 0 && (!poses || !outs))
        return fail(RL_ERR_INVALID, "rl_calc_range_fan: null pointer");
    if (!h->reps.empty()) return n_poses ? multi_fan(h, poses, nullptr, n_poses, fov, num_rays, outs, hits, steps) : RL_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    return fan_host(h, poses, n_poses, fov, num_rays, outs, hits, steps, nullptr, 0.0, nullptr);
}

extern "C" int rl_calc_range_many_fan(rl_method *h, const float *ins_rows3, float *outs, int n_rows,
                                      float fov, int num_rays)
{
    if (!h) return fail(RL_ERR_INVALID, "null method handle");
    if (num_rays <= 0) return fail(RL_ERR_INVALID, "num_rays must be > 0");
    if (n_rows < 0) return fail(RL_ERR_INVALID, "n_rows must be >= 0");
    // n_poses = ins.shape[0] / num_rays; trailing rows that do
    // not make a whole fan are left untouched
    const int n_poses = n_rows / num_rays;
    int rc = check_fan_args(h, n_poses, fov, num_rays);
    if (rc) return rc;
    if (n_poses > 0 && (!ins_rows3 || !outs))
        return fail(RL_ERR_INVALID, "rl_calc_range_many_fan: null pointer");
    if (!h->reps.empty()) return n_poses ? multi_fan(h, nullptr, ins_rows3, n_poses, fov, num_rays, outs, nullptr, nullptr) : RL_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    // gather the live row of every pose (row p*num_rays): 12 B per pose cross PCIe,
    // not the reference's 12 B per ray (scripts/scan_simulator.py:39-40)
    h->h_poses.resize((size_t)n_poses * 3);
    for (int p = 0; p < n_poses; ++p) {
        const float *row = ins_rows3 + (size_t)p * num_rays * 3;
        h->h_poses[3 * (size_t)p] = row[0];
        h->h_poses[3 * (size_t)p + 1] = row[1];
        h->h_poses[3 * (size_t)p + 2] = row[2];
    }
    return fan_host(h, h->h_poses.data(), n_poses, fov, num_rays, outs, nullptr, nullptr, nullptr,
                    0.0, nullptr);
}

extern "C" int rl_calc_range_many(rl_method *h, const float *ins, float *outs, int n)
{
    if (!h) return fail(RL_ERR_INVALID, "null method handle");
    if (n < 0) return fail(RL_ERR_INVALID, "n must be >= 0");
    if (n == 0) return RL_OK;
    if (!ins || !outs) return fail(RL_ERR_INVALID, "rl_calc_range_many: null pointer");
    if (!h->reps.empty()) return multi_rays(h, ins, outs, n);
    return rays_host(h, ins, outs, n);
}

extern "C" int rl_check_collision_many(rl_method *h, const float *poses, int n_poses, float fov,
                                       int num_rays, const double *edge, double crash_thresh,
                                       int *first_crashed, float *ranges_or_null)
{
    int rc = check_fan_args(h, n_poses, fov, num_rays);
    if (rc) return rc;
    if (!first_crashed || !edge || (n_poses > 0 && !poses))
        return fail(RL_ERR_INVALID, "rl_check_collision_many: null pointer");
    if (n_poses == 0) {
        *first_crashed = -1;
        return RL_OK;
    }
    if (!h->reps.empty())
        return multi_crash(h, poses, 1, n_poses, fov, num_rays, edge, crash_thresh, first_crashed, ranges_or_null, true);
    if (h->kind != RL_RM && h->kind != RL_RM_GPU)      // generic: scan, then one crash pass
        return rl_check_collision_groups(h, poses, 1, n_poses, fov, num_rays, edge, crash_thresh,
                                         first_crashed, ranges_or_null);
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    return fan_host(h, poses, n_poses, fov, num_rays, ranges_or_null, nullptr, nullptr, edge,
                    crash_thresh, first_crashed);
}


extern "C" int rl_check_collision_groups(rl_method *h, const float *poses, int n_groups, int group,
                                         float fov, int num_rays, const double *edge,
                                         double crash_thresh, int *first_crashed, float *ranges_or_null)
{
    if (n_groups < 0 || group <= 0) return fail(RL_ERR_INVALID, "n_groups >= 0 and group > 0 required");
    const long n_poses_l = (long)n_groups * group;
    if (n_poses_l > INT_MAX) return fail(RL_ERR_INVALID, "too many poses");
    const int n_poses = (int)n_poses_l;
    int rc = check_fan_args(h, n_poses, fov, num_rays);
    if (rc) return rc;
    if (n_groups == 0) return RL_OK;
    if (!poses || !edge || !first_crashed) return fail(RL_ERR_INVALID, "rl_check_collision_groups: null pointer");
    if (!h->reps.empty())
        return multi_crash(h, poses, n_groups, group, fov, num_rays, edge, crash_thresh, first_crashed, ranges_or_null, false);
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    if ((rc = set_device(h->map))) return rc;
    const size_t n_rays = (size_t)n_poses * num_rays;
    if ((rc = h->poses.ensure((size_t)n_poses * 12)) || (rc = h->outs.ensure(n_rays * 4)) ||
        (rc = upload_edge(h, edge, num_rays)) || (rc = h->flag.ensure((size_t)n_groups * 4)))
        return rc;
    HIPCHK(hipMemcpyAsync(h->poses.p, poses, (size_t)n_poses * 12, hipMemcpyHostToDevice, h->stream));
    rc = crash_groups_device(h, (const float *)h->poses.p, n_groups, group, fov, num_rays,
                             (const double *)h->edge.p, crash_thresh, (int *)h->flag.p,
                             (float *)h->outs.p, true, h->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(first_crashed, h->flag.p, (size_t)n_groups * 4, hipMemcpyDeviceToHost, h->stream));
    if (ranges_or_null)
        HIPCHK(hipMemcpyAsync(ranges_or_null, h->outs.p, n_rays * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RL_OK;
}

