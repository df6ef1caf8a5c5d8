// abi_fan.hip — the method handle of libscan_amd.so (C ABI: include/scanlib.h): options, derived tables, launch planning,
// every fan / ray launch, the device-pointer entry points, the single-device host-pointer paths, the fused crash test.
//
// Replaces, for the scan path only, range_libc's PyRayMarching / PyRayMarchingGPU / PyBresenhamsLine / PyCDDTCast /
// PyGiantLUTCast objects that the reference builds at scripts/scan_simulator.py:72-76, scripts/ros_interface.py:210 and
// scripts/two_player/scan.py:45-46.  There is no CPU fallback in this library.
#include "abi_internal.h"
#include "scan_kernels.h"
#include "crash_kernels.h"
#include "launch_plan.h"

static_assert(plan::WG == scan::WG && plan::STREAM_HDR == scan::STREAM_HDR && plan::STRIPE_BINS == scan::STRIPE_BINS &&
                  plan::STRIPE_MAX_PER_LANE == scan::STRIPE_MAX_PER_LANE && plan::DRAIN_CAP == scan::DRAIN_CAP &&
                  plan::DRAIN_FIELDS == scan::DRAIN_FIELDS && plan::INLINE_REC_BYTES == (int)sizeof(scan::BlockRec),
              "launch_plan.h and rm_kernels.h disagree about the stream kernels' LDS layout");

// ------------------------------------------------------------------------------
// method
// ------------------------------------------------------------------------------
extern "C" int rl_method_create(rl_map *m, int kind, float max_range_px, int theta_disc,
                                rl_method **out)
{
    if (!m || !out) return fail(RL_ERR_INVALID, "rl_method_create: null pointer");
    if (!(max_range_px > 0.0f)) return fail(RL_ERR_INVALID, "max_range_px must be > 0");
    if (kind < RL_BRESENHAM || kind > RL_GIANT_LUT)
        return fail(RL_ERR_INVALID, "unknown range method kind %d", kind);
    if ((kind == RL_CDDT || kind == RL_GIANT_LUT) && (theta_disc < 2 || theta_disc > 65536))
        return fail(RL_ERR_INVALID, "theta_disc must be in [2, 65536] for CDDT / GiantLUT (got %d)",
                    theta_disc);
    rl_method *h = new (std::nothrow) rl_method();
    if (!h) return fail(RL_ERR_NOMEM, "out of host memory");
    h->map = m;
    h->kind = kind;
    h->max_range = max_range_px;
    h->theta_disc = theta_disc;
    h->step_coeff = kind == RL_RM_GPU ? 1.0f : 0.999f;   // kernels.cu STEP_COEFF vs RayMarching (also seeds the LUT)
    h->variant = plan::default_variant(kind);             // RL_RM: the upstream-literal arithmetic; the others canonical
    if (!m->reps.empty()) {
        // multi-device: one ordinary method per device replica of the map + one worker thread per extra device
        std::vector<int> devs;
        for (rl_map *rm : m->reps) {
            rl_method *r = nullptr;
            const int rc = rl_method_create(rm, kind, max_range_px, theta_disc, &r);
            if (rc) {
                const std::string keep = last_error();
                rl_method_destroy(h);
                set_last_error(keep);
                return rc;
            }
            h->reps.push_back(r);
            devs.push_back(rm->device);
        }
        h->pool = std::make_unique<MultiPool>();
        h->pool->start(devs);
        *out = h;
        return RL_OK;
    }
    if (kind == RL_CDDT) {
        // the map starts keeping its edge list (and rebuilds it with every rl_map_update)
        std::lock_guard<std::mutex> lk(m->mu);
        std::unique_lock<std::shared_mutex> wl(m->tables_mu);
        if (!m->want_edges) {
            m->want_edges = true;
            int rc_ = hipSetDevice(m->device) == hipSuccess ? map_build_tables(m) : fail(RL_ERR_HIP, "hipSetDevice failed");
            if (rc_) {
                m->want_edges = false;
                delete h;
                return rc_;
            }
        }
    }
    if (hipSetDevice(m->device) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        rl_method_destroy(h);
        return fail(RL_ERR_HIP, "stream/event creation failed");
    }
    *out = h;
    return RL_OK;
}

extern "C" void rl_method_destroy(rl_method *h)
{
    if (!h) return;
    if (!h->reps.empty() || h->pool) {
        h->pool.reset();                       // (joins the workers: no job is in flight, the caller owns the handle)
        for (rl_method *r : h->reps) rl_method_destroy(r);
        delete h;
        return;
    }
    if (h->map) (void)hipSetDevice(h->map->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->poses.release();
    h->outs.release();
    h->hits.release();
    h->steps.release();
    h->edge.release();
    h->flag.release();
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->pin_flag) (void)hipHostFree(h->pin_flag);
    for (LaunchCtx &c : h->ctx) c.release();
    for (TableDep *d : {&h->pdt_dep, &h->lut_dep, &h->cddt_dep, &h->blpad_dep})
        if (d->ev) (void)hipEventDestroy(d->ev);
    for (auto &ft : h->fan_tabs) {
        if (ft.dep.ev) (void)hipEventDestroy(ft.dep.ev);
        ft.tab.release();
    }
    h->pdt.release();
    for (DevBuf *b : {&h->cmap, &h->cval, &h->cidx, &h->ctab, &h->cnum}) b->release();
    if (h->pin_cnum) (void)hipHostFree(h->pin_cnum);
    h->blpad.release();
    h->lut.release();
    for (DevBuf *b : {&h->cd_cos, &h->cd_sin, &h->cd_trans, &h->cd_width, &h->cd_boff, &h->cd_offsets,
                      &h->cd_xs2, &h->cd_cursor, &h->cd_tmp, &h->cd_hdr, &h->cd_tab})
        b->release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->slice_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int rl_method_kind(const rl_method *h) { return h ? h->kind : -1; }

extern "C" int rl_method_n_devices(const rl_method *h) { return h ? (h->reps.empty() ? 1 : (int)h->reps.size()) : 0; }

extern "C" rl_method *rl_method_replica(rl_method *h, int i)
{
    if (!h) return nullptr;
    if (h->reps.empty()) return i == 0 ? h : nullptr;
    return (i >= 0 && i < (int)h->reps.size()) ? h->reps[i] : nullptr;
}

static int cddt_table_stats(rl_method *h, const char *name, int64_t *value_out);

// how many devices of a multi-device handle a batch of n_poses is cut over
int multi_parts(const rl_method *h, long n_poses)
{
    const long by_size = n_poses / std::max(h->multi_min_poses, 1);
    return (int)std::max<long>(1, std::min<long>((long)h->reps.size(), by_size));
}

static int multi_needs_replica(const char *fn)
{
    return fail(RL_ERR_INVALID, "%s: device pointers belong to one device — on a multi-device handle call it with "
                                "rl_method_replica(h, i)", fn);
}

extern "C" int rl_set_noise(rl_method *h, float std, uint64_t seed, uint64_t ray_offset)
{
    if (!h) return fail(RL_ERR_INVALID, "rl_set_noise: null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    h->noise_std = std;
    h->noise_seed = seed;
    h->ray_offset = ray_offset;
    return RL_OK;
}

extern "C" int rl_method_set_option(rl_method *h, const char *name, int value)
{
    if (!h || !name) return fail(RL_ERR_INVALID, "rl_method_set_option: null pointer");
    if (!h->reps.empty()) {
        std::lock_guard<std::mutex> lk(h->mu);
        if (!strcmp(name, "multi_min_poses")) {
            h->multi_min_poses = value < 1 ? 1 : value;
            return RL_OK;
        }
        for (rl_method *r : h->reps) {
            const int rc = rl_method_set_option(r, name, value);
            if (rc) return rc;
        }
        return RL_OK;
    }
    std::lock_guard<std::mutex> lk(h->mu);
    if (!strcmp(name, "variant")) h->variant = value < 0 ? plan::default_variant(h->kind) : value;
    else if (!strcmp(name, "grid_mult")) h->grid_mult = value < 1 ? 1 : value;
    else if (!strcmp(name, "low_water")) h->low_water = value < 0 ? -1 : (value > 63 ? 63 : value);
    else if (!strcmp(name, "sort_poses")) h->sort_poses = value != 0;
    else if (!strcmp(name, "debug_stamps")) h->debug_stamps = value != 0;
    else if (!strcmp(name, "drain_prio")) h->drain_prio = value != 0;
    else if (!strcmp(name, "spec_drain")) h->spec_drain = value < 0 ? 0 : (value > 64 ? 64 : value);
    else if (!strcmp(name, "spec_stretch")) h->spec_stretch = value < 1 ? 1 : (value > 4096 ? 4096 : value);
    else if (!strcmp(name, "drain_cap")) h->drain_cap = value < 1 ? 1 : (value > 64 ? 64 : value);
    else if (!strcmp(name, "group_drain")) h->group_drain = value < 0 ? 0 : (value > 16 ? 16 : value);
    else if (!strcmp(name, "handoff")) h->handoff = value != 0;
    else if (!strcmp(name, "handoff_cap")) h->handoff_cap = value >= 64 ? 64 : (value >= 32 ? 32 : (value >= 16 ? 16 : 8));
    else if (!strcmp(name, "handoff_wg")) h->handoff_wg = value >= 256 ? 256 : (value >= 128 ? 128 : 64);
    else if (!strcmp(name, "drain_stretch")) h->drain_stretch = value < 1 ? 1 : (value > 4096 ? 4096 : value);
    else if (!strcmp(name, "nt_store")) h->nt_store = value != 0;
    else if (!strcmp(name, "timing")) h->timing = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(name, "bin_multi_min")) h->bin_multi_min = value;
    else if (!strcmp(name, "bin_ppw")) h->bin_ppw = value < 256 ? 256 : (value > 8192 ? 8192 : value);
    else if (!strcmp(name, "tile_stripe")) h->tile_stripe = value < 0 ? -1 : (value > 4096 ? 4096 : value);
    else if (!strcmp(name, "inline_prep")) h->inline_prep = value != 0;
    else if (!strcmp(name, "bin_generic")) h->bin_generic = value != 0;
    else if (!strcmp(name, "tiled")) h->tiled = value != 0;
    else if (!strcmp(name, "code_map")) h->code_map = value == 2 ? 2 : 0;
    else if (!strcmp(name, "code_min_rays")) h->code_min_rays = value < 0 ? 0 : value;
    else if (!strcmp(name, "tail_pct")) h->tail_pct = value < 0 ? 0 : (value > 75 ? 75 : value);
    else if (!strcmp(name, "tail_wg_pct")) h->tail_wg_pct = value < 10 ? 10 : (value > 400 ? 400 : value);
    else if (!strcmp(name, "pinned_max_rays")) h->pinned_max_rays = value < 0 ? 0 : value;
    else if (!strcmp(name, "direct_max_rays")) h->direct_max_rays = value < 0 ? 0 : value;
    else if (!strcmp(name, "overlap_min_rays")) h->overlap_min_rays = value < 0 ? 0 : value;
    else if (!strcmp(name, "inline_map_kb")) h->inline_map_kb = value < 0 ? 0 : value;
    else if (!strcmp(name, "stripe_max")) h->stripe_max = value < 0 ? 0 : value;
    else if (!strcmp(name, "order_inline")) h->order_inline = value != 0;
    else if (!strcmp(name, "run_log2")) h->run_log2 = value < 0 ? -1 : value > 8 ? 8 : value;
    else if (!strcmp(name, "slice_log2")) h->slice_log2 = value < 8 ? 8 : (value > 30 ? 30 : value);
    else if (!strcmp(name, "inline_max")) h->inline_max = value;
    else if (!strcmp(name, "lut_debug")) { h->lut_debug = value; h->lp.debug = value; h->cdp.debug = value; }
    else if (!strcmp(name, "wg_threads")) h->wg_threads = value >= 1024 ? 1024 : (value >= 512 ? 512 : 256);
    else if (!strcmp(name, "xcd_bands")) h->xcd_bands = value < 1 ? 1 : value;
    else if (!strcmp(name, "slots")) h->slots = value >= 3 ? 3 : (value == 2 ? 2 : (value == 1 ? 1 : 0));
    else if (!strcmp(name, "cddt_bins")) h->cddt_bins_kernel = value != 0;
    else if (!strcmp(name, "cddt_search")) h->cddt_search = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (!strcmp(name, "cddt_sort")) h->cddt_sort = value != 0;
    else if (!strcmp(name, "cddt_theta_min")) h->cddt_theta_min = value < 0 ? 0 : value;
    else if (!strcmp(name, "cddt_lds_sort")) {
        int v = 128;                                   // a power of two in [128, CDDT_LDS_SORT]: the bitonic network pads to one
        while (v * 2 <= value && v * 2 <= (int)CDDT_LDS_SORT) v *= 2;
        h->cddt_lds_sort = v;
        h->cddt_epoch = ~0ull;
    }
    else return fail(RL_ERR_INVALID, "unknown option '%s'", name);
    return RL_OK;
}

extern "C" int rl_method_get_info(rl_method *h, const char *name, int64_t *value_out)
{
    if (!h || !name || !value_out) return fail(RL_ERR_INVALID, "rl_method_get_info: null pointer");
    if (!strcmp(name, "n_devices")) { *value_out = h->reps.empty() ? 1 : (int64_t)h->reps.size(); return RL_OK; }
    if (!h->reps.empty()) {
        if (!strcmp(name, "multi_min_poses")) { *value_out = h->multi_min_poses; return RL_OK; }
        return rl_method_get_info(h->reps[0], name, value_out);
    }
    if (!strcmp(name, "cddt_values") || !strcmp(name, "cddt_buckets") || !strcmp(name, "cddt_nonempty_buckets"))
        return cddt_table_stats(h, name, value_out);
    if (!strcmp(name, "n_cu")) *value_out = h->map->n_cu;
    else if (!strcmp(name, "clock_khz")) *value_out = h->map->clock_khz;
    else if (!strcmp(name, "variant")) *value_out = h->variant;
    else if (!strcmp(name, "grid_mult")) *value_out = h->grid_mult;
    else if (!strcmp(name, "low_water")) *value_out = h->low_water;
    else if (!strcmp(name, "sort_poses")) *value_out = h->sort_poses;
    else if (!strcmp(name, "debug_stamps")) *value_out = h->debug_stamps;
    else if (!strcmp(name, "drain_prio")) *value_out = h->drain_prio;
    else if (!strcmp(name, "spec_drain")) *value_out = h->spec_drain;
    else if (!strcmp(name, "spec_stretch")) *value_out = h->spec_stretch;
    else if (!strcmp(name, "drain_cap")) *value_out = h->drain_cap;
    else if (!strcmp(name, "group_drain")) *value_out = h->group_drain;
    else if (!strcmp(name, "handoff")) *value_out = h->handoff;
    else if (!strcmp(name, "handoff_cap")) *value_out = h->handoff_cap;
    else if (!strcmp(name, "handoff_wg")) *value_out = h->handoff_wg;
    else if (!strcmp(name, "drain_stretch")) *value_out = h->drain_stretch;
    else if (!strcmp(name, "nt_store")) *value_out = h->nt_store;
    else if (!strcmp(name, "timing")) *value_out = h->timing;
    else if (!strcmp(name, "bin_multi_min")) *value_out = h->bin_multi_min;
    else if (!strcmp(name, "bin_ppw")) *value_out = h->bin_ppw;
    else if (!strcmp(name, "tile_stripe")) *value_out = h->tile_stripe;
    else if (!strcmp(name, "inline_prep")) *value_out = h->inline_prep;
    else if (!strcmp(name, "bin_generic")) *value_out = h->bin_generic;
    else if (!strcmp(name, "tiled")) *value_out = h->tiled;
    else if (!strcmp(name, "code_map")) *value_out = h->code_map;
    else if (!strcmp(name, "code_min_rays")) *value_out = h->code_min_rays;
    else if (!strcmp(name, "tail_pct")) *value_out = h->tail_pct;
    else if (!strcmp(name, "tail_wg_pct")) *value_out = h->tail_wg_pct;
    else if (!strcmp(name, "code_entries")) *value_out = h->code_n;
    else if (!strcmp(name, "pinned_max_rays")) *value_out = h->pinned_max_rays;
    else if (!strcmp(name, "direct_max_rays")) *value_out = h->direct_max_rays;
    else if (!strcmp(name, "overlap_min_rays")) *value_out = h->overlap_min_rays;
    else if (!strcmp(name, "inline_map_kb")) *value_out = h->inline_map_kb;
    else if (!strcmp(name, "stripe_max")) *value_out = h->stripe_max;
    else if (!strcmp(name, "order_inline")) *value_out = h->order_inline;
    else if (!strcmp(name, "run_log2")) *value_out = h->run_log2;
    else if (!strcmp(name, "slice_log2")) *value_out = h->slice_log2;
    else if (!strcmp(name, "inline_max")) *value_out = h->inline_max;
    else if (!strcmp(name, "wg_threads")) *value_out = h->wg_threads;
    else if (!strcmp(name, "last_grid")) *value_out = h->last_grid;
    else if (!strcmp(name, "xcd_bands")) *value_out = h->xcd_bands;
    else if (!strcmp(name, "slots")) *value_out = h->slots;
    else if (!strcmp(name, "cddt_bins")) *value_out = h->cddt_bins_kernel;
    else if (!strcmp(name, "cddt_search")) *value_out = h->cddt_search;
    else if (!strcmp(name, "cddt_sort")) *value_out = h->cddt_sort;
    else if (!strcmp(name, "cddt_theta_min")) *value_out = h->cddt_theta_min;
    else if (!strcmp(name, "cddt_lds_sort")) *value_out = h->cddt_lds_sort;
    else if (!strcmp(name, "map_epoch")) *value_out = (int64_t)h->map->epoch;
    else return fail(RL_ERR_INVALID, "unknown info '%s'", name);
    return RL_OK;
}

// ------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------
static FanParams make_fan(const rl_method *h, int n_poses, float fov, int num_rays)
{
    FanParams f{};
    f.n_poses = n_poses;
    f.num_rays = num_rays;
    f.amin = -0.5f * fov;
    f.inc = fov / (float)num_rays;
    f.max_range = h->max_range;
    f.step_coeff = h->step_coeff;
    f.noise_std = h->noise_std;
    f.noise_seed = h->noise_seed;
    f.ray_offset = h->ray_offset;
    return f;
}

int check_fan_args(const rl_method *h, int n_poses, float fov, int num_rays)
{
    if (!h) return fail(RL_ERR_INVALID, "null method handle");
    if (n_poses < 0) return fail(RL_ERR_INVALID, "n_poses must be >= 0");
    if (num_rays <= 0) return fail(RL_ERR_INVALID, "num_rays must be > 0");
    if (num_rays > 7680)      // 8 B per beam in LDS next to the other per-workgroup state (<= 64 KiB)
        return fail(RL_ERR_UNSUPPORTED, "num_rays %d exceeds the LDS fan table (7680 beams)", num_rays);
    if (!(fov == fov)) return fail(RL_ERR_INVALID, "fov is NaN");
    return RL_OK;
}


// ------------------------------------------------------------------------------
// launch contexts and table dependencies (see LaunchCtx / TableDep)
// ------------------------------------------------------------------------------
static int acquire_ctx(rl_method *h, hipStream_t stream, LaunchCtx **out)
{
    LaunchCtx *pick = nullptr;
    for (LaunchCtx &c : h->ctx)
        if (c.bound && c.stream == stream) { pick = &c; break; }
    if (!pick)
        for (LaunchCtx &c : h->ctx)
            if (!c.bound) { pick = &c; break; }
    if (!pick) {
        for (LaunchCtx &c : h->ctx)
            if (!pick || c.last_use < pick->last_use) pick = &c;
        // hand-over: whatever the old stream still has in flight on this scratch must finish first
        HIPCHK(hipDeviceSynchronize());
        // (the theta-major CDDT scratch R[bin][pose] is the one large buffer of a context — up to 2 GiB —: a
        //  context that changes hands gives it back instead of pinning it for the handle's lifetime)
        pick->cddt_r.release();
    }
    pick->bound = true;
    pick->stream = stream;
    pick->last_use = ++h->use_clock;
    *out = pick;
    return RL_OK;
}

// after (re)building a table on `stream`
static int table_built(TableDep &d, hipStream_t stream)
{
    if (!d.ev) HIPCHK(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(d.ev, stream));
    d.built_on = stream;
    d.pending = true;
    return RL_OK;
}

// before a launch on `stream` reads the table
static int table_wait(TableDep &d, hipStream_t stream)
{
    if (!d.pending || stream == d.built_on) return RL_OK;      // (same stream: stream order)
    if (hipEventQuery(d.ev) == hipSuccess) {
        d.pending = false;
        return RL_OK;
    }
    HIPCHK(hipStreamWaitEvent(stream, d.ev, 0));
    return RL_OK;
}

// ------------------------------------------------------------------------------
// derived tables (built lazily on the launch stream, rebuilt when the map changed)
// ------------------------------------------------------------------------------
// (cos, sin) of the beam angles of fan f: one small table per (fov, num_rays) the handle is called
// with — four are kept, the least recently used one is rebuilt (after a device synchronisation:
// launches of other streams may still read it) when a fifth fan shows up
static int ensure_fan_table(rl_method *h, const FanParams &f, float fov, hipStream_t stream, const float2 **out)
{
    uint32_t bits;
    memcpy(&bits, &fov, sizeof bits);
    rl_method::FanTab *slot = nullptr;
    for (auto &ft : h->fan_tabs)
        if (ft.tab.p && ft.fov_bits == bits && ft.num_rays == f.num_rays) slot = &ft;
    if (slot) {
        slot->last_use = ++h->fan_clock;
        *out = (const float2 *)slot->tab.p;
        return table_wait(slot->dep, stream);
    }
    for (auto &ft : h->fan_tabs)
        if (!ft.tab.p) { slot = &ft; break; }
    if (!slot) {
        slot = &h->fan_tabs[0];
        for (auto &ft : h->fan_tabs)
            if (ft.last_use < slot->last_use) slot = &ft;
        HIPCHK(hipDeviceSynchronize());
    }
    int rc = slot->tab.ensure((size_t)f.num_rays * sizeof(float2));
    if (rc) return rc;
    hipLaunchKernelGGL(fan_table_kernel, dim3((f.num_rays + 255) / 256), dim3(256), 0, stream, f, (float2 *)slot->tab.p);
    slot->fov_bits = bits;
    slot->num_rays = f.num_rays;
    slot->last_use = ++h->fan_clock;
    *out = (const float2 *)slot->tab.p;
    return table_built(slot->dep, stream);
}

static int ensure_lut(rl_method *h, hipStream_t stream)
{
    rl_map *m = h->map;
    if (h->lut_epoch == m->epoch && h->lut.p) return table_wait(h->lut_dep, stream);
    HIPCHK(hipDeviceSynchronize());     // launches of other streams may still read the old table
    const size_t n = (size_t)m->rows * m->cols * h->theta_disc;
    int rc = h->lut.ensure(n * sizeof(uint16_t) + 64);      // + slack: rows are read in 16-B pieces
    if (rc) return rc;
    LutParams &lp = h->lp;
    lp.lut = (uint16_t *)h->lut.p;
    lp.theta_disc = h->theta_disc;
    lp.bins_per_rad = (float)h->theta_disc * 0.15915494309189535f;
    lp.bin_width = 6.283185307179586f / (float)h->theta_disc;
    lp.quant = 65535.0f / h->max_range;
    lp.dequant = h->max_range / 65535.0f;
    lp.debug = h->lut_debug;
    const long cells = (long)m->rows * m->cols;
    const int grid = (int)std::min(cells, (long)m->n_cu * 16);
    hipLaunchKernelGGL(lut_build_kernel, dim3(grid), dim3(256), 0, stream, m->mp, lp, h->max_range,
                       h->step_coeff, 0, m->rows);
    HIPCHK(hipGetLastError());
    h->lut_epoch = m->epoch;
    return table_built(h->lut_dep, stream);
}

// CDDT table of the current map, ENQUEUED on `stream` with no host synchronisation and no read-back:
// the two-player front-end rebuilds it before every scan (scripts/two_player/rcs_two_player.py:110-121).
// Sizes the host needs are known without asking the device: bucket counts follow from the map shape,
// and the number of stored values is bounded by 3 per (edge cell, theta bin) — a cell's footprint
// (half-width <= sqrt(2)/2) covers at most 3 buckets — with the edge count kept by the map.
static int ensure_cddt(rl_method *h, hipStream_t stream)
{
    rl_map *m = h->map;
    if (h->cddt_epoch == m->epoch && h->cd_tab.p) return table_wait(h->cddt_dep, stream);
    if (h->cd_tab.p) HIPCHK(hipDeviceSynchronize());     // launches of other streams may still read the old table
    const int td = h->theta_disc, nb = (td + 1) / 2;
    int rc;
    if (h->cd_geom_rows != m->rows || h->cd_geom_cols != m->cols) {
        // per-bin geometry: depends on the map SHAPE only, uploaded once
        h->cd_h_cos.resize(nb); h->cd_h_sin.resize(nb); h->cd_h_trans.resize(nb);
        h->cd_h_width.resize(nb); h->cd_h_boff.resize(nb + 1);
        uint32_t nbk = 0;
        const float W = (float)m->cols, H = (float)m->rows;
        for (int a = 0; a < nb; ++a) {
            float s, c;
            host_sincosf((float)a * (6.283185307179586f / (float)td), s, c);
            h->cd_h_cos[a] = c;
            h->cd_h_sin[a] = s;
            // buckets = height of the rotated map's bounding box; translation lifts the lowest
            // rotated corner to bucket 0
            h->cd_h_width[a] = (int)ceilf((fabsf(W * s) + fabsf(H * c)) - CDDT_EPS) + 1;
            const float lt = H * c, rt = fmaf(W, s, H * c), rb = W * s;
            h->cd_h_trans[a] = fmaxf(0.0f, -fminf(lt, fminf(rt, rb)) - CDDT_EPS);
            h->cd_h_boff[a] = nbk;
            nbk += (uint32_t)h->cd_h_width[a];
        }
        h->cd_h_boff[nb] = nbk;
        h->cd_buckets = nbk;
        if ((rc = h->cd_cos.ensure(nb * 4)) || (rc = h->cd_sin.ensure(nb * 4)) ||
            (rc = h->cd_trans.ensure(nb * 4)) || (rc = h->cd_width.ensure(nb * 4)) ||
            (rc = h->cd_boff.ensure((nb + 1) * 4)) || (rc = h->cd_offsets.ensure(((size_t)nbk + 1) * 4)) ||
            (rc = h->cd_cursor.ensure(((size_t)nbk + 1) * 4)))
            return rc;
        HIPCHK(hipMemcpyAsync(h->cd_cos.p, h->cd_h_cos.data(), nb * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(h->cd_sin.p, h->cd_h_sin.data(), nb * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(h->cd_trans.p, h->cd_h_trans.data(), nb * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(h->cd_width.p, h->cd_h_width.data(), nb * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(h->cd_boff.p, h->cd_h_boff.data(), (nb + 1) * 4, hipMemcpyHostToDevice, stream));
        h->cd_geom_rows = m->rows;
        h->cd_geom_cols = m->cols;
        h->cd_counts_clean = false;
    }
    const uint32_t nbk = h->cd_buckets;
    const size_t cap = std::max<size_t>((size_t)m->n_edges * nb * 3, 1);    // stored values, upper bound
    if (cap > (size_t)INT_MAX) return fail(RL_ERR_UNSUPPORTED, "CDDT table too large (%zu values)", cap);
    if ((rc = h->cd_xs2.ensure(cap * 4))) return rc;
    // the blocked table the queries read (cddt_kernels.h, CddtParams): leaves of 32 values + separator lines;
    // upper bound: every bucket pads its last leaf and, with more than one leaf, its last separator line
    const size_t tab_lines = cap / 32 + cap / 1024 + 2 * (size_t)nbk + 2;
    if (tab_lines > (size_t)UINT32_MAX) return fail(RL_ERR_UNSUPPORTED, "CDDT table too large (%zu lines)", tab_lines);
    if ((rc = h->cd_hdr.ensure((size_t)nbk * 8)) || (rc = h->cd_tab.ensure(tab_lines * 128))) return rc;
    CddtParams &cp = h->cdp;
    cp.theta_disc = td;
    cp.n_bins = nb;
    cp.cosv = (const float *)h->cd_cos.p;
    cp.sinv = (const float *)h->cd_sin.p;
    cp.trans = (const float *)h->cd_trans.p;
    cp.width = (const int *)h->cd_width.p;
    cp.bucket_off = (const uint32_t *)h->cd_boff.p;
    cp.offsets = (uint32_t *)h->cd_offsets.p;
    cp.xs = (float *)h->cd_xs2.p;
    cp.hdr = (uint2 *)h->cd_hdr.p;
    cp.tab = (float *)h->cd_tab.p;
    cp.bins_per_rad = (float)td * 0.15915494309189535f;
    cp.debug = h->lut_debug;
    if ((rc = h->cd_tmp.ensure(((size_t)nbk + 2 * nb + 64) * 4))) return rc;   // big-bucket count, bin totals (values, lines), list
    if (!h->cd_sort_attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cddt_sort_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CDDT_LDS_SORT * sizeof(float))));
        h->cd_sort_attr = true;
    }
    // bucket counters: zeroed once; every complete build returns them to zero (FILL subtracts what
    // COUNT added), so a rebuild starts without a memset
    if (!h->cd_counts_clean) {
        HIPCHK(hipMemsetAsync(h->cd_cursor.p, 0, ((size_t)nbk + 1) * 4, stream));
        h->cd_counts_clean = true;
    }
    // one workgroup per (chunk of edge cells, theta bin), bucket histogram of the bin in LDS
    int wmax = 0;
    for (int a = 0; a < nb; ++a) wmax = std::max(wmax, h->cd_h_width[a]);
    const size_t lds_fill = (size_t)wmax * 2 * sizeof(uint32_t);
    if (lds_fill > 150 * 1024) return fail(RL_ERR_UNSUPPORTED, "CDDT: map too large for the LDS bucket histogram");
    if (lds_fill > 48 * 1024) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cddt_project_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cddt_project_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill));
    }
    const dim3 pgrid((unsigned)std::max<uint32_t>(1u, (m->n_edges + CDDT_CHUNK - 1) / CDDT_CHUNK), (unsigned)nb);
    // count -> exclusive scan (CSR offsets) -> fill -> sort every bucket
    hipLaunchKernelGGL(cddt_project_kernel<false>, pgrid, dim3(256), lds_fill / 2, stream, cp,
                       (const uint32_t *)m->d_edges, (const uint32_t *)m->d_n_edges, (uint32_t *)h->cd_cursor.p);
    // [0] big-bucket count (zeroed by the sort's last reader), [1..64) spare, [64..64+nb) bin totals, then the list
    uint32_t *big_count = (uint32_t *)h->cd_tmp.p, *bin_total = (uint32_t *)h->cd_tmp.p + 64;
    uint32_t *bin_lines = bin_total + nb, *big_list = bin_lines + nb;
    HIPCHK(hipMemsetAsync(big_count, 0, 4, stream));
    hipLaunchKernelGGL(cddt_scan_bins_kernel, dim3(nb), dim3(256), 0, stream, cp, (const uint32_t *)h->cd_cursor.p,
                       bin_total, bin_lines);
    hipLaunchKernelGGL(cddt_scan_add_kernel, dim3(nb), dim3(256), 0, stream, cp, (const uint32_t *)h->cd_cursor.p,
                       (const uint32_t *)bin_total, (const uint32_t *)bin_lines, big_list, big_count);
    hipLaunchKernelGGL(cddt_project_kernel<true>, pgrid, dim3(256), lds_fill, stream, cp,
                       (const uint32_t *)m->d_edges, (const uint32_t *)m->d_n_edges, (uint32_t *)h->cd_cursor.p);
    // two launches of the sort kernel: the large buckets (workgroup each, 64 KB of LDS) and the small ones
    // (wave each, no LDS — in one launch the LDS size of the large path would cap everybody's occupancy)
    const uint32_t n_big_wg = (uint32_t)m->n_cu;
    hipLaunchKernelGGL(cddt_sort_kernel, dim3(n_big_wg), dim3(256), (size_t)h->cddt_lds_sort * sizeof(float), stream,
                       (const uint32_t *)h->cd_offsets.p, nbk, (const float *)h->cd_xs2.p, (const uint2 *)h->cd_hdr.p,
                       (float *)h->cd_tab.p, (const uint32_t *)big_list, (const uint32_t *)big_count, n_big_wg,
                       (uint32_t)h->cddt_lds_sort);
    const int sgrid = (int)std::max(1L, std::min(((long)nbk + 3) / 4, (long)m->n_cu * 32));
    hipLaunchKernelGGL(cddt_sort_kernel, dim3(sgrid), dim3(256), 0, stream,
                       (const uint32_t *)h->cd_offsets.p, nbk, (const float *)h->cd_xs2.p, (const uint2 *)h->cd_hdr.p,
                       (float *)h->cd_tab.p, (const uint32_t *)big_list, (const uint32_t *)big_count, 0u,
                       (uint32_t)h->cddt_lds_sort);
    HIPCHK(hipGetLastError());
    h->cddt_epoch = m->epoch;
    return table_built(h->cddt_dep, stream);
}

// diagnostics (bench.py's algorithmic bytes of a CDDT ray): stored values, buckets and non-empty buckets of the
// current table, from the CSR offsets the build leaves behind (builds the table if needed; synchronises)
static int cddt_table_stats(rl_method *h, const char *name, int64_t *value_out)
{
    if (h->kind != RL_CDDT) return fail(RL_ERR_INVALID, "not a CDDT method");
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    int rc = set_device(h->map);
    if (rc) return rc;
    if ((rc = ensure_cddt(h, h->stream))) return rc;
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint32_t> off((size_t)h->cd_buckets + 1);
    HIPCHK(hipMemcpy(off.data(), h->cd_offsets.p, off.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int64_t nonempty = 0;
    for (size_t b = 0; b < (size_t)h->cd_buckets; ++b) nonempty += off[b + 1] > off[b];
    if (!strcmp(name, "cddt_values")) *value_out = (int64_t)off[h->cd_buckets];
    else if (!strcmp(name, "cddt_buckets")) *value_out = (int64_t)h->cd_buckets;
    else *value_out = nonempty;
    return RL_OK;
}

// K2b's padded bit maps (normal + transposed), rebuilt when the map changed
static int ensure_blpad(rl_method *h, hipStream_t stream)
{
    rl_map *m = h->map;
    if (h->blpad_epoch == m->epoch && h->blpad.p) return table_wait(h->blpad_dep, stream);
    if (h->blpad.p) HIPCHK(hipDeviceSynchronize());     // launches of other streams may still read the old copy
    const int reach = (int)h->max_range + 3 + 2;        // cells a walk can get away from its origin (cap + margin)
    const int near = reach;                             // origins up to here outside the map are still covered
    const int pad = near + reach + 2;
    const int pad32 = (pad + 31) / 32;                  // major-axis padding in words
    const int stride_n = (m->cols + 31) / 32 + 2 * pad32, prow_n = m->rows + 2 * pad;
    const int stride_t = (m->rows + 31) / 32 + 2 * pad32, prow_t = m->cols + 2 * pad;
    const size_t words_n = (size_t)stride_n * prow_n, words_t = (size_t)stride_t * prow_t;
    if ((words_n + words_t) * 4 > (size_t)1 << 31) return fail(RL_ERR_UNSUPPORTED, "map too large for the padded bit maps");
    int rc = h->blpad.ensure((words_n + words_t) * 4);
    if (rc) return rc;
    uint32_t *out_n = (uint32_t *)h->blpad.p, *out_t = out_n + words_n;
    hipLaunchKernelGGL(bl_pad_bits_kernel, dim3((stride_n + 255) / 256, prow_n), dim3(256), 0, stream, m->d_occ,
                       m->rows, m->cols, 0, pad, pad32, stride_n, prow_n, out_n);
    hipLaunchKernelGGL(bl_pad_bits_kernel, dim3((stride_t + 255) / 256, prow_t), dim3(256), 0, stream, m->d_occ,
                       m->rows, m->cols, 1, pad, pad32, stride_t, prow_t, out_t);
    HIPCHK(hipGetLastError());
    BlPad &bp = h->blp;
    bp.bits = out_n;
    bp.stride_n = stride_n;
    bp.stride_t = stride_t;
    bp.k_n = (uint32_t)(((size_t)pad * stride_n + pad32) * 4);
    bp.k_t = (uint32_t)((words_n + (size_t)pad * stride_t + pad32) * 4);
    bp.near = (float)near;
    h->blpad_epoch = m->epoch;
    return table_built(h->blpad_dep, stream);
}

static BlParams make_bl(const rl_method *h, int num_rays, size_t &lds_bytes)
{
    BlParams bp{};
    // window radius: a walk takes at most (int)max_range + 3 unit steps, but its float coordinate can gain
    // one more cell on the way — x0 + 1 + 1 + ... rounds UP when it crosses a power of two with a
    // fraction just below 1 (127.99999 + 1 -> 129.0) — found by the 30-minute fuzz of round 2 as a stale
    // LDS read one row outside a window sized with no margin; two cells of margin now
    bp.R = (int)std::ceil(h->max_range) + 5;
    bp.ww = ((2 * bp.R + 32 + 31) / 32) | 1;
    const size_t win = (size_t)(2 * bp.R + 1) * bp.ww * sizeof(uint32_t);
    const size_t fan = (size_t)num_rays * sizeof(float2);
    bp.use_lds = (win + fan) <= 150 * 1024;
    lds_bytes = fan + (bp.use_lds ? win : 0);
    return bp;
}

// pose records in map-tile order (rec_sorted / order) by the binning pass the launch plan names
// (rl_binning, plan::binning_for); walk_outside = Bresenham semantics (origins outside the map still walk)
static int bin_poses(rl_method *h, LaunchCtx &cx, const float *d_poses, int n_poses, int walk_outside,
                     hipStream_t stream, int binning)
{
    const bool keys_only = binning == RL_BIN_SMALL_KEYS;
    const rl_map *m = h->map;
    int rc;
    // ray marching: the sample every ray of a pose takes at t = 0, read once per pose with the record
    // (pose_first_step); the Bresenham walk (walk_outside) has no use for it
    float *d0 = nullptr;
    if (!walk_outside && !keys_only) {
        if ((rc = cx.d0.ensure((size_t)n_poses * sizeof(float)))) return rc;
        d0 = (float *)cx.d0.p;
    }
    const float coeff = h->step_coeff;
    if ((rc = cx.rec.ensure((size_t)n_poses * sizeof(PoseRec)))) return rc;
    if ((rc = cx.order.ensure((size_t)n_poses * sizeof(uint32_t)))) return rc;
    if ((rc = cx.keys.ensure((size_t)n_poses * sizeof(uint32_t)))) return rc;
    if ((rc = cx.rec_sorted.ensure((size_t)n_poses * sizeof(PoseRec)))) return rc;
    const int do_sort = (binning == RL_BIN_GRID_UNSORTED || (binning == RL_BIN_GENERIC && !(h->sort_poses && n_poses >= 64))) ? 0 : 1;
    int shift = 6;
    while ((long)((m->cols >> shift) + 1) * ((m->rows >> shift) + 1) > 8192) ++shift;
    // (tile_key: tile rows in stripes walked column by column; -1: as many tile rows as an XCD band of evenly spread poses holds)
    auto striped = [&](int tx, int sh) {
        const int tiles_y = (m->rows >> sh) + 1;
        int rps = h->tile_stripe < 0 ? std::max(1, (m->rows >> sh) / std::max(1, h->xcd_bands)) : h->tile_stripe;
        if (rps >= tiles_y || !do_sort) rps = 0;
        return tx | (rps << 16);
    };
    const int tiles_x = striped((m->cols >> shift) + 1, shift);
    const int n_tiles = ((m->cols >> shift) + 1) * ((m->rows >> shift) + 1);
    if (binning == RL_BIN_GRID_SORT || binning == RL_BIN_GRID_UNSORTED) {
        const int ppw = h->bin_ppw;
        const int n_wg = (n_poses + ppw - 1) / ppw;
        if (do_sort) {
            // grid-wide binning on coarse tiles (<= 1024): per-workgroup LDS histograms ->
            // one scan over (tile, workgroup) -> scatter from LDS cursors
            int cshift = shift;
            while ((long)((m->cols >> cshift) + 1) * ((m->rows >> cshift) + 1) > 1024) ++cshift;
            const int ctx = striped((m->cols >> cshift) + 1, cshift);
            const int cnt = ((m->cols >> cshift) + 1) * ((m->rows >> cshift) + 1);
            const size_t n_ctr = (size_t)cnt * n_wg;
            if ((rc = cx.hist.ensure((n_ctr + cnt) * sizeof(uint32_t)))) return rc;     // counters, then tile totals
            hipLaunchKernelGGL(pose_prep_kernel, dim3(n_wg), dim3(256), (size_t)cnt * 4, stream,
                               m->mp, d_poses, n_poses, (PoseRec *)cx.rec.p,
                               (uint32_t *)cx.keys.p, (uint32_t *)cx.hist.p, n_wg, cshift, ctx,
                               cnt, (uint32_t *)nullptr, walk_outside, ppw, (float *)nullptr, coeff);
            uint32_t *tile_total = (uint32_t *)cx.hist.p + n_ctr;
            hipLaunchKernelGGL(tile_scan_a_kernel, dim3(cnt), dim3(256), 0, stream, (uint32_t *)cx.hist.p, n_wg,
                               tile_total);
            hipLaunchKernelGGL(pose_scatter_kernel, dim3(n_wg), dim3(256), (size_t)cnt * 4, stream,
                               n_poses, (const PoseRec *)cx.rec.p, (const uint32_t *)cx.keys.p,
                               (const uint32_t *)cx.hist.p, (const uint32_t *)tile_total, n_wg, cnt, (PoseRec *)cx.rec_sorted.p,
                               (uint32_t *)cx.order.p, ppw, m->mp, d0, coeff);
        } else {
            // caller's order kept: one fully parallel pass, records land in place
            hipLaunchKernelGGL(pose_prep_kernel, dim3(n_wg), dim3(256), 0, stream, m->mp, d_poses,
                               n_poses, (PoseRec *)cx.rec_sorted.p, (uint32_t *)nullptr,
                               (uint32_t *)nullptr, n_wg, shift, tiles_x, n_tiles,
                               (uint32_t *)cx.order.p, walk_outside, ppw, d0, coeff);
        }
    } else if (binning == RL_BIN_SMALL_KEYS || binning == RL_BIN_SMALL_RECORDS) {
        if (keys_only)
            hipLaunchKernelGGL(pose_bin_small_kernel<true>, dim3(1), dim3(1024),
                               (size_t)(n_tiles + 1024) * sizeof(uint32_t), stream, m->mp, d_poses,
                               n_poses, (PoseRec *)cx.rec_sorted.p, (uint32_t *)cx.order.p, shift,
                               tiles_x, n_tiles, walk_outside, (float *)nullptr, coeff);
        else
            hipLaunchKernelGGL(pose_bin_small_kernel<false>, dim3(1), dim3(1024),
                               (size_t)(n_tiles + 1024) * sizeof(uint32_t), stream, m->mp, d_poses,
                               n_poses, (PoseRec *)cx.rec_sorted.p, (uint32_t *)cx.order.p, shift,
                               tiles_x, n_tiles, walk_outside, d0, coeff);
    } else {
        hipLaunchKernelGGL(pose_bin_kernel, dim3(1), dim3(1024),
                           (size_t)(n_tiles + 1024) * sizeof(uint32_t), stream, m->mp, d_poses,
                           n_poses, (PoseRec *)cx.rec.p, (PoseRec *)cx.rec_sorted.p,
                           (uint32_t *)cx.order.p, (uint32_t *)cx.keys.p, shift, tiles_x,
                           n_tiles, do_sort, walk_outside, d0, coeff);
    }
    return RL_OK;
}

static FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f{};
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    f.d = d;
    return f;
}

// the options of a handle that shape a launch, as the planner takes them
static rl_plan_opts opts_of(const rl_method *h)
{
    rl_plan_opts o;
    plan::default_opts(o);
    o.variant = h->variant;
    o.grid_mult = h->grid_mult;
    o.wg_threads = h->wg_threads;
    o.low_water = h->low_water;
    o.sort_poses = h->sort_poses;
    o.xcd_bands = h->xcd_bands;
    o.slots = h->slots;
    o.tiled = h->tiled;
    o.inline_prep = h->inline_prep;
    o.inline_max = h->inline_max;
    o.inline_map_kb = h->inline_map_kb;
    o.stripe_max = h->stripe_max;
    o.order_inline = h->order_inline;
    o.bin_multi_min = h->bin_multi_min;
    o.bin_generic = h->bin_generic;
    o.run_log2 = h->run_log2;
    o.cddt_bins = h->cddt_bins_kernel;
    o.cddt_sort = h->cddt_sort;
    o.cddt_theta_min = h->cddt_theta_min;
    o.cddt_search = h->cddt_search;
    o.lut_debug = h->lut_debug;
    o.debug_stamps = h->debug_stamps;
    o.slice_log2 = h->slice_log2;
    o.code_map = h->code_map;
    o.code_min_rays = h->code_min_rays;
    o.tail_pct = h->tail_pct;
    o.tail_wg_pct = h->tail_wg_pct;
    o.code_entries = (h->code_map && h->code_built == h->code_map && h->pdt_epoch == h->map->epoch) ? h->code_n : 0;
    return o;
}

static int plan_for(const rl_method *h, int n_poses, int num_rays, bool aux, bool crash, rl_launch_plan *out)
{
    plan::In in;
    in.kind = h->kind;
    in.n_cu = h->map->n_cu;
    in.rows = h->map->rows;
    in.cols = h->map->cols;
    in.theta_disc = h->theta_disc;
    in.max_range = h->max_range;
    in.o = opts_of(h);
    in.n_poses = n_poses;
    in.num_rays = num_rays;
    in.aux = aux;
    in.crash = crash;
    return plan::plan_fan(in, out);
}

// step map of a ray-marching method (the EDT padded, holding the march's step), rebuilt when the map
// or the layout option changed
static int ensure_step_map(rl_method *h, hipStream_t stream)
{
    const rl_map *m = h->map;
    int rc;
    // (the planner marches on the row-major copy when the tiled geometry does not fit the address arithmetic:
    //  plan::tiled_fit — elongated maps whose pitch would need K > 24, tables beyond 4 GiB)
    const plan::TiledFit fit = plan::tiled_fit(m->rows, m->cols, h->max_range);
    const int want_tiled = (h->tiled && fit.ok) ? 1 : 0;
    if (h->pdt_epoch == m->epoch && h->pdt.p && h->pdt_tiled == want_tiled && h->code_built == h->code_map)
        return table_wait(h->pdt_dep, stream);
    if (h->pdt.p) HIPCHK(hipDeviceSynchronize());   // launches of other streams may still read the old copy
    h->pad = (int)std::ceil(h->max_range) + 2;
    if (want_tiled) {
        TiledGeom tg{};
        tg.pad = h->pad = fit.pad;
        tg.padr = fit.padr;
        tg.pcols = fit.pcols;
        tg.prows = fit.prows;
        tg.K = fit.K;
        const size_t bytes = fit.bytes;
        const uint32_t M = 4u + (1u << (tg.K - 2));
        h->pstride = (int)M;
        h->pdt_mask = 0xCu | (~0u << tg.K);
        h->pdt_k4 = (uint32_t)tg.padr * M;
        h->pdt_base_off = (size_t)h->pad << 4;
        if ((rc = h->pdt.ensure(bytes))) return rc;
        hipLaunchKernelGGL(pad_dt_tiled_kernel, dim3((tg.pcols + 255) / 256, tg.prows), dim3(256), 0, stream,
                           m->d_dt, m->rows, m->cols, (float *)h->pdt.p, tg, h->step_coeff);
    } else {
        h->pstride = (m->cols + 2 * h->pad + 31) & ~31;
        const int prow = m->rows + 2 * h->pad;
        // (row-major address: v_mad_i32_i24 r * stride + c, then << 2 in 32 bits)
        if (h->pstride >= (1 << 23) || (size_t)prow * h->pstride >= ((size_t)1 << 30))
            return fail(RL_ERR_UNSUPPORTED, "map %dx%d with max_range %g is too large for the step map", m->rows, m->cols,
                        h->max_range);
        if ((rc = h->pdt.ensure((size_t)prow * h->pstride * sizeof(float)))) return rc;
        hipLaunchKernelGGL(pad_dt_kernel, dim3((h->pstride + 255) / 256, prow), dim3(256), 0,
                           stream, m->d_dt, m->rows, m->cols, (float *)h->pdt.p, h->pad,
                           h->pstride, h->step_coeff);
        h->pdt_k4 = (uint32_t)(((size_t)h->pad * h->pstride + h->pad) * 4);
        h->pdt_mask = 0;
        h->pdt_base_off = 0;
    }
    // the CODE map next to it: palette of the map's distinct steps (mark -> scan), then the tiled map of their codes.
    // The palette size decides the launches' LDS, so the host reads it back here (a map build, not a scan).
    h->code_n = 0;
    const plan::CodeFit cf = plan::code_fit(m->rows, m->cols, h->max_range, 1);
    if (h->code_map == 2 && want_tiled && cf.ok) {
        const uint32_t cap = (uint32_t)plan::CODE_MAX_ENTRIES;
        if ((rc = h->cval.ensure((size_t)cf.nb * sizeof(float)))) return rc;
        if ((rc = h->cidx.ensure((size_t)cf.nb * sizeof(uint32_t)))) return rc;
        if ((rc = h->ctab.ensure((size_t)cap * sizeof(float)))) return rc;
        if ((rc = h->cnum.ensure(2 * sizeof(uint32_t)))) return rc;
        if (!h->pin_cnum) HIPCHK(hipHostMalloc((void **)&h->pin_cnum, 2 * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(h->cval.p, 0, (size_t)cf.nb * sizeof(float), stream));
        const size_t n_cells = (size_t)m->rows * m->cols;
        hipLaunchKernelGGL(code_mark_kernel, dim3((unsigned)std::min<size_t>((n_cells + 255) / 256, (size_t)m->n_cu * 8)), dim3(256),
                           0, stream, m->d_dt, n_cells, (float *)h->cval.p, cf.nb, h->step_coeff, h->max_range);
        hipLaunchKernelGGL(code_scan_kernel, dim3(1), dim3(1024), 0, stream, (const float *)h->cval.p, cf.nb, h->step_coeff,
                           (uint32_t *)h->cidx.p, (float *)h->ctab.p, cap, (uint32_t *)h->cnum.p);
        HIPCHK(hipMemcpyAsync(h->pin_cnum, h->cnum.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (h->pin_cnum[1] == 0u && h->pin_cnum[0] <= cap) {
            TiledGeom tg{};
            tg.pad = cf.pad;
            tg.padr = cf.padr;
            tg.pcols = cf.pcols;
            tg.prows = cf.prows;
            tg.K = cf.K;
            if ((rc = h->cmap.ensure(cf.bytes))) return rc;
            hipLaunchKernelGGL((pad_code_tiled_kernel<1>), dim3((tg.pcols + 255) / 256, tg.prows), dim3(256), 0, stream,
                               m->d_dt, m->rows, m->cols, h->cmap.p, tg, h->step_coeff, h->max_range,
                               (const uint32_t *)h->cidx.p, cf.nb, (const uint32_t *)h->cnum.p);
            const uint32_t M = (1u << cf.es) + (1u << (cf.K - 3));
            h->cstride = (int)M;
            h->cmask = (7u << cf.es) | (~0u << cf.K);
            h->ck4 = (uint32_t)cf.padr * M;
            h->cbase_off = (size_t)cf.pad << (3 + cf.es);
            h->code_n = (int)h->pin_cnum[0];
        }
    }
    h->code_built = h->code_map;
    h->pdt_epoch = m->epoch;
    h->pdt_tiled = want_tiled;
    return table_built(h->pdt_dep, stream);
}

// the stream-kernel instantiation a plan names
template <bool A, bool C, int N, bool I, bool T, int S, bool L = false, int CD = 0>
static void launch_rm_stream(const rl_launch_plan &pl, hipStream_t stream, const PadMap &pm, const FanParams &f,
                             const StreamParams &sp, float *d_out, int32_t *d_hits, uint16_t *d_steps,
                             const CrashParams &cp)
{
    // (more dynamic LDS than HIP's default cap — fans of several thousand beams, with the crash table —: opt in,
    //  as the BL / occ / CDDT kernels do; the attribute is sticky per function and device, the call is cheap)
    if (pl.lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rm_fan_stream_kernel<A, C, N, I, T, S, L, CD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, pl.lds_bytes);
    hipLaunchKernelGGL((rm_fan_stream_kernel<A, C, N, I, T, S, L, CD>), dim3(pl.grid), dim3(N), (size_t)pl.lds_bytes, stream,
                       pm, f, sp, d_out, d_hits, d_steps, cp);
}

static int dispatch_rm_stream(const rl_launch_plan &pl, hipStream_t stream, const PadMap &pm, const FanParams &f,
                              const StreamParams &sp, float *d_out, int32_t *d_hits, uint16_t *d_steps,
                              const CrashParams &cp)
{
    const bool inl = pl.record_source != 0, tiled = pl.tiled != 0, aux = pl.aux != 0, crash = pl.crash != 0;
    const int nt = pl.block;
#define RM_ARGS pl, stream, pm, f, sp, d_out, d_hits, d_steps, cp
    if (pl.kernel == RL_K_RM_STREAM_LIT) {     // upstream-literal arithmetic (variant 3): INLINE records, tiled step map
        if (pl.slots >= 2 && pl.code == 2) { if (crash) launch_rm_stream<false, true, 1024, true, true, 2, true, 2>(RM_ARGS);
                                             else launch_rm_stream<false, false, 1024, true, true, 2, true, 2>(RM_ARGS); }
        else if (pl.slots >= 2) { if (crash) launch_rm_stream<false, true, 1024, true, true, 2, true>(RM_ARGS);
                             else launch_rm_stream<false, false, 1024, true, true, 2, true>(RM_ARGS); }
        else               { if (crash) launch_rm_stream<false, true, 1024, true, true, 1, true>(RM_ARGS);
                             else launch_rm_stream<false, false, 1024, true, true, 1, true>(RM_ARGS); }
    } else if (pl.code == 2 && pl.slots == 2 && tiled && !aux && nt == 1024) {     // two rays per lane on the u16 code map
        if (inl) { if (crash) launch_rm_stream<false, true, 1024, true, true, 2, false, 2>(RM_ARGS);
                   else launch_rm_stream<false, false, 1024, true, true, 2, false, 2>(RM_ARGS); }
        else     { if (crash) launch_rm_stream<false, true, 1024, false, true, 2, false, 2>(RM_ARGS);
                   else launch_rm_stream<false, false, 1024, false, true, 2, false, 2>(RM_ARGS); }
    } else if (pl.slots == 3) {                // three rays per lane: plain ranges, 1024 lanes
        if (!inl) launch_rm_stream<false, false, 1024, false, true, 3>(RM_ARGS);
        else if (tiled) launch_rm_stream<false, false, 1024, true, true, 3>(RM_ARGS);
        else launch_rm_stream<false, false, 1024, true, false, 3>(RM_ARGS);
    } else if (pl.slots == 2) {                // two rays per lane: ranges / fused crash test, tiled step map
#define RM_S2(C)                                                                     \
    do {                                                                             \
        if (inl) launch_rm_stream<false, C, 1024, true, true, 2>(RM_ARGS);           \
        else if (nt == 1024) launch_rm_stream<false, C, 1024, false, true, 2>(RM_ARGS); \
        else if (nt == 512) launch_rm_stream<false, C, 512, false, true, 2>(RM_ARGS);   \
        else launch_rm_stream<false, C, 256, false, true, 2>(RM_ARGS);               \
    } while (0)
        if (inl && nt == 512 && !crash) launch_rm_stream<false, false, 512, true, true, 2>(RM_ARGS);   // (A/B: wg_threads 512)
        else if (crash) RM_S2(true); else RM_S2(false);
#undef RM_S2
    } else {
#define RM_S1(A, C, T)                                                               \
    do {                                                                             \
        if (inl) launch_rm_stream<A, C, 1024, true, T, 1>(RM_ARGS);                  \
        else if (nt == 1024) launch_rm_stream<A, C, 1024, false, T, 1>(RM_ARGS);     \
        else if (nt == 512) launch_rm_stream<A, C, 512, false, T, 1>(RM_ARGS);       \
        else launch_rm_stream<A, C, 256, false, T, 1>(RM_ARGS);                      \
    } while (0)
#define RM_S1_T(A, C) do { if (tiled) RM_S1(A, C, true); else RM_S1(A, C, false); } while (0)
        if (crash) { if (aux) RM_S1_T(true, true); else RM_S1_T(false, true); }
        else       { if (aux) RM_S1_T(true, false); else RM_S1_T(false, false); }
#undef RM_S1_T
#undef RM_S1
    }
#undef RM_ARGS
    return RL_OK;
}

// enqueue the fan kernels on `stream`; all pointers are device pointers.  What is launched is decided by
// plan::plan_fan (launch_plan.h); this function only executes the plan.
// audit mode (variant 3): the per-map constants of range_libc's RangeMethod, double arithmetic with the host's libm
// (as the CPU checker's upstream-literal statement computes them)
static LiteralParams make_literal(const rl_map *m)
{
    LiteralParams lt;
    const double wa = (double)m->mp.wa;
    lt.rotation_const = (float)(-1.0 * wa - 3.0 * M_PI / 2.0);
    lt.wsin = (float)sin(wa);
    lt.wcos = (float)cos(wa);
    return lt;
}

static int launch_fan(rl_method *h, const float *d_poses, int n_poses, float fov, int num_rays,
                      float *d_out, int32_t *d_hits, uint16_t *d_steps, const CrashParams *crash,
                      hipStream_t stream);

// what a method family's launch function needs: the call's arguments, the plan, the launch context
struct FanLaunch {
    rl_method *h;
    const rl_map *m;
    const rl_launch_plan &pl;
    LaunchCtx *cx;
    FanParams f;
    const float *d_poses;
    int n_poses;
    float fov;
    int num_rays;
    float *d_out;
    int32_t *d_hits;
    uint16_t *d_steps;
    const CrashParams *crash;
    hipStream_t stream;
    bool aux;
};
#define FAN_LAUNCH_LOCALS                                                                                              \
    rl_method *h = L.h; const rl_map *m = L.m; const rl_launch_plan &pl = L.pl; LaunchCtx *cx = L.cx;                   \
    const FanParams &f = L.f; const float *d_poses = L.d_poses; const int n_poses = L.n_poses; const float fov = L.fov; \
    const int num_rays = L.num_rays; float *d_out = L.d_out; int32_t *d_hits = L.d_hits; uint16_t *d_steps = L.d_steps; \
    const CrashParams *crash = L.crash; hipStream_t stream = L.stream; const bool aux = L.aux;                          \
    const dim3 grid(pl.grid), block(pl.block); const size_t lds = (size_t)pl.lds_bytes; int rc = RL_OK;                \
    (void)h; (void)m; (void)cx; (void)f; (void)d_poses; (void)n_poses; (void)fov; (void)num_rays; (void)d_out;           \
    (void)d_hits; (void)d_steps; (void)crash; (void)stream; (void)aux; (void)grid; (void)block; (void)lds; (void)rc

// RL_K_LUT_LDS, RL_K_LUT_FAN
static int launch_lut(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    if ((rc = ensure_lut(h, stream))) return rc;
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
#define LAUNCH_LL(N, C) hipLaunchKernelGGL((lut_fan_lds_kernel<N, C>), grid, block, lds, stream, m->mp, f, h->lp, d_poses, d_out)
    if (pl.kernel == RL_K_LUT_LDS) {
        if (pl.ch == 12) { if (pl.nl == 1) LAUNCH_LL(1, 12); else if (pl.nl == 2) LAUNCH_LL(2, 12); else LAUNCH_LL(3, 12); }
        else             { if (pl.nl == 1) LAUNCH_LL(1, 17); else if (pl.nl == 2) LAUNCH_LL(2, 17); else LAUNCH_LL(3, 17); }
    } else if (pl.ch == 12)
        hipLaunchKernelGGL((lut_fan_kernel<12>), grid, block, 0, stream, m->mp, f, h->lp, d_poses, d_out);
    else
        hipLaunchKernelGGL((lut_fan_kernel<17>), grid, block, 0, stream, m->mp, f, h->lp, d_poses, d_out);
#undef LAUNCH_LL
    return RL_OK;
}

// RL_K_CDDT_BINS
static int launch_cddt_bins(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    if ((rc = ensure_cddt(h, stream))) return rc;
    // tile-ordered poses in XCD bands: neighbouring origins hit neighbouring buckets (L2 reuse)
    const uint32_t *d_order = nullptr;
    if (pl.binning != RL_BIN_NONE) {
        if ((rc = bin_poses(h, *cx, d_poses, n_poses, 0, stream, pl.binning))) return rc;
        d_order = (const uint32_t *)cx->order.p;
    }
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
    hipLaunchKernelGGL(cddt_fan_bins_kernel, grid, block, lds, stream, m->mp, f, h->cdp, d_poses, d_out, d_order,
                       pl.bands, pl.nl, pl.ch);
    return RL_OK;
}

// RL_K_CDDT_THETA
static int launch_cddt_theta(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    if ((rc = ensure_cddt(h, stream))) return rc;
    // scratch of the launch context: R[raw bin][pose] behind the per-pose records {gx, gy, first bin, bins}
    const bool fused = !strcmp(pl.name, "scan::cddt_theta_fused_kernel");     // (the planner's decision: cddt_search 2 and the tile fits LDS)
    const size_t prep_bytes = (((size_t)n_poses * 16) + 255) & ~(size_t)255;
    if ((rc = cx->cddt_r.ensure(prep_bytes + (fused ? 0 : (size_t)h->cdp.theta_disc * (size_t)n_poses * sizeof(float))))) return rc;
    float4 *d_prep = (float4 *)cx->cddt_r.p;
    float *d_r = (float *)((char *)cx->cddt_r.p + prep_bytes);
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
    hipLaunchKernelGGL(cddt_theta_prep_kernel, dim3((unsigned)std::max(1, std::min((n_poses + 255) / 256, m->n_cu * 8))),
                       dim3(256), 0, stream, m->mp, f, h->cdp, d_poses, d_prep);
    if (fused) {
        hipLaunchKernelGGL(cddt_theta_fused_kernel, grid, block, lds, stream, m->mp, f, h->cdp, d_poses,
                           (const float4 *)d_prep, d_out, pl.nl);
        return RL_OK;
    }
    if (h->cddt_search)
        hipLaunchKernelGGL(cddt_theta_search2_kernel, grid, block, 0, stream, m->mp, f, h->cdp, d_poses,
                           (const float4 *)d_prep, d_r, pl.bands);
    else
        hipLaunchKernelGGL(cddt_theta_search_kernel, grid, block, 0, stream, m->mp, f, h->cdp, d_poses,
                           (const float4 *)d_prep, d_r, pl.bands);
    const int n_grp = (n_poses + (1 << pl.ch) - 1) >> pl.ch;
    hipLaunchKernelGGL(cddt_theta_fan_kernel, dim3((unsigned)std::max(1, std::min(n_grp, m->n_cu * 8))), block, lds,
                       stream, m->mp, f, h->cdp, d_poses, (const float *)d_r, d_out, pl.ch, pl.nl);
    return RL_OK;
}

// RL_K_CDDT_RAYS
static int launch_cddt_rays(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    if ((rc = ensure_cddt(h, stream))) return rc;
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
    hipLaunchKernelGGL(cddt_fan_kernel, grid, block, 0, stream, m->mp, f, h->cdp, d_poses, d_out);
    return RL_OK;
}

// RL_K_BL_STREAM
static int launch_bl_stream(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    // K2b: stream schedule on the cache-resident bit map
    if ((rc = ensure_blpad(h, stream))) return rc;
    if ((rc = bin_poses(h, *cx, d_poses, n_poses, 1, stream, pl.binning))) return rc;
    StreamParams sp{};
    sp.rec = (const PoseRec *)cx->rec_sorted.p;
    sp.order = (const uint32_t *)cx->order.p;
    sp.div_B = make_fastdiv((uint32_t)num_rays);
    sp.low_water = h->low_water >= 0 ? h->low_water : 12;
    sp.n_bands = pl.bands;
    sp.plain_store = !h->nt_store;
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
    if (aux)
        hipLaunchKernelGGL((bl_fan_stream_kernel<true, 1024>), grid, block, lds, stream, m->mp, f, sp, h->blp, d_out,
                           d_hits, d_steps);
    else
        hipLaunchKernelGGL((bl_fan_stream_kernel<false, 1024>), grid, block, lds, stream, m->mp, f, sp, h->blp, d_out,
                           d_hits, d_steps);
    return RL_OK;
}

// RL_K_BL_LDS, RL_K_OCC_LDS
static int launch_bl_lds(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    size_t lds_bl = 0;
    BlParams bp = make_bl(h, num_rays, lds_bl);
    if (lds_bl > 48 * 1024) {          // more dynamic LDS than the default cap: opt in
        const void *fa = pl.kernel == RL_K_BL_LDS ? reinterpret_cast<const void *>(&bl_fan_kernel<true>)
                                                  : reinterpret_cast<const void *>(&occ_fan_lds_kernel<true>);
        const void *fb = pl.kernel == RL_K_BL_LDS ? reinterpret_cast<const void *>(&bl_fan_kernel<false>)
                                                  : reinterpret_cast<const void *>(&occ_fan_lds_kernel<false>);
        HIPCHK(hipFuncSetAttribute(fa, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bl));
        HIPCHK(hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bl));
    }
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
    if (pl.kernel == RL_K_BL_LDS) {
        if (aux) hipLaunchKernelGGL((bl_fan_kernel<true>), grid, block, lds_bl, stream, m->mp, f, bp, d_poses, d_out, d_hits, d_steps);
        else     hipLaunchKernelGGL((bl_fan_kernel<false>), grid, block, lds_bl, stream, m->mp, f, bp, d_poses, d_out, d_hits, d_steps);
    } else {
        // occ_fan_lds: unit-step march on an LDS-resident occupancy window (A/B partner, approximate)
        if (aux) hipLaunchKernelGGL((occ_fan_lds_kernel<true>), grid, block, lds_bl, stream, m->mp, f, bp, d_poses, d_out, d_hits, d_steps);
        else     hipLaunchKernelGGL((occ_fan_lds_kernel<false>), grid, block, lds_bl, stream, m->mp, f, bp, d_poses, d_out, d_hits, d_steps);
    }
    return RL_OK;
}

// RL_K_RM_LITERAL
static int launch_rm_literal(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    const LiteralParams lt = make_literal(m);
    const long n_rays = (long)n_poses * num_rays;
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
    if (aux) hipLaunchKernelGGL((rm_literal_kernel<true, false>), grid, block, 0, stream, m->mp, f, lt, d_poses, n_rays, d_out, d_hits, d_steps);
    else     hipLaunchKernelGGL((rm_literal_kernel<false, false>), grid, block, 0, stream, m->mp, f, lt, d_poses, n_rays, d_out, d_hits, d_steps);
    return RL_OK;
}

// RL_K_RM_CHUNK
static int launch_rm_chunk(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    CrashParams cp{nullptr, 0.0, nullptr, 1};
    if (crash) cp = *crash;
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
#define LAUNCH_CHUNK(A, C) hipLaunchKernelGGL((rm_fan_kernel<A, C>), grid, block, lds, stream, m->mp, f, d_poses, d_out, d_hits, d_steps, cp)
    if (crash) { if (aux) LAUNCH_CHUNK(true, true); else LAUNCH_CHUNK(false, true); }
    else       { if (aux) LAUNCH_CHUNK(true, false); else LAUNCH_CHUNK(false, false); }
#undef LAUNCH_CHUNK
    return RL_OK;
}

// RL_K_RM_STREAM_LIT, RL_K_RM_STREAM
static int launch_rm_stream_family(const FanLaunch &L)
{
    FAN_LAUNCH_LOCALS;
    // (1) per-pose records + tile-ordered permutation, (2) banded lane-refill march
    CrashParams cp{nullptr, 0.0, nullptr, 1};
    if (crash) cp = *crash;
    if ((rc = cx->rec.ensure((size_t)n_poses * sizeof(PoseRec)))) return rc;
    if ((rc = cx->order.ensure((size_t)n_poses * sizeof(uint32_t)))) return rc;
    if ((rc = cx->keys.ensure((size_t)n_poses * sizeof(uint32_t)))) return rc;
    if ((rc = ensure_step_map(h, stream))) return rc;
    if (pl.binning != RL_BIN_NONE &&
        (rc = bin_poses(h, *cx, d_poses, n_poses, 0, stream, pl.binning)))
        return rc;
    PadMap pm{};
    pm.pdt = (const float *)((const char *)h->pdt.p + h->pdt_base_off);
    pm.stride = h->pstride;
    pm.nstride = (int)h->pdt_mask;
    pm.pad = h->pad;
    pm.k4 = h->pdt_k4;
    pm.div_stride = make_fastdiv((uint32_t)h->pstride);
    pm.res = m->res;
    if (pl.code) {                               // the march reads the map of palette codes: its base and address constants
        if (h->code_n <= 0 || pl.code_entries != h->code_n)
            return fail(RL_ERR_INVALID, "internal: code-map plan (%d entries) without a matching palette (%d)", pl.code_entries, h->code_n);
        pm.pdt = (const float *)((const char *)h->cmap.p + h->cbase_off);
        pm.stride = h->cstride;
        pm.nstride = (int)h->cmask;
        pm.k4 = h->ck4;
    }
    StreamParams sp{};
    sp.code_tab = (const float *)h->ctab.p;
    sp.code_n = pl.code ? h->code_n : 0;
    sp.tail_g1 = pl.gen1 > 0 ? pl.gen1 / std::max(pl.bands, 1) : 0;
    sp.tail_pct = std::min(h->tail_pct, h->tail_wg_pct);
    sp.rec = (const PoseRec *)cx->rec_sorted.p;
    sp.order = (const uint32_t *)cx->order.p;
    sp.d0 = (const float *)cx->d0.p;
    if ((rc = ensure_fan_table(h, f, fov, stream, &sp.fan_tab))) return rc;
    sp.div_B = make_fastdiv((uint32_t)num_rays);
    sp.low_water = h->low_water >= 0 ? h->low_water : ((pl.record_source != 0 && pl.slots >= 2) ? 20 : 12);
    sp.n_bands = pl.bands;
    sp.raw_poses = d_poses;
    sp.map = m->d_mp;
    sp.k_max = pl.k_max;
    sp.cpp = (uint32_t)((num_rays + 63) / 64);
    sp.div_cpp = make_fastdiv(sp.cpp);
    sp.drain_prio = h->drain_prio;
    sp.spec_drain = h->spec_drain;
    sp.spec_stretch = h->spec_stretch;
    sp.drain_cap = h->drain_cap;
    sp.drain_stretch = h->drain_stretch;
    sp.group_drain = h->group_drain;
    if (pl.kernel == RL_K_RM_STREAM_LIT) sp.lit = make_literal(m);
    sp.plain_store = !h->nt_store;
    sp.dbg = nullptr;
    const int waves_per_wg = pl.block / 64;
    if (h->debug_stamps) {
        if ((rc = cx->dbg.ensure((size_t)pl.grid * waves_per_wg * 4 * sizeof(uint64_t)))) return rc;
        sp.dbg = (unsigned long long *)cx->dbg.p;
        h->last_dbg = cx->dbg.p;
    }
    sp.stripe = pl.record_source == 2 ? 1 : pl.record_source == 3 ? 2 : 0;
    sp.run_log2 = pl.run_log2;
    h->last_grid = pl.grid * waves_per_wg / WAVES_PER_WG;
    // hand-off march (several rays per lane on the tiled step map): dry waves leave their last rays in the launch
    // context's leftover list — one region of handoff_cap records per wave of the main grid —, the second launch
    // finishes them
    const bool handoff = h->handoff && pl.slots >= 2 && pl.tiled && h->spec_drain > 0 && !h->debug_stamps &&
                         pl.kernel == RL_K_RM_STREAM && !pl.code;   // (the leftover kernel marches the canonical arithmetic on the float32 map)
    int cap_log2 = 4;
    const int n_src = pl.grid * waves_per_wg;
    if (handoff) {
        cap_log2 = h->handoff_cap >= 64 ? 6 : (h->handoff_cap >= 32 ? 5 : (h->handoff_cap >= 16 ? 4 : 3));
        if ((rc = cx->left_rec.ensure(((size_t)n_src << cap_log2) * sizeof(LeftoverRec)))) return rc;
        if ((rc = cx->left_cnt.ensure((size_t)n_src * sizeof(uint32_t)))) return rc;
        sp.left_rec = (LeftoverRec *)cx->left_rec.p;
        sp.left_cnt = (uint32_t *)cx->left_cnt.p;
        sp.left_cap_log2 = cap_log2;
        sp.drain_cap = std::min(sp.drain_cap, 1 << cap_log2);
    }
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));   // march kernel(s) alone
    if ((rc = dispatch_rm_stream(pl, stream, pm, f, sp, d_out, d_hits, d_steps, cp))) return rc;
    if (handoff) {
        const int lw = h->handoff_wg / 64;                              // leftover waves per workgroup
        const int n_lw = (n_src + (64 >> cap_log2) - 1) / (64 >> cap_log2);
        const dim3 lgrid((unsigned)((n_lw + lw - 1) / lw)), lblock((unsigned)h->handoff_wg);
        if (crash)
            hipLaunchKernelGGL((rm_leftover_kernel<true>), lgrid, lblock, 0, stream, pm, f, (const LeftoverRec *)sp.left_rec,
                               (const uint32_t *)sp.left_cnt, n_src, cap_log2, h->drain_stretch, sp.plain_store, d_out, cp);
        else
            hipLaunchKernelGGL((rm_leftover_kernel<false>), lgrid, lblock, 0, stream, pm, f, (const LeftoverRec *)sp.left_rec,
                               (const uint32_t *)sp.left_cnt, n_src, cap_log2, h->drain_stretch, sp.plain_store, d_out, cp);
    }
    return RL_OK;
}

// a batch cut into pose slices (plan: slices > 1), each its own launch sequence on the stream
static int launch_fan_sliced(rl_method *h, const rl_launch_plan &pl, const float *d_poses, int n_poses, float fov,
                             int num_rays, float *d_out, int32_t *d_hits, uint16_t *d_steps, const CrashParams *crash,
                             hipStream_t stream)
{
    int rc = RL_OK;
    // pose slices below 2^slice_log2 rays, each its own launch sequence
    if (crash && crash->group != 0)
        return fail(RL_ERR_UNSUPPORTED, "a fused crash test over %d poses in the upstream-literal mode needs the per-pose "
                                        "mark form (rl_check_collision_groups*), not one roll-out of that length", n_poses);
    const int per = pl.slice_poses;
    const uint64_t base_off = h->ray_offset;
    // one event pair around the whole sliced sequence (the per-slice pairs would leave the
    // last slice only)
    const int timing = h->timing;
    h->timing = 0;
    if (timing) HIPCHK(hipEventRecord(h->ev0, stream));
    rc = RL_OK;
    for (int p0 = 0; p0 < n_poses && rc == RL_OK; p0 += per) {
        const int np = std::min(per, n_poses - p0);
        const size_t r0 = (size_t)p0 * num_rays;
        h->ray_offset = base_off + r0;               // noise stays keyed by the global ray id
        // (a fused crash test reaches a sliced launch only in per-pose-mark form — the upstream-literal mode's
        //  slices: slice k marks poses p0 .. p0 + np - 1 through a shifted mark array)
        CrashParams cps{nullptr, 0.0, nullptr, 1, 0};
        if (crash) {
            cps = *crash;
            cps.first_crashed = crash->first_crashed + p0;
        }
        rc = launch_fan(h, d_poses + (size_t)p0 * 3, np, fov, num_rays, d_out ? d_out + r0 : nullptr,
                        d_hits ? d_hits + 2 * r0 : nullptr, d_steps ? d_steps + r0 : nullptr,
                        crash ? &cps : nullptr, stream);
    }
    h->ray_offset = base_off;
    h->timing = timing;
    if (timing && rc == RL_OK) { HIPCHK(hipEventRecord(h->ev1, stream)); h->timed = true; }
    return rc;
}

static int launch_fan(rl_method *h, const float *d_poses, int n_poses, float fov, int num_rays,
                      float *d_out, int32_t *d_hits, uint16_t *d_steps, const CrashParams *crash,
                      hipStream_t stream)
{
    if (n_poses == 0) return RL_OK;
    const rl_map *m = h->map;
    const bool aux = d_hits || d_steps;
    if (h->kind != RL_RM && h->kind != RL_RM_GPU) {
        if (crash) return fail(RL_ERR_UNSUPPORTED, "fused crash test needs a ray-marching method");
        if (aux && h->kind != RL_BRESENHAM)
            return fail(RL_ERR_UNSUPPORTED, "hit cells / step counts exist only for RM and Bresenham");
    }
    rl_launch_plan pl;
    int rc = RL_OK;
    // (a code-map handle plans with its palette size: the step map and the palette are built before the plan)
    if (h->code_map && (h->kind == RL_RM || h->kind == RL_RM_GPU) && h->variant >= 1 && (rc = ensure_step_map(h, stream))) return rc;
    rc = plan_for(h, n_poses, num_rays, aux, crash != nullptr, &pl);
    if (rc == RL_ERR_UNSUPPORTED)
        return fail(rc, (h->variant >= 2 && crash) ? "the fused crash test needs variant 0 or 1 (not the occupancy-window or the audit kernel)"
                        : h->variant == 2 ? "occupancy window of max_range %g does not fit LDS (num_rays %d)"
                                          : "the beam tables of max_range %g, num_rays %d exceed a workgroup's LDS (160 KB)",
                    h->max_range, num_rays);
    if (rc) return fail(rc, "launch planning failed");
    if (pl.slices > 1) return launch_fan_sliced(h, pl, d_poses, n_poses, fov, num_rays, d_out, d_hits, d_steps, crash, stream);
    LaunchCtx *cx = nullptr;
    if ((rc = acquire_ctx(h, stream, &cx))) return rc;
    h->last_plan = pl;
    FanParams f = make_fan(h, n_poses, fov, num_rays);
    if (h->timing == 1) HIPCHK(hipEventRecord(h->ev0, stream));
    const FanLaunch L{h, m, pl, cx, f, d_poses, n_poses, fov, num_rays, d_out, d_hits, d_steps, crash, stream, aux};
    switch (pl.kernel) {
    case RL_K_LUT_LDS:
    case RL_K_LUT_FAN:
        rc = launch_lut(L);
        break;
    case RL_K_CDDT_BINS:
        rc = launch_cddt_bins(L);
        break;
    case RL_K_CDDT_THETA:
        rc = launch_cddt_theta(L);
        break;
    case RL_K_CDDT_RAYS:
        rc = launch_cddt_rays(L);
        break;
    case RL_K_BL_STREAM:
        rc = launch_bl_stream(L);
        break;
    case RL_K_BL_LDS:
    case RL_K_OCC_LDS:
        rc = launch_bl_lds(L);
        break;
    case RL_K_RM_LITERAL:
        rc = launch_rm_literal(L);
        break;
    case RL_K_RM_CHUNK:
        rc = launch_rm_chunk(L);
        break;
    case RL_K_RM_STREAM_LIT:
    case RL_K_RM_STREAM:
        rc = launch_rm_stream_family(L);
        break;
    default:
        return fail(RL_ERR_INVALID, "launch plan names no kernel");
    }
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    if (h->timing) { HIPCHK(hipEventRecord(h->ev1, stream)); h->timed = true; }
    return RL_OK;
}

static int launch_rays(rl_method *h, const float *d_ins, long n, float *d_out, int32_t *d_hits,
                       uint16_t *d_steps, hipStream_t stream)
{
    if (n == 0) return RL_OK;
    const rl_map *m = h->map;
    FanParams f = make_fan(h, 0, 0.0f, 1);
    long want = (n + WG - 1) / WG;
    long cap = (long)m->n_cu * h->grid_mult;
    int grid = (int)std::max(1L, std::min(want, cap));
    if (h->timing) HIPCHK(hipEventRecord(h->ev0, stream));
    int rc;
    if (h->kind == RL_GIANT_LUT) {
        if ((rc = ensure_lut(h, stream))) return rc;
        hipLaunchKernelGGL(lut_rays_kernel, dim3(grid), dim3(256), 0, stream, m->mp, f, h->lp, d_ins,
                           n, d_out);
    } else if (h->kind == RL_CDDT) {
        if ((rc = ensure_cddt(h, stream))) return rc;
        hipLaunchKernelGGL(cddt_rays_kernel, dim3(grid), dim3(256), 0, stream, m->mp, f, h->cdp,
                           d_ins, n, d_out);
    } else if (h->kind == RL_BRESENHAM) {
        hipLaunchKernelGGL(bl_rays_kernel, dim3(grid), dim3(256), 0, stream, m->mp, f, d_ins, n,
                           d_out);
    } else if (h->variant == 3) {
        // audit mode: the upstream 2-argument form stated literally, one lane per row
        const LiteralParams lt = make_literal(m);
        if (d_hits || d_steps)
            hipLaunchKernelGGL((rm_literal_kernel<true, true>), dim3(grid), dim3(256), 0, stream, m->mp, f, lt, d_ins, n, d_out, d_hits, d_steps);
        else
            hipLaunchKernelGGL((rm_literal_kernel<false, true>), dim3(grid), dim3(256), 0, stream, m->mp, f, lt, d_ins, n, d_out, d_hits, d_steps);
    } else if (h->variant >= 1 && n <= INT_MAX) {
        // a ray is a pose with one beam at alpha = 0: fan(num_rays = 1, fov = 0) gives exactly
        // (cos, sin) of the heading as direction, and the stream kernel packs 64 rays per block
        return launch_fan(h, d_ins, (int)n, 0.0f, 1, d_out, d_hits, d_steps, nullptr, stream);
    } else {
        hipLaunchKernelGGL(rm_rays_kernel, dim3(grid), dim3(WG), 0, stream, m->mp, f, d_ins, n,
                           d_out, d_hits, d_steps);
    }
    HIPCHK(hipGetLastError());
    if (h->timing) { HIPCHK(hipEventRecord(h->ev1, stream)); h->timed = true; }
    return RL_OK;
}

// ------------------------------------------------------------------------------
// launch planning through the C ABI (pure host arithmetic: works without a device)
// ------------------------------------------------------------------------------
extern "C" int rl_plan_default_opts(rl_plan_opts *out)
{
    if (!out) return fail(RL_ERR_INVALID, "rl_plan_default_opts: null pointer");
    plan::default_opts(*out);
    return RL_OK;
}

extern "C" int rl_plan_fan(int kind, int n_cu, int rows, int cols, float max_range_px, int theta_disc,
                           const rl_plan_opts *opts_or_null, int n_poses, int num_rays, int want_aux,
                           int want_crash, rl_launch_plan *out)
{
    if (!out) return fail(RL_ERR_INVALID, "rl_plan_fan: null pointer");
    if (kind < RL_BRESENHAM || kind > RL_GIANT_LUT) return fail(RL_ERR_INVALID, "unknown range method kind %d", kind);
    if (n_cu <= 0 || rows <= 0 || cols <= 0 || n_poses < 0 || num_rays <= 0 || !(max_range_px > 0.0f))
        return fail(RL_ERR_INVALID, "rl_plan_fan: bad shape arguments");
    plan::In in;
    in.kind = kind;
    in.n_cu = n_cu;
    in.rows = rows;
    in.cols = cols;
    in.max_range = max_range_px;
    in.theta_disc = theta_disc;
    if (opts_or_null) in.o = *opts_or_null; else plan::default_opts(in.o);
    in.n_poses = n_poses;
    in.num_rays = num_rays;
    in.aux = want_aux != 0;
    in.crash = want_crash != 0;
    if (in.crash && kind != RL_RM && kind != RL_RM_GPU)
        return fail(RL_ERR_UNSUPPORTED, "fused crash test needs a ray-marching method");
    const int rc = plan::plan_fan(in, out);
    if (rc) return fail(rc, "no kernel of this variant serves the request");
    return RL_OK;
}

extern "C" int rl_method_plan_fan(rl_method *h, int n_poses, int num_rays, int want_aux, int want_crash,
                                  rl_launch_plan *out)
{
    if (!h || !out) return fail(RL_ERR_INVALID, "rl_method_plan_fan: null pointer");
    if (n_poses < 0 || num_rays <= 0) return fail(RL_ERR_INVALID, "rl_method_plan_fan: bad shape arguments");
    if (!h->reps.empty()) {                     // what ONE device launches for its block of the batch
        long lo, hi;
        block_of(n_poses, 0, multi_parts(h, n_poses), lo, hi);
        return rl_method_plan_fan(h->reps[0], (int)(hi - lo), num_rays, want_aux, want_crash, out);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    int rc = RL_OK;
    if (h->code_map && (h->kind == RL_RM || h->kind == RL_RM_GPU) && h->variant >= 1) {
        if ((rc = set_device(h->map))) return rc;
        std::shared_lock<std::shared_mutex> tl(h->map->tables_mu);
        if ((rc = ensure_step_map(h, h->stream))) return rc;
    }
    rc = plan_for(h, n_poses, num_rays, want_aux != 0, want_crash != 0, out);
    if (rc) return fail(rc, "no kernel of this variant serves the request");
    return RL_OK;
}

extern "C" int rl_method_last_plan(rl_method *h, rl_launch_plan *out)
{
    if (!h || !out) return fail(RL_ERR_INVALID, "rl_method_last_plan: null pointer");
    if (!h->reps.empty()) return rl_method_last_plan(h->reps[0], out);
    std::lock_guard<std::mutex> lk(h->mu);
    *out = h->last_plan;
    return RL_OK;
}

extern "C" int rl_launch_contexts(void) { return N_LAUNCH_CTX; }

extern "C" int rl_calc_range_fan_device(rl_method *h, const float *d_poses, int n_poses, float fov,
                                        int num_rays, float *d_outs, int32_t *d_hits,
                                        uint16_t *d_steps, void *hip_stream)
{
    int rc = check_fan_args(h, n_poses, fov, num_rays);
    if (rc) return rc;
    if (!h->reps.empty()) return multi_needs_replica("rl_calc_range_fan_device");
    if (n_poses > 0 && (!d_poses || !d_outs))
        return fail(RL_ERR_INVALID, "rl_calc_range_fan_device: null device pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    rc = set_device(h->map);
    if (rc) return rc;
    return launch_fan(h, d_poses, n_poses, fov, num_rays, d_outs, d_hits, d_steps, nullptr,
                      (hipStream_t)hip_stream);
}

extern "C" int rl_calc_range_many_device(rl_method *h, const float *d_ins, float *d_outs, int n,
                                         void *hip_stream)
{
    if (!h) return fail(RL_ERR_INVALID, "null method handle");
    if (n < 0) return fail(RL_ERR_INVALID, "n must be >= 0");
    if (!h->reps.empty()) return multi_needs_replica("rl_calc_range_many_device");
    if (n > 0 && (!d_ins || !d_outs))
        return fail(RL_ERR_INVALID, "rl_calc_range_many_device: null device pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    int rc = set_device(h->map);
    if (rc) return rc;
    return launch_rays(h, d_ins, n, d_outs, nullptr, nullptr, (hipStream_t)hip_stream);
}

// host-pointer forms ---------------------------------------------------------------
// per-pose crash marks: an int per pose that is never cleared between launches — every launch
// writes its own epoch (zeroed when the buffer grows or the epoch wraps)
static int pose_marks(rl_method *h, int n_poses, hipStream_t stream, int &mark, int **d_marks)
{
    LaunchCtx *cx = nullptr;
    int rc = acquire_ctx(h, stream, &cx);
    if (rc) return rc;
    const size_t cap_before = cx->pose_first.cap;
    rc = cx->pose_first.ensure((size_t)n_poses * sizeof(int));
    if (rc) return rc;
    if (cx->pose_first.cap != cap_before || cx->crash_epoch >= INT_MAX - 1) {
        HIPCHK(hipMemsetAsync(cx->pose_first.p, 0, cx->pose_first.cap, stream));
        cx->crash_epoch = 0;
    }
    mark = ++cx->crash_epoch;
    *d_marks = (int *)cx->pose_first.p;
    return RL_OK;
}

// pinned, device-mapped host staging of at least `bytes` (small host calls run zero-copy through it)
static int pin_ensure(rl_method *h, size_t bytes)
{
    if (bytes <= h->pin_cap) return RL_OK;
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_cap = 0;
    if (hipHostMalloc(&h->pin, bytes * 2, hipHostMallocDefault) != hipSuccess)
        return fail(RL_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes * 2);
    h->pin_cap = bytes * 2;
    return RL_OK;
}

// car-outline table -> h->edge, re-sent only when its contents changed since the last call
int upload_edge(rl_method *h, const double *edge, int num_rays)
{
    const size_t cap_before = h->edge.cap;
    int rc = h->edge.ensure((size_t)num_rays * sizeof(double));
    if (rc) return rc;
    if (h->edge.cap != cap_before || h->edge_host.size() != (size_t)num_rays ||
        memcmp(h->edge_host.data(), edge, (size_t)num_rays * sizeof(double)) != 0) {
        HIPCHK(hipMemcpyAsync(h->edge.p, edge, (size_t)num_rays * sizeof(double), hipMemcpyHostToDevice,
                              h->stream));
        h->edge_host.assign(edge, edge + num_rays);
    }
    return RL_OK;
}

int fan_host(rl_method *h, const float *poses, int n_poses, float fov, int num_rays,
                    float *outs, int32_t *hits, uint16_t *steps, const double *edge,
                    double crash_thresh, int *first_crashed)
{
    const size_t n_rays = (size_t)n_poses * num_rays;
    int rc = set_device(h->map);
    if (rc) return rc;
    if (n_poses == 0) {
        if (first_crashed) *first_crashed = -1;
        return RL_OK;
    }
    // small calls: zero-copy through pinned host memory (scan() 45 -> ~25 us host-visible)
    // output buffer inside a pinned block of rl_host_alloc: the kernel writes the ranges straight into it
    const bool direct_out = outs && !hits && !steps && n_rays <= (size_t)h->direct_max_rays &&
                            in_host_block(outs, n_rays * sizeof(float), h->map->device);
    const bool zc = !hits && !steps && (direct_out || n_rays <= (size_t)h->pinned_max_rays);
    const size_t off_out = ((size_t)n_poses * 3 * sizeof(float) + 255) & ~(size_t)255;
    const size_t off_end = off_out + (direct_out ? 0 : ((n_rays * sizeof(float) + 255) & ~(size_t)255));
    if (zc) {
        if ((rc = pin_ensure(h, off_end))) return rc;
        memcpy(h->pin, poses, (size_t)n_poses * 3 * sizeof(float));
    } else {
        if ((rc = h->poses.ensure((size_t)n_poses * 3 * sizeof(float)))) return rc;
        if (outs || !first_crashed)
            if ((rc = h->outs.ensure(n_rays * sizeof(float)))) return rc;
        if (hits && (rc = h->hits.ensure(n_rays * 2 * sizeof(int32_t)))) return rc;
        if (steps && (rc = h->steps.ensure(n_rays * sizeof(uint16_t)))) return rc;
        HIPCHK(hipMemcpyAsync(h->poses.p, poses, (size_t)n_poses * 3 * sizeof(float),
                              hipMemcpyHostToDevice, h->stream));
    }
    const float *d_poses = zc ? (const float *)h->pin : (const float *)h->poses.p;
    CrashParams cp{nullptr, 0.0, nullptr, 1, 0};
    const bool crash_direct = first_crashed && n_poses <= 512;
    if (first_crashed) {
        if ((rc = upload_edge(h, edge, num_rays))) return rc;
        if ((rc = h->flag.ensure(sizeof(int)))) return rc;
        if (!h->pin_flag && hipHostMalloc((void **)&h->pin_flag, 64, hipHostMallocDefault) != hipSuccess)
            return fail(RL_ERR_NOMEM, "hipHostMalloc(64) failed");
        cp.edge = (const double *)h->edge.p;
        cp.thresh = crash_thresh;
        if (crash_direct) {
            // one roll-out: atomicMin straight into the result word (few poses, little contention)
            hipLaunchKernelGGL(fill_int_kernel, dim3(1), dim3(64), 0, h->stream, (int *)h->flag.p, 1, INT_MAX);
            cp.first_crashed = (int *)h->flag.p;
            cp.group = n_poses;
        } else {
            // big batches: the kernel marks crashed poses (a word per pose), the first one is reduced
            // on the device afterwards (see crash_reduce_kernel)
            if ((rc = pose_marks(h, n_poses, h->stream, cp.mark, &cp.first_crashed))) return rc;
            cp.group = 0;
        }
    }
    float *d_out = (outs || !first_crashed)
                       ? (direct_out ? outs : zc ? (float *)((char *)h->pin + off_out) : (float *)h->outs.p)
                       : nullptr;
    if (outs && !zc && !first_crashed && !hits && !steps && h->overlap_min_rays > 0 &&
        n_rays >= (size_t)h->overlap_min_rays && n_poses >= 4 && !h->timing &&
        (h->kind == RL_RM || h->kind == RL_RM_GPU || h->kind == RL_BRESENHAM)) {
        // big plain scans are bound by the 4 B per ray going back over PCIe: four pose slices, the copy of slice k on
        // a second stream while slice k+1 marches (the march of a 65536-pose batch is ~10 % of the call).  The table
        // methods keep one launch: their kernels take 2-4 % of the call, and the theta-major CDDT search wants the
        // whole batch (>= 32768 poses) in one launch
        constexpr int S = 4;
        if (!h->copy_stream) HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        const int per = (n_poses + S - 1) / S;
        const uint64_t base_off = h->ray_offset;
        rc = RL_OK;
        for (int k = 0, p0 = 0; p0 < n_poses && rc == RL_OK; ++k, p0 += per) {
            const int np = std::min(per, n_poses - p0);
            const size_t r0 = (size_t)p0 * num_rays, nr = (size_t)np * num_rays;
            h->ray_offset = base_off + r0;               // noise stays keyed by the global ray id
            rc = launch_fan(h, d_poses + (size_t)p0 * 3, np, fov, num_rays, d_out + r0, nullptr, nullptr, nullptr, h->stream);
            if (rc) break;
            if (!h->slice_ev[k] && hipEventCreateWithFlags(&h->slice_ev[k], hipEventDisableTiming) != hipSuccess) {
                rc = fail(RL_ERR_HIP, "hipEventCreate failed");
                break;
            }
            if (hipEventRecord(h->slice_ev[k], h->stream) != hipSuccess ||
                hipStreamWaitEvent(h->copy_stream, h->slice_ev[k], 0) != hipSuccess ||
                hipMemcpyAsync(outs + r0, d_out + r0, nr * sizeof(float), hipMemcpyDeviceToHost, h->copy_stream) != hipSuccess)
                rc = fail(RL_ERR_HIP, "sliced device-to-host copy failed");
        }
        h->ray_offset = base_off;
        // (both streams are drained whatever happened; a kernel fault or a copy error that only surfaces here
        //  must not come back as RL_OK with garbage in `outs`)
        const hipError_t e_launch = hipGetLastError();
        const hipError_t e_march = hipStreamSynchronize(h->stream);
        const hipError_t e_copy = hipStreamSynchronize(h->copy_stream);
        if (rc == RL_OK && (e_launch != hipSuccess || e_march != hipSuccess || e_copy != hipSuccess))
            rc = fail(RL_ERR_HIP, "sliced host-pointer scan failed: launch %s, march stream %s, copy stream %s",
                      hipGetErrorString(e_launch), hipGetErrorString(e_march), hipGetErrorString(e_copy));
        return rc;
    }
    rc = launch_fan(h, d_poses, n_poses, fov, num_rays, d_out,
                    hits ? (int32_t *)h->hits.p : nullptr, steps ? (uint16_t *)h->steps.p : nullptr,
                    first_crashed ? &cp : nullptr, h->stream);
    if (rc) return rc;
    if (outs && !zc)
        HIPCHK(hipMemcpyAsync(outs, h->outs.p, n_rays * sizeof(float), hipMemcpyDeviceToHost,
                              h->stream));
    if (hits)
        HIPCHK(hipMemcpyAsync(hits, h->hits.p, n_rays * 2 * sizeof(int32_t), hipMemcpyDeviceToHost,
                              h->stream));
    if (steps)
        HIPCHK(hipMemcpyAsync(steps, h->steps.p, n_rays * sizeof(uint16_t), hipMemcpyDeviceToHost,
                              h->stream));
    if (first_crashed) {
        if (!crash_direct)
            hipLaunchKernelGGL(crash_reduce_kernel, dim3(1), dim3(64), 0, h->stream,
                               (const int *)cp.first_crashed, cp.mark, 1, n_poses, (int *)h->flag.p);
        HIPCHK(hipMemcpyAsync(h->pin_flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    int flag = first_crashed ? *h->pin_flag : 0;
    if (crash_direct && flag == INT_MAX) flag = -(n_poses + 1);
    if (zc && !direct_out) {
        if (outs) memcpy(outs, (char *)h->pin + off_out, n_rays * sizeof(float));
    }
    if (first_crashed) *first_crashed = flag;      // first crashed pose, or -(n_poses + 1)
    return RL_OK;
}


// rl_calc_range_many on one device: (x, y, theta) rows in, ranges out
int rays_host(rl_method *h, const float *ins, float *outs, int n)
{
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    int rc = set_device(h->map);
    if (rc) return rc;
    if (n <= h->pinned_max_rays) {                       // one scan's worth of rows: zero-copy
        const size_t off_out = ((size_t)n * 3 * sizeof(float) + 255) & ~(size_t)255;
        if ((rc = pin_ensure(h, off_out + (size_t)n * sizeof(float)))) return rc;
        memcpy(h->pin, ins, (size_t)n * 3 * sizeof(float));
        float *p_out = (float *)((char *)h->pin + off_out);
        if ((rc = launch_rays(h, (const float *)h->pin, n, p_out, nullptr, nullptr, h->stream))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(outs, p_out, (size_t)n * sizeof(float));
        return RL_OK;
    }
    if ((rc = h->poses.ensure((size_t)n * 3 * sizeof(float)))) return rc;
    if ((rc = h->outs.ensure((size_t)n * sizeof(float)))) return rc;
    HIPCHK(hipMemcpyAsync(h->poses.p, ins, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice,
                          h->stream));
    rc = launch_rays(h, (const float *)h->poses.p, n, (float *)h->outs.p, nullptr, nullptr,
                     h->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(outs, h->outs.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RL_OK;
}

extern "C" int rl_method_read_lut(rl_method *h, int row0, int row1, uint16_t *out)
{
    if (!h || !out) return fail(RL_ERR_INVALID, "rl_method_read_lut: null pointer");
    if (h->kind != RL_GIANT_LUT) return fail(RL_ERR_INVALID, "not a GiantLUT method");
    if (!h->reps.empty()) return rl_method_read_lut(h->reps[0], row0, row1, out);
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    int rc = set_device(h->map);
    if (rc) return rc;
    if (row0 < 0 || row1 > h->map->rows || row0 > row1)
        return fail(RL_ERR_INVALID, "row range [%d,%d) outside the map", row0, row1);
    if ((rc = ensure_lut(h, h->stream))) return rc;
    const size_t per_row = (size_t)h->map->cols * h->theta_disc;
    HIPCHK(hipMemcpyAsync(out, (const uint16_t *)h->lut.p + (size_t)row0 * per_row,
                          (size_t)(row1 - row0) * per_row * sizeof(uint16_t), hipMemcpyDeviceToHost,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RL_OK;
}

extern "C" int rl_debug_read_stamps(rl_method *h, uint64_t *out, int max_words)
{
    if (!h || !out) return fail(RL_ERR_INVALID, "rl_debug_read_stamps: null pointer");
    if (!h->reps.empty()) return rl_debug_read_stamps(h->reps[0], out, max_words);
    std::lock_guard<std::mutex> lk(h->mu);
    int rc = set_device(h->map);
    if (rc) return rc;
    size_t words = (size_t)h->last_grid * WAVES_PER_WG * 4;
    if (!h->last_dbg || words == 0) return fail(RL_ERR_INVALID, "no stamps recorded (set debug_stamps=1)");
    if ((size_t)max_words < words) words = (size_t)max_words;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, h->last_dbg, words * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return (int)words;
}

// ------------------------------------------------------------------------------
// grouped crash test and the roll-out generator ("next" rows)
// ------------------------------------------------------------------------------
// d_first[g] <- first crashed pose of group g, or INT_MAX when none (finalize = false), or
// -(group+1) (finalize = true).  Ray-marching methods fuse the test into the march kernel; the
// others scan into d_ranges (required then) and run one pass over the ranges.
int crash_groups_device(rl_method *h, const float *d_poses, int n_groups, int group, float fov,
                               int num_rays, const double *d_edge, double thresh, int *d_first,
                               float *d_ranges, bool finalize, hipStream_t stream)
{
    (void)finalize;
    const int n_poses = n_groups * group;
    // the kernels mark crashed POSES (one word each, no contended atomics); groups are reduced after
    int rc;
    int mark;
    int *d_pose_first = nullptr;
    if ((rc = pose_marks(h, n_poses, stream, mark, &d_pose_first))) return rc;
    if (h->kind == RL_RM || h->kind == RL_RM_GPU) {
        CrashParams cp{d_edge, thresh, d_pose_first, 0, mark};
        if ((rc = launch_fan(h, d_poses, n_poses, fov, num_rays, d_ranges, nullptr, nullptr, &cp, stream)))
            return rc;
    } else {
        if (!d_ranges) return fail(RL_ERR_INVALID, "this range method needs a ranges buffer for the crash test");
        if ((rc = launch_fan(h, d_poses, n_poses, fov, num_rays, d_ranges, nullptr, nullptr, nullptr, stream)))
            return rc;
        const int grid = (int)std::max(1L, std::min(((long)n_poses + 3) / 4, (long)h->map->n_cu * 8));
        hipLaunchKernelGGL(crash_groups_kernel, dim3(grid), dim3(256), 0, stream, d_ranges, d_edge,
                           thresh, n_poses, num_rays, 0, mark, d_pose_first);
    }
    const int rgrid = (int)std::max(1L, std::min(((long)n_groups + 3) / 4, (long)h->map->n_cu * 8));
    hipLaunchKernelGGL(crash_reduce_kernel, dim3(rgrid), dim3(256), 0, stream, d_pose_first, mark,
                       n_groups, group, d_first);
    HIPCHK(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_check_collision_groups_device(rl_method *h, const float *d_poses, int n_groups,
                                                int group, float fov, int num_rays,
                                                const double *d_edge, double crash_thresh,
                                                int *d_first_crashed, float *d_ranges_or_null,
                                                void *hip_stream)
{
    if (n_groups < 0 || group <= 0) return fail(RL_ERR_INVALID, "n_groups >= 0 and group > 0 required");
    if ((long)n_groups * group > INT_MAX) return fail(RL_ERR_INVALID, "too many poses");
    int rc = check_fan_args(h, n_groups * group, fov, num_rays);
    if (rc) return rc;
    if (n_groups == 0) return RL_OK;
    if (!h->reps.empty()) return multi_needs_replica("rl_check_collision_groups_device");
    if (!d_poses || !d_edge || !d_first_crashed)
        return fail(RL_ERR_INVALID, "rl_check_collision_groups_device: null device pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    std::shared_lock<std::shared_mutex> ml(h->map->tables_mu);
    if ((rc = set_device(h->map))) return rc;
    return crash_groups_device(h, d_poses, n_groups, group, fov, num_rays, d_edge, crash_thresh,
                               d_first_crashed, d_ranges_or_null, true, (hipStream_t)hip_stream);
}


extern "C" int rl_last_kernel_ms(rl_method *h, float *ms_out)
{
    if (!h || !ms_out) return fail(RL_ERR_INVALID, "rl_last_kernel_ms: null pointer");
    if (!h->reps.empty()) return rl_last_kernel_ms(h->reps[0], ms_out);
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->timed) return fail(RL_ERR_INVALID, "no launch has been timed on this handle (set option \"timing\"=1 first)");
    int rc = set_device(h->map);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms_out, h->ev0, h->ev1));
    return RL_OK;
}

