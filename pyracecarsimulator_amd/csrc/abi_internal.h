// abi_internal.h — what the translation units of libscan_amd.so share: the handles behind include/scanlib.h's opaque
// pointers, the error slot, and the few internal entry points one unit calls in another.
//   abi_map.hip    errors, pinned host blocks, rl_map_* (EDT, bit map, edge list)
//   abi_fan.hip    rl_method_*: options, derived tables, the launch planner's C ABI, every fan / ray launch, the
//                  device-pointer entry points, the single-device host-pointer paths, the fused crash test
//   abi_multi.hip  the host-pointer entry points and their multi-device forms (one pose block per device)
//   abi_car.hip    roll-out generator, FollowGap, 16-bit ranges, probes, the car-outline table
#pragma once
// (the units are built with -fvisibility=hidden: only the C ABI leaves the library)
#pragma GCC visibility push(default)
#include "../../include/scanlib.h"
#pragma GCC visibility pop
#include "scan_params.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace scan;

// ------------------------------------------------------------------------------
// errors (abi_map.hip): one message per thread
// ------------------------------------------------------------------------------
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const std::string &last_error();
void set_last_error(const std::string &msg);

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)


// is [p, p+bytes) inside a block of rl_host_alloc?  `device` >= 0: a launch on that device is about to write it
bool in_host_block(const void *p, size_t bytes, int device = -1);

// ------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------
struct rl_map {
    // a MULTI-DEVICE map (rl_map_create_multi) owns no device memory itself: it holds one ordinary map per
    // device in `reps` (the same device may appear several times) and its own fields describe the shape only
    std::vector<rl_map *> reps;
    int device = 0;
    int clock_khz = 0;
    int rows = 0, cols = 0;
    float res = 0, ox = 0, oy = 0, oyaw = 0;
    uint8_t *d_occ = nullptr;
    uint8_t *d_occ_base = nullptr;   // rl_map_stamp_cells: the occupancy as created / last rl_map_update'd (made at the first stamp)
    int32_t *d_stamp = nullptr;      // ... the cell indices of the stamp in place (restored by the next one)
    int32_t *pin_stamp = nullptr;    // ... pinned, device-mapped landing buffer of a call's indices (no staging copy)
    int stamp_cap = 0, n_stamped = 0;
    int *d_g = nullptr;          // EDT pass-1 scratch
    float *d_dt = nullptr;
    uint32_t *d_bits = nullptr;
    int bits_stride = 0;
    hipStream_t stream = nullptr;
    std::atomic<uint64_t> epoch{0};   // bumped by rl_map_update; derived tables rebuild lazily
    MapParams mp{};
    MapParams *d_mp = nullptr;   // device copy (kernels that take the map by pointer)
    // edge cells (occupied with a free 4-neighbour), the input of every CDDT table of this map: built
    // with the other map tables once a CDDT method exists, so that a table rebuild knows the count on
    // the host without a read-back of its own (rl_map_update synchronises anyway)
    bool want_edges = false;
    uint32_t *d_edges = nullptr, *d_n_edges = nullptr;
    uint32_t *pin_n_edges = nullptr;
    uint32_t n_edges = 0;
    int n_cu = 256;
    std::mutex mu;
    // readers: every launch path of every method of this map (held for the whole call, i.e. until
    // the results of a host-pointer call have landed); writer: rl_map_update while it rewrites
    // occ / EDT / bit map.  A map callback thread and a scan thread may share the objects
    // (scripts/ros_interface.py:107-115).
    std::shared_mutex tables_mu;
    // MULTI-DEVICE map only.  readers: the multi_* host-pointer entry points of every method of this map, for the
    // whole batch (every device's block); writer: rl_map_update while it walks the replicas — so one batch is
    // never scanned partly on the old and partly on the new occupancy.  `broken`: an update failed after some
    // replicas had already taken the new cells; the handle then refuses every further call instead of answering
    // from two different maps.
    std::shared_mutex multi_mu;
    std::atomic<bool> broken{false};
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return RL_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(RL_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want,
                                         hipGetErrorString(e));
        cap = want;
        return RL_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Per-launch scratch of a method (pose records, tile order, binning histograms, crash marks) is
// kept PER STREAM: a *_device call only enqueues work, so a second call on another stream may run
// concurrently with the first on the GPU (bench.py pipelines consecutive batches on two streams so
// that batch k+1 fills the CUs batch k's tail leaves idle).  Calls on one stream reuse one context
// in stream order.  More distinct streams than contexts: the least recently used context is handed
// over after a device synchronisation (rare, and needs no handle of the old stream, which the
// caller may have destroyed).
struct LaunchCtx {
    hipStream_t stream = nullptr;
    bool bound = false;
    uint64_t last_use = 0;
    DevBuf rec, rec_sorted, order, keys, hist, pose_first, dbg, d0, cddt_r;   // cddt_r: theta-major CDDT, R[raw bin][pose]
    DevBuf left_rec, left_cnt;     // hand-off march: the leftover list (rm_leftover_kernel), one region per wave of the main grid
    int crash_epoch = 0;           // mark value of the last per-pose crash launch (pose_marks)
    void release()
    {
        for (DevBuf *b : {&rec, &rec_sorted, &order, &keys, &hist, &pose_first, &dbg, &d0, &cddt_r, &left_rec, &left_cnt}) b->release();
    }
};
constexpr int N_LAUNCH_CTX = 8;      // (HIP's default 4 hardware queues carry 4 concurrent streams; GPU_MAX_HW_QUEUES=8 carries 8)

// A derived table (step map, GiantLUT, CDDT) is built lazily on the stream of the call that needs
// it first; launches on OTHER streams must not start before the build has finished.
struct TableDep {
    hipEvent_t ev = nullptr;
    hipStream_t built_on = nullptr;
    bool pending = false;
};


// ------------------------------------------------------------------------------
// Several devices behind one handle (rl_map_create_multi): a single-process caller — the reference's
// scanMany / checkCollisionMany callers are ONE Python process (scripts/mcts.py:237,
// scripts/scan_simulator.py:113-135) — hands over one pose batch and every device scans a contiguous
// block of it.  One persistent worker thread per device (bound to it with hipSetDevice once) runs the
// ordinary single-device entry point on that device's replica handle; job 0 runs on the calling thread.
// Threads and streams only: nothing is forked or re-executed after the GPU has been initialised.
// ------------------------------------------------------------------------------
struct MultiPool {
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;
        bool has = false, done = false, stop = false;
        int rc = 0;
        std::string err;
        int device = 0;
    };
    std::vector<std::unique_ptr<Worker>> w;

    void start(const std::vector<int> &devices)
    {
        for (size_t i = 1; i < devices.size(); ++i) {       // (block 0 is the caller's)
            auto wk = std::make_unique<Worker>();
            wk->device = devices[i];
            Worker *raw = wk.get();
            wk->th = std::thread([raw]() {
                (void)hipSetDevice(raw->device);
                std::unique_lock<std::mutex> lk(raw->mu);
                for (;;) {
                    raw->cv.wait(lk, [raw]() { return raw->has || raw->stop; });
                    if (raw->stop) return;
                    raw->has = false;
                    lk.unlock();
                    const int rc = raw->job();
                    std::string msg = rc ? last_error() : std::string();
                    lk.lock();
                    raw->rc = rc;
                    raw->err = std::move(msg);
                    raw->done = true;
                    raw->cv.notify_all();
                }
            });
            w.push_back(std::move(wk));
        }
    }

    // jobs[0] on the caller, jobs[i] on worker i-1; the first failure (lowest block) is reported, its message
    // becomes the caller's rl_last_error
    int run(std::vector<std::function<int()>> &jobs)
    {
        if (jobs.size() > w.size() + 1)
            return fail(RL_ERR_INVALID, "internal: %zu pose blocks for %zu devices", jobs.size(), w.size() + 1);
        const size_t k = jobs.size();
        for (size_t i = 1; i < k; ++i) {
            Worker &x = *w[i - 1];
            std::lock_guard<std::mutex> lk(x.mu);
            x.job = std::move(jobs[i]);
            x.has = true;
            x.done = false;
            x.cv.notify_all();
        }
        int rc = jobs.empty() ? RL_OK : jobs[0]();
        std::string err = rc ? last_error() : std::string();
        for (size_t i = 1; i < k; ++i) {
            Worker &x = *w[i - 1];
            std::unique_lock<std::mutex> lk(x.mu);
            x.cv.wait(lk, [&x]() { return x.done; });
            if (rc == RL_OK && x.rc != RL_OK) {
                rc = x.rc;
                err = x.err;
            }
        }
        if (rc) set_last_error(err);
        return rc;
    }

    ~MultiPool()
    {
        for (auto &x : w) {
            {
                std::lock_guard<std::mutex> lk(x->mu);
                x->stop = true;
                x->cv.notify_all();
            }
            if (x->th.joinable()) x->th.join();
        }
    }
};

// contiguous block of `rank` when n items are cut into `parts` (the same split as workloads.shard_range)
static inline void block_of(long n, int rank, int parts, long &lo, long &hi)
{
    const long base = n / parts, rem = n % parts;
    lo = rank * base + std::min<long>(rank, rem);
    hi = lo + base + (rank < rem ? 1 : 0);
}

struct rl_method {
    // multi-device method (created on a multi-device map): one ordinary method per device + the worker pool;
    // the parent keeps kind / noise / options and owns no device memory
    std::vector<rl_method *> reps;
    std::unique_ptr<MultiPool> pool;
    int multi_min_poses = 512;   // a device is only brought in per this many poses: waking a worker costs ~18 us
                                 // (profiles/r04/host_pointer_rate.txt: a 200-pose roll-out cut over three contexts 60 vs
                                 // 42 us), a 512-pose block's transfer alone ~50 us — the reference's roll-out stays on one device
    rl_map *map = nullptr;
    int kind = 0;
    float max_range = 0;
    float step_coeff = 0.999f;
    int theta_disc = 0;
    float noise_std = 0;
    uint64_t noise_seed = 0, ray_offset = 0;
    int variant = 1;             // 0: chunk kernel (K1); 1: binned + banded + lane-refill stream kernel (K1b)
    int grid_mult = 8;           // workgroups per CU for the persistent launches (8 resident: <= 80 SGPRs, <= 64 VGPRs)
    int low_water = -1;          // stream kernel: refill when <= this many lanes (per ray slot) still march.  -1 = auto: 12, and 20
                                 // for the several-rays-per-lane launches that derive their records in LDS (small and mid-size
                                 // batches: +3 % with four in flight; big batches lose 3-6 % above 12: profiles/r03/ab_low_water.txt)
    int sort_poses = 1;          // stream kernel: order poses by map tile
    int xcd_bands = 8;           // stream kernel: bands of the sorted list, one per XCD
    int timing = 0;              // 1: HIP events around every launch sequence (rl_last_kernel_ms);
                                 // 2: around the march kernel only (pose binning excluded)
    int lut_debug = 0;
    int drain_prio = 0;
    int spec_drain = 8;          // one ray per lane: value-speculating drain loop once <= this many lanes are live (0 = off)
    int spec_stretch = 16;       //   ... after this many plain samples, and between two attempts whose first prediction failed
    int drain_cap = 64;          // several rays per lane: compact a wave's last rays into one slot from <= this many (<= 64)
    int drain_stretch = 8;       //   ... plain samples between two speculation attempts of the compacted rays
    int group_drain = 0;         //   ... and from 2 N / N live rays down 2 / 4 lanes per ray, 8 / 16 samples per round trip (N <= 16; 0: off)
    int handoff = 0;             // several rays per lane, 1: a dry wave hands its last <= handoff_cap rays to rm_leftover_kernel (the
                                 // next launch on the stream) instead of draining them in place (0: drain in place)
    int handoff_cap = 16;        //   ... rays per wave handed over (8, 16, 32 or 64)
    int handoff_wg = 256;        //   ... workgroup size of the leftover launch (64, 128 or 256)
    int nt_store = 1;            // ranges leave the stream kernels with non-temporal stores (0: plain — a consumer kernel reads them next)
    int wg_threads = 1024;       // stream kernel: workgroup size (256/512/1024) sharing one ray stream
    // GiantLUT (K3)
    DevBuf lut;
    uint64_t lut_epoch = ~0ull;
    LutParams lp{};
    // CDDT (K3b)
    DevBuf cd_cos, cd_sin, cd_trans, cd_width, cd_boff, cd_offsets, cd_xs2, cd_cursor, cd_tmp, cd_hdr, cd_tab;
    uint64_t cddt_epoch = ~0ull;
    CddtParams cdp{};
    uint32_t cd_buckets = 0;
    std::vector<float> cd_h_cos, cd_h_sin, cd_h_trans;     // per-bin constants (host copies stay alive:
    std::vector<int> cd_h_width;                           //  their uploads are asynchronous)
    std::vector<uint32_t> cd_h_boff;
    int cd_geom_rows = -1, cd_geom_cols = -1;              // map shape the constants were made for
    bool cd_sort_attr = false;
    bool cd_counts_clean = false;                          // bucket counters are all zero (see ensure_cddt)
    int slots = 0;               // stream kernel: rays per lane; 2 (3: inline form only) = plain-range launches on the tiled step
                                 // map keep two loads in flight per lane and compact a dry wave's last rays into one slot;
                                 // 0 = auto (launch_plan.h: 2 from 2^23 rays per launch up, from 2^20 on maps beyond the
                                 // small-map bound; callers that keep several launches in flight set 2: +15..30 %)
    int cddt_theta_min = 32768;                            // poses per launch from which the CDDT look-ups run theta-major (0: never)
    int cddt_search = 1;                                   // theta-major search kernel: 1 = look-ups prepared once per pose (round 5), 0 = round 4's
    int cddt_sort = 0;                                     // per-bin fan kernel walks the poses in map-tile order, XCD bands
                                                           // (measured: -13 % at 4096 poses - the binning launch and no
                                                           // reuse at that density -, +3 % at 32768: off by default)
    int cddt_lds_sort = (int)CDDT_LDS_SORT;                // buckets up to this size are sorted in LDS (diagnostics: lower it)
    int cddt_bins_kernel = 1;                              // 1: one query per (pose, theta bin); 0: per ray
    DevBuf blpad;                // K2b: padded normal + transposed bit maps (bl_pad_bits_kernel)
    BlPad blp{};
    uint64_t blpad_epoch = ~0ull;
    TableDep blpad_dep;
    DevBuf pdt;                  // EDT with a border of `pad` cells of -1 (stream kernel)
    int pad = 0, pstride = 0;    // pstride: elements per row (row-major) | M (tiled, see pdt_tiled_byte)
    uint64_t pdt_epoch = ~0ull;  // map epoch the padded copy was built from
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // big host-pointer calls: the D2H copy of pose slice k overlaps the march of slice k+1
    hipEvent_t slice_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int overlap_min_rays = 1 << 24;      // ... from this many rays per call (0 = never); below ~16 k poses the slices cost more than they hide
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    DevBuf poses, outs, hits, steps, edge, flag;
    LaunchCtx ctx[N_LAUNCH_CTX];
    uint64_t use_clock = 0;
    TableDep pdt_dep, lut_dep, cddt_dep;
    // beam-direction tables (cos, sin per beam) of the fans this handle has been called with
    struct FanTab {
        uint32_t fov_bits = 0;
        int num_rays = 0;
        uint64_t last_use = 0;
        DevBuf tab;
        TableDep dep;
    } fan_tabs[4];
    uint64_t fan_clock = 0;
    // small host calls (scan(): one pose, scanMany(): a roll-out): poses and ranges go through ONE
    // pinned, device-mapped host buffer the kernels read / write directly — no staging copies
    void *pin = nullptr;
    size_t pin_cap = 0;
    int pinned_max_rays = 262144; // 0 = always stage through device buffers
    int direct_max_rays = 1 << 21; // a result buffer in a pinned block of rl_host_alloc is written by the kernel itself
                                  // up to this many rays (a 200-pose roll-out: 41 vs 61 us, 1024 poses: 113 vs 132); larger
                                  // batches go HBM -> DMA into the pinned block, which moves 4 B per ray faster than the
                                  // kernel's stores over PCIe (4096 poses: 394 vs 449 us; a tie at 2048:
                                  // profiles/r04/host_pointer_rate.txt)
    std::vector<double> edge_host; // the car-outline table last uploaded to `edge` (re-sent only when it changes)
    int *pin_flag = nullptr;       // pinned landing slot for the crash index
    int bin_multi_min = 8192;    // batches at least this large bin poses with grid-wide kernels
    int tile_stripe = -1;        // binning order: tile rows per stripe walked column-major (rm_kernels.h tile_key); 0: row-major, -1: by xcd_bands
    int bin_ppw = POSES_PER_WG;  // ... poses per workgroup of those kernels
    int order_inline = 1;        // big maps, stripe_max..8192 poses: keys-only binning launch + INLINE march
    int stripe_max = 1536;       // big maps, inline_max..stripe_max poses: no binning launch, workgroups compact
                                 // their own row-stripe band of the pose list (0 = off)
    int inline_map_kb = 2048;    // maps up to this size (f32 cells) never take the binning launch while the records fit LDS
    int run_log2 = -1;           // stream interleave granularity: runs of 2^run_log2 blocks; -1 = by batch size
    int tiled = 1;               // step map with 4 rows interleaved (a 128-B line = 4x8 cells); 0 = row-major
    int pdt_tiled = -1;          // layout the padded copy was built with
    uint32_t pdt_k4 = 0, pdt_mask = 0;
    size_t pdt_base_off = 0;     // tiled: the column bias (pad << 4 bytes) folded into the base address
    // the CODE map (option code_map = 2): the step map as u16 palette codes + the palette (rm_kernels.h), built next to
    // the float32 step map by ensure_step_map; code_n = palette entries with the two stop codes, 0 = none (option off,
    // geometry does not fit, or more distinct steps than plan::CODE_MAX_ENTRIES)
    int code_map = 2;
    int code_min_rays = 1 << 22;
    int tail_pct = 0, tail_wg_pct = 50;
    DevBuf cmap, cval, cidx, ctab, cnum;
    uint32_t *pin_cnum = nullptr;
    int code_n = 0;
    int code_built = -1;         // code_map value the tables were built for
    int cstride = 0;             // M of the code map's address
    uint32_t ck4 = 0, cmask = 0;
    size_t cbase_off = 0;
    int slice_log2 = 30;         // launches are cut into pose slices below 2^slice_log2 rays
    int bin_generic = 0;         // diagnostics: force the generic single-workgroup binning kernel
    int inline_prep = 1;         // tiny batches: no binning launch, workgroups derive their own records
    int inline_max = 512;        //   ... below this many poses (measured: wins below ~512, loses above)
    int debug_stamps = 0;        // diagnostics: per-wave start/end stamps of the stream kernel
    int last_grid = 0;
    void *last_dbg = nullptr;    // stamps buffer of the last launch (in its context)
    rl_launch_plan last_plan{};  // what the last fan launch of this handle was planned as (plan::plan_fan)
    std::vector<float> h_poses;
    std::mutex mu;
};

int set_device(const rl_map *m);
int map_build_tables(rl_map *m);                    // EDT + bit map (+ edge list once a CDDT method exists); abi_map.hip
void host_sincosf(float x, float &s, float &c);      // host twin of scan::det_sincosf (abi_map.hip)

// ------------------------------------------------------------------------------
// abi_fan.hip, called from abi_multi.hip / abi_car.hip
// ------------------------------------------------------------------------------
int check_fan_args(const rl_method *h, int n_poses, float fov, int num_rays);
int fan_host(rl_method *h, const float *poses, int n_poses, float fov, int num_rays, float *outs, int32_t *hits,
             uint16_t *steps, const double *edge, double crash_thresh, int *first_crashed);
int crash_groups_device(rl_method *h, const float *d_poses, int n_groups, int group, float fov, int num_rays,
                        const double *d_edge, double thresh, int *d_first, float *d_ranges, bool finalize,
                        hipStream_t stream);
int upload_edge(rl_method *h, const double *edge, int num_rays);
int rays_host(rl_method *h, const float *ins, float *outs, int n);
int multi_parts(const rl_method *h, long n_poses);
