// scanlib.hip — host side of libscan_amd.so (C ABI declared in include/scanlib.h).
//
// Replaces, for the scan path only, range_libc's PyOMap / PyRayMarching /
// PyRayMarchingGPU / PyCDDTCast objects that the reference builds at
// scripts/scan_simulator.py:72-76, scripts/ros_interface.py:210 and
// scripts/two_player/scan.py:45-46.  There is no CPU fallback in this library.
#include "../../include/scanlib.h"
#include "scan_kernels.h"
#include "car_kernels.h"
#include "consumer_kernels.h"
#include "probe_kernels.h"
#include "launch_plan.h"

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

static_assert(plan::WG == scan::WG && plan::STREAM_HDR == scan::STREAM_HDR && plan::STRIPE_BINS == scan::STRIPE_BINS &&
                  plan::STRIPE_MAX_PER_LANE == scan::STRIPE_MAX_PER_LANE && plan::DRAIN_CAP == scan::DRAIN_CAP &&
                  plan::DRAIN_FIELDS == scan::DRAIN_FIELDS && plan::INLINE_REC_BYTES == (int)sizeof(scan::BlockRec),
              "launch_plan.h and rm_kernels.h disagree about the stream kernels' LDS layout");

// ------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------
static thread_local std::string g_err = "";

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

extern "C" const char *rl_last_error(void) { return g_err.c_str(); }
extern "C" const char *rl_version(void) { return "scanlib-amd 0.5 (gfx950)"; }

extern "C" int rl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------
// pinned host blocks handed to callers (rl_host_alloc): a scan whose output buffer lies in one of
// them is written by the kernel directly — no staging copy on the way back
// ------------------------------------------------------------------------------
static std::mutex g_host_mu;
struct HostBlock {
    char *p;
    size_t bytes;
    uint64_t devices;      // devices that may have work in flight on the block: where it was allocated and
};                         // every device a scan was launched from with its output inside the block
static std::vector<HostBlock> g_host_blocks;

// is [p, p+bytes) inside a block of rl_host_alloc?  `device` >= 0: a launch on that device is about to write it
static bool in_host_block(const void *p, size_t bytes, int device = -1)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (auto &b : g_host_blocks)
        if ((const char *)p >= b.p && (const char *)p + bytes <= b.p + b.bytes) {
            if (device >= 0 && device < 64) b.devices |= 1ull << device;
            return true;
        }
    return false;
}

extern "C" int rl_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return fail(RL_ERR_INVALID, "rl_host_alloc: bad arguments");
    if (rl_device_count() <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device available");
    void *p = nullptr;
    // (portable + mapped: every device of a multi-device handle writes its pose block's ranges straight into it)
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess)
        return fail(RL_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    memset(p, 0, bytes);
    int cur = 0;
    (void)hipGetDevice(&cur);
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        g_host_blocks.push_back(HostBlock{(char *)p, bytes, (cur >= 0 && cur < 64) ? 1ull << cur : 0ull});
    }
    *out = p;
    return RL_OK;
}

extern "C" int rl_host_free(void *p)
{
    if (!p) return RL_OK;
    uint64_t devices = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = std::find_if(g_host_blocks.begin(), g_host_blocks.end(),
                               [&](const HostBlock &b) { return b.p == (char *)p; });
        if (it == g_host_blocks.end()) return fail(RL_ERR_INVALID, "rl_host_free: not a block of rl_host_alloc");
        devices = it->devices;
        g_host_blocks.erase(it);
    }
    // a kernel may still be writing into it: wait for the devices that were handed the block — not for every
    // visible device (a rank of an N-GPU job would create contexts on, and stall, its neighbours' GPUs)
    int ndev = 0, cur = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && hipGetDevice(&cur) == hipSuccess) {
        for (int d = 0; d < ndev && d < 64; ++d)
            if (((devices >> d) & 1ull) && hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
        (void)hipSetDevice(cur);
    }
    HIPCHK(hipHostFree(p));
    return RL_OK;
}

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
                    std::string msg = rc ? g_err : std::string();
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
        std::string err = rc ? g_err : std::string();
        for (size_t i = 1; i < k; ++i) {
            Worker &x = *w[i - 1];
            std::unique_lock<std::mutex> lk(x.mu);
            x.cv.wait(lk, [&x]() { return x.done; });
            if (rc == RL_OK && x.rc != RL_OK) {
                rc = x.rc;
                err = x.err;
            }
        }
        if (rc) g_err = err;
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

static int set_device(const rl_map *m)
{
    HIPCHK(hipSetDevice(m->device));
    return RL_OK;
}

// ------------------------------------------------------------------------------
// map
// ------------------------------------------------------------------------------
static void host_sincosf(float x, float &s, float &c)
{
    // host twin of scan::det_sincosf (same operations; this TU is built with
    // -ffp-contract=off and fmaf is a single rounding on the host too)
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float P1 = 0x1.921fb6p+0f, P2 = -0x1.777a5cp-25f, P3 = -0x1.ee59dap-50f;
    float k = rintf(x * TWO_OVER_PI);
    float r = fmaf(-k, P1, x);
    r = fmaf(-k, P2, r);
    r = fmaf(-k, P3, r);
    float z = r * r;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    float sr = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    float ss = (q & 1) ? cr : sr;
    float cc = (q & 1) ? sr : cr;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    s = ss;
    c = cc;
}

static int map_build_tables(rl_map *m)
{
    // K0: exact EDT + bit-packed occupancy, all on the device
    const int rows = m->rows, cols = m->cols;
    hipLaunchKernelGGL(edt_cols_kernel, dim3((cols + 63) / 64), dim3(1024), 0, m->stream,
                       m->d_occ, rows, cols, m->d_g);
    hipLaunchKernelGGL(edt_rows_kernel, dim3(rows), dim3(256), (size_t)cols * sizeof(int),
                       m->stream, m->d_g, rows, cols, m->d_dt);
    hipLaunchKernelGGL(pack_bits_kernel, dim3((m->bits_stride + 255) / 256, rows), dim3(256), 0,
                       m->stream, m->d_occ, rows, cols, m->bits_stride, m->d_bits);
    if (m->want_edges) {
        if (!m->d_edges) {
            HIPCHK(hipMalloc((void **)&m->d_edges, (size_t)rows * cols * sizeof(uint32_t)));
            HIPCHK(hipMalloc((void **)&m->d_n_edges, 256));
            HIPCHK(hipHostMalloc((void **)&m->pin_n_edges, 64, hipHostMallocDefault));
        }
        HIPCHK(hipMemsetAsync(m->d_n_edges, 0, 4, m->stream));
        hipLaunchKernelGGL(cddt_edges_kernel, dim3((cols + 255) / 256, (rows + EDGE_ROWS_PER_WG - 1) / EDGE_ROWS_PER_WG),
                           dim3(256), 0, m->stream,
                           m->d_occ, rows, cols, m->d_n_edges, m->d_edges);
        HIPCHK(hipMemcpyAsync(m->pin_n_edges, m->d_n_edges, 4, hipMemcpyDeviceToHost, m->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->want_edges) m->n_edges = *m->pin_n_edges;
    return RL_OK;
}

extern "C" int rl_map_create(const uint8_t *occ, int rows, int cols, float res, float ox,
                             float oy, float oyaw, int device, rl_map **out)
{
    if (!occ || !out) return fail(RL_ERR_INVALID, "rl_map_create: null pointer");
    if (rows <= 0 || cols <= 0 || rows > 16384 || cols > 16384)
        return fail(RL_ERR_INVALID, "rl_map_create: rows/cols must be in [1,16384] (got %dx%d)",
                    rows, cols);
    if (!(res > 0.0f)) return fail(RL_ERR_INVALID, "rl_map_create: resolution must be > 0");
    int ndev = rl_device_count();
    if (ndev <= 0)
        return fail(RL_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(RL_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    rl_map *m = new (std::nothrow) rl_map();
    if (!m) return fail(RL_ERR_NOMEM, "out of host memory");
    m->device = device;
    m->rows = rows;
    m->cols = cols;
    m->res = res;
    m->ox = ox;
    m->oy = oy;
    m->oyaw = oyaw;
    m->bits_stride = (cols + 31) / 32;
    auto bail = [&](int code) {
        rl_map_destroy(m);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(RL_ERR_HIP, "hipSetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        m->n_cu = prop.multiProcessorCount;
        m->clock_khz = prop.clockRate;
    }
    const size_t n = (size_t)rows * cols;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&m->d_occ, n) != hipSuccess ||
        hipMalloc((void **)&m->d_g, n * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&m->d_dt, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&m->d_bits, (size_t)rows * m->bits_stride * sizeof(uint32_t)) !=
            hipSuccess)
        return bail(fail(RL_ERR_NOMEM, "device allocation for a %dx%d map failed", rows, cols));
    if (hipMemcpyAsync(m->d_occ, occ, n, hipMemcpyHostToDevice, m->stream) != hipSuccess)
        return bail(fail(RL_ERR_HIP, "map upload failed"));
    int rc = map_build_tables(m);
    if (rc != RL_OK) return bail(rc);

    MapParams &p = m->mp;
    p.dt = m->d_dt;
    p.bits = m->d_bits;
    p.bits_stride = m->bits_stride;
    p.rows = rows;
    p.cols = cols;
    p.frows = (float)rows;
    p.fcols = (float)cols;
    p.res = res;
    p.inv_res = (float)(1.0 / (double)res);
    p.ox = ox;
    p.oy = oy;
    p.wa = -oyaw;                                   // PyOMap: world_angle = -yaw
    host_sincosf(p.wa, p.wa_sin, p.wa_cos);
    if (hipMalloc((void **)&m->d_mp, sizeof(MapParams)) != hipSuccess ||
        hipMemcpy(m->d_mp, &m->mp, sizeof(MapParams), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(RL_ERR_NOMEM, "map parameter upload failed"));
    *out = m;
    return RL_OK;
}

extern "C" int rl_map_create_multi(const uint8_t *occ, int rows, int cols, float res, float ox, float oy,
                                   float oyaw, const int *devices, int n_devices, rl_map **out)
{
    if (!occ || !out || !devices) return fail(RL_ERR_INVALID, "rl_map_create_multi: null pointer");
    if (n_devices < 1 || n_devices > 64) return fail(RL_ERR_INVALID, "rl_map_create_multi: 1..64 devices (got %d)", n_devices);
    rl_map *m = new (std::nothrow) rl_map();
    if (!m) return fail(RL_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < n_devices; ++i) {
        rl_map *r = nullptr;
        const int rc = rl_map_create(occ, rows, cols, res, ox, oy, oyaw, devices[i], &r);
        if (rc) {
            const std::string keep = g_err;
            rl_map_destroy(m);
            g_err = keep;
            return rc;
        }
        m->reps.push_back(r);
    }
    const rl_map *r0 = m->reps[0];
    m->device = r0->device;
    m->rows = rows;
    m->cols = cols;
    m->res = res;
    m->ox = ox;
    m->oy = oy;
    m->oyaw = oyaw;
    m->n_cu = r0->n_cu;
    m->clock_khz = r0->clock_khz;
    m->mp = r0->mp;
    *out = m;
    return RL_OK;
}

extern "C" int rl_map_n_devices(const rl_map *m) { return m ? (m->reps.empty() ? 1 : (int)m->reps.size()) : 0; }

extern "C" rl_map *rl_map_replica(rl_map *m, int i)
{
    if (!m) return nullptr;
    if (m->reps.empty()) return i == 0 ? m : nullptr;
    return (i >= 0 && i < (int)m->reps.size()) ? m->reps[i] : nullptr;
}

extern "C" int rl_map_update(rl_map *m, const uint8_t *occ)
{
    if (!m || !occ) return fail(RL_ERR_INVALID, "rl_map_update: null pointer");
    if (!m->reps.empty()) {
        std::lock_guard<std::mutex> lk(m->mu);
        // exclusive against every multi_* call in progress: a batch sees ONE occupancy on all of its devices
        std::unique_lock<std::shared_mutex> wl(m->multi_mu);
        if (m->broken.load()) return fail(RL_ERR_INVALID, "multi-device map is inconsistent after a failed update: destroy it");
        for (size_t i = 0; i < m->reps.size(); ++i) {
            const int rc = rl_map_update(m->reps[i], occ);
            if (rc) {
                // replicas [0, i) hold the new cells, the others the old ones: no roll-back (the old cells are
                // gone from the host) — the handle is marked and refuses further scans
                if (i > 0) {
                    m->broken.store(true);
                    const std::string keep = g_err;
                    return fail(rc, "rl_map_update failed on replica %zu of %zu after %zu replica(s) had been updated — "
                                    "the multi-device map is now invalid: %s", i, m->reps.size(), i, keep.c_str());
                }
                return rc;
            }
        }
        m->epoch++;
        return RL_OK;
    }
    std::lock_guard<std::mutex> lk(m->mu);
    // exclusive: no host-pointer call of any method of this map is in progress; the device
    // synchronisation covers launches the asynchronous *_device entry points left in flight
    std::unique_lock<std::shared_mutex> wl(m->tables_mu);
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyAsync(m->d_occ, occ, (size_t)m->rows * m->cols, hipMemcpyHostToDevice,
                          m->stream));
    rc = map_build_tables(m);
    if (rc) return rc;
    m->epoch++;
    return RL_OK;
}

extern "C" void rl_map_destroy(rl_map *m)
{
    if (!m) return;
    if (!m->reps.empty()) {
        for (rl_map *r : m->reps) rl_map_destroy(r);
        delete m;
        return;
    }
    (void)hipSetDevice(m->device);
    if (m->d_occ) (void)hipFree(m->d_occ);
    if (m->d_g) (void)hipFree(m->d_g);
    if (m->d_dt) (void)hipFree(m->d_dt);
    if (m->d_bits) (void)hipFree(m->d_bits);
    if (m->d_mp) (void)hipFree(m->d_mp);
    if (m->d_edges) (void)hipFree(m->d_edges);
    if (m->d_n_edges) (void)hipFree(m->d_n_edges);
    if (m->pin_n_edges) (void)hipHostFree(m->pin_n_edges);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int rl_map_rows(const rl_map *m) { return m ? m->rows : 0; }
extern "C" int rl_map_cols(const rl_map *m) { return m ? m->cols : 0; }
extern "C" int rl_map_device(const rl_map *m) { return m ? m->device : -1; }

extern "C" int rl_map_get_dt(rl_map *m, float *dt_out)
{
    if (!m || !dt_out) return fail(RL_ERR_INVALID, "rl_map_get_dt: null pointer");
    if (!m->reps.empty()) return rl_map_get_dt(m->reps[0], dt_out);
    std::lock_guard<std::mutex> lk(m->mu);
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(hipMemcpy(dt_out, m->d_dt, (size_t)m->rows * m->cols * sizeof(float),
                     hipMemcpyDeviceToHost));
    return RL_OK;
}

extern "C" int rl_map_get_occ(rl_map *m, uint8_t *occ_out)
{
    if (!m || !occ_out) return fail(RL_ERR_INVALID, "rl_map_get_occ: null pointer");
    if (!m->reps.empty()) return rl_map_get_occ(m->reps[0], occ_out);
    std::lock_guard<std::mutex> lk(m->mu);
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(hipMemcpy(occ_out, m->d_occ, (size_t)m->rows * m->cols, hipMemcpyDeviceToHost));
    return RL_OK;
}

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
    if (!m->reps.empty()) {
        // multi-device: one ordinary method per device replica of the map + one worker thread per extra device
        std::vector<int> devs;
        for (rl_map *rm : m->reps) {
            rl_method *r = nullptr;
            const int rc = rl_method_create(rm, kind, max_range_px, theta_disc, &r);
            if (rc) {
                const std::string keep = g_err;
                rl_method_destroy(h);
                g_err = keep;
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
static int multi_parts(const rl_method *h, long n_poses)
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
    if (!strcmp(name, "variant")) h->variant = value;
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
    else if (!strcmp(name, "inline_prep")) h->inline_prep = value != 0;
    else if (!strcmp(name, "bin_generic")) h->bin_generic = value != 0;
    else if (!strcmp(name, "tiled")) h->tiled = value != 0;
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
    else if (!strcmp(name, "cddt_search")) h->cddt_search = value != 0;
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
    else if (!strcmp(name, "inline_prep")) *value_out = h->inline_prep;
    else if (!strcmp(name, "bin_generic")) *value_out = h->bin_generic;
    else if (!strcmp(name, "tiled")) *value_out = h->tiled;
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

static int check_fan_args(const rl_method *h, int n_poses, float fov, int num_rays)
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
    const int tiles_x = (m->cols >> shift) + 1;
    const int n_tiles = tiles_x * ((m->rows >> shift) + 1);
    if (binning == RL_BIN_GRID_SORT || binning == RL_BIN_GRID_UNSORTED) {
        const int ppw = h->bin_ppw;
        const int n_wg = (n_poses + ppw - 1) / ppw;
        if (do_sort) {
            // grid-wide binning on coarse tiles (<= 1024): per-workgroup LDS histograms ->
            // one scan over (tile, workgroup) -> scatter from LDS cursors
            int cshift = shift;
            while ((long)((m->cols >> cshift) + 1) * ((m->rows >> cshift) + 1) > 1024) ++cshift;
            const int ctx = (m->cols >> cshift) + 1;
            const int cnt = ctx * ((m->rows >> cshift) + 1);
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
    if (h->pdt_epoch == m->epoch && h->pdt.p && h->pdt_tiled == want_tiled) return table_wait(h->pdt_dep, stream);
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
    h->pdt_epoch = m->epoch;
    h->pdt_tiled = want_tiled;
    return table_built(h->pdt_dep, stream);
}

// the stream-kernel instantiation a plan names
template <bool A, bool C, int N, bool I, bool T, int S, bool L = false>
static void launch_rm_stream(const rl_launch_plan &pl, hipStream_t stream, const PadMap &pm, const FanParams &f,
                             const StreamParams &sp, float *d_out, int32_t *d_hits, uint16_t *d_steps,
                             const CrashParams &cp)
{
    // (more dynamic LDS than HIP's default cap — fans of several thousand beams, with the crash table —: opt in,
    //  as the BL / occ / CDDT kernels do; the attribute is sticky per function and device, the call is cheap)
    if (pl.lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rm_fan_stream_kernel<A, C, N, I, T, S, L>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, pl.lds_bytes);
    hipLaunchKernelGGL((rm_fan_stream_kernel<A, C, N, I, T, S, L>), dim3(pl.grid), dim3(N), (size_t)pl.lds_bytes, stream,
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
        if (pl.slots >= 2) { if (crash) launch_rm_stream<false, true, 1024, true, true, 2, true>(RM_ARGS);
                             else launch_rm_stream<false, false, 1024, true, true, 2, true>(RM_ARGS); }
        else               { if (crash) launch_rm_stream<false, true, 1024, true, true, 1, true>(RM_ARGS);
                             else launch_rm_stream<false, false, 1024, true, true, 1, true>(RM_ARGS); }
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
    const size_t prep_bytes = (((size_t)n_poses * 16) + 255) & ~(size_t)255;
    if ((rc = cx->cddt_r.ensure(prep_bytes + (size_t)h->cdp.theta_disc * (size_t)n_poses * sizeof(float)))) return rc;
    float4 *d_prep = (float4 *)cx->cddt_r.p;
    float *d_r = (float *)((char *)cx->cddt_r.p + prep_bytes);
    if (h->timing == 2) HIPCHK(hipEventRecord(h->ev0, stream));
    hipLaunchKernelGGL(cddt_theta_prep_kernel, dim3((unsigned)std::max(1, std::min((n_poses + 255) / 256, m->n_cu * 8))),
                       dim3(256), 0, stream, m->mp, f, h->cdp, d_poses, d_prep);
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
    StreamParams sp{};
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
                         pl.kernel == RL_K_RM_STREAM;      // (the leftover kernel marches the canonical arithmetic)
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
    int rc = plan_for(h, n_poses, num_rays, aux, crash != nullptr, &pl);
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
    const int rc = plan_for(h, n_poses, num_rays, want_aux != 0, want_crash != 0, out);
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
static int upload_edge(rl_method *h, const double *edge, int num_rays)
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

static int fan_host(rl_method *h, const float *poses, int n_poses, float fov, int num_rays,
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
static int crash_groups_device(rl_method *h, const float *d_poses, int n_groups, int group, float fov,
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
            const std::string keep = g_err;
            rl_car_destroy(c);
            g_err = keep;
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


// ---------------------------------------------------------------- FollowGap
struct rl_followgap {
    int device = 0;
    FollowGapParams P{};
    int window_size = 0;           // kept for the caller; FollowGap::eval never reads it
    int n_cu = 256;                // (queried once: hipGetDeviceProperties costs the host tens of microseconds per call)
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
