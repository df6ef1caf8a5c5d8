// launch_plan.h — which kernel, grid, LDS size and pose-binning pass a fan call takes.
//
// ONE pure function of (range method, CU count, map shape, options, batch shape): no HIP call, no
// handle state, so it is tested on a box without a GPU (tests/test_host.py, through rl_plan_fan)
// for every BASELINE.json configuration and for the reference's own 200-pose roll-out batch
// (/root/reference/params.yaml:126, scripts/mcts.py:214-237).  launch_fan (abi_fan.hip) executes
// exactly the plan this returns; the thresholds in here are measured optima (profiles/r03/
// plan_sweep*.txt), not derived constants.
#pragma once
#include "../../include/scanlib.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace plan {

// layout constants shared with rm_kernels.h (static_asserts in abi_fan.hip tie them together)
constexpr int WG = 256;                  // threads of the unit workgroup grid_mult counts in
constexpr int STREAM_HDR = 66;           // LDS header words of the stream kernels
constexpr int STRIPE_BINS = 64;
constexpr int STRIPE_MAX_PER_LANE = 8;
constexpr int INLINE_LDS_BUDGET = 72 * 1024;   // LDS an INLINE workgroup may use (two 1024-lane workgroups per CU)
constexpr int INLINE_REC_BYTES = 32;     // one BlockRec per owned 64-ray block (rm_kernels.h)
constexpr int LIT_SLICE_POSES = 4096;    // upstream-literal stream form: poses per slice of a batch too large for one INLINE launch
constexpr int CODE_MAX_ENTRIES = 4096;   // palette entries (16 KB of LDS) a code-map launch may carry; u16 codes hold 4 * index
constexpr int DRAIN_CAP = 64, DRAIN_FIELDS = 7;   // several rays per lane: per-wave compaction scratch of the drain phase
                                                  // (7 dwords per ray, 9 with the fused crash test)

struct In {
    int kind = RL_RM_GPU, n_cu = 256, rows = 0, cols = 0, theta_disc = 0;
    float max_range = 300.0f;
    rl_plan_opts o{};
    int n_poses = 0, num_rays = 0;
    bool aux = false, crash = false;
};

// The arithmetic a range method runs by default.  RL_RM is range_libc's CPU RayMarching (scripts/scan_simulator.py:72-73,
// simple_params.yaml:114): it computes what numpy_calc_range + RayMarching::calc_range compute — per-ray float32 angle,
// glibc sinf / cosf, un-fused march — i.e. variant 3, the upstream-literal statement (the checker's rm_fan_libm).  RL_RM_GPU
// stands for kernels.cu, whose device trig cannot be known here: the canonical arithmetic (variant 1).  Everything
// else has one form.  Option "variant" overrides either way.
inline int default_variant(int kind) { return kind == RL_RM ? 3 : 1; }

inline void default_opts(rl_plan_opts &o)
{
    std::memset(&o, 0, sizeof o);
    o.variant = -1;           // the default of the kind (default_variant): resolved by plan_fan
    o.grid_mult = 8;          // workgroups (x256 threads) per CU of a persistent launch
    o.wg_threads = 1024;
    o.low_water = -1;                          // (auto: abi_fan.hip)
    o.sort_poses = 1;
    o.xcd_bands = 8;
    o.slots = 0;              // auto
    o.tiled = 1;
    o.inline_prep = 1;
    o.inline_max = 512;
    o.inline_map_kb = 2048;
    o.stripe_max = 1536;      // (profiles/r03/plan_sweep.txt: row stripes tie with the keys-only binning launch at 1024
                              //  poses and lose 8..20 % at 2048..2560 on both big maps)
    o.order_inline = 1;
    o.bin_multi_min = 8192;
    o.bin_generic = 0;
    o.run_log2 = -1;
    o.cddt_bins = 1;
    o.cddt_sort = 0;
    o.cddt_theta_min = 32768;
    o.cddt_search = 1;
    o.lut_debug = 0;
    o.debug_stamps = 0;
    o.slice_log2 = 30;
    o.code_map = 2;           // u16 palette codes wherever the palette fits (profiles/r06/ab_code_map.txt)
    o.code_min_rays = 1 << 22;   // (cfg2's 4096 x 1081 and up)
    o.code_entries = 0;       // (a handle fills in its map's palette size)
    o.tail_pct = 0;
    o.tail_wg_pct = 50;
}

// Options as the planner may use them: every field a division, a shift or a template choice depends on is
// brought into its valid range — the same clamps rl_method_set_option applies to a handle, so a caller of the
// public rl_plan_fan with a zeroed or hand-filled rl_plan_opts gets a plan (of the nearest valid options)
// instead of a division by zero, an undefined shift or a block size no kernel was instantiated for.
inline rl_plan_opts sanitized(rl_plan_opts o)
{
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    o.variant = o.variant < 0 ? -1 : clampi(o.variant, 0, 3);
    o.grid_mult = clampi(o.grid_mult, 1, 64);
    o.wg_threads = o.wg_threads >= 1024 ? 1024 : (o.wg_threads >= 512 ? 512 : 256);
    o.low_water = o.low_water < 0 ? -1 : clampi(o.low_water, 0, 63);
    o.sort_poses = o.sort_poses != 0;
    o.xcd_bands = clampi(o.xcd_bands, 1, 64);
    o.slots = clampi(o.slots, 0, 3);
    o.tiled = o.tiled != 0;
    o.inline_prep = o.inline_prep != 0;
    o.inline_max = clampi(o.inline_max, 0, 1 << 30);
    o.inline_map_kb = clampi(o.inline_map_kb, 0, 1 << 30);
    o.stripe_max = clampi(o.stripe_max, 0, 1 << 30);
    o.order_inline = o.order_inline != 0;
    o.bin_multi_min = clampi(o.bin_multi_min, 1, 1 << 30);
    o.bin_generic = o.bin_generic != 0;
    o.run_log2 = o.run_log2 < 0 ? -1 : clampi(o.run_log2, 0, 8);
    o.cddt_bins = o.cddt_bins != 0;
    o.cddt_sort = o.cddt_sort != 0;
    o.cddt_theta_min = clampi(o.cddt_theta_min, 0, 1 << 30);
    o.cddt_search = clampi(o.cddt_search, 0, 2);
    o.slice_log2 = clampi(o.slice_log2, 8, 30);
    o.code_map = o.code_map == 2 ? 2 : 0;          // (u16 codes; 1 = u8 codes is not instantiated)
    o.code_entries = clampi(o.code_entries, 0, 1 << 30);
    o.code_min_rays = clampi(o.code_min_rays, 0, 1 << 30);
    o.tail_wg_pct = clampi(o.tail_wg_pct, 10, 400);
    o.tail_pct = clampi(o.tail_pct, 0, std::min(o.tail_wg_pct, 75));   // (a second-generation workgroup never owns more than a first-generation one)
    return o;
}

// The power-of-two-pitch tiled step map (pad_dt_tiled_kernel) of a rows x cols map with a border for max_range:
// whether its geometry fits the march's address arithmetic — the table below 4 GiB (32-bit byte offsets) and
// K <= 24, because the row term comes from v_mad_i32_i24 with M = 4 + 2^(K-2) (a signed 24-bit operand) — and
// the geometry itself.  Shared by the planner (which falls back to the row-major map: tiled = 0, one ray per
// lane) and ensure_step_map (which builds what the planner said).
struct TiledFit {
    bool ok;
    int pad, padr, pcols, prows, K;
    size_t bytes;
};
inline TiledFit tiled_fit(int rows, int cols, float max_range)
{
    TiledFit t{};
    t.pad = (((int)std::ceil(max_range) + 2) + 7) & ~7;      // 128-B lines line up with the border
    t.padr = t.pad + 4;                                       // one slack group in front: offsets stay positive
    t.pcols = cols + 2 * t.pad;
    t.prows = (rows + 2 * t.pad + 4 + 3) & ~3;
    int lg = 3;                                               // power-of-two pitch >= padded cols and padded rows
    while ((1L << lg) < (long)std::max(t.pcols, t.prows) && lg < 30) ++lg;
    t.K = lg + 4;
    t.bytes = ((size_t)(t.prows >> 2)) << t.K;
    t.ok = t.K <= 24 && t.bytes <= ((size_t)1 << 32) && max_range < 1.0e6f;
    return t;
}

// The tiled CODE map (pad_code_tiled_kernel: groups of 8 rows, 2^es bytes per cell — es 1: u16, 0: u8): the same
// checks for its address arithmetic — M = 2^es + 2^(K-3) a signed 24-bit operand, r' * M below 2^31, the table
// below 4 GiB.  nb: slots of the palette's d^2 histogram — every step below max_range comes from an EDT value
// below max_range / coeff, coeff >= 0.999.
struct CodeFit {
    bool ok;
    int pad, padr, pcols, prows, K, es;
    size_t bytes;
    unsigned nb;
};
inline CodeFit code_fit(int rows, int cols, float max_range, int es)
{
    CodeFit t{};
    t.es = es;
    t.pad = (((int)std::ceil(max_range) + 2) + 15) & ~15;
    t.padr = t.pad + 8;
    t.pcols = cols + 2 * t.pad;
    t.prows = (rows + 2 * t.pad + 8 + 7) & ~7;
    int lg = 3;
    while ((1L << lg) < (long)std::max(t.pcols, t.prows) && lg < 30) ++lg;
    t.K = lg + 3 + es;
    t.bytes = ((size_t)(t.prows >> 3)) << t.K;
    const double lim = ((double)max_range / 0.999) * ((double)max_range / 0.999) * 1.001 + 4.0;
    t.nb = lim < 2097152.0 ? (unsigned)lim : 0u;
    t.ok = t.K <= 25 && 2 * lg + es + 1 < 31 && t.bytes <= ((size_t)1 << 32) && t.nb != 0u;
    return t;
}

// device limit a launch's dynamic LDS must stay within (gfx950: 160 KB per workgroup)
constexpr int DEVICE_LDS_BYTES = 160 * 1024;

// the binning pass a batch of n_poses takes when one is needed (bin_poses in abi_fan.hip)
inline bool keys_only_ok(const rl_plan_opts &o, int n_poses)
{
    return o.sort_poses && n_poses >= 64 && n_poses < o.bin_multi_min && n_poses <= 8192 && !o.bin_generic;
}

inline int binning_for(const rl_plan_opts &o, int n_poses, bool keys_only)
{
    const bool do_sort = o.sort_poses && n_poses >= 64;
    if (n_poses >= o.bin_multi_min) return do_sort ? RL_BIN_GRID_SORT : RL_BIN_GRID_UNSORTED;
    if (do_sort && n_poses <= 8192 && !o.bin_generic) return keys_only ? RL_BIN_SMALL_KEYS : RL_BIN_SMALL_RECORDS;
    return RL_BIN_GENERIC;
}

// Bresenham / occupancy window in LDS (make_bl in abi_fan.hip)
inline void bl_window(float max_range, int num_rays, int &R, int &ww, bool &use_lds, size_t &lds_bytes)
{
    R = (int)std::ceil(max_range) + 5;
    ww = ((2 * R + 32 + 31) / 32) | 1;
    const size_t win = (size_t)(2 * R + 1) * ww * sizeof(uint32_t);
    const size_t fan = (size_t)num_rays * 8;
    use_lds = (win + fan) <= 150 * 1024;
    lds_bytes = fan + (use_lds ? win : 0);
}

inline const char *tf(bool b) { return b ? "true" : "false"; }

inline int plan_one(const In &in, rl_launch_plan *p)
{
    const rl_plan_opts &o = in.o;
    const int n_poses = in.n_poses, num_rays = in.num_rays, n_cu = in.n_cu;
    const long rays = (long)n_poses * num_rays;
    p->aux = in.aux;
    p->crash = in.crash;
    p->slots = 1;
    p->bands = 1;
    p->slices = 1;
    p->slice_poses = n_poses;
    // the tiled step map only where its geometry fits the march's address arithmetic (tiled_fit): very elongated or
    // huge maps march on the row-major copy
    const bool tiled_opt = o.tiled != 0 && ((in.kind != RL_RM && in.kind != RL_RM_GPU) ||
                                           tiled_fit(in.rows, in.cols, in.max_range).ok);
    p->tiled = tiled_opt;
    if (in.kind == RL_GIANT_LUT) {
        // (two generations of workgroups: the second fills in behind the first's ragged end — a lone cfg3 launch
        //  0.0843 -> 0.0786 ms, 840 -> 901 Grays/s; a third generation adds nothing: profiles/r04/lut_grid_sweep.txt)
        p->grid = (int)std::max(1L, std::min(((long)n_poses + 3) / 4, (long)n_cu * o.grid_mult * 2));
        p->block = 256;
        const int td = in.theta_disc;
        const int nl = (td / 2 + 255) / 256;                 // 16-B loads per lane for one row
        const bool lds_ok = (td % 2 == 0) && nl <= 3 && num_rays <= 17 * 64 && !(o.lut_debug & 4);
        p->ch = num_rays <= 12 * 64 ? 12 : 17;
        if (lds_ok) {
            p->kernel = RL_K_LUT_LDS;
            p->nl = nl;
            p->lds_bytes = 4 * nl * 256 * (int)sizeof(uint32_t);
            std::snprintf(p->name, sizeof p->name, "scan::lut_fan_lds_kernel<%d, %d>", nl, p->ch);
        } else {
            p->kernel = RL_K_LUT_FAN;
            std::snprintf(p->name, sizeof p->name, "scan::lut_fan_kernel<%d>", p->ch);
        }
        return RL_OK;
    }
    if (in.kind == RL_CDDT) {
        if (o.cddt_bins && in.theta_disc <= num_rays && in.theta_disc <= 4096 && o.cddt_theta_min > 0 &&
            n_poses >= o.cddt_theta_min && (long)n_poses * in.theta_disc <= (1L << 29)) {   // (R[bin][pose]: <= 2 GiB)
            // theta-major: all poses against one table bin at a time, bins pinned to XCDs; the fan kernel takes
            // 2^ch poses per pass (theta_disc x poses floats of LDS, at most 16 K)
            int ppb_log2 = 5;
            while (ppb_log2 > 0 && ((in.theta_disc | 1) << ppb_log2) > 16384) --ppb_log2;
            p->kernel = RL_K_CDDT_THETA;
            p->block = 256;
            // (eight generations of workgroups: the units are short and of uneven cost — bucket sizes —, later
            //  generations level the end: cfg3 341 -> 364 Grays/s serial, ~350 -> 382-405 with four launches in
            //  flight; 4 and 16 generations are 2-4 % behind: profiles/r04/cddt_grid_sweep.txt)
            // (a unit of work = one table bin x 128 poses; 256 poses with the round-5 search kernel, whose waves take 64 each)
            if (o.cddt_search == 2 && (((in.theta_disc | 1) + 1) * 64 * (int)sizeof(float)) <= 65536) {
                // search + fan fused: a workgroup owns a tile of 64 poses, its per-bin results live in LDS only
                // ((theta_disc | 1) + 1) x 64 floats: 29 KB at theta_disc 112 -> five tiles per CU)
                p->ch = 6;
                p->nl = in.theta_disc | 1;
                p->lds_bytes = (p->nl + 1) * 64 * (int)sizeof(float);
                const int per_cu = std::max(1, std::min(8, (160 * 1024) / std::max(p->lds_bytes, 1)));
                p->grid = (int)std::max(1L, std::min((long)(n_poses + 63) / 64, (long)n_cu * per_cu));
                p->bands = 1;
                std::snprintf(p->name, sizeof p->name, "scan::cddt_theta_fused_kernel");
                return RL_OK;
            }
            const int unit_poses = o.cddt_search ? 256 : 128;
            p->grid = (int)std::max(1L, std::min((long)((in.theta_disc + 1) / 2) * ((n_poses + unit_poses - 1) / unit_poses), (long)n_cu * 64));
            p->bands = (p->grid >= o.xcd_bands && (in.theta_disc + 1) / 2 >= o.xcd_bands) ? std::max(o.xcd_bands, 1) : 1;
            p->ch = ppb_log2;
            p->nl = in.theta_disc | 1;                 // LDS row stride of the fan kernel
            p->lds_bytes = ((p->nl + 1) << ppb_log2) * (int)sizeof(float);      // R rows of ppb poses + their headings
            std::snprintf(p->name, sizeof p->name, o.cddt_search ? "scan::cddt_theta_search2_kernel" : "scan::cddt_theta_search_kernel");
            return RL_OK;
        }
        if (o.cddt_bins && in.theta_disc <= num_rays && in.theta_disc <= 8192) {
            // one lane per (pose, TABLE bin) — a table bin answers both raw bins half a turn apart —, pp poses
            // per workgroup pass; the grid keeps every CU's 2048 lanes occupied
            const int n_tb = (in.theta_disc + 1) / 2;
            const int bnt = n_tb <= 256 ? 256 : 1024;
            const int lpp = std::min(bnt, ((n_tb + 63) / 64) * 64);
            const int pp = bnt / lpp;
            p->kernel = RL_K_CDDT_BINS;
            p->block = bnt;
            p->nl = lpp;                               // lanes per pose
            p->ch = pp;                                // poses per workgroup pass
            p->grid = (int)std::max(1L, std::min(((long)n_poses + pp - 1) / pp, (long)n_cu * (2048 / bnt)));
            p->lds_bytes = pp * in.theta_disc * (int)sizeof(float);
            if (o.sort_poses && o.cddt_sort && n_poses >= 512 && p->grid >= o.xcd_bands) {
                p->binning = binning_for(o, n_poses, keys_only_ok(o, n_poses));
                p->bands = o.xcd_bands;
            }
            std::snprintf(p->name, sizeof p->name, "scan::cddt_fan_bins_kernel");
        } else {
            p->kernel = RL_K_CDDT_RAYS;
            p->block = 256;
            p->grid = (int)std::max(1L, std::min((long)n_poses, (long)n_cu * o.grid_mult));
            std::snprintf(p->name, sizeof p->name, "scan::cddt_fan_kernel");
        }
        return RL_OK;
    }
    if (in.kind == RL_BRESENHAM) {
        if (o.variant >= 1 && rays < (1L << 30)) {
            // K2b: stream schedule on the cache-resident padded bit maps
            p->kernel = RL_K_BL_STREAM;
            p->binning = binning_for(o, n_poses, false);
            p->bands = n_poses >= 64 ? o.xcd_bands : 1;
            const long n_blocks = (rays + 63) / 64;
            p->grid = (int)std::max((long)p->bands, std::min((n_blocks + 15) / 16, (long)n_cu * o.grid_mult / 4));
            p->block = 1024;
            p->lds_bytes = num_rays * 8 + 8;
            std::snprintf(p->name, sizeof p->name, "scan::bl_fan_stream_kernel<%s, 1024>", tf(in.aux));
        } else {
            int R, ww;
            bool use_lds;
            size_t lds;
            bl_window(in.max_range, num_rays, R, ww, use_lds, lds);
            p->kernel = RL_K_BL_LDS;
            p->grid = (int)std::max(1L, std::min((long)n_poses, (long)n_cu * 2));
            p->block = 256;
            p->lds_bytes = (int)lds;
            std::snprintf(p->name, sizeof p->name, "scan::bl_fan_kernel<%s>", tf(in.aux));
        }
        return RL_OK;
    }
    // ---- ray marching (RL_RM / RL_RM_GPU)
    const long cpp = (num_rays + 63) / 64;
    const long n_chunks = (o.variant >= 1) ? (rays + 63) / 64 : (long)n_poses * cpp;
    const bool stream_ok = rays < (1L << 30);
    if (o.variant == 3) {
        // upstream-literal arithmetic.  Production form: the stream kernel's schedule with per-ray libm directions at
        // claim time (template argument LIT) — whenever the records can be derived in LDS (INLINE: every batch of up to
        // 8192 poses and fans of >= 64 beams; plan_fan cuts larger batches into pose slices) on the tiled step map and no
        // diagnostics are asked for.  Everything else: the one-lane-per-ray kernel (literal_kernels.h).
        if (!in.aux && tiled_opt) {
            In as_stream = in;
            as_stream.o.variant = 1;
            if (as_stream.o.slots == 3) as_stream.o.slots = 2;
            rl_launch_plan q;
            std::memset(&q, 0, sizeof q);
            if (plan_one(as_stream, &q) == RL_OK && q.kernel == RL_K_RM_STREAM && q.record_source != 0 && q.tiled &&
                q.block == 1024 && q.slots <= 2) {
                *p = q;
                p->kernel = RL_K_RM_STREAM_LIT;
                std::snprintf(p->name, sizeof p->name, "scan::rm_fan_stream_kernel<false, %s, 1024, true, true, %d, true, %d>",
                              tf(in.crash), q.slots, q.code);
                return RL_OK;
            }
        }
        if (in.crash) return RL_ERR_UNSUPPORTED;
        p->kernel = RL_K_RM_LITERAL;
        p->grid = (int)std::max(1L, std::min((rays + WG - 1) / WG, (long)n_cu * o.grid_mult));
        p->block = WG;
        p->tiled = 0;
        std::snprintf(p->name, sizeof p->name, "scan::rm_literal_kernel<%s, false>", tf(in.aux));
        return RL_OK;
    }
    if (o.variant == 2) {
        int R, ww;
        bool use_lds;
        size_t lds;
        bl_window(in.max_range, num_rays, R, ww, use_lds, lds);
        if (in.crash || !use_lds) return RL_ERR_UNSUPPORTED;
        p->kernel = RL_K_OCC_LDS;
        p->grid = (int)std::max(1L, std::min((long)n_poses, (long)n_cu * 2));
        p->block = 256;
        p->lds_bytes = (int)lds;
        std::snprintf(p->name, sizeof p->name, "scan::occ_fan_lds_kernel<%s>", tf(in.aux));
        return RL_OK;
    }
    if (!(o.variant >= 1 && stream_ok)) {
        p->kernel = RL_K_RM_CHUNK;
        p->grid = (int)std::max(1L, std::min((n_chunks + 3) / 4, (long)n_cu * o.grid_mult));
        p->block = WG;
        p->lds_bytes = num_rays * 8;
        std::snprintf(p->name, sizeof p->name, "scan::rm_fan_kernel<%s, %s>", tf(in.aux), tf(in.crash));
        return RL_OK;
    }
    // K1b.  (1) where the pose records come from, (2) the persistent grid, (3) rays per lane
    const int bands = n_poses >= 64 ? o.xcd_bands : 1;
    int nt = o.wg_threads;
    // small batches: no binning launch, workgroups derive the records of their own blocks ... and
    // whenever the map sits in every XCD's L2: tile order buys nothing there
    const bool small_map = (size_t)in.rows * in.cols * sizeof(float) <= (size_t)o.inline_map_kb * 1024;
    bool inl = o.inline_prep && num_rays >= 64 && (small_map || (n_poses < o.inline_max && n_poses < o.bin_multi_min));
    // big maps, mid-size batches: every workgroup compacts the poses of its own band (a row stripe
    // of the map) from the caller's list
    const bool stripe = o.inline_prep && num_rays >= 64 && !inl && bands > 1 && o.sort_poses && n_poses >= 64 &&
                        n_poses <= std::min(o.stripe_max, 1024 * STRIPE_MAX_PER_LANE);
    if (stripe) inl = true;
    // big maps, up to 8192 poses: keys-only binning launch + INLINE march that takes its pose ids
    // from the tile order
    const bool order_inl = o.inline_prep && o.order_inline && num_rays >= 64 && !inl && bands > 1 &&
                           keys_only_ok(o, n_poses);
    if (order_inl) inl = true;
    int k_max = 0, inl_rl = 0;
    size_t lds_extra = 0;
    // INLINE launches cut every pose's fan into cpp blocks of 64 rays (the last one partly filled): a block
    // never straddles two poses, so ONE record per block serves every ray slot of it
    const long n_blocks_inl = (long)n_poses * cpp;
    // several rays per lane (decided from the ray count before the record source, which needs the LDS size):
    // every wave of a workgroup gets DRAIN_FIELDS x DRAIN_CAP dwords of compaction scratch behind the tables
    auto tables_bytes = [&](size_t tabw) {
        return (((size_t)STREAM_HDR + tabw + (in.crash ? 4 : 2) * (size_t)num_rays + 7) & ~(size_t)7) * sizeof(float);
    };
    size_t tables_b = tables_bytes(0);
    const size_t drain_wave = (size_t)(DRAIN_FIELDS + (in.crash ? 2 : 0)) * DRAIN_CAP * 4;
    // auto: two rays per lane from 2^23 rays up, and from 2^20 on maps beyond the small-map bound (long rays:
    // +6 % on a lone 2049^2 launch of 1024 ... 16 384 poses since dry waves compact their last rays; colombia's
    // short rays lose 3 % — profiles/r03/sweep_serial_slots.txt)
    // (round 5, tools/r05/small_batch_sweep.py: since dry waves compact their last rays, two rays per lane win at EVERY
    //  batch size on maps beyond the small-map bound — one scan 13.5 -> 12.7 us, the reference's 200-pose roll-out
    //  20.5 -> 18.8 us, 1024 poses 26.4 -> 24.2 us on 2049^2; 200 poses 21.9 -> 19.4 us, 4096 poses 54 -> 48.6 us on
    //  4096^2 —, and still lose 3..8 % on colombia below 4096 poses)
    int slots_req = o.slots ? o.slots : ((rays >= (1L << 23) || !small_map) ? 2 : 1);
    // the code map (u16 palette codes + the palette in LDS): the two-rays-per-lane kernels of 1024 lanes on the tiled
    // layout, when the handle knows the palette fits (code_n)
    const int code_n = o.code_entries;
    int code = (o.code_map == 2 && code_n >= 2 && code_n <= CODE_MAX_ENTRIES && rays >= (long)o.code_min_rays && !in.aux && tiled_opt && slots_req == 2 &&
                o.wg_threads == 1024) ? 2 : 0;
    const size_t tabw = ((size_t)code_n + 1) & ~(size_t)1;
    if (code && tables_bytes(tabw) + 16 * drain_wave > (size_t)INLINE_LDS_BUDGET) code = 0;
    if (code) tables_b = tables_bytes(tabw);
    // (a fan whose tables leave no room for the scratch of 16 waves marches one ray per lane)
    if (slots_req >= 2 && (in.aux || !tiled_opt || tables_b + 16 * drain_wave > (size_t)INLINE_LDS_BUDGET)) slots_req = 1;
    const bool multi = slots_req >= 2;                                // <=> the launch takes 2 or 3 rays per lane
    auto drain_bytes = [&](int nthreads) { return multi ? (size_t)(nthreads / 64) * drain_wave : (size_t)0; };
    // (INLINE launches: 1024-lane workgroups; 512 with wg_threads = 512 — an A/B of round 5: half the waves wait for the
    //  slowest wave of their workgroup, twice the workgroups share a band's stream)
    const int inl_nt = (o.wg_threads == 512 && slots_req == 2 && !in.crash && !in.aux) ? 512 : 1024;
    const size_t inl_tables = tables_b + drain_bytes(inl_nt);
    if (inl) {
        nt = inl_nt;
        const long g_min = std::max(1L, std::min((n_blocks_inl + 15) / 16, (long)n_cu * o.grid_mult * WG / nt) / bands);
        const long seg_chunks_max = (((long)n_poses + bands - 1) / bands) * cpp;
        const long grid_i = std::max((long)bands, std::min((n_blocks_inl + 15) / 16, std::max((long)n_cu * o.grid_mult * WG / nt, 1L)));
        inl_rl = o.run_log2;
        if (inl_rl < 0) {
            inl_rl = 0;                               // (stripe batches are small: single blocks)
            if (!stripe)
                for (; inl_rl < 5 && ((n_blocks_inl / grid_i) >> (inl_rl + 1)) >= 16; ++inl_rl) {}
        }
        const long seg_runs_max = (seg_chunks_max + (1L << inl_rl) - 1) >> inl_rl;
        const long k_blocks = ((seg_runs_max + g_min - 1) / g_min) << inl_rl;
        k_max = (int)k_blocks + 1;
        if (stripe)                                   // band list + histogram / wave counts / cuts
            lds_extra = ((size_t)(n_poses + bands - 1) / bands + 2 + STRIPE_BINS + 3 * (nt / 64) + 4) * 4;
        if (inl_tables + (size_t)k_max * INLINE_REC_BYTES + lds_extra + 32 > (size_t)INLINE_LDS_BUDGET) inl = false;
    }
    if (!inl) {
        nt = o.wg_threads;
        p->binning = binning_for(o, n_poses, false);
        p->record_source = 0;
    } else if (order_inl) {
        p->binning = binning_for(o, n_poses, true);
        p->record_source = 3;
    } else {
        p->binning = RL_BIN_NONE;
        p->record_source = stripe ? 2 : 1;
    }
    const int waves_per_wg = nt / 64;
    const long n_blocks = inl ? n_blocks_inl : n_chunks;
    const long want_q = (n_blocks + waves_per_wg - 1) / waves_per_wg;
    const long cap_q = (long)n_cu * o.grid_mult * WG / nt;
    p->grid = (int)std::max((long)bands, std::min(want_q, std::max(cap_q, 1L)));
    // two generations: only launches that fill the machine (every resident slot taken, whole bands of workgroups) and have
    // work to split (>= 32 blocks per workgroup)
    if (o.tail_pct > 0 && bands > 1 && want_q >= cap_q && (long)p->grid == cap_q && p->grid % bands == 0 &&
        n_blocks / p->grid >= 32) {
        const int g1 = p->grid / bands, g2 = std::max(1, g1 * o.tail_wg_pct / 100);
        p->gen1 = p->grid;
        p->grid += g2 * bands;
    }
    p->block = nt;
    int rl2 = o.run_log2;
    if (rl2 < 0) {
        const long per_wg = n_blocks / std::max(p->grid, 1);
        for (rl2 = 0; rl2 < 5 && (per_wg >> (rl2 + 1)) >= 16; ++rl2) {}
    }
    p->run_log2 = inl ? inl_rl : rl2;          // (the inline LDS record table was sized for inl_rl)
    p->bands = bands;
    p->k_max = inl ? k_max : 0;
    const int slots = slots_req;
    bool a = in.aux, c = in.crash, t = tiled_opt;
    int s = 1;
    if (slots == 3 && !in.aux && !in.crash && tiled_opt && (inl || nt == 1024)) {
        s = 3;
        nt = 1024;
    } else if (slots >= 2 && !in.aux && tiled_opt) {
        s = 2;
    }
    p->kernel = RL_K_RM_STREAM;
    p->code = (s == 2 && t) ? code : 0;
    p->code_entries = p->code ? code_n : 0;
    p->slots = s;
    p->tiled = t;
    p->block = inl ? (s == 2 ? inl_nt : 1024) : nt;
    // (several rays per lane: the compaction scratch sits between the tables and the block records)
    p->lds_bytes = (int)(inl ? inl_tables + (size_t)k_max * INLINE_REC_BYTES + lds_extra
                             : (s >= 2 ? tables_b + drain_bytes(p->block)
                                       : ((size_t)STREAM_HDR + (in.crash ? 4 : 2) * (size_t)num_rays) * sizeof(float)));
    // (the label is the real symbol: all eight template arguments, as rocprofv3's kernel trace prints them)
    std::snprintf(p->name, sizeof p->name, "scan::rm_fan_stream_kernel<%s, %s, %d, %s, %s, %d, false, %d>", tf(a), tf(c),
                  p->block, tf(inl), tf(t), s, p->code);
    // (a fan of ~20 000 beams: the beam tables alone exceed a workgroup's LDS — say so instead of failing the launch)
    if (p->lds_bytes > DEVICE_LDS_BYTES) return RL_ERR_UNSUPPORTED;
    return RL_OK;
}

inline int plan_fan(const In &in_raw, rl_launch_plan *p)
{
    In in = in_raw;
    in.o = sanitized(in_raw.o);
    if (in.o.variant < 0) in.o.variant = default_variant(in.kind);
    std::memset(p, 0, sizeof *p);
    if (in.n_poses <= 0 || in.num_rays <= 0) {
        std::snprintf(p->name, sizeof p->name, "(nothing to launch)");
        return RL_OK;
    }
    // the stream kernels index rays with 32-bit byte offsets: batches of 2^30 rays or more go
    // through in pose slices, each its own launch sequence
    const long slice_rays = 1L << in.o.slice_log2;
    if (in.o.variant == 3 && !in.aux && in.n_poses > LIT_SLICE_POSES && in.num_rays >= 64 &&
        (in.kind == RL_RM || in.kind == RL_RM_GPU)) {
        // upstream-literal mode: the stream form derives its records in LDS — batches the planner cannot take that way
        // in one launch (beyond the keys-only binning's 8191 poses on a big map, or beyond the LDS budget) run as pose
        // slices of 4096, one launch sequence each (launch_fan; a fused crash test marks poses, so its slices only shift
        // the mark array)
        int rc = plan_one(in, p);
        if (rc == RL_OK && p->kernel == RL_K_RM_STREAM_LIT) return rc;
        In first = in;
        first.n_poses = LIT_SLICE_POSES;
        std::memset(p, 0, sizeof *p);
        rc = plan_one(first, p);
        if (rc == RL_OK && p->kernel == RL_K_RM_STREAM_LIT) {
            p->slices = (in.n_poses + LIT_SLICE_POSES - 1) / LIT_SLICE_POSES;
            p->slice_poses = LIT_SLICE_POSES;
            return rc;
        }
        std::memset(p, 0, sizeof *p);
    }
    if ((long)in.n_poses * in.num_rays >= slice_rays && in.o.variant >= 1 && !in.crash && in.n_poses > 1) {
        const int per = (int)std::max(1L, (slice_rays - 1) / in.num_rays);
        In first = in;
        first.n_poses = std::min(per, in.n_poses);
        const int rc = plan_one(first, p);
        p->slices = (in.n_poses + per - 1) / per;
        p->slice_poses = per;
        return rc;
    }
    return plan_one(in, p);
}

}  // namespace plan
