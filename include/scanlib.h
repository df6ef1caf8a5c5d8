/*
 * scanlib.h — C ABI of libscan_amd.so: the MI355X (gfx950) 2D lidar range library.
 *
 * Drop-in boundary for the ONE hot path of felrock/PyRacecarSimulator:
 *   ScanSimulator2D.scan / scanMany            scripts/scan_simulator.py:88-135
 *     -> range_libc.PyOMap / PyRayMarching / PyRayMarchingGPU / PyCDDTCast
 *        .calc_range_many(...)                  scripts/scan_simulator.py:72-76,103-106,130-133
 *                                               scripts/two_player/scan.py:45-46,69-70
 * range_libc is a Cython module; its FFI for this path is "float32 C-contiguous
 * numpy buffers + scalars" (SURVEY.md §8b).  Every entry point below is what a
 * ctypes / Cython stub for that path binds (INTEGRATION.md shows the stubs).
 *
 * Rules of the boundary
 *   - plain pointers and sizes only; no C++/torch types; nothing throws across it;
 *   - every function returns RL_OK (0) or a negative rl_status; rl_last_error()
 *     gives the thread-local message of the last failure;
 *   - host-pointer entry points are synchronous (results are in the caller's buffer
 *     on return) and never retain the pointers; *_device entry points take device
 *     pointers + a hipStream_t (as void*) and only enqueue work;
 *   - there is NO CPU fallback: without a usable HIP device rl_map_create fails.
 *   - a handle may be shared by threads (each call locks the handle), as the
 *     reference's rospy callbacks do (scripts/ros_interface.py:115,142,189);
 *     rl_map_update waits for every scan in progress on the map's methods (host calls
 *     hold a shared lock until their results have landed; launches the *_device entry
 *     points left in flight are waited for with a device synchronisation);
 *   - streams: *_device calls on ONE method handle may use different streams and then run
 *     concurrently on the GPU (that is how bench.py pipelines consecutive pose batches).
 *     Per-launch scratch is kept per stream (rl_launch_contexts() = 8 streams per handle without any
 *     synchronisation, more are served after a device synchronisation), lazily built
 *     tables are guarded by events.  The caller's own buffers (poses, ranges) are the
 *     caller's to order.  A stream must not be destroyed while a launch enqueued on it
 *     through this library is still running.
 *
 * Coordinates: occ[r*cols + c], r = row = world y, c = col = world x, row 0 at the
 * smallest world y (the layout of nav_msgs/OccupancyGrid.data that PyOMap reads).
 */
#ifndef SCANLIB_H
#define SCANLIB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rl_map rl_map;        /* replaces range_libc.PyOMap                */
typedef struct rl_method rl_method;  /* replaces range_libc.Py<RangeMethod>       */

typedef enum rl_status {
    RL_OK = 0,
    RL_ERR_INVALID = -1,     /* bad argument (null pointer, size, kind, ...)      */
    RL_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range         */
    RL_ERR_HIP = -3,         /* a HIP runtime call failed (message has details)   */
    RL_ERR_UNSUPPORTED = -4, /* e.g. num_rays larger than the kernel's LDS fan    */
    RL_ERR_NOMEM = -5
} rl_status;

/* Range methods.  Names follow range_libc's classes. */
typedef enum rl_kind {
    RL_BRESENHAM = 0,   /* BresenhamsLine: LDS-tiled bit-packed occupancy march (K2)   */
    RL_RM = 1,          /* PyRayMarching    scan_simulator.py:72-73  step_coeff 0.999  */
    RL_RM_GPU = 2,      /* PyRayMarchingGPU scan_simulator.py:74-76  step_coeff 1.0    */
    RL_CDDT = 3,        /* PyCDDTCast       two_player/scan.py:46                      */
    RL_GIANT_LUT = 4    /* GiantLUTCast: u16 [row][col][theta] table, fan-contiguous   */
} rl_kind;

/* ---- library -------------------------------------------------------------- */
const char *rl_version(void);
const char *rl_last_error(void);          /* thread-local, never NULL              */
int rl_device_count(void);                /* 0 when no HIP device is usable        */

/* ---- map: range_libc.PyOMap(map_msg)  scripts/ros_interface.py:210 ----------
 * occ: rows*cols bytes, nonzero = occupied (the reference feeds {0,255} after its
 * binarisation, scripts/ros_interface.py:80-86; PyOMap tests data > 10).
 * res/ox/oy/oyaw: map_msg.info.resolution / origin position / yaw
 * (scripts/ros_interface.py:212-220).  The exact Euclidean distance transform
 * (range_libc DistanceTransform) is built on the device at creation.           */
int rl_map_create(const uint8_t *occ, int rows, int cols, float res, float ox, float oy,
                  float oyaw, int device, rl_map **out);
/* ---- several devices behind ONE handle (SURVEY.md section 8b "device_mask") -----------------------
 * The reference's caller of this path is ONE Python process (scripts/mcts.py:237 ->
 * scripts/racecar_simulator_v2.py:146-167 -> scripts/scan_simulator.py:113-135), so a drop-in that wants
 * the other GPUs of the node cannot ask it to become N processes.  rl_map_create_multi uploads the map to
 * every device of `devices` (an index may repeat: several contexts on one GPU) and builds the tables on
 * each; rl_method_create on that map returns a handle whose HOST-pointer entry points —
 * rl_calc_range_fan, rl_calc_range_many_fan, rl_calc_range_many, rl_check_collision_many,
 * rl_check_collision_groups, rl_car_rollout_check — cut the batch into contiguous pose blocks, one per
 * device (a device is brought in per `multi_min_poses` = 512 poses, option of the handle), run them
 * concurrently (one worker thread per device, nothing is forked) and let every device write its block of
 * the results straight into the caller's buffer.  Results are bit-identical to the single-device call:
 * noise is keyed by the global ray id, crash indices are global.  With a result buffer from
 * rl_host_alloc every device stores over its own PCIe link (4 B per ray) and nothing touches xGMI, so the
 * host-pointer scan is EXPECTED to scale with the number of links — modelled, not measured: the pool's boxes have one
 * GPU, the tests name device 0 several times there (and distinct devices wherever more are visible); the crash forms
 * return 4 B per roll-out.
 * The *_device entry points take device memory of ONE device: call them with rl_method_replica(h, i)
 * (a borrowed per-device handle; rl_map_replica likewise).  rl_map_update updates every replica.        */
int rl_map_create_multi(const uint8_t *occ, int rows, int cols, float res, float ox, float oy, float oyaw,
                        const int *devices, int n_devices, rl_map **out);
int rl_map_n_devices(const rl_map *m);                 /* 1 for a map of rl_map_create                    */
/* ... and with the results in DEVICE memory of one of the handle's devices (round 6): the reference's consumer of a
 * roll-out batch is one process (scripts/mcts.py:237 -> scripts/racecar_simulator_v2.py:146-167 -> Car::isCrashed); when
 * what comes next is itself a kernel on ONE GPU (a policy network, the tree update), the ranges — or the fused crash
 * indices — of every device should land in that GPU's HBM, not in host memory.  h: a method of a multi-device map;
 * poses (and edge): HOST pointers; `consumer`: index of the replica (0 .. rl_method_n_devices - 1) whose device owns
 * d_outs_on_consumer (n_poses * num_rays floats) / d_first_on_consumer (n_groups ints).  Every device marches its
 * contiguous pose block into its own HBM in `chunks` pieces (0 = 4) and sends each piece to the consumer with
 * hipMemcpyPeerAsync on a second stream — device to device over xGMI where peer access exists — while the next piece
 * marches; the consumer's own block is marched in place.  Synchronous: the data is in the consumer's memory on return.
 * Bit-identical to the single-device scan (noise keyed by the global ray id, crash indices global).  Tested with
 * repeated device indices on one GPU (the peer copy degenerates to a device-to-device copy) and with distinct devices
 * wherever more than one is visible; rates over xGMI are MODELLED, not measured.                 */
int rl_calc_range_fan_multi_device(rl_method *h, const float *poses, int n_poses, float fov, int num_rays, int consumer,
                                   float *d_outs_on_consumer, int chunks);
int rl_check_collision_groups_multi_device(rl_method *h, const float *poses, int n_groups, int group, float fov,
                                           int num_rays, const double *edge, double crash_thresh, int consumer,
                                           int *d_first_on_consumer);
rl_map *rl_map_replica(rl_map *m, int i);              /* NULL when i is out of range                     */
/* replace the occupancy (same shape) and rebuild the distance transform: the
 * per-scan rebuild of scripts/two_player/rcs_two_player.py:110-121 and the
 * updateMap stub of scripts/scan_simulator.py:81-86.  Methods created from the
 * map see the new data (CDDT / GiantLUT tables are rebuilt lazily).            */
int rl_map_update(rl_map *m, const uint8_t *occ);
/* ... the two-player tick without the grid crossing PCIe (round 6): the occupancy becomes the BASE map — as created or
 * last rl_map_update'd — with the n cells flat_idx[i] = row * cols + col set to `value` (nonzero = occupied; the
 * reference writes 255), and every table is rebuilt on the device.  Indices outside the grid are skipped, as the
 * reference's own guard skips them (rcs_two_player.py:113).  A stamp replaces the previous one (`ego_map[:] = org_map`,
 * rcs_two_player.py:110).  n = 0 restores the base map.                                                                */
int rl_map_stamp_cells(rl_map *m, const int32_t *flat_idx, int n, uint8_t value);
void rl_map_destroy(rl_map *m);
int rl_map_rows(const rl_map *m);
int rl_map_cols(const rl_map *m);
int rl_map_device(const rl_map *m);
/* test hooks: copy the device-built tables back (float32 rows*cols / u8 rows*cols) */
int rl_map_get_dt(rl_map *m, float *dt_out);
int rl_map_get_occ(rl_map *m, uint8_t *occ_out);

/* ---- method: range_libc.PyRayMarching(omap, mrx) etc. -----------------------
 * max_range_px: scripts/racecar_simulator_v2.py:196 (int(scan_max_range/res));
 * theta_disc: only for RL_CDDT / RL_GIANT_LUT (two_player/rcs_two_player.py:121). */
int rl_method_create(rl_map *m, int kind, float max_range_px, int theta_disc, rl_method **out);
void rl_method_destroy(rl_method *h);
int rl_method_kind(const rl_method *h);
int rl_method_n_devices(const rl_method *h);           /* devices behind the handle (rl_map_create_multi)  */
rl_method *rl_method_replica(rl_method *h, int i);     /* per-device handle for the *_device entry points  */

/* upstream 2-arg calc_range_many(ins, outs): one (x, y, theta) world row per ray.
 * scripts/two_player/scan.py:69-70.  ins: n*3 floats, outs: n floats (metres).  */
int rl_calc_range_many(rl_method *h, const float *ins_n3, float *outs_n, int n);

/* the fork's 4-arg calc_range_many(ins, outs, fov, num_rays) exactly as the
 * reference calls it (scripts/scan_simulator.py:103-106,130-133): ins has n_rows =
 * n_poses*num_rays rows of 3 floats, pose p lives in row p*num_rays (all other rows
 * are ignored), beam j of pose p is cast at theta_p - fov/2 + j*fov/num_rays and
 * written to outs[p*num_rays + j] (layout consumed by racecar/src/racecar.cpp:320).
 * Only the n_poses live rows cross PCIe.                                         */
int rl_calc_range_many_fan(rl_method *h, const float *ins_rows3, float *outs, int n_rows,
                           float fov, int num_rays);

/* dense form of the same call: poses[p*3..] -> outs[p*num_rays + j].
 * hit_cells (2 ints per ray: col,row or -1,-1) and steps (samples per ray) are
 * optional diagnostics for RL_RM / RL_RM_GPU / RL_BRESENHAM; pass NULL otherwise. */
int rl_calc_range_fan(rl_method *h, const float *poses_p3, int n_poses, float fov, int num_rays,
                      float *outs, int32_t *hit_cells_or_null, uint16_t *steps_or_null);

/* Optional: pinned host memory for result buffers.  A host-pointer scan whose `outs` lies inside a
 * block from rl_host_alloc is written by the kernel directly (no staging copy on the way back:
 * scanMany(200) 63 -> ~40 us) up to 2^21 rays per call and by DMA from HBM beyond that (faster from ~2000 poses up).  ScanSimulator2D keeps its cached output vectors
 * (scripts/scan_simulator.py:32-40) in such blocks.  rl_host_free waits for the device first.      */
int rl_host_alloc(size_t bytes, void **out);
int rl_host_free(void *p);

/* device-resident, asynchronous forms (pointers are device memory on the map's
 * device, stream is a hipStream_t or NULL for the default stream).               */
int rl_calc_range_fan_device(rl_method *h, const float *d_poses_p3, int n_poses, float fov,
                             int num_rays, float *d_outs, int32_t *d_hit_cells_or_null,
                             uint16_t *d_steps_or_null, void *hip_stream);
int rl_calc_range_many_device(rl_method *h, const float *d_ins_n3, float *d_outs_n, int n,
                              void *hip_stream);

/* Gaussian range noise (scripts/scan_simulator.py:33,109; scan_std params.yaml:32):
 * out += N(0, std) from a counter-based generator keyed by (seed, global ray id +
 * ray_offset) so a sharded batch reproduces the unsharded one.  std <= 0 disables. */
int rl_set_noise(rl_method *h, float std, uint64_t seed, uint64_t ray_offset);

/* Fused crash test (Car::isCrashed racecar/src/racecar.cpp:305-328 over the output
 * of scanMany, scripts/racecar_simulator_v2.py:146-167): scans n_poses poses and
 * returns in *first_crashed the index of the first pose with any beam j where
 * (double)range - edge[j] < crash_thresh, else -(n_poses+1).  ranges_or_null gets
 * the ranges too when non-NULL.  edge: num_rays doubles (setCarEdgeDistances).     */
int rl_check_collision_many(rl_method *h, const float *poses_p3, int n_poses, float fov,
                            int num_rays, const double *edge, double crash_thresh,
                            int *first_crashed, float *ranges_or_null);

/* The same test per roll-out of a batch: poses holds n_groups roll-outs of `group` consecutive
 * poses (scripts/mcts.py:228-231 stores 200 per roll-out); first_crashed[g] = index inside
 * roll-out g of its first crashed pose, else -(group+1).  Works for every range method.       */
int rl_check_collision_groups(rl_method *h, const float *poses_p3, int n_groups, int group,
                              float fov, int num_rays, const double *edge, double crash_thresh,
                              int *first_crashed, float *ranges_or_null);

/* device-resident, asynchronous form: every pointer is device memory, d_first_crashed gets
 * n_groups ints; d_ranges_or_null may be NULL for RL_RM / RL_RM_GPU (the test is fused into the
 * march kernel, ranges need not be stored at all) and is required scratch for the other methods. */
int rl_check_collision_groups_device(rl_method *h, const float *d_poses_p3, int n_groups, int group,
                                     float fov, int num_rays, const double *d_edge,
                                     double crash_thresh, int *d_first_crashed,
                                     float *d_ranges_or_null, void *hip_stream);

/* ---- scan consumer: Follow-the-Gap steering --------------------------
 * Replaces followgap.PyFollowGap(ws, md, ma, angle_inc).eval(lidar, size)
 * (followgap/followgap.pyx:23-31 -> FollowGap::eval, followgap/followgap.hpp:104-129; built at
 * scripts/mcts.py:97-99, scripts/two_player/simple_driver.py:31, called at scripts/mcts.py:267,
 * simple_driver.py:51) for a BATCH of scans: n_scans rows of `size` float32 ranges in, one
 * steering angle per scan out, one wave per scan.  Bit-identical to the reference's compiled
 * header (tests/golden/followgap_ref.npz).  size < 10 -> RL_ERR_INVALID (the reference indexes
 * out of bounds there); a gap consisting of the single last beam reads beam size-1 where the
 * reference reads one past the array.  Scans of up to 1280 beams run the one-bit-per-beam kernel
 * (consumer_kernels.h: followgap_bits_kernel), longer ones the per-beam walk (followgap_kernel);
 * RL_FOLLOWGAP_WALK=1 in the environment at create time forces the walk (A/B, diagnostics).      */
typedef struct rl_followgap rl_followgap;
int rl_followgap_create(int device, int window_size, float max_distance, float max_angle,
                        float angle_inc, rl_followgap **out);
void rl_followgap_destroy(rl_followgap *g);
int rl_followgap_eval(rl_followgap *g, const float *scans, int n_scans, int size, float *angles);
/* scans and angles resident on the device (e.g. the ranges a scan call just wrote)              */
int rl_followgap_eval_device(rl_followgap *g, const float *d_scans, int n_scans, int size,
                             float *d_angles, void *hip_stream);

/* ---- roll-out pose generator ---------------------------------------
 * The step in front of scanMany in MCTS.rollout (scripts/mcts.py:214-231): 200 x
 * {Car::control, Car::updatePosition(dt)} (racecar/src/racecar.cpp:53-98,118-237,294-303) per
 * roll-out, one GPU lane per roll-out in float64.
 * car_params: the 17 constructor arguments of Car in order (racecar/include/racecar.hpp:32-36).
 * states: 11 doubles per roll-out in Car::getState layout (racecar.cpp:355-376).
 * actions: (speed, steer) pairs, ceil(n_steps/action_every) per roll-out.                      */
typedef struct rl_car rl_car;
int rl_car_create(int device, const double *car_params17, rl_car **out);
/* the same over several devices (pair it with a method of rl_map_create_multi on the SAME device list):
 * rl_car_rollout and rl_car_rollout_check then cut the roll-outs into contiguous blocks, one per device  */
int rl_car_create_multi(const int *devices, int n_devices, const double *car_params17, rl_car **out);
void rl_car_destroy(rl_car *c);
/* poses_out: n_rollouts*n_steps*3 float32 (x, y, theta of the car after each step);
 * states_out (optional): final states; velocities_out (optional): state[3] after each step.   */
int rl_car_rollout(rl_car *c, const double *states_in, const double *actions, int n_rollouts,
                   int n_steps, int action_every, double dt, float *poses_out,
                   double *states_out_or_null, double *velocities_out_or_null);
/* roll-outs -> poses -> scan -> per-roll-out crash index without the poses or the ranges ever
 * leaving the device: what MCTS.rollout + checkCollisionMany compute (scripts/mcts.py:202-245,
 * scripts/racecar_simulator_v2.py:146-167), for n_rollouts roll-outs in one call.              */
int rl_car_rollout_check(rl_car *c, rl_method *h, const double *states_in, const double *actions,
                         int n_rollouts, int n_steps, int action_every, double dt, float fov,
                         int num_rays, const double *edge, double crash_thresh, int *first_crashed,
                         double *states_out_or_null, double *velocities_out_or_null);

/* Car::setCarEdgeDistances (racecar/src/racecar.cpp:239-292; called at
 * scripts/racecar_simulator_v2.py:47-50): distance from the lidar to the car's outline along each of
 * num_rays beams starting one increment after min_ang — the table every crash test above takes as
 * `edge`.  Host arithmetic (a one-off table), no device needed; the reference's quirks are kept
 * (shifted by one beam, pi = 3.145, about -1016 m for a beam at exactly 0 rad).                   */
int rl_car_edge_distances(int num_rays, double min_ang, double ang_inc, double scan_dist_to_base,
                          double width, double wheelbase, double *edge_out);
/* Car::isCrashed (racecar/src/racecar.cpp:305-328) over ranges already on the host: index of the
 * first of n_scans scans with a beam j where (double)range - edge[j] < crash_thresh, else
 * -(n_scans+1).  (Scanned batches use the fused device test: rl_check_collision_*.)               */
int rl_car_is_crashed(const float *ranges, int num_rays, int n_scans, const double *edge,
                      double crash_thresh, int *first_crashed);

/* device time of the last enqueued launch sequence of this handle, from HIP events
 * recorded on the launch stream (blocks until that work has finished).  Events are only
 * recorded after rl_method_set_option(h, "timing", 1): they cost microseconds per launch.
 * "timing" = 2 brackets the ray-marching kernel alone (the pose-binning launches in front of it
 * excluded), which is the duration a kernel trace reports for it.                              */
int rl_last_kernel_ms(rl_method *h, float *ms_out);

/* tuning / diagnostics: integer options by name.  None changes a result bit; defaults are the
 * measured optima on MI355X.
 *   schedule   variant (1 stream kernel | 0 chunk-per-wave | 2 occ_fan_lds: unit steps on an LDS occupancy
 *              window, approximate | 3 the UPSTREAM-LITERAL arithmetic of RL_RM / RL_RM_GPU — range_libc's CPU
 *              statement: per-ray glibc sinf / cosf, un-fused products and sums —, the ONE option that changes result
 *              bits: onto the checker's libm form.  A production mode since round 5 — same entry points, same stream
 *              kernel schedule, fused crash test and noise included, ~0.87x the canonical rate — and since round 6 the
 *              DEFAULT of RL_RM (the class that names range_libc's CPU RayMarching); RL_RM_GPU defaults to 1; a negative
 *              value restores the kind's default), grid_mult, wg_threads, low_water (-1 auto), run_log2 (-1 auto), xcd_bands,
 *              sort_poses, tiled (step-map layout), slots (rays per lane: 1 | 2 | 3 | 0 auto),
 *              code_map (2: launches of >= code_min_rays rays read the step map as 16-bit palette codes, palette in LDS —
 *              the same sample sequence on half the bytes; 0: float32 steps), code_min_rays (default 2^22),
 *              tail_pct / tail_wg_pct (a second generation of workgroups for whole-machine launches: measured, off),
 *              cddt_bins (one look-up per pose and table bin), cddt_theta_min (poses from which the look-ups
 *              run theta-major: all poses against one table bin at a time), cddt_lds_sort
 *   binning    inline_prep, inline_max, inline_map_kb, stripe_max, order_inline, bin_multi_min,
 *              bin_generic, bin_ppw (poses per workgroup of the grid-wide binning kernels), tile_stripe (order of the map
 *              tiles the poses are binned by: N > 0 = tile rows in stripes of N, a stripe walked column by column, so that
 *              a band of the pose list sweeps its part of the map once; 0 = row-major; -1, the default = the tile rows
 *              one of xcd_bands bands of evenly spread poses holds)
 *   launches   slice_log2 (pose slices below 2^n rays), pinned_max_rays (zero-copy host calls), direct_max_rays (a result
 *              buffer in a block of rl_host_alloc is written by the kernel itself up to this many rays, by DMA beyond),
 *              overlap_min_rays (plain host-pointer scans of at least this many rays — default 2^24 — run as four pose
 *              slices, the device-to-host copy of one overlapping the march of the next; 0 = never)
 *              spec_drain / spec_stretch (one ray per lane: value-speculating drain loop from <= N live lanes,
 *              plain samples between attempts); drain_cap / drain_stretch (several rays per lane: a wave whose
 *              stream is dry compacts its last <= N rays (<= 64) into one ray per lane and finishes them with
 *              that loop); nt_store (1: the stream kernels' ranges leave with non-temporal stores — write-once data
 *              that would otherwise displace the step map from the L2; set 0 when the next kernel on the stream
 *              reads the ranges back at once, e.g. rl_followgap_eval_device)
 *   diagnosis  timing (1 launch sequence | 2 main kernel only), debug_stamps, drain_prio, lut_debug (bits: 1 skip the
 *              gathers / searches, 2 skip the range stores, 8 non-temporal GiantLUT range stores, 16 PLAIN instead of
 *              non-temporal GiantLUT row loads — the A/B partner of the default)
 *   multi-device handles: every option goes to every device's replica; multi_min_poses (poses per device
 *              from which another device is brought in, default 512) belongs to the handle itself.
 * rl_method_get_info additionally answers n_devices, n_cu, clock_khz, last_grid, map_epoch, code_entries (palette entries of
 * the handle's map incl. the two stop codes; 0 = no code map: option off, geometry or palette does not fit) and, for RL_CDDT
 * (builds the table if needed, synchronises): cddt_values, cddt_buckets, cddt_nonempty_buckets.          */
int rl_method_set_option(rl_method *h, const char *name, int value);
int rl_method_get_info(rl_method *h, const char *name, int64_t *value_out);
/* ---- launch planning -------------------------------------------------------------------------
 * Which kernel, grid, LDS size and pose-binning pass a fan call of (n_poses x num_rays) takes is
 * decided by ONE pure function of the map shape, the device's CU count and the options above —
 * no device, no handle state: rl_plan_fan can be called (and is tested) on a box without a GPU.
 * rl_method_plan_fan applies it with a handle's current options (a ray-marching handle with the code map on builds its
 * step map first — the one thing a plan needs from the device is the map's palette size); every launch goes through the
 * same function and rl_method_last_plan returns the plan the last launch of the handle used
 * (`name` is the kernel as a rocprofv3 kernel trace prints it, template arguments included).     */
/* Which fields still carry weight (round 4; every default is a measured optimum, profiles/r03/plan_sweep.txt):
 *   set by callers in production   grid_mult + slots (a caller that keeps several launches in flight: 3 and 2,
 *                                  INTEGRATION.md), slice_log2 (only to force slicing in tests)
 *   thresholds of the planner      inline_max, inline_map_kb, stripe_max, bin_multi_min, cddt_theta_min, xcd_bands,
 *                                  low_water (-1 = automatic) — change them only with a sweep in hand
 *   arithmetic                     variant 3: range_libc's CPU arithmetic stated literally — glibc sinf / cosf per ray,
 *                                  un-fused products and sums — bit-identical to the checker's libm form; the stream
 *                                  kernel's schedule (rm_fan_stream_kernel<.., LIT>), 0.87x the canonical rate; -1 (the
 *                                  struct's default) = the kind's default: 3 for RL_RM, 1 otherwise
 *   kernel selection for A/B       variant (0 chunk kernel, 2 occ_fan_lds), group_drain, handoff (round 5's measured and
 *                                  rejected drain forms), cddt_search (0 = round 4's search kernel), tiled (0 = row-major step map; the
 *                                  planner clears it by itself when the tiled geometry does not fit), cddt_bins
 *   diagnostics only               wg_threads, sort_poses, inline_prep, order_inline, bin_generic, run_log2,
 *                                  cddt_sort, lut_debug, debug_stamps
 * Out-of-range values are clamped by the planner (a zeroed struct is valid input).                              */
typedef struct rl_plan_opts {
    int variant, grid_mult, wg_threads, low_water, sort_poses, xcd_bands, slots, tiled;
    int inline_prep, inline_max, inline_map_kb, stripe_max, order_inline, bin_multi_min, bin_generic;
    int run_log2, cddt_bins, cddt_sort, lut_debug, debug_stamps, slice_log2, cddt_theta_min;
    int cddt_search;     /* theta-major CDDT: 1 = look-ups prepared once per pose, picked up by the 8-lane groups
                            (cddt_theta_search2_kernel + cddt_theta_fan_kernel, round 5), 0 = every group prepares its own
                            (round 4), 2 = search and fan of a 64-pose tile fused in one workgroup, per-bin results in
                            LDS only (cddt_theta_fused_kernel)                                                            */
    int code_map;        /* ray marching: 2 = the step map as 16-bit palette codes, the palette (exact float32 steps) in LDS
                            (rm_fan_stream_kernel<..., CODE = 2>: the same sample sequence, half the bytes per cell);
                            0 = float32 steps.  Takes effect where the map's palette fits (code_entries)                */
    int code_min_rays;   /* ... from this many rays per launch: below it the look-up's latency in every dependent sample costs a
                            lone launch more than the smaller footprint buys (profiles/r06/ab_code_map.txt)              */
    int tail_pct;        /* ray marching, launches that fill the machine: this share (%) of every band's work goes to a SECOND
                            generation of tail_wg_pct % as many workgroups, dispatched as resident ones finish (0 = off)    */
    int tail_wg_pct;
    int code_entries;    /* entries of the map's step palette with its two stop codes — a handle knows it once its step map
                            is built (rl_method_get_info "code_entries", filled in by rl_method_plan_fan); 0 = unknown or
                            too many: the device-less rl_plan_fan then plans the float32 map                          */
} rl_plan_opts;

typedef enum rl_kernel_id {
    RL_K_NONE = 0,
    RL_K_RM_CHUNK = 1,      /* rm_fan_kernel<AUX, CRASH>: one 64-beam chunk per wave (variant 0)          */
    RL_K_RM_STREAM = 2,     /* rm_fan_stream_kernel<AUX, CRASH, NT, INLINE, TILED, SLOTS, LIT, CODE> (default) */
    RL_K_OCC_LDS = 3,       /* occ_fan_lds_kernel<AUX> (variant 2)                                        */
    RL_K_BL_STREAM = 4,     /* bl_fan_stream_kernel<AUX, 1024>                                            */
    RL_K_BL_LDS = 5,        /* bl_fan_kernel<AUX> (variant 0)                                             */
    RL_K_LUT_LDS = 6,       /* lut_fan_lds_kernel<NL, CH>                                                 */
    RL_K_LUT_FAN = 7,       /* lut_fan_kernel<CH>                                                         */
    RL_K_CDDT_BINS = 8,     /* cddt_fan_bins_kernel                                                       */
    RL_K_CDDT_RAYS = 9,     /* cddt_fan_kernel                                                            */
    RL_K_CDDT_THETA = 10,   /* cddt_theta_search[2]_kernel + cddt_theta_fan_kernel | cddt_theta_fused_kernel (large batches) */
    RL_K_RM_LITERAL = 11,   /* rm_literal_kernel<AUX, RAYS>: upstream-literal arithmetic, one lane per ray — variant 3 with
                               diagnostics (hit cells / sample counts), the 2-argument per-ray form, fans below 64 beams */
    RL_K_RM_STREAM_LIT = 12 /* rm_fan_stream_kernel<false, CRASH, 1024, true, true, SLOTS, true, CODE>: variant 3 in production —
                               the upstream-literal arithmetic on the stream kernel's schedule (ranges, fused crash test,
                               noise; two rays per lane; batches above 8192 poses in pose slices)                 */
} rl_kernel_id;

typedef enum rl_binning {
    RL_BIN_NONE = 0,          /* the march kernel derives the pose records of its own blocks (LDS)        */
    RL_BIN_SMALL_KEYS = 1,    /* pose_bin_small_kernel<true>: tile order only, one workgroup              */
    RL_BIN_SMALL_RECORDS = 2, /* pose_bin_small_kernel<false>: records in tile order, one workgroup       */
    RL_BIN_GRID_SORT = 3,     /* pose_prep -> tile_scan_a -> pose_scatter (grid-wide)                    */
    RL_BIN_GRID_UNSORTED = 4, /* pose_prep only (caller's order kept)                                     */
    RL_BIN_GENERIC = 5        /* pose_bin_kernel: one workgroup, any size                                 */
} rl_binning;

typedef struct rl_launch_plan {
    int kernel;          /* rl_kernel_id                                                                  */
    int grid, block;     /* workgroups, threads per workgroup                                             */
    int lds_bytes;       /* dynamic LDS per workgroup                                                     */
    int binning;         /* rl_binning pass in front of the march (0 = none)                              */
    int record_source;   /* RL_K_RM_STREAM: 0 records binned in HBM | 1 derived in LDS, caller's order |
                            2 derived in LDS, row-stripe bands compacted by every workgroup |
                            3 derived in LDS, tile order from the keys-only binning pass                  */
    int slots;           /* rays per lane                                                                 */
    int bands, run_log2, k_max, tiled, aux, crash;
    int nl, ch;          /* GiantLUT: 16-B loads per lane and row, 64-beam chunks per pose; CDDT per-bin
                            kernel: lanes per pose, poses per workgroup pass                              */
    int slices;          /* > 1: the batch goes through in this many pose slices of slice_poses poses,    */
    int slice_poses;     /*      each its own launch sequence planned like this one (for its own size)    */
    int code;            /* RL_K_RM_STREAM[_LIT]: 2 = marches on the 16-bit code map, 0 = float32 step map  */
    int code_entries;    /*      ... palette entries the workgroups copy to LDS                           */
    int gen1;            /* > 0: the first gen1 workgroups are the resident generation, grid - gen1 follow (tail_pct) */
    char name[192];
} rl_launch_plan;

int rl_plan_default_opts(rl_plan_opts *out);
/* kind: rl_kind; want_aux: hit cells / step counts requested; want_crash: fused crash test.
 * opts_or_null = defaults.  Pure host arithmetic.                                                 */
int rl_plan_fan(int kind, int n_cu, int rows, int cols, float max_range_px, int theta_disc,
                const rl_plan_opts *opts_or_null, int n_poses, int num_rays, int want_aux,
                int want_crash, rl_launch_plan *out);
int rl_method_plan_fan(rl_method *h, int n_poses, int num_rays, int want_aux, int want_crash,
                       rl_launch_plan *out);
int rl_method_last_plan(rl_method *h, rl_launch_plan *out);
/* launch contexts (per-stream scratch sets) a handle keeps: streams beyond this count are served
 * after a device synchronisation (callers that pipeline batches stay at or below it).            */
int rl_launch_contexts(void);

/* test hook (RL_GIANT_LUT): builds the table if needed and copies rows [row0,row1) of
 * uint16 lut[row][col][theta_bin] (entry = rint(min(range_px,max_range)*65535/max_range)). */
int rl_method_read_lut(rl_method *h, int row0, int row1, uint16_t *out);
/* diagnostics: after a launch with option "debug_stamps"=1, copies 4 words per wave of the
 * stream kernel (start, end in 100 MHz ticks; services<<32 | longest drain chain;
 * (drain start - start)<<32 | blocks<<8 | band);
 * returns the number of words copied (>= 0) or a negative rl_status.                       */
int rl_debug_read_stamps(rl_method *h, uint64_t *out, int max_words);

/* ---- 16-bit ranges for the multi-GPU exchange (opt-in, LOSSY) ---------------------------------
 * The all-gather of ranges BASELINE.json's north_star names moves 4 B per ray over xGMI; these two
 * streaming passes let a caller exchange 2 B per ray instead: q = rint(clamp(r, 0, max) * 65535 / max),
 * r' = q * max / 65535.  Error <= max/131070 plus float32 rounding (< 0.12 mm at the reference's 15 m, params.yaml:39) —
 * inside north_star's one-cell tolerance, but NOT bit-exact: only bench.py --gather ranges_u16 and
 * distributed.ShardedScan(mode="ranges_u16") use them, and both label their results.
 * Device pointers (any element alignment; 16-byte aligned pairs take the wide path); asynchronous on
 * hip_stream.                                                                                       */
int rl_ranges_to_u16_device(int device, const float *d_ranges, size_t n, float max_range_m,
                            uint16_t *d_out_u16, void *hip_stream);
int rl_ranges_from_u16_device(int device, const uint16_t *d_in_u16, size_t n, float max_range_m,
                              float *d_ranges, void *hip_stream);

/* diagnostics: how fast a CU of this device retires a wave-wide global_load_dword whose
 * `active_lanes` live lanes read unrelated cells of a cache-resident table — the instruction the
 * ray-marching kernels are bound by (tools/probes/tcp_probe3.hip is the same
 * probe stand-alone).  Returns active lanes per clock per CU; chip peak in samples/s =
 * lanes_per_clk_per_cu * n_cu * clock_hz.  bench.py reports `roofline_gather` against it.        */
int rl_probe_gather_rate(int device, int active_lanes, double *lanes_per_clk_per_cu,
                         double *clock_hz_or_null, int *n_cu_or_null);

/* diagnostics: HBM stream rates of this device with hand-written 16-B-per-lane kernels on two buffers of
 * `bytes`: gbs_out5 = {copy, read-only, write-only, copy with non-temporal stores, fill with non-temporal
 * stores} in GB/s (copy counts read + write).  The practical ceiling next to the 8 TB/s spec of
 * bench.py's roofline (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).                     */
int rl_probe_hbm(int device, size_t bytes, double *gbs_out5);
/* ... the same kernels with NON-TEMPORAL loads (what the GiantLUT row fetch uses): gbs_out3 = {read-only, copy, copy
 * with non-temporal stores}.                                                                          */
int rl_probe_hbm_nt(int device, size_t bytes, double *gbs_out3);

/* diagnostics: the audit mode's sinf / cosf (glibc's algorithm in double precision, csrc/literal_kernels.h) of n
 * host floats, evaluated on the device — tests hold it against the host's libm.                        */
int rl_probe_literal_sincosf(int device, const float *x, size_t n, float *sin_out, float *cos_out);

#ifdef __cplusplus
}
#endif
#endif /* SCANLIB_H */
