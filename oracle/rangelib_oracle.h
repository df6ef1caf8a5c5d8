/*
 * rangelib_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the ray-casting algorithms behind
 *   ScanSimulator2D.scan / scanMany          (/root/reference/scripts/scan_simulator.py:88-135)
 *   -> range_libc.Py{RayMarching,RayMarchingGPU,CDDTCast}.calc_range_many
 *      (call sites scripts/scan_simulator.py:72-76,103-106,130-133;
 *       scripts/two_player/scan.py:45-46,69-70)
 *
 * PARITY UNPINNED: range_libc (github.com/felrock/range_libc, fork of
 * github.com/kctess5/range_libc, no pinned version — reference README.md:22,
 * .gitignore:6) is an un-vendored dependency that is absent from
 * /root/reference, and the reference holds no tests or golden vectors for this
 * path.  This file restates range_libc's *published* algorithm (RangeLib.h:
 * OMap, DistanceTransform, RayMarching, BresenhamsLine, CDDTCast, GiantLUTCast;
 * kernels.cu: cuda_ray_marching) as recalled in SURVEY.md §8(a) rows a6-a14 and
 * Appendix A, and anchors the conventions (fan angles, output layout,
 * max_range_px, binarisation) on the reference's own call sites.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pyracecarsimulator_amd/) never does.
 *
 * Conventions
 *   grid   occ[r*cols + c], r = row = world y, c = col = world x, row 0 = min y
 *   rays   marched in (col,row) space along (cos th, sin th), th = world heading - yaw
 *   trig   one deterministic float32 routine (orc_sincosf) built from fmaf only,
 *          so a GPU restatement can reproduce every bit; no libm trig.
 */
#ifndef RANGELIB_ORACLE_H
#define RANGELIB_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_map {
    int rows, cols;
    const uint8_t *occ;      /* borrowed, rows*cols, nonzero = occupied           */
    float res;               /* world_scale  (metres / cell)                      */
    float ox, oy;            /* world origin                                      */
    float wa_cos, wa_sin;    /* cos/sin of world_angle = -yaw (orc_sincosf)       */
    float wa;                /* world_angle = -yaw                                */
    float inv_res;           /* (float)(1.0 / (double)res)                        */
} orc_map;

/* range_libc PyOMap(OccupancyGrid) — scripts/ros_interface.py:210; row a6 */
void orc_map_init(orc_map *m, const uint8_t *occ, int rows, int cols,
                  float res, float ox, float oy, float oyaw);

/* deterministic float32 sin/cos (Cody-Waite + minimax, fmaf only) */
void orc_sincosf(float x, float *s, float *c);

/* exact Euclidean distance transform in cells: dt = sqrtf((float)d2); row a7.
 * Maps without any occupied cell: every dt = 1e10f (Felzenszwalb INF=1e20). */
void orc_edt(const uint8_t *occ, int rows, int cols, float *dt);
/* squared distances as exact integers (UINT32_MAX = no obstacle anywhere) */
void orc_edt_sq(const uint8_t *occ, int rows, int cols, uint32_t *d2);

/* ---- RayMarching (rows a8/a9/a10/a11) ----------------------------------
 * step_coeff 0.999f = CPU RayMarching, 1.0f = kernels.cu (RayMarchingGPU).
 * hits: 2 ints per ray (col,row) or (-1,-1) on a miss; steps: DT samples read.
 * hits/steps may be NULL.  nthreads<=1: serial loop (faithful to upstream).   */
void orc_rm_fan(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                const float *poses, int n_poses, float fov, int num_rays,
                float *ranges, int32_t *hits, uint16_t *steps, int nthreads);
void orc_rm_rays(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                 const float *ins, int n, float *ranges, int32_t *hits, uint16_t *steps,
                 int nthreads);
/* upstream-literal variant (libm cosf/sinf of -th + rot_const, calc_range(y,x,.)):
 * CPU-only cross-check of the canonical form, tolerance-compared in tests.      */
void orc_rm_rays_libm(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                      const float *ins, int n, float *ranges, int32_t *hits_or_null,
                      uint16_t *steps_or_null);
void orc_rm_fan_libm(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                     const float *poses, int n_poses, float fov, int num_rays, float *ranges,
                     int32_t *hits_or_null, uint16_t *steps_or_null);

/* the host libm's sinf / cosf in bulk; the statement of glibc's algorithm the product's audit mode runs on the
 * device (csrc/literal_kernels.h); and how many float inputs the two differ on (0 on glibc >= 2.28, x86-64 + FMA) */
/* BresenhamsLine in the upstream-literal form (libm trig, un-fused end point and hit distance; parity unpinned) */
void orc_bl_rays_libm(const orc_map *m, float max_range_px, const float *ins, int n, float *ranges, int32_t *hits,
                      uint16_t *steps);
void orc_bl_fan_libm(const orc_map *m, float max_range_px, const float *poses, int n_poses, float fov, int num_rays,
                     float *ranges, int32_t *hits, uint16_t *steps);
void orc_libm_sincosf(const float *x, long n, float *s_out, float *c_out);
void orc_lit_sincosf(const float *x, long n, float *s_out, float *c_out);
long orc_libm_restatement_check(uint32_t first, uint32_t step, long *bad_cos, int nthreads);

/* ---- BresenhamsLine (row a12) ------------------------------------------ */
void orc_bl_fan(const orc_map *m, float max_range_px,
                const float *poses, int n_poses, float fov, int num_rays,
                float *ranges, int32_t *hits, uint16_t *steps, int nthreads);
void orc_bl_rays(const orc_map *m, float max_range_px,
                 const float *ins, int n, float *ranges, int32_t *hits, uint16_t *steps,
                 int nthreads);

/* ---- GiantLUTCast (row a14): u16 table [row][col][theta_bin] ------------- *
 * entry = rint(min(range_px, max_range) * 65535/max_range), range from
 * RayMarching with the given step_coeff cast from the cell's integer corner
 * (x=(float)c, y=(float)r) at grid heading bin*2pi/theta_disc.                */
void orc_lut_build(const orc_map *m, const float *dt, float max_range_px, float step_coeff,
                   int theta_disc, int r0, int r1, uint16_t *lut /* (r1-r0)*cols*theta_disc */,
                   int nthreads);
void orc_lut_fan(const orc_map *m, const uint16_t *lut, int theta_disc, float max_range_px,
                 const float *poses, int n_poses, float fov, int num_rays,
                 float *ranges, int nthreads);
void orc_lut_pose_cells(const orc_map *m, const float *poses, int n_poses, int *rows_out, int *cols_out);
void orc_lut_fan_rows(const orc_map *m, const uint16_t *pose_rows /* n_poses*theta_disc */,
                      int theta_disc, float max_range_px, const float *poses, int n_poses,
                      float fov, int num_rays, float *ranges);
void orc_lut_rays(const orc_map *m, const uint16_t *lut, int theta_disc, float max_range_px,
                  const float *ins, int n, float *ranges, int nthreads);

/* ---- CDDTCast (row a13) --------------------------------------------------- */
typedef struct orc_cddt {
    int theta_disc;          /* bins over [0, 2pi); only bins < theta_disc/2 are stored */
    int n_bins;              /* ceil(theta_disc/2)                                      */
    int *lut_width;          /* per bin: number of buckets                              */
    float *lut_translation;  /* per bin                                                 */
    float *cosv, *sinv;      /* per bin: orc_sincosf(bin * 2pi/theta_disc)              */
    int64_t *bucket_off;     /* per bin: first bucket index in offsets[]                */
    int64_t *offsets;        /* CSR: bucket -> [start,end) in xs                        */
    float *xs;               /* sorted unique lut-space x of edge-cell centres          */
    int64_t n_buckets, n_xs;
    int literal;             /* 1: built by orc_cddt_build_libm (libm trig, un-fused projection, upstream's bin rule) */
} orc_cddt;
orc_cddt *orc_cddt_build(const orc_map *m, int theta_disc);
/* upstream-literal statements of the table methods (libm cosf / sinf per bin, un-fused projection, roundf bin rule):
 * how far the canonical tables (== the device's, bit for bit) are from a libm build — tests/test_oracle.py,
 * tests/golden/table_libm_forms.npz, tests/test_gpu_parity.py::test_device_tables_vs_upstream_literal_libm_forms */
orc_cddt *orc_cddt_build_libm(const orc_map *m, int theta_disc);
void orc_cddt_rays_libm(const orc_map *m, const orc_cddt *c, float max_range_px, const float *ins, int n, float *ranges);
void orc_lut_build_libm(const orc_map *m, const float *dt, float max_range_px, float step_coeff, int theta_disc,
                        int r0, int r1, uint16_t *lut, int nthreads);
void orc_lut_fan_rows_libm(const orc_map *m, const uint16_t *pose_rows, int theta_disc, float max_range_px,
                           const float *poses, int n_poses, float fov, int num_rays, float *ranges);
void orc_cddt_free(orc_cddt *c);
void orc_cddt_fan(const orc_map *m, const orc_cddt *c, float max_range_px,
                  const float *poses, int n_poses, float fov, int num_rays,
                  float *ranges, int nthreads);
void orc_cddt_rays(const orc_map *m, const orc_cddt *c, float max_range_px,
                   const float *ins, int n, float *ranges, int nthreads);

/* ---- consumers of ranges ("next" rows) ----------------------- */
/* Car::setCarEdgeDistances racecar/src/racecar.cpp:239-292 (PI=3.145, table
 * shifted by one increment) and Car::isCrashed :305-328.                       */
void orc_edge_distances(int num_rays, double min_ang, double inc, double scan_dist_to_base,
                        double width, double wheelbase, double *edge);
/* FollowGap::eval (followgap/followgap.hpp:104-129); NaN when size < 10 */
float orc_followgap_eval(const float *lidar, int size, float max_distance, float max_angle,
                         float angle_inc);
int orc_is_crashed(const float *rays, int num_rays, int poses, const double *edge,
                   double crash_thresh);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
