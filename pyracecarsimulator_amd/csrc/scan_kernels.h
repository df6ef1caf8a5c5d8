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
//
// The kernels live in one header per family:
//   edt_kernels.h   K0   exact EDT
//   rm_kernels.h    K1 / K1b ray marching: chunk-per-wave kernels, pose binning, step map, the hand-scheduled
//                   march and drain loops, the persistent stream kernel
//   lut_kernels.h   K3   GiantLUT build + fan kernels
//   bl_kernels.h    K2 / K2b Bresenham, occ_fan_lds
//   cddt_kernels.h  K3b  CDDT table build, pose-major and theta-major fan kernels
//   literal_kernels.h    the audit mode of the ray-marching methods: upstream-literal arithmetic with glibc's sinf / cosf
#pragma once
#include "scan_device.h"
#include "edt_kernels.h"
#include "rm_kernels.h"
#include "lut_kernels.h"
#include "bl_kernels.h"
#include "cddt_kernels.h"
#include "literal_kernels.h"
