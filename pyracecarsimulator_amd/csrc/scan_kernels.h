// scan_kernels.h — gfx950 kernels of the lidar scan path.
//
// Product path of the ray-marching methods: K1b, rm_fan_stream_kernel (rm_kernels.h) — a persistent grid of 1024-lane
// workgroups, each draining ONE stream of 64-ray blocks through an LDS counter with lane refill, two rays per lane, the
// tile-ordered pose list cut into one band per XCD, a hand-scheduled gfx950 march loop on the 4x8-cell-tiled step map
// (DESIGN.md section 4).  K1 (rm_fan_kernel: a chunk of 64 consecutive beams per wave, no refill) is variant 0, the
// round-1 shape kept as an A/B partner and for launches the stream kernel cannot index (>= 2^30 rays without slicing).
//
// The kernels live in one header per family (this header = what abi_fan.hip launches; edt_kernels.h — K0, the exact EDT, the
// bit map and the edge-cell list — belongs to abi_map.hip):
//   rm_kernels.h    K1 / K1b ray marching: chunk-per-wave kernels, pose binning, step map, the hand-scheduled
//                   march and drain loops, the persistent stream kernel
//   lut_kernels.h   K3   GiantLUT build + fan kernels
//   bl_kernels.h    K2 / K2b Bresenham, occ_fan_lds
//   cddt_kernels.h  K3b  CDDT table build, pose-major and theta-major fan kernels
//   literal_kernels.h    the audit mode of the ray-marching methods: upstream-literal arithmetic with glibc's sinf / cosf
#pragma once
#include "scan_device.h"
#include "rm_kernels.h"
#include "lut_kernels.h"
#include "bl_kernels.h"
#include "cddt_kernels.h"
#include "literal_kernels.h"
