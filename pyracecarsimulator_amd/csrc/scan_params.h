// scan_params.h — kernel-parameter structs and launch constants the host handles keep (rl_method holds a LutParams,
// a CddtParams and a BlPad): shared by every translation unit of libscan_amd.so, no kernels in here.
#pragma once
#include "scan_device.h"

namespace scan {

// K3 GiantLUT (lut_kernels.h)
struct LutParams {
    uint16_t *lut;
    int theta_disc;
    float bins_per_rad;      // theta_disc / 2pi (float)
    float bin_width;         // 2pi / theta_disc
    float quant, dequant;    // 65535/max_range, max_range/65535
    int debug;               // diagnostics only: bit0 skip table loads, bit1 skip range stores
};

// K3b CDDT (cddt_kernels.h)
struct CddtParams {
    int theta_disc, n_bins;
    const float *cosv, *sinv, *trans;   // per bin
    const int *width;                   // per bin: buckets
    const uint32_t *bucket_off;         // per bin: first bucket (n_bins + 1)
    uint32_t *offsets;                  // per bucket: [start, end) in xs (n_buckets + 1)   (build intermediate)
    float *xs;                          // CSR values as projected, unsorted                 (build intermediate)
    // what the queries read: the blocked table.  A bucket of n values owns a run of 128-B lines starting at
    // line hdr[b].x: its values in LEAVES of 32 (sorted, the last one padded with +inf), and — more than one
    // leaf — in front of them the SEPARATORS, the first value of every leaf, 32 per line (padded with +inf).
    // A query reads the header, one separator line and one leaf line: two table lines instead of the 3.6 a
    // bisection over the packed CSR run touched, three dependent loads instead of eight.
    uint2 *hdr;                         // per bucket: {first line, n}
    float *tab;
    float bins_per_rad;
    int debug;                          // diagnostics only: bit0 skip the bucket searches, bit1 skip the range stores
};

constexpr uint32_t CDDT_LDS_SORT = 16384;     // buckets up to this many values are sorted in LDS (cddt_sort_kernel)

// K2b Bresenham on the stream machinery (bl_kernels.h): the padded normal + transposed bit maps
struct BlPad {
    const uint32_t *bits;       // both padded copies in one buffer
    uint32_t k_n, k_t;          // byte offset of the word holding cell (0, 0): normal / transposed copy
    int stride_n, stride_t;     // words per padded row
    float near;                 // origins with -near < g < dim + near are covered by the padding
};

// grid-wide pose binning (rm_kernels.h)
constexpr int POSES_PER_WG = 512;      // (2048 while one workgroup scanned all the counters; with the per-tile
                                       //  scan 256..1024 are equally good and 4..13 % ahead of that)

}  // namespace scan
