// crash_kernels.h — Car::isCrashed over finished ranges and the reduction of the march kernels' per-pose crash marks
// (the fused test itself lives in the march kernels, rm_kernels.h).  Part of abi_fan.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scan {

// per-group first crashed pose over finished ranges (Car::isCrashed racecar.cpp:305-328 applied to
// each roll-out of a batch): first[g] = min{k : exists j, (double)r[(g*G+k)*B + j] - edge[j] < thresh}
// or -(G+1).  Called with group = 0 (per-pose marks, see CrashParams); crash_reduce_kernel then finds
// every group's first marked pose.
__global__ __launch_bounds__(256) void crash_groups_kernel(const float *__restrict__ ranges,
                                                           const double *__restrict__ edge,
                                                           double thresh, int n_poses, int num_rays,
                                                           int group, int mark, int *__restrict__ first)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    for (int p = wave; p < n_poses; p += n_waves) {
        bool crashed = false;
        for (int j = lane; j < num_rays; j += 64)
            crashed |= ((double)ranges[(size_t)p * num_rays + j] - edge[j]) < thresh;
        if (__ballot(crashed) && lane == 0) {
            if (group == 0) first[p] = mark;
            else atomicMin(&first[p / group], p % group);
        }
    }
}

// per-pose crash marks -> first crashed pose of every group (roll-out): one wave per group.
// The march kernels mark POSES (pose_mark[p] = this launch's epoch, each pose its own word, so the
// array never needs clearing): a word per group was
// the target of thousands of same-address atomics, which the L2 retires one at a time (~10 per us)
// — measured +100 % kernel time at 4096 poses in 32 groups, +9 % with a word per pose.
__global__ __launch_bounds__(256) void crash_reduce_kernel(const int *__restrict__ pose_mark, int mark,
                                                           int n_groups, int group, int *__restrict__ first)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    for (int g = wave; g < n_groups; g += n_waves) {
        int best = 0x7fffffff;
        const int *row = pose_mark + (size_t)g * group;
        for (int k = lane; k < group && best == 0x7fffffff; k += 256) {      // 4 independent loads per trip
            const int m0 = row[k];
            const int m1 = k + 64 < group ? row[k + 64] : mark - 1;
            const int m2 = k + 128 < group ? row[k + 128] : mark - 1;
            const int m3 = k + 192 < group ? row[k + 192] : mark - 1;
            best = m0 == mark ? k : m1 == mark ? k + 64 : m2 == mark ? k + 128 : m3 == mark ? k + 192 : best;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off));
        if (lane == 0) first[g] = best == 0x7fffffff ? -(group + 1) : best;
    }
}

__global__ void fill_int_kernel(int *p, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace scan
