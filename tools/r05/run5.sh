#!/bin/bash
set -u
OUT=gpurun_out/r05_run5; mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== cur group_drain=1"; python tools/gpu_stamps_pipe.py --pipeline 1 --grid-mult 8 --slots 2 --steps 12 --opt group_drain=1
echo "== cur group_drain=0"; python tools/gpu_stamps_pipe.py --pipeline 1 --grid-mult 8 --slots 2 --steps 12 --opt group_drain=0
echo "== r5a"; SCANLIB_SO=$PWD/tools/ab/libscan_amd_r5a.so python tools/gpu_stamps_pipe.py --pipeline 1 --grid-mult 8 --slots 2 --steps 12
} > $OUT/stamps.txt 2>&1
grep -v amdgpu.ids $OUT/stamps.txt
