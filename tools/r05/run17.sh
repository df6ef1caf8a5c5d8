#!/bin/bash
set -u
OUT=gpurun_out/r05_run17; mkdir -p $OUT
export TMPDIR=/tmp
python tools/r05/small_batch_sweep.py cfg2 1,16,64,128,200,300,400,600 2>&1 | grep -v amdgpu.ids | tee $OUT/small_batch_cfg2.txt
python tools/r05/small_batch_sweep.py cfg5 1,64,200,400 2>&1 | grep -v amdgpu.ids | tee $OUT/small_batch_cfg5.txt
python tools/r05/small_batch_sweep.py cfg4 1,64,200,400 2>&1 | grep -v amdgpu.ids | tee $OUT/small_batch_cfg4.txt
