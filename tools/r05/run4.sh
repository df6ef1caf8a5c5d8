#!/bin/bash
set -u
OUT=gpurun_out/r05_run4; mkdir -p $OUT
export TMPDIR=/tmp
{
for r in 1 2; do
  python tools/r05/ab_lone.py group_drain=0
  python tools/r05/ab_lone.py group_drain=1
  SCANLIB_SO=$PWD/tools/ab/libscan_amd_r5a.so python tools/r05/ab_lone.py
  SCANLIB_SO=$PWD/tools/ab/libscan_amd_r4base.so python tools/r05/ab_lone.py
done
} > $OUT/ab_lone.txt 2>&1
grep -v amdgpu.ids $OUT/ab_lone.txt
FUZZ_TRACE=1 timeout 200 python tests/gpu_fuzz.py --seconds 120 --seed 505 > $OUT/fuzz_trace.log 2>&1; echo "fuzz all rc $?"; tail -4 $OUT/fuzz_trace.log
FUZZ_TRACE=1 FUZZ_NO_GROUP=1 timeout 200 python tests/gpu_fuzz.py --seconds 120 --seed 505 > $OUT/fuzz_nogroup.log 2>&1; echo "fuzz nogroup rc $?"; tail -4 $OUT/fuzz_nogroup.log
FUZZ_TRACE=1 FUZZ_NO_HANDOFF=1 timeout 200 python tests/gpu_fuzz.py --seconds 120 --seed 505 > $OUT/fuzz_nohandoff.log 2>&1; echo "fuzz nohandoff rc $?"; tail -4 $OUT/fuzz_nohandoff.log
FUZZ_TRACE=1 FUZZ_NO_HANDOFF=1 FUZZ_NO_GROUP=1 timeout 200 python tests/gpu_fuzz.py --seconds 120 --seed 505 > $OUT/fuzz_neither.log 2>&1; echo "fuzz neither rc $?"; tail -4 $OUT/fuzz_neither.log
