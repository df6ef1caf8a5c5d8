#!/bin/bash
set -u
OUT=gpurun_out/r05_run31; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2600 python tests/gpu_fuzz.py --seconds 2400 --seed 31337 > $OUT/fuzz_40min.log 2>&1; tail -2 $OUT/fuzz_40min.log
