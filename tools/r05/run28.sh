#!/bin/bash
set -u
OUT=gpurun_out/r05_run28; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1400 python tests/gpu_fuzz.py --seconds 1200 --seed 777001 > $OUT/fuzz_20min.log 2>&1; tail -3 $OUT/fuzz_20min.log
