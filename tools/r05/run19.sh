#!/bin/bash
set -u
OUT=gpurun_out/r05_run19; mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python tests/gpu_fuzz.py --seconds 2700 --seed 90210 > $OUT/fuzz_45min.log 2>&1; tail -2 $OUT/fuzz_45min.log
