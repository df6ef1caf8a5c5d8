#!/bin/bash
set -u
OUT=gpurun_out/r05_run27; mkdir -p $OUT
export TMPDIR=/tmp
timeout 4800 python tests/gpu_fuzz.py --seconds 4500 --seed 777001 > $OUT/fuzz_75min.log 2>&1; tail -2 $OUT/fuzz_75min.log
