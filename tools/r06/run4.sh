#!/bin/bash
# round 6, run 4: the whole GPU suite (code map + RM literal defaults, device-resident multi-device exchange, stamp) + a short fuzz
set -u
OUT=gpurun_out/r06_run4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
timeout 420 python tests/gpu_fuzz.py --seconds 300 --seed 606 > $OUT/fuzz_300s.log 2>&1
tail -3 $OUT/fuzz_300s.log
