#!/bin/bash
set -u
OUT=gpurun_out/r06_last; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "two_player" > $OUT/pytest_two_player.txt 2>&1; tail -2 $OUT/pytest_two_player.txt
python tools/r06/tick_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cddt_latency.txt
