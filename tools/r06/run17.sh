#!/bin/bash
set -u
OUT=gpurun_out/r06_run17; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "falls_back or two_player" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
python tools/r06/cddt_sorted_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cddt_sorted_probe.txt
bash tools/prof_trace_cmd.sh r06_run17/kt_sorted tools/r06/cddt_sorted_probe.py > /dev/null 2>&1
grep "cddt_theta" $OUT/kt_sorted/kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
