#!/bin/bash
# round 6, run 24: strengthened FollowGap parity (both kernels), then the steer timeline with followgap_bits_kernel
set -u
OUT=gpurun_out/r06_run24; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -x -q -k "followgap or steer" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
RL_FOLLOWGAP_WALK=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "followgap" > $OUT/pytest_walk.txt 2>&1; tail -3 $OUT/pytest_walk.txt
for mode in pipe serial; do
  O2=$PWD/$OUT/$mode; mkdir -p $O2
  A="--gather steer --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-other-configs"
  [ $mode = serial ] && A="$A --pipeline 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O2/trace -- python3 bench.py $A > $O2/bench.json 2> $O2/bench.err
  find $O2/trace -name '*kernel_stats.csv' -exec cp {} $O2/kernel_stats.csv \;
  rm -rf $O2/trace
  head -5 $O2/kernel_stats.csv | cut -c1-60,150-330
done
