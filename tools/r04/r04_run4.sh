#!/bin/bash
set -u
O=gpurun_out/r04_run4; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "schedule or cfg2 or edge or golden or crash" > $O/pytest_a.log 2>&1; echo "a rc=$?" >> $O/rc.txt
echo "== serial cfg2: cur vs r4base" > $O/ab.txt
bash tools/ab_env.sh tools/ab/libscan_amd_r4base.so 3 --pipeline 1 --no-extras >> $O/ab.txt 2>&1
echo "== pipelined cfg2 20 steps: cur vs r4base" >> $O/ab.txt
bash tools/ab_env.sh tools/ab/libscan_amd_r4base.so 3 --steps 20 --warmup 5 --no-extras >> $O/ab.txt 2>&1
echo "== pipelined cfg2 300 steps" >> $O/ab.txt
bash tools/ab_env.sh tools/ab/libscan_amd_r4base.so 2 --no-extras >> $O/ab.txt 2>&1
echo "== serial 2048 / 8000 poses, cfg5 4096 poses" >> $O/ab.txt
bash tools/ab_env.sh tools/ab/libscan_amd_r4base.so 2 --pipeline 1 --poses 2048 --no-extras >> $O/ab.txt 2>&1
bash tools/ab_env.sh tools/ab/libscan_amd_r4base.so 2 --pipeline 1 --poses 8000 --no-extras >> $O/ab.txt 2>&1
bash tools/ab_env.sh tools/ab/libscan_amd_r4base.so 2 --pipeline 1 --workload cfg5 --poses 4096 --no-extras >> $O/ab.txt 2>&1
cat $O/rc.txt; tail -3 $O/pytest_a.log; cat $O/ab.txt
