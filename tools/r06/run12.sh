#!/bin/bash
# round 6, run 12: stripe / order A/B (run11), the bench matrix of the final build, the driver's command with its legs, bench-shape tests
set -u
export TMPDIR=/tmp
bash tools/r06/run11.sh
bash tools/bench_matrix.sh r06_bench > gpurun_out/r06_bench_summary.txt 2>&1
mv gpurun_out/r06_bench_summary.txt gpurun_out/r06_bench/SUMMARY.txt
cat gpurun_out/r06_bench/SUMMARY.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench/driver_cmd_full.json 2> gpurun_out/r06_bench/driver_cmd_full.err; echo "driver rc $?"
timeout 1500 python -m pytest tests/test_gpu_bench_shape.py -x -q > gpurun_out/r06_bench/pytest_bench_shape.txt 2>&1; tail -5 gpurun_out/r06_bench/pytest_bench_shape.txt
