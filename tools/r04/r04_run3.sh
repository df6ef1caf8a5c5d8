#!/bin/bash
# round 4: evidence refresh on the current build — bench matrix, kernel traces, PMC of the headline shape
set -u
bash tools/bench_matrix.sh r04_bench > gpurun_out/r04_bench_summary.txt 2>&1
cp gpurun_out/r04_bench_summary.txt gpurun_out/r04_bench/SUMMARY.txt
python bench.py --no-cpu-baseline --gather crash > gpurun_out/r04_bench/cfg2_crash.json 2>/dev/null
python bench.py --no-cpu-baseline --gather steer > gpurun_out/r04_bench/cfg2_steer.json 2>/dev/null
python bench.py --no-cpu-baseline --gather crash --steps 20 --warmup 5 > gpurun_out/r04_bench/cfg2_crash_steps20.json 2>/dev/null
python bench.py --no-cpu-baseline --gather crash --workload cfg4 --poses 131072 --steps 40 --warmup 4 > gpurun_out/r04_bench/cfg4_shard131072_crash.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench/driver_cmd.json 2>/dev/null
KEEP=2500 bash tools/prof_kernel_trace.sh r04_kt_cfg2_driver_cmd --steps 20 --warmup 5 > /dev/null 2>&1
python tools/trace_overlap.py gpurun_out/r04_kt_cfg2_driver_cmd/kernel_trace_tail.csv > gpurun_out/r04_kt_cfg2_driver_cmd/overlap.txt 2>&1
bash tools/prof_kernel_trace.sh r04_kt_cfg2_serial --pipeline 1 > /dev/null 2>&1
bash tools/prof_pmc.sh r04_pmc_cfg2_slots2 --grid-mult 3 --opt slots=2 > /dev/null 2>&1
bash tools/prof_pmc.sh r04_pmc_cfg2_crash_slots2 --grid-mult 3 --opt slots=2 --gather crash > /dev/null 2>&1
cat gpurun_out/r04_bench_summary.txt | tail -30
