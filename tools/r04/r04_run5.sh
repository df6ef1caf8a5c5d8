#!/bin/bash
# round 4, second session: evidence of the build with non-temporal range stores / GiantLUT row loads —
# bench matrix, PMC passes of the headline shapes, kernel traces of the driver's command and the serial schedule
set -u
bash tools/bench_matrix.sh r04b_bench > gpurun_out/r04b_bench_summary.txt 2>&1
O=gpurun_out/r04b_bench
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || echo "FAILED $name"; }
run cfg2_crash --gather crash
run cfg2_crash_steps20 --gather crash --steps 20 --warmup 5
run cfg2_steer --gather steer
run cfg4_shard131072_crash --workload cfg4 --poses 131072 --steps 40 --warmup 4 --gather crash
python bench.py --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err
bash tools/prof_pmc.sh r04b_pmc_cfg2_slots2 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg2_serial --pipeline 1 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg3_glt --workload cfg3 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg3_rmgpu --workload cfg3 --method RMGPU > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg4_shard --workload cfg4 --poses 131072 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg5_shard_pipe --workload cfg5 --poses 32768 > /dev/null
bash tools/prof_kernel_trace.sh r04b_kt_cfg2_driver_cmd --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r04b_kt_cfg2_serial --pipeline 1 --steps 100 > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r04b_kt_cfg3_glt --workload cfg3 --steps 60 > /dev/null 2>&1
cat gpurun_out/r04b_bench_summary.txt | tail -30
ls gpurun_out/r04b_*
