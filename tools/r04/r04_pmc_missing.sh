#!/bin/bash
# round 4: PMC passes for the BASELINE configs whose roofline.traffic was null (VERDICT r03 missing #4)
set -u
mkdir -p gpurun_out/r04_base
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_base/bench_steps20.json 2> gpurun_out/r04_base/bench_steps20.err
python bench.py --pipeline 1 --no-cpu-baseline > gpurun_out/r04_base/bench_serial.json 2> gpurun_out/r04_base/bench_serial.err
bash tools/prof_pmc.sh r04_pmc_cfg4_shard --workload cfg4 --poses 131072 > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg5_shard --workload cfg5 --poses 32768 > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg3_rmgpu --workload cfg3 --method RMGPU > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg2_bl --method BL > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg4_4096 --workload cfg4 --poses 4096 > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg5 --workload cfg5 > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg4_1M --workload cfg4 > /dev/null
ls gpurun_out/r04_pmc_*
