#!/bin/bash
# round-5 evidence run on the final build: GPU suite, driver command, bench matrix, kernel traces, PMC passes, long fuzz
set -u
OUT=gpurun_out/r05_final; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err
bash tools/bench_matrix.sh r05_final/bench > $OUT/bench_SUMMARY.txt 2>&1
B="--no-cpu-baseline --no-other-configs"
python bench.py $B --steps 300 --warmup 20 > $OUT/bench/cfg2_300steps.json 2> /dev/null
python bench.py $B --variant 3 --steps 300 --warmup 20 > $OUT/bench/cfg2_variant3_literal.json 2> /dev/null
python bench.py $B --variant 3 --pipeline 1 > $OUT/bench/cfg2_variant3_literal_serial.json 2> /dev/null
python bench.py $B --gather crash > $OUT/bench/cfg2_crash.json 2> /dev/null
python bench.py $B --gather steer > $OUT/bench/cfg2_steer.json 2> /dev/null
tail -30 $OUT/bench_SUMMARY.txt
PAT="rm_fan|pose_" KEEP=80 bash tools/prof_trace_cmd.sh r05_final/kt_cfg2_driver_cmd bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs > /dev/null 2>&1
PAT="rm_fan|pose_" KEEP=80 bash tools/prof_trace_cmd.sh r05_final/kt_cfg2_serial bench.py --pipeline 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --no-verify > /dev/null 2>&1
PAT="rm_fan|pose_" KEEP=80 bash tools/prof_trace_cmd.sh r05_final/kt_cfg2_literal bench.py --variant 3 --pipeline 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --no-verify > /dev/null 2>&1
PAT="cddt_" KEEP=60 bash tools/prof_trace_cmd.sh r05_final/kt_cfg3_cddt bench.py --workload cfg3 --method CDDT --pipeline 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --no-verify > /dev/null 2>&1
head -4 $OUT/kt_cfg2_driver_cmd/kernel_stats.csv $OUT/kt_cfg2_serial/kernel_stats.csv $OUT/kt_cfg2_literal/kernel_stats.csv $OUT/kt_cfg3_cddt/kernel_stats.csv
bash tools/prof_pmc.sh r05_final/pmc_cfg2_slots2 --no-other-configs --grid-mult 3 --opt slots=2 > /dev/null 2>&1
bash tools/prof_pmc.sh r05_final/pmc_cfg2_literal --no-other-configs --grid-mult 3 --opt slots=2 --variant 3 > /dev/null 2>&1
bash tools/prof_pmc.sh r05_final/pmc_cfg3_cddt --no-other-configs --workload cfg3 --method CDDT --pipeline 1 > /dev/null 2>&1
python - <<'PY'
import json
for t in ("pmc_cfg2_slots2","pmc_cfg2_literal","pmc_cfg3_cddt"):
    try:
        d=json.load(open('gpurun_out/r05_final/%s/pmc_summary.json'%t))
        for k,v in d.items():
            if 'rm_fan_stream' in k or 'cddt_theta' in k: print(t,k,{c:round(x) for c,x in v.items() if c in ('SQ_INSTS_VALU','SQ_WAIT_ANY','SQ_WAVE_CYCLES','FETCH_SIZE','WRITE_SIZE','SQ_INSTS_VMEM_RD','TCP_TOTAL_CACHE_ACCESSES_sum','_dispatches')})
    except Exception as e: print(t, "ERR", e)
PY
timeout 2000 python tests/gpu_fuzz.py --seconds 1800 --seed 5050 > $OUT/fuzz_30min.log 2>&1; tail -2 $OUT/fuzz_30min.log
