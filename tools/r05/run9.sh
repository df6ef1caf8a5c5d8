#!/bin/bash
set -u
OUT=gpurun_out/r05_run9; mkdir -p $OUT
export TMPDIR=/tmp
python tools/r05/cddt_search_ab.py > $OUT/cddt_search_ab.txt 2>&1; grep -v amdgpu.ids $OUT/cddt_search_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cddt" > $OUT/pytest_cddt.txt 2>&1; tail -4 $OUT/pytest_cddt.txt
PAT="cddt_" KEEP=60 bash tools/prof_trace_cmd.sh r05_run9/kt_cfg3_cddt bench.py --workload cfg3 --method CDDT --pipeline 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
head -8 $OUT/kt_cfg3_cddt/kernel_stats.csv
timeout 300 python bench.py --workload cfg3 --method CDDT --steps 60 --warmup 8 --no-cpu-baseline --no-extras > $OUT/cfg3_cddt_pipe.json 2> $OUT/cfg3_cddt_pipe.err
timeout 300 python bench.py --workload cfg3 --method CDDT --steps 60 --warmup 8 --no-cpu-baseline --no-extras --pipeline 1 > $OUT/cfg3_cddt_serial.json 2> $OUT/cfg3_cddt_serial.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err
python - <<'PY'
import json
for f in ("cfg3_cddt_pipe","cfg3_cddt_serial"):
    d=json.loads(open('gpurun_out/r05_run9/%s.json'%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["verified"], d["config"].get("kernel"))
d=json.loads(open('gpurun_out/r05_run9/driver_cmd.json').read().strip().splitlines()[-1])
print("driver:", d["value"], d["ms_per_step"], d["verified"], "roofline frac", d["roofline"]["frac"], d.get("roofline_tcp"))
for k,v in d.get("other_configs",{}).items(): print(k, {q:v.get(q) for q in ("mrays_s","ms_per_step","frac","frac_hbm","verified","leg_seconds","error")})
PY
