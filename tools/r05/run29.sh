#!/bin/bash
set -u
OUT=gpurun_out/r05_run29; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err
timeout 300 python bench.py --workload cfg3 --method CDDT --steps 40 --no-cpu-baseline --no-other-configs > $OUT/cfg3_CDDT.json 2>/dev/null
timeout 300 python bench.py --workload cfg3 --method CDDT --steps 40 --pipeline 1 --no-cpu-baseline --no-other-configs > $OUT/cfg3_CDDT_serial.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_run29/driver_cmd.json').read().strip().splitlines()[-1])
print("driver:", d["value"], d["ms_per_step"], d["verified"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for k,v in d.get("other_configs",{}).items(): print(k, {q:v.get(q) for q in ("mrays_s","ms_per_step","frac_hbm","verified")})
for f in ("cfg3_CDDT","cfg3_CDDT_serial"):
    d=json.loads(open('gpurun_out/r05_run29/%s.json'%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"].get("frac_hbm"), d["verified"])
PY
