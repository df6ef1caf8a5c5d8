#!/bin/bash
set -u
OUT=gpurun_out/r05_run11; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench300.json 2> $OUT/bench300.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_run11/driver_cmd.json').read().strip().splitlines()[-1])
print("driver:", d["value"], d["ms_per_step"], d["verified"], "roofline frac", d["roofline"]["frac"])
for k,v in d.get("other_configs",{}).items(): print(k, {q:v.get(q) for q in ("mrays_s","ms_per_step","frac","verified","leg_seconds","error")})
d=json.loads(open('gpurun_out/r05_run11/bench300.json').read().strip().splitlines()[-1]); print("300:", d["value"], d["ms_per_step"], d["verified"])
PY
