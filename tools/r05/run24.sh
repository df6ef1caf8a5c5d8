#!/bin/bash
set -u
OUT=gpurun_out/r05_run24; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd_$rep.json 2> $OUT/driver_cmd.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r05_run24/driver_cmd_$rep.json').read().strip().splitlines()[-1])
print("driver:", d["value"], d["ms_per_step"], d["verified"])
for k,v in d.get("other_configs",{}).items(): print(k, {q:v.get(q) for q in ("mrays_s","ms_per_step","frac_hbm","verified","leg_seconds")})
PY
done
timeout 600 python -m pytest tests/test_gpu_bench_shape.py -x -q 2>&1 | tail -2
