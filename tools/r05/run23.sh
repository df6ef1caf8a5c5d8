#!/bin/bash
# last check of the round's final tree: GPU suite, smoke(), the driver's command
set -u
OUT=gpurun_out/r05_run23; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_run23/driver_cmd.json').read().strip().splitlines()[-1])
print("driver:", d["value"], d["ms_per_step"], d["verified"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_source"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
for k,v in d.get("other_configs",{}).items(): print(k, {q:v.get(q) for q in ("mrays_s","ms_per_step","frac","frac_hbm","verified")})
print("literal_mode", d.get("literal_mode",{}).get("value"), d.get("literal_mode",{}).get("vs_canonical"))
PY
