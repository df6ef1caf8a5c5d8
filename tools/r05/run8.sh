#!/bin/bash
set -u
OUT=gpurun_out/r05_run8; mkdir -p $OUT
export TMPDIR=/tmp
python tools/r05/steer_side_stream.py > $OUT/steer_side_stream.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/r05/steer_side_stream.py > $OUT/steer_side_stream_hwq8.txt 2>&1
grep -v amdgpu.ids $OUT/steer_side_stream.txt $OUT/steer_side_stream_hwq8.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_run8/driver_cmd.json').read().strip().splitlines()[-1])
print("driver:", d["value"], d["ms_per_step"], d["verified"], "roofline frac", d["roofline"]["frac"])
for k,v in d.get("other_configs",{}).items(): print(k, {q:v.get(q) for q in ("mrays_s","ms_per_step","frac","frac_hbm","verified","leg_seconds","error")})
PY
bash tools/prof_pmc.sh r05_run8/pmc_cfg2_slots2 --grid-mult 3 --opt slots=2 > /dev/null 2>&1
bash tools/prof_pmc.sh r05_run8/pmc_cfg2_literal --grid-mult 3 --opt slots=2 --variant 3 > /dev/null 2>&1
python - <<'PY'
import json
for t in ("pmc_cfg2_slots2","pmc_cfg2_literal"):
    d=json.load(open('gpurun_out/r05_run8/%s/pmc_summary.json'%t))
    for k,v in d.items():
        if 'rm_fan_stream' in k: print(t,k,{c:round(x) for c,x in v.items() if c in ('SQ_INSTS_VALU','SQ_WAIT_ANY','SQ_WAVE_CYCLES','FETCH_SIZE','WRITE_SIZE','SQ_INSTS_VMEM_RD','TCP_TOTAL_CACHE_ACCESSES_sum','_dispatches')})
PY
