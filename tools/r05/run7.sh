#!/bin/bash
# round 5: the driver's command, the 300-step line, kernel traces and PMC passes of the headline kernel and of the literal mode
set -u
OUT=gpurun_out/r05_prof; mkdir -p $OUT
export TMPDIR=/tmp
S=$(date +%s.%N)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err
E=$(date +%s.%N); echo "driver command wall seconds: $(echo "$E - $S" | bc)" > $OUT/driver_cmd_wall.txt
timeout 300 python bench.py --gpus 1 --steps 300 --warmup 15 --no-other-configs --cpu-seconds 2 > $OUT/cfg2_300.json 2> $OUT/cfg2_300.err
bash tools/prof_kernel_trace.sh r05_prof/kt_cfg2_driver_cmd --steps 20 --warmup 5 --no-other-configs --no-extras > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r05_prof/kt_cfg2_serial --pipeline 1 --steps 300 --no-other-configs --no-extras > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r05_prof/kt_cfg2_literal --steps 300 --variant 3 --no-other-configs --no-extras > /dev/null 2>&1
bash tools/prof_pmc.sh r05_prof/pmc_cfg2_slots2 > /dev/null 2>&1
bash tools/prof_pmc.sh r05_prof/pmc_cfg2_literal --variant 3 > /dev/null 2>&1
python tools/trace_overlap.py $OUT/kt_cfg2_driver_cmd/kernel_trace_tail.csv > $OUT/kt_cfg2_driver_cmd/overlap.txt 2>&1
cat $OUT/driver_cmd_wall.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_prof/driver_cmd.json').read().strip().splitlines()[-1])
print("driver:", d["value"], d["ms_per_step"], d["verified"], "roofline frac", d["roofline"]["frac"])
print("literal_mode:", d.get("literal_mode"))
for k,v in d.get("other_configs",{}).items(): print(k, {q:v.get(q) for q in ("mrays_s","ms_per_step","frac","frac_hbm","verified","leg_seconds","error")}, v.get("vs_exact_rm"))
d=json.loads(open('gpurun_out/r05_prof/cfg2_300.json').read().strip().splitlines()[-1])
print("300:", d["value"], d["ms_per_step"], d["verified"], d["roofline"]["frac"], d.get("literal_mode",{}).get("value"))
PY
head -6 $OUT/kt_cfg2_serial/kernel_stats.csv; head -6 $OUT/kt_cfg2_driver_cmd/kernel_stats.csv; head -5 $OUT/kt_cfg2_literal/kernel_stats.csv; cat $OUT/kt_cfg2_driver_cmd/overlap.txt
python - <<'PY'
import json
for t in ("pmc_cfg2_slots2","pmc_cfg2_literal"):
    d=json.load(open('gpurun_out/r05_prof/%s/pmc_summary.json'%t))
    for k,v in d.items():
        if 'rm_fan_stream' in k: print(t,k,{c:round(x) for c,x in v.items() if c in ('SQ_INSTS_VALU','SQ_WAIT_ANY','SQ_WAVE_CYCLES','FETCH_SIZE','WRITE_SIZE','SQ_INSTS_VMEM_RD','TCP_TOTAL_CACHE_ACCESSES_sum','TCC_HIT_sum','TCC_MISS_sum','_dispatches')})
PY
