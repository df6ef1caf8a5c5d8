#!/bin/bash
# round 6, run 25 (followgap_bits_kernel in): steer lines, the driver's command with its legs, the steer PMC pass, resources
set -u
OUT=gpurun_out/r06_run25; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-other-configs"
timeout 300 python bench.py $B --gather steer > $OUT/cfg2_steer.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --gather steer --steps 300 --warmup 15 > $OUT/cfg2_steer_300.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --gather steer --pipeline 1 > $OUT/cfg2_steer_serial.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --gather steer --workload cfg4 --poses 131072 --steps 10 --warmup 2 > $OUT/cfg4_shard_steer.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --steps 300 --warmup 15 > $OUT/cfg2_300.json 2>> $OUT/err.txt
for i in 1 2; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2>> $OUT/err.txt; done
bash tools/prof_pmc.sh r06_run25/pmc_cfg2_steer --no-other-configs --grid-mult 3 --opt slots=2 --gather steer > /dev/null 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run25/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-22s %10.0f [%8.0f..%8.0f] %.4f ms frac %.3f ver %s" % (f.split('/')[-1][:-5], d["value"], d["value_min"], d["value_max"], d["ms_per_step"], d["roofline"]["frac"], d["verified"]))
        oc=d.get("other_configs")
        if oc: print("   legs:", {k:(round(v.get("value",0)) if isinstance(v,dict) else v) for k,v in oc.items()})
        if "steer_mode" in d: print("   steer_mode", d["steer_mode"].get("value"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
