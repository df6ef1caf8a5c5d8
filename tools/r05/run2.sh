#!/bin/bash
set -u
OUT=gpurun_out/r05_run2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/r05/handoff_sweep.py cfg2 4096 > $OUT/handoff_sweep.txt 2>&1
for ho in 0 1; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt handoff=$ho > $OUT/bench20_ho$ho.json 2> $OUT/bench20_ho$ho.err
  timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --opt handoff=$ho > $OUT/bench300_ho$ho.json 2> $OUT/bench300_ho$ho.err
done
for cap in 8 32; do
  timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --opt handoff=1 --opt handoff_cap=$cap > $OUT/bench300_ho1_cap$cap.json 2> $OUT/bench300_ho1_cap$cap.err
done
for gm in 4 5; do
  timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --opt handoff=1 --grid-mult $gm > $OUT/bench300_ho1_gm$gm.json 2> $OUT/bench300_ho1_gm$gm.err
done
timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --opt handoff=1 --pipeline 1 > $OUT/bench300_ho1_serial.json 2> $OUT/bench300_ho1_serial.err
timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --opt handoff=0 --pipeline 1 > $OUT/bench300_ho0_serial.json 2> $OUT/bench300_ho0_serial.err
cat $OUT/handoff_sweep.txt
for f in $OUT/bench*.json; do echo -n "$f: "; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("verified"))
except Exception as e: print("ERR", e)
PY
done
tail -3 $OUT/*.err | head -60
