#!/bin/bash
set -u
OUT=gpurun_out/r05_run3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "schedule or crash or golden or edge" > $OUT/pytest_subset.txt 2>&1
tail -5 $OUT/pytest_subset.txt
timeout 600 python tools/r05/group_drain_ab.py > $OUT/group_drain_ab.txt 2>&1
cat $OUT/group_drain_ab.txt
for gd in 0 1; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt group_drain=$gd > $OUT/bench20_gd$gd.json 2> $OUT/bench20_gd$gd.err
  timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --opt group_drain=$gd > $OUT/bench300_gd$gd.json 2> $OUT/bench300_gd$gd.err
  timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --pipeline 1 --opt group_drain=$gd > $OUT/bench300_serial_gd$gd.json 2> $OUT/bench300_serial_gd$gd.err
done
for f in $OUT/bench*.json; do echo -n "$f: "; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("verified"))
except Exception as e: print("ERR", e)
PY
done
timeout 400 python tests/gpu_fuzz.py --seconds 240 --seed 505 > $OUT/fuzz_240s.log 2>&1; tail -3 $OUT/fuzz_240s.log
