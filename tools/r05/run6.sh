#!/bin/bash
set -u
OUT=gpurun_out/r05_run6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "literal or audit or schedule or crash or tables or exact" > $OUT/pytest_subset.txt 2>&1
tail -15 $OUT/pytest_subset.txt
timeout 600 python tools/r05/group_drain_ab.py > $OUT/group_drain_ab.txt 2>&1
grep -v amdgpu.ids $OUT/group_drain_ab.txt
timeout 300 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-other-configs > $OUT/bench300.json 2> $OUT/bench300.err
timeout 300 python bench.py --gpus 1 --steps 300 --warmup 15 --cpu-seconds 2 --variant 3 --no-other-configs > $OUT/bench300_variant3.json 2> $OUT/bench300_variant3.err
timeout 300 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --pipeline 1 --variant 3 > $OUT/bench300_variant3_serial.json 2> $OUT/bench300_variant3_serial.err
for f in $OUT/bench*.json; do echo -n "$f: "; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("verified"), d.get("literal_mode"), d["config"].get("kernel"))
except Exception as e: print("ERR", e)
PY
done
tail -3 $OUT/bench300_variant3.err
