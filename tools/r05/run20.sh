#!/bin/bash
set -u
OUT=gpurun_out/r05_run20; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for pr in 0 2 3; do
python tools/r05/ab_lone.py drain_prio=$pr 2>&1 | grep -v amdgpu.ids | grep "slots 2"
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs --opt drain_prio=$pr > $OUT/bench300_prio${pr}_$rep.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --opt drain_prio=$pr > $OUT/bench20_prio${pr}_$rep.json 2> /dev/null
done
done | tee $OUT/ab_lone.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_run20/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"])
    except Exception as e: print(f, "ERR", e)
PY
