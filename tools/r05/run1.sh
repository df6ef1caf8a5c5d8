#!/bin/bash
# round 5, GPU call 1: atomic probe, run_log2 sweep, baseline bench lines
set -u
OUT=gpurun_out/r05_run1; mkdir -p $OUT
export TMPDIR=/tmp
hipcc -O2 --offload-arch=gfx950 -o /tmp/atomic_probe tools/probes/atomic_probe.hip && timeout 120 /tmp/atomic_probe > $OUT/atomic_probe.txt 2>&1
timeout 300 python tools/r05/runlog_sweep.py 4096 > $OUT/runlog_sweep.txt 2>&1
for rl in -1 0 2; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt run_log2=$rl > $OUT/bench20_rl$rl.json 2> $OUT/bench20_rl$rl.err
  timeout 200 python bench.py --gpus 1 --steps 300 --warmup 15 --no-cpu-baseline --no-extras --opt run_log2=$rl > $OUT/bench300_rl$rl.json 2> $OUT/bench300_rl$rl.err
done
tail -n 40 $OUT/atomic_probe.txt $OUT/runlog_sweep.txt
for f in $OUT/bench*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("verified"))
except Exception as e: print("ERR", e)
PY
done
