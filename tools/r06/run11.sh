#!/bin/bash
# round 6, run 11: does the tile order still pay on the code map?  cfg2 serial / pipelined with the keys-only binning launch
# (default), row stripes compacted in the march kernel (stripe_max 8192: no binning launch), caller's order (sort_poses 0)
set -u
OUT=gpurun_out/r06_run11; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-other-configs --no-verify"
for rep in 1 2; do
for o in "stripe_max=1536" "stripe_max=8192" "sort_poses=0"; do
  tag=$(echo $o | tr '=' '_')
  timeout 200 python bench.py $B --pipeline 1 --steps 100 --warmup 10 --opt $o > $OUT/serial_${tag}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 300 --warmup 20 --bursts 9 --opt $o > $OUT/s300_${tag}_$rep.json 2>> $OUT/err.txt
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run11/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-34s %10.0f  %.4f ms  lone %.4f %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["roofline"]["serial"]["kernel_ms"], d["roofline"]["kernel"][-40:]))
    except Exception as e: print(f, "ERR", e)
PY
