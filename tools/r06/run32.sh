#!/bin/bash
# round 6, run 32: the literal kernel's service round is expensive (glibc sinf / cosf per claimed ray under a partial EXEC mask):
# does it want a lower low_water (fewer, fuller service rounds) than the canonical kernel's 20?
set -u
OUT=gpurun_out/r06_run32; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-other-configs --method RM"
for rep in 1 2; do
for lw in 4 8 12 16 20 28; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --opt low_water=$lw > $OUT/rm_s300_lw${lw}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --pipeline 1 --steps 100 --warmup 10 --opt low_water=$lw > $OUT/rm_serial_lw${lw}_$rep.json 2>> $OUT/err.txt
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run32/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms ver %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
