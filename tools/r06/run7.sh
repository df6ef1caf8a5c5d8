#!/bin/bash
# round 6, run 7: low_water / grid / depth sweep on the code-map kernel (cfg2 300 steps, 20 steps; cfg5 shard)
set -u
OUT=gpurun_out/r06_run7; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-other-configs --no-verify"
for lw in 12 16 20 24 28 36; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --bursts 9 --opt low_water=$lw > $OUT/cfg2_s300_lw$lw.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 --bursts 7 --opt low_water=$lw > $OUT/cfg5s_lw$lw.json 2>> $OUT/err.txt
done
for gm in 2 3 4; do for P in 3 4 5 6; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --bursts 9 --grid-mult $gm --pipeline $P > $OUT/cfg2_s300_gm${gm}_P$P.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 20 --warmup 5 --bursts 15 --grid-mult $gm --pipeline $P > $OUT/cfg2_s20_gm${gm}_P$P.json 2>> $OUT/err.txt
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run7/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -2 $OUT/err.txt
