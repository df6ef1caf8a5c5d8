#!/bin/bash
# round 6, run 19: the INEXACT 8-bit code-map probe (code_map 1: the 253 most frequent steps exact, the rest rounded down) against
# the u16 code map and the float32 map — what an escape-free u8 map would buy in cache lines (no verification: results differ)
set -u
OUT=gpurun_out/r06_run19; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-other-configs --no-verify"
for rep in 1 2 3; do
for cm in 2 1; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --bursts 9 --opt code_map=$cm > $OUT/cfg2_s300_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 20 --warmup 5 --opt code_map=$cm > $OUT/cfg2_s20_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 --bursts 7 --opt code_map=$cm > $OUT/cfg5s_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --poses 32768 --steps 40 --warmup 5 --bursts 7 --opt code_map=$cm > $OUT/cfg2_32k_cm${cm}_$rep.json 2>> $OUT/err.txt
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run19/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms lone %.4f  %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["roofline"]["serial"]["kernel_ms"], d["roofline"]["kernel"][-22:]))
    except Exception as e: print(f, "ERR", e)
PY
tail -2 $OUT/err.txt
