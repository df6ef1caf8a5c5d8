#!/bin/bash
# round 6, run 33: steer on big batches in pose chunks (scan chunk, FollowGap chunk, ... on one stream: the ranges FollowGap reads are
# still in the Infinity Cache) — experimental environment switch RL_STEER_CHUNK
set -u
OUT=gpurun_out/r06_run33; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-other-configs --gather steer"
for rep in 1 2; do
for ck in 0 8192 16384 32768; do
  export RL_STEER_CHUNK=$ck
  timeout 200 python bench.py $B --workload cfg4 --poses 131072 --steps 10 --warmup 2 > $OUT/cfg4s_steer_ck${ck}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 --pipeline 1 > $OUT/cfg5s_steer_serial_ck${ck}_$rep.json 2>> $OUT/err.txt
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run33/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-34s %10.0f  %.4f ms ver %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
