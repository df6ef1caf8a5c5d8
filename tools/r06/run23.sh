#!/bin/bash
# round 6, run 23: followgap_bits_kernel (one bit per beam) vs followgap_kernel (RL_FOLLOWGAP_WALK=1): parity, then steer A/B
set -u
OUT=gpurun_out/r06_run23; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -x -q -k "followgap or steer" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-extras --no-other-configs --gather steer"
for rep in 1; do
for w in 0 1; do
  export RL_FOLLOWGAP_WALK=$w
  timeout 200 python bench.py $B --steps 300 --warmup 20 > $OUT/steer_s300_walk${w}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 20 --warmup 5 > $OUT/steer_s20_walk${w}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --pipeline 1 --steps 100 --warmup 10 > $OUT/steer_serial_walk${w}_$rep.json 2>> $OUT/err.txt
done
done
unset RL_FOLLOWGAP_WALK
timeout 200 python bench.py --no-cpu-baseline --no-extras --no-other-configs --steps 300 --warmup 20 > $OUT/plain_s300.json 2>> $OUT/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run23/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms ver %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
