#!/bin/bash
# round 6, run 28: binning order — tile rows in stripes walked column by column (tile_stripe) vs row-major, verified lines
set -u
OUT=gpurun_out/r06_run28; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "every_kernel_schedule or code_map" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2; do
for ts in 0 -1 2 8; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --opt tile_stripe=$ts > $OUT/cfg2_s300_ts${ts}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 20 --warmup 5 --opt tile_stripe=$ts > $OUT/cfg2_s20_ts${ts}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --pipeline 1 --steps 100 --warmup 10 --opt tile_stripe=$ts > $OUT/cfg2_serial_ts${ts}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 --opt tile_stripe=$ts > $OUT/cfg5s_ts${ts}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --poses 32768 --steps 40 --warmup 5 --opt tile_stripe=$ts > $OUT/cfg2_32k_ts${ts}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg4 --poses 131072 --steps 10 --warmup 2 --opt tile_stripe=$ts > $OUT/cfg4s_ts${ts}_$rep.json 2>> $OUT/err.txt
done
done
timeout 300 python bench.py $B --workload cfg5 --steps 10 --warmup 2 --opt tile_stripe=0 > $OUT/cfg5_ts0.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --workload cfg5 --steps 10 --warmup 2 --opt tile_stripe=-1 > $OUT/cfg5_ts-1.json 2>> $OUT/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run28/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms ver %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
