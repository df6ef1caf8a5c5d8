#!/bin/bash
# round 6, run 31: tile sizes of the binning order (environment overrides of an experimental build): finer coarse tiles for the
# grid-wide binning (RL_COARSE_TILES_MAX), the small path's minimum tile (RL_TILE_SHIFT_MIN)
set -u
OUT=gpurun_out/r06_run31; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2; do
for cm in 1024 4096 8192; do
  export RL_COARSE_TILES_MAX=$cm
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 > $OUT/cfg5s_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --poses 32768 --steps 40 --warmup 5 > $OUT/cfg2_32k_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg4 --poses 131072 --steps 10 --warmup 2 > $OUT/cfg4s_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B --workload cfg5 --steps 10 --warmup 2 > $OUT/cfg5_cm${cm}_$rep.json 2>> $OUT/err.txt
done
unset RL_COARSE_TILES_MAX
for sh in 5 6 7; do
  export RL_TILE_SHIFT_MIN=$sh
  timeout 200 python bench.py $B --steps 300 --warmup 20 > $OUT/cfg2_s300_sh${sh}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --pipeline 1 --steps 100 --warmup 10 > $OUT/cfg2_serial_sh${sh}_$rep.json 2>> $OUT/err.txt
done
unset RL_TILE_SHIFT_MIN
for rl in 0 1 2 3; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --opt run_log2=$rl > $OUT/cfg2_s300_rl${rl}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 --opt run_log2=$rl > $OUT/cfg5s_rl${rl}_$rep.json 2>> $OUT/err.txt
done
for lw in 16 24 28; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --opt low_water=$lw > $OUT/cfg2_s300_lw${lw}_$rep.json 2>> $OUT/err.txt
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run31/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms ver %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
