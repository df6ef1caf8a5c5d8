#!/bin/bash
# round 6, run 30: sorted pose + first step from the keys-only binning launch (one round trip in the march's prologue instead of
# three) vs the previous build (libscan_amd_base.so, SCANLIB_SO): parity, then interleaved A/B
set -u
OUT=gpurun_out/r06_run30; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "every_kernel_schedule or code_map or golden or cfg1 or crash" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 300 python tests/gpu_fuzz.py --seconds 120 --seed 3001 > $OUT/fuzz_120s.log 2>&1; tail -2 $OUT/fuzz_120s.log
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2 3; do
for so in new base; do
  if [ $so = base ]; then export SCANLIB_SO=$PWD/pyracecarsimulator_amd/libscan_amd_base.so; else unset SCANLIB_SO; fi
  timeout 200 python bench.py $B --pipeline 1 --steps 100 --warmup 10 > $OUT/cfg2_serial_${so}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 300 --warmup 20 > $OUT/cfg2_s300_${so}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 20 --warmup 5 > $OUT/cfg2_s20_${so}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --method RM --pipeline 1 --steps 100 --warmup 10 > $OUT/cfg2_RM_serial_${so}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --gather crash --pipeline 1 --steps 100 --warmup 10 > $OUT/cfg2_crash_serial_${so}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --poses 2048 --pipeline 1 --steps 100 --warmup 10 > $OUT/cfg2_2048_serial_${so}_$rep.json 2>> $OUT/err.txt
done
done
unset SCANLIB_SO
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run30/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-30s %10.0f  %.4f ms ver %s lone %.4f" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
