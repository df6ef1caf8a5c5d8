#!/bin/bash
# round 6, run 20: the EXACT 8-bit code map (escape to the float32 map) — parity, a short fuzz, then A/B against u16 (verified)
set -u
OUT=gpurun_out/r06_run20; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "u8_code or code_map or every_kernel_schedule or crash" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 300 python tests/gpu_fuzz.py --seconds 180 --seed 77 > $OUT/fuzz_180s.log 2>&1; tail -2 $OUT/fuzz_180s.log
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2 3; do
for cm in 2 1; do
  timeout 200 python bench.py $B --steps 300 --warmup 20 --opt code_map=$cm > $OUT/cfg2_s300_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 20 --warmup 5 --opt code_map=$cm > $OUT/cfg2_s20_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --pipeline 1 --steps 100 --warmup 10 --opt code_map=$cm > $OUT/cfg2_serial_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 --opt code_map=$cm > $OUT/cfg5s_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --poses 32768 --steps 40 --warmup 5 --opt code_map=$cm > $OUT/cfg2_32k_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --gather crash --steps 300 --warmup 20 --opt code_map=$cm > $OUT/cfg2_crash_cm${cm}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg4 --poses 131072 --steps 10 --warmup 2 --opt code_map=$cm > $OUT/cfg4s_cm${cm}_$rep.json 2>> $OUT/err.txt
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run20/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms ver %s lone %.4f  %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"], d["roofline"]["kernel"][-22:]))
    except Exception as e: print(f, "ERR", e)
PY
tail -2 $OUT/err.txt
