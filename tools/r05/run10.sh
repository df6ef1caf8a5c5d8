#!/bin/bash
set -u
OUT=gpurun_out/r05_run10; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "schedule or crash" > $OUT/pytest_sched.txt 2>&1; tail -4 $OUT/pytest_sched.txt
for rep in 1 2; do
SCANLIB_SO=tools/ab/libscan_amd_r5a.so python tools/r05/ab_lone.py 2>&1 | grep -v amdgpu.ids
python tools/r05/ab_lone.py split_service=0 2>&1 | grep -v amdgpu.ids
python tools/r05/ab_lone.py split_service=1 2>&1 | grep -v amdgpu.ids
done | tee $OUT/ab_lone.txt
for ss in 0 1; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --opt split_service=$ss > $OUT/bench20_ss$ss.json 2> $OUT/bench20_ss$ss.err
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs --opt split_service=$ss > $OUT/bench300_ss$ss.json 2> $OUT/bench300_ss$ss.err
timeout 300 python bench.py --steps 300 --warmup 20 --pipeline 1 --no-cpu-baseline --no-extras --no-other-configs --opt split_service=$ss > $OUT/bench300_serial_ss$ss.json 2> $OUT/bench300_serial_ss$ss.err
done
SCANLIB_SO=tools/ab/libscan_amd_r5a.so timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench300_r5a.json 2> $OUT/bench300_r5a.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_run10/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["verified"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 300 python tests/gpu_fuzz.py --seconds 180 --seed 1010 > $OUT/fuzz_180s.log 2>&1; tail -2 $OUT/fuzz_180s.log
