#!/bin/bash
set -u
OUT=gpurun_out/r05_run22; mkdir -p $OUT
export TMPDIR=/tmp
SCANLIB_SO=tools/ab/libscan_amd_fly.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "schedule or literal_mode or golden or crash or edge or beam_counts" > $OUT/pytest_fly.txt 2>&1; tail -4 $OUT/pytest_fly.txt
for rep in 1 2; do
for so in pyracecarsimulator_amd/libscan_amd.so tools/ab/libscan_amd_fly.so; do
SCANLIB_SO=$so python tools/r05/ab_lone.py 2>&1 | grep -v amdgpu.ids | grep "slots 2"
SCANLIB_SO=$so timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench300_$(basename $so .so)_$rep.json 2> $OUT/err.txt
SCANLIB_SO=$so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench20_$(basename $so .so)_$rep.json 2>> $OUT/err.txt
done
done | tee $OUT/ab_lone.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_run22/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"])
    except Exception as e: print(f, "ERR", e)
PY
SCANLIB_SO=tools/ab/libscan_amd_fly.so timeout 300 python tests/gpu_fuzz.py --seconds 150 --seed 2222 > $OUT/fuzz_fly.log 2>&1; tail -2 $OUT/fuzz_fly.log
