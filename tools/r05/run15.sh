#!/bin/bash
set -u
OUT=gpurun_out/r05_run15; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "schedule or literal_mode or crash" > $OUT/pytest_sched.txt 2>&1; tail -4 $OUT/pytest_sched.txt
(SCANLIB_SO=tools/ab/libscan_amd_r5b.so python tools/r05/pool_ab.py 0,0 2>&1 | grep -v amdgpu.ids
python tools/r05/pool_ab.py 2>&1 | grep -v amdgpu.ids) | tee $OUT/pool_ab.txt
for so in tools/ab/libscan_amd_r5b.so pyracecarsimulator_amd/libscan_amd.so; do
SCANLIB_SO=$so timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench300_$(basename $so .so).json 2> /dev/null
SCANLIB_SO=$so timeout 300 python bench.py --steps 300 --warmup 20 --pipeline 1 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench300_serial_$(basename $so .so).json 2> /dev/null
done
for pool in 15 25; do
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs --opt pool=$pool > $OUT/bench300_pool$pool.json 2> /dev/null
timeout 300 python bench.py --steps 300 --warmup 20 --pipeline 1 --no-cpu-baseline --no-extras --no-other-configs --opt pool=$pool > $OUT/bench300_serial_pool$pool.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --opt pool=$pool > $OUT/bench20_pool$pool.json 2> /dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_run15/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 400 python tests/gpu_fuzz.py --seconds 240 --seed 1515 > $OUT/fuzz_240s.log 2>&1; tail -2 $OUT/fuzz_240s.log
