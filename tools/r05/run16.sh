#!/bin/bash
set -u
OUT=gpurun_out/r05_run16; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for so in pyracecarsimulator_amd/libscan_amd.so tools/ab/libscan_amd_sgpr96.so; do
SCANLIB_SO=$so python tools/r05/ab_lone.py 2>&1 | grep -v amdgpu.ids
SCANLIB_SO=$so timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench300_$(basename $so .so)_$rep.json 2> /dev/null
SCANLIB_SO=$so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs > $OUT/bench20_$(basename $so .so)_$rep.json 2> /dev/null
done
done | tee $OUT/ab_lone.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_run16/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"])
    except Exception as e: print(f, "ERR", e)
PY
