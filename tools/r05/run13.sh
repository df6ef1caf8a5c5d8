#!/bin/bash
set -u
OUT=gpurun_out/r05_run13; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/r05/cddt_overlap_ab.py > $OUT/cddt_overlap_ab.txt 2>&1; grep -v amdgpu.ids $OUT/cddt_overlap_ab.txt
for ov in 0 4; do
timeout 300 python bench.py --workload cfg3 --method CDDT --steps 60 --warmup 8 --no-cpu-baseline --no-extras --no-other-configs --opt cddt_overlap=$ov > $OUT/cfg3_cddt_pipe_ov$ov.json 2> $OUT/cfg3_cddt_pipe_ov$ov.err
timeout 300 python bench.py --workload cfg3 --method CDDT --steps 60 --warmup 8 --no-cpu-baseline --no-extras --no-other-configs --pipeline 1 --opt cddt_overlap=$ov > $OUT/cfg3_cddt_serial_ov$ov.json 2> $OUT/cfg3_cddt_serial_ov$ov.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_run13/cfg3*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["verified"], d["config"].get("kernel"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cddt" > $OUT/pytest_cddt.txt 2>&1; tail -3 $OUT/pytest_cddt.txt
