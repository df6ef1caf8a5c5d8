#!/bin/bash
# round 6, run 8: CDDT theta-major search with shared compares — parity, then cfg3 rates + kernel trace; tail / stamp tests
set -u
OUT=gpurun_out/r06_run8; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cddt or every_kernel_schedule or two_player or code_map" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 600 python -m pytest tests/test_gpu_multi_device.py -x -q > $OUT/pytest_multi.txt 2>&1; tail -3 $OUT/pytest_multi.txt
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2; do
timeout 300 python bench.py $B --workload cfg3 --method CDDT --theta-disc 112 --steps 64 --warmup 8 > $OUT/cfg3_cddt112_$rep.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --workload cfg3 --method CDDT --theta-disc 112 --pipeline 1 --steps 32 --warmup 4 > $OUT/cfg3_cddt112_serial_$rep.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --workload cfg3 --method CDDT --theta-disc 108 --steps 64 --warmup 8 > $OUT/cfg3_cddt108_$rep.json 2>> $OUT/err.txt
done
bash tools/prof_kernel_trace.sh r06_run8/kt_cfg3_cddt --no-extras --no-other-configs --workload cfg3 --method CDDT --theta-disc 112 --pipeline 1 --steps 20 --warmup 3 > /dev/null 2>&1
head -8 $OUT/kt_cfg3_cddt/kernel_stats.csv
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run8/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms verified %s lone %.4f" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -2 $OUT/err.txt
