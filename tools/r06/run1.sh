#!/bin/bash
# round 6, run 1: the code map (u16 palette codes + LDS palette) — parity first, then the A/B against the float32 step map
set -u
OUT=gpurun_out/r06_run1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "code_map or every_kernel_schedule" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2 3; do
for cm in 0 2; do
timeout 300 python bench.py --steps 20 --warmup 5 $B --opt code_map=$cm > $OUT/cfg2_s20_cm${cm}_$rep.json 2> $OUT/err.txt
timeout 300 python bench.py --steps 300 --warmup 20 $B --opt code_map=$cm > $OUT/cfg2_s300_cm${cm}_$rep.json 2>> $OUT/err.txt
timeout 300 python bench.py --steps 100 --warmup 10 --pipeline 1 $B --opt code_map=$cm > $OUT/cfg2_serial_cm${cm}_$rep.json 2>> $OUT/err.txt
timeout 300 python bench.py --workload cfg5 --poses 32768 --steps 40 --warmup 5 $B --opt code_map=$cm > $OUT/cfg5shard_cm${cm}_$rep.json 2>> $OUT/err.txt
done
done
timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 $B --opt code_map=0 > $OUT/cfg5_cm0.json 2>> $OUT/err.txt
timeout 300 python bench.py --workload cfg5 --steps 10 --warmup 2 $B --opt code_map=2 > $OUT/cfg5_cm2.json 2>> $OUT/err.txt
timeout 300 python bench.py --workload cfg4 --poses 131072 --steps 10 --warmup 2 $B --opt code_map=0 > $OUT/cfg4shard_cm0.json 2>> $OUT/err.txt
timeout 300 python bench.py --workload cfg4 --poses 131072 --steps 10 --warmup 2 $B --opt code_map=2 > $OUT/cfg4shard_cm2.json 2>> $OUT/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run1/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"], d.get("plan",{}).get("name") if isinstance(d.get("plan"),dict) else "")
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
