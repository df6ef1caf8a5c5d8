#!/bin/bash
set -u
for p in 2 4; do for g in 4 8 16; do python bench.py --no-cpu-baseline --workload cfg3 --steps 60 --pipeline $p --grid-mult $g 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('GLT pipeline $p grid_mult $g', d['value'], d['ms_per_step'], d.get('verified'))
"; done; done
for g in 8 16 32; do python bench.py --no-cpu-baseline --workload cfg3 --steps 60 --pipeline 1 --grid-mult $g 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('GLT serial grid_mult $g', d['value'], d['ms_per_step'], d.get('verified'))
"; done
bash tools/bench_matrix.sh r04c_bench > gpurun_out/r04c_bench_summary.txt 2>&1
O=gpurun_out/r04c_bench
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || echo "FAILED $name"; }
run cfg2_crash --gather crash
run cfg2_crash_steps20 --gather crash --steps 20 --warmup 5
run cfg2_steer --gather steer
run cfg4_shard131072_crash --workload cfg4 --poses 131072 --steps 40 --warmup 4 --gather crash
python bench.py --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err
tail -24 gpurun_out/r04c_bench_summary.txt | cut -c1-150
