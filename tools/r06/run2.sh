#!/bin/bash
# round 6, run 2: the whole GPU suite on the tree with the code map and RM = upstream-literal as defaults; then the
# code-map A/B over batch sizes / maps / arithmetic (serial and four in flight)
set -u
OUT=gpurun_out/r06_run2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -8 $OUT/pytest.txt
B="--no-cpu-baseline --no-extras --no-other-configs"
run() { # tag, args...
  local tag=$1; shift
  for cm in 0 2; do
    timeout 300 python bench.py $B "$@" --opt code_map=$cm > $OUT/${tag}_cm${cm}.json 2>> $OUT/err.txt
  done
}
run cfg2_1 --poses 1 --pipeline 1 --steps 200 --warmup 20
run cfg2_200 --poses 200 --pipeline 1 --steps 200 --warmup 20
run cfg2_1024 --poses 1024 --steps 200 --warmup 20
run cfg2_1024s --poses 1024 --pipeline 1 --steps 200 --warmup 20
run cfg2_2048 --poses 2048 --steps 200 --warmup 20
run cfg2_32k --poses 32768 --steps 40 --warmup 5
run cfg2_32ks --poses 32768 --pipeline 1 --steps 40 --warmup 5
run cfg2_RM --method RM --steps 300 --warmup 20
run cfg2_RMv1 --method RM --variant 1 --steps 300 --warmup 20
run cfg2_lit --variant 3 --steps 300 --warmup 20
run cfg2_crash --gather crash --steps 300 --warmup 20
run cfg2_steer --gather steer --steps 300 --warmup 20
run cfg3_RMGPU --workload cfg3 --method RMGPU --steps 10 --warmup 2
run cfg4_200 --workload cfg4 --poses 200 --pipeline 1 --steps 200 --warmup 20
run cfg4_4096 --workload cfg4 --poses 4096 --steps 200 --warmup 20
run cfg4_4096s --workload cfg4 --poses 4096 --pipeline 1 --steps 200 --warmup 20
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-24s %10.0f  %.4f ms  verified %s  lone %.4f  %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"], d["roofline"].get("kernel","")))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
