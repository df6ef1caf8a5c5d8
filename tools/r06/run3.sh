#!/bin/bash
# round 6, run 3: pipelined palette look-ups (RL_CODE_PIPE, the in-tree build) against the in-line form
# (tools/probes/bin/libscan_nopipe.so) and against the float32 map, with the code map forced at every batch size
set -u
OUT=gpurun_out/r06_run3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "code_map or every_kernel_schedule or golden or literal" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
B="--no-cpu-baseline --no-extras --no-other-configs --opt code_min_rays=0"
run() { # tag, args...
  local tag=$1; shift
  for rep in 1 2; do
  unset SCANLIB_SO
  timeout 300 python bench.py $B "$@" --opt code_map=0 > $OUT/${tag}_f32_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B "$@" --opt code_map=2 > $OUT/${tag}_pipe_$rep.json 2>> $OUT/err.txt
  SCANLIB_SO=$PWD/tools/probes/bin/libscan_nopipe.so timeout 300 python bench.py $B "$@" --opt code_map=2 > $OUT/${tag}_inline_$rep.json 2>> $OUT/err.txt
  done
}
run cfg2_s20 --steps 20 --warmup 5
run cfg2_s300 --steps 300 --warmup 20
run cfg2_serial --pipeline 1 --steps 100 --warmup 10
run cfg2_200 --poses 200 --pipeline 1 --steps 200 --warmup 20
run cfg2_1024s --poses 1024 --pipeline 1 --steps 200 --warmup 20
run cfg2_2048s --poses 2048 --pipeline 1 --steps 200 --warmup 20
run cfg2_2048 --poses 2048 --steps 200 --warmup 20
run cfg5shard --workload cfg5 --poses 32768 --steps 40 --warmup 5
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run3/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-24s %10.0f  %.4f ms  verified %s  lone %.4f  %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"], d["roofline"].get("kernel","")))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
