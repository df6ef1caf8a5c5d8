#!/bin/bash
# round 6, run 14: fused crash kernels with pose | beam in one register (less scratch) — parity, then A/B against the previous build
set -u
OUT=gpurun_out/r06_run14; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "crash or rollout or literal_mode or schedule or code_map" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 600 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_dist.py -x -q > $OUT/pytest2.txt 2>&1; tail -3 $OUT/pytest2.txt
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2 3; do
for which in cur prev; do
  if [ $which = cur ]; then unset SCANLIB_SO; else export SCANLIB_SO=$PWD/tools/probes/bin/libscan_prev.so; fi
  timeout 300 python bench.py $B --gather crash --steps 300 --warmup 20 > $OUT/crash_s300_${which}_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B --gather crash --method RM --steps 300 --warmup 20 > $OUT/crash_lit_s300_${which}_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B --gather crash --workload cfg4 --poses 131072 --steps 10 --warmup 2 > $OUT/crash_cfg4shard_${which}_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B --gather crash --poses 32768 --steps 40 --warmup 4 > $OUT/crash_32k_${which}_$rep.json 2>> $OUT/err.txt
done
done
unset SCANLIB_SO
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run14/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-30s %10.0f  %.4f ms  verified %s  lone %.4f" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
