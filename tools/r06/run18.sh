#!/bin/bash
# drain parameters on the code-map kernel (lone launches, then the pipelined bench for the best few)
set -u
OUT=gpurun_out/r06_run18; mkdir -p $OUT
export TMPDIR=/tmp
python tools/r06/ab_lone.py "" "drain_cap=32" "drain_cap=16" "drain_stretch=4" "drain_stretch=16" "drain_cap=32,drain_stretch=4" "group_drain=8" "low_water=12" "low_water=28" 2>&1 | grep -v amdgpu.ids | grep "cfg2 n   4096\|cfg5\|cfg2 n  32768" | tee $OUT/ab_lone_drain.txt
B="--no-cpu-baseline --no-extras --no-other-configs --no-verify"
for o in "drain_cap=64" "drain_cap=32" "drain_stretch=4" "drain_stretch=16"; do
  tag=$(echo $o | tr '=' '_')
  timeout 200 python bench.py $B --steps 20 --warmup 5 --opt $o > $OUT/s20_$tag.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --steps 300 --warmup 20 --bursts 9 --opt $o > $OUT/s300_$tag.json 2>> $OUT/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run18/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-30s %10.0f  %.4f ms" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"]))
    except Exception as e: print(f, "ERR", e)
PY
