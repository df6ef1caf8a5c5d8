#!/bin/bash
# round 6, run 9: CDDT search kernel A/B on one box: this tree vs the previous commit (tools/probes/bin/libscan_prev.so)
set -u
OUT=gpurun_out/r06_run9; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for which in cur prev; do
  if [ $which = cur ]; then unset SCANLIB_SO; else export SCANLIB_SO=$PWD/tools/probes/bin/libscan_prev.so; fi
  bash tools/prof_kernel_trace.sh r06_run9/kt_${which}_$rep --no-extras --no-other-configs --workload cfg3 --method CDDT --theta-disc 112 --pipeline 1 --steps 20 --warmup 3 > /dev/null 2>&1
  echo "$which $rep: $(grep search2 $OUT/kt_${which}_$rep/kernel_stats.csv | cut -d, -f2-4) | fan $(grep theta_fan $OUT/kt_${which}_$rep/kernel_stats.csv | cut -d, -f4)"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-other-configs --workload cfg3 --method CDDT --theta-disc 112 --steps 64 --warmup 8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('   pipelined', d['value'], d['ms_per_step'], d['verified'])
"
done
done
unset SCANLIB_SO
