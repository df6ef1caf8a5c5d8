#!/bin/bash
# round 6, run 22: where do steer's +10 us per step go?  kernel trace of bench.py --gather steer, pipelined and serial
set -u
export TMPDIR=/tmp
for mode in pipe serial; do
  OUT=$PWD/gpurun_out/r06_run22/$mode; mkdir -p $OUT
  A="--gather steer --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-other-configs"
  [ $mode = serial ] && A="$A --pipeline 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $A > $OUT/bench.json 2> $OUT/bench.err
  find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
  find $OUT/trace -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
  rm -rf $OUT/trace
  (head -1 $OUT/kernel_trace.csv; grep -E "rm_fan_stream|followgap|pose_bin" $OUT/kernel_trace.csv | tail -1200) > $OUT/kernel_trace_tail.csv
  rm -f $OUT/kernel_trace.csv
  head -8 $OUT/kernel_stats.csv | cut -c1-200
  python3 - $OUT/kernel_trace_tail.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:40],r.get("Queue_Id","?"),r.get("Stream_Id","?")) for r in rows]
ev.sort()
t0=ev[-120][0]
for s,e,n,q,st in ev[-120:-60]:
    print("%9.2f %9.2f  %7.2f us  q%s s%s %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,q,st,n.replace("void scan::","")))
PY
done
