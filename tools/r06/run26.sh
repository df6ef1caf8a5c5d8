#!/bin/bash
# round 6, run 26: 2-minute fuzz with FollowGap in every case; kernel trace of the roll-out chain (where do its 4.4 ms go?)
set -u
OUT=gpurun_out/r06_run26; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python tests/gpu_fuzz.py --seconds 120 --seed 2601 > $OUT/fuzz_120s.log 2>&1; tail -2 $OUT/fuzz_120s.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/r06/rollout_loop.py 6 > $OUT/rollout.log 2>&1
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/trace
python3 - $OUT/kernel_trace.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:60]) for r in rows]
ev.sort()
# the last call: from the last rollout_kernel on
idx=max(i for i,e in enumerate(ev) if "rollout_kernel" in e[2])
t0=ev[idx][0]
for s,e,n in ev[idx:]:
    print("%9.1f %9.1f  %8.1f us  %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,n.replace("void scan::","")))
PY
rm -f $OUT/kernel_trace.csv
