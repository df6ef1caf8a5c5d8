#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; summaries land in gpurun_out/<tag>/
# usage (on the GPU box, from the repo root): bash tools/prof_kernel_trace.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace" -name '*kernel_trace.csv' -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/trace"
# keep the trace small: only our kernels
if [ -f "$OUT/kernel_trace.csv" ]; then
  (head -1 "$OUT/kernel_trace.csv"; grep -E "rm_fan|pose_bin|rm_rays|edt_|bl_|lut_|cddt_" "$OUT/kernel_trace.csv" | tail -${KEEP:-1500}) > "$OUT/kernel_trace_tail.csv"
  rm -f "$OUT/kernel_trace.csv"
fi
cat "$OUT/kernel_stats.csv" 2>/dev/null | head -20
cat "$OUT/bench.json"
