#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script: tools/prof_trace_cmd.sh <tag> <script.py> [args...]
# keeps kernel_stats.csv and the last 400 rows of our kernels (with queue ids and timestamps).
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log"
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace" -name '*kernel_trace.csv' -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/trace"
if [ -f "$OUT/kernel_trace.csv" ]; then
  (head -1 "$OUT/kernel_trace.csv"; grep -E "${PAT:-rm_fan|pose_bin|pose_prep|pose_scatter|tile_scan|rm_rays|bl_|lut_|cddt_|crash_}" "$OUT/kernel_trace.csv" | tail -${KEEP:-400}) > "$OUT/kernel_trace_tail.csv"
  rm -f "$OUT/kernel_trace.csv"
fi
head -12 "$OUT/kernel_stats.csv" 2>/dev/null
tail -5 "$OUT/stdout.log"
