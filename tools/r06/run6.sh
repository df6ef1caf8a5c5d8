#!/bin/bash
# round 6, run 6: two generations of workgroups (tail_pct) on lone launches; schedule matrix with it
set -u
OUT=gpurun_out/r06_run6; mkdir -p $OUT
export TMPDIR=/tmp
python tools/r06/ab_lone.py "tail_pct=0" "tail_pct=15" "tail_pct=25" "tail_pct=35" "tail_pct=25,tail_wg_pct=100" "tail_pct=50,tail_wg_pct=100" "tail_pct=40,tail_wg_pct=200" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_lone_tail.txt
