#!/bin/bash
set -u
OUT=gpurun_out/r05_run21; mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-other-configs --no-verify"
for lw in 12 16 20 24 28 20; do
python bench.py $B --steps 300 --warmup 20 --opt low_water=$lw 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('low_water $lw: 300 steps', d['value'], d['ms_per_step'])"
python bench.py $B --steps 20 --warmup 5 --opt low_water=$lw 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('low_water $lw: 20 steps', d['value'], d['ms_per_step'])"
done
for gm in 2 3 4; do for pl in 3 4 6; do
python bench.py $B --steps 20 --warmup 5 --grid-mult $gm --pipeline $pl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid_mult $gm pipeline $pl: 20 steps', d['value'], d['ms_per_step'])"
done; done
