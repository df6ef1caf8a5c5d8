#!/bin/bash
# cfg2 default: grid_mult x steps in flight x rays per lane on the final kernels
for p in 3 4; do for gm in 2 3 4 5; do for sl in 2 3; do
python bench.py --no-cpu-baseline --pipeline $p --grid-mult $gm --opt slots=$sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $p grid_mult $gm slots $sl', d['value'], d['ms_per_step'])"
done; done; done
