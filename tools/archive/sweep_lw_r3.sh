#!/bin/bash
# low_water x burst length on the round-3 kernel (cheaper claims move the optimum up)
for lw in 8 12 16 20 24 28; do
  for a in "--steps 20 --warmup 5" "" "--pipeline 1 --steps 100"; do
    python bench.py --no-cpu-baseline --no-extras --no-verify --opt low_water=$lw $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('low_water $lw [$a]', d['value'], d['ms_per_step'], d['value_min'], d['value_max'])"
  done
done
