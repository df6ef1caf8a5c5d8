#!/bin/bash
# cfg2 default (pipelined, three rays per lane) and 32k poses against low_water
for lw in 6 8 10 12 16 20 24; do
  for a in "" "--poses 32768 --steps 100"; do
    python bench.py --no-cpu-baseline --opt low_water=$lw $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('low_water $lw $a', d['value'], d['ms_per_step'])"
  done
done
