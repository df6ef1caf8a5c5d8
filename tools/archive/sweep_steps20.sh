#!/bin/bash
# the driver's command is `bench.py --gpus 1 --steps 20 --warmup 5`: a 20-step burst, where the fill
# and the drain of the launch pipeline are a tenth of the timed region.  Pipeline depth x grid x rays
# per lane at that length and at 300 steps (median of 25 bursts each).
CFGS=${CFGS:-"4 3 3;4 3 2;4 4 2;4 2 2;4 6 2;3 3 2;3 4 2;5 3 2;5 2 2;6 3 2;6 2 2;8 2 2;2 4 2;2 8 2;4 3 1;4 8 1;1 8 1"}
IFS=';' read -ra LIST <<< "$CFGS"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  for st in "--steps 20 --warmup 5" "--steps 300"; do
    python bench.py --no-cpu-baseline --no-extras --no-verify $st --pipeline $1 --grid-mult $2 --opt slots=$3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $1 grid_mult $2 slots $3 [$st]', d['value'], d['ms_per_step'], d['value_min'], d['value_max'])"
  done
done
