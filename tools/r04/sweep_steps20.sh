#!/bin/bash
# the driver's 20-step burst over steps in flight x grid x low_water on the round-4 bench (no HIP events in the bursts)
for rep in 1 2; do
for p in 3 4 5 6 8; do for gm in 2 3 4; do
  python bench.py --no-cpu-baseline --no-extras --no-verify --steps 20 --warmup 5 --pipeline $p --grid-mult $gm 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('rep $rep pipeline $p grid_mult $gm', d['value'], d['ms_per_step'], d['value_min'], d['value_max'])
"
done; done; done
