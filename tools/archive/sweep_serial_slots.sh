#!/bin/bash
# a lone launch (one stream): one ray per lane at grid_mult 8 (the default) against two / three rays per lane
# with the compaction of a dry wave's last rays, over the grid size
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], r['value_min'], r['value_max'])"; }
for a in "" "--poses 2048" "--poses 1024" "--workload cfg4 --poses 4096" "--poses 16384 --steps 100"; do
  for cfg in "1 8" "2 8" "2 6" "2 4" "2 3" "3 4" "1 8"; do
    set -- $cfg
    python bench.py --no-cpu-baseline --no-extras --no-verify --pipeline 1 $a --opt slots=$1 --grid-mult $2 2>/dev/null | line "[serial $a] slots=$1 grid_mult=$2"
  done
done
