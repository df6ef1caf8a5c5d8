#!/bin/bash
# two or three rays per lane in the pipelined launches, per workload (one box, alternating)
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], r['value_min'], r['value_max'])"; }
for a in "--poses 2048" "--poses 8192" "--workload cfg4 --poses 4096" "--workload cfg5 --poses 4096 --steps 40" "--workload cfg1 --poses 4096" "--method RM" "--poses 1024"; do
  for s in 2 3 2 3; do
    python bench.py --no-cpu-baseline --no-extras --no-verify $a --opt slots=$s 2>/dev/null | line "[$a] slots=$s"
  done
done
