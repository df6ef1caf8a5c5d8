#!/bin/bash
# does the compaction of a dry wave's last rays (drain_cap; 1 = as good as off) pay on every workload?
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], r['value_min'], r['value_max'])"; }
for a in "--workload cfg4 --poses 4096" "--workload cfg5 --poses 4096 --steps 40" "--workload cfg1 --poses 4096" "--poses 2048" "--poses 8192"; do
  for s in 2 3; do
    for cap in 1 24 64 1 64; do
      python bench.py --no-cpu-baseline --no-extras --no-verify $a --opt slots=$s --opt drain_cap=$cap 2>/dev/null | line "[$a] slots=$s drain_cap=$cap"
    done
  done
done
