#!/bin/bash
# low_water 12 (the default so far) against 20 / 24 / 32 per workload, two rays per lane with the drain compaction
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], r['value_min'], r['value_max'])"; }
for a in "" "--steps 20 --warmup 5" "--workload cfg4 --poses 4096" "--poses 32768 --steps 100" "--poses 2048" "--workload cfg5 --poses 32768 --steps 60" "--workload cfg4 --poses 131072 --steps 40" "--method RM"; do
  for lw in 12 20 24 32 12 24; do
    python bench.py --no-cpu-baseline --no-extras --no-verify $a --opt low_water=$lw 2>/dev/null | line "[$a] low_water=$lw"
  done
done
