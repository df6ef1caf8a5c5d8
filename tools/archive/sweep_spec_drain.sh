#!/bin/bash
# the value-speculating drain loop of the one-ray-per-lane kernel: off (spec_drain 0) vs on from <= N live
# lanes, after / between stretches of `spec_stretch` plain samples
for cfg in "0 16" "8 16" "8 8" "8 32" "16 16" "64 16" "8 4" "0 16"; do
  set -- $cfg
  for a in "--pipeline 1 --steps 100" "--pipeline 1 --steps 100 --poses 2048" "--pipeline 1 --steps 100 --workload cfg4 --poses 4096" "--pipeline 1 --steps 40 --workload cfg5 --poses 4096"; do
    python bench.py --no-cpu-baseline --no-extras --opt spec_drain=$1 --opt spec_stretch=$2 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec_drain $1 stretch $2 [$a]', d['value'], d['ms_per_step'], 'lone kernel', d['roofline']['serial']['kernel_ms'], d['verified'])"
  done
done
