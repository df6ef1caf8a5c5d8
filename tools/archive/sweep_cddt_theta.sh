#!/bin/bash
# CDDT: pose-major (cddt_theta_min=0) against theta-major (=1) over the batch size, cfg3 map, lone launches and
# four in flight
line() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], r['roofline']['kernel'])"; }
for n in 4096 8192 16384 32768 65536 262144; do
  for tm in 0 1; do
    for pl in "--pipeline 1" ""; do
      python bench.py --no-cpu-baseline --no-extras --no-verify --workload cfg3 --method CDDT --steps 30 --poses $n $pl --opt cddt_theta_min=$tm 2>/dev/null | line "poses=$n theta_min=$tm [$pl]"
    done
  done
done
