#!/bin/bash
# Several rays per lane, stream dry: from how many live rays per wave on is compacting them into one slot worth it
# (drain_cap), when does the one-slot loop hand over to the speculating loop (spec_drain), and how long are the
# plain stretches between two speculation attempts (spec_stretch)?  bench.py lines, 25-burst medians, driver
# burst (20 steps) and pipelined steady state (300 steps).
for cfg in "24 64 16" "32 64 16" "48 64 16" "64 64 16" "64 64 8" "64 64 4" "64 64 32" "64 48 16" "24 8 16" "24 64 16"; do
  set -- $cfg
  for st in 20 300; do
    python bench.py --steps $st --warmup 5 --no-extras --no-verify --opt drain_cap=$1 --opt spec_drain=$2 --opt spec_stretch=$3 2>/dev/null | \
      python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('drain_cap=$1 spec_drain=$2 spec_stretch=$3 steps=$st', r['value'], r['ms_per_step'])"
  done
done
