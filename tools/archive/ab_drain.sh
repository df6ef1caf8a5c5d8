#!/bin/bash
# A/B on one box: the committed build (compile-time drain cap 24, 9 fields) against the current one at several
# drain_cap / spec_stretch settings
OLD=tools/ab/libscan_amd_cap24.so
line() { python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('$1', r['value'], r['ms_per_step'])"; }
for r in 1 2; do
  for st in 20 300; do
    SCANLIB_SO=$OLD python bench.py --steps $st --warmup 5 --no-extras --no-verify --no-cpu-baseline 2>/dev/null | line "old steps=$st"
    for cfg in "24 16" "64 16" "64 8" "64 4" "48 8"; do
      set -- $cfg
      python bench.py --steps $st --warmup 5 --no-extras --no-verify --no-cpu-baseline --opt drain_cap=$1 --opt spec_stretch=$2 2>/dev/null | line "new drain_cap=$1 spec_stretch=$2 steps=$st"
    done
  done
done
