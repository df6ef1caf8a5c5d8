#!/bin/bash
# A/B of two library builds on one GPU box without touching the in-tree file: SCANLIB_SO selects the build
# (pyracecarsimulator_amd/_lib.py).  usage: tools/ab_env.sh <other.so> <rounds> [bench args...]
# prints: which value ms_per_step serial_kernel_ms verified
OTHER=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for which in cur other; do
    if [ $which = cur ]; then unset SCANLIB_SO; else export SCANLIB_SO=$OTHER; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$which', d['value'], d['ms_per_step'], d.get('roofline',{}).get('serial',{}).get('kernel_ms'), d.get('verified'), d.get('value_min'), d.get('value_max'))
"
  done
done
unset SCANLIB_SO
