#!/bin/bash
# A/B of two library builds given as files: tools/ab_two.sh <a.so> <b.so> <rounds> [bench args...]
A=$1; B=$2; ROUNDS=$3; shift 3
for r in $(seq 1 $ROUNDS); do
  for which in A B; do
    if [ $which = A ]; then export SCANLIB_SO=$A; else export SCANLIB_SO=$B; fi
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
