#!/bin/bash
# A/B two builds of libscan_amd.so on the same GPU box: tools/ab_so.sh <other.so> [bench args...]
# runs bench.py alternately with the in-tree library and with <other.so> (3 rounds).
OTHER=$1; shift
CUR=pyracecarsimulator_amd/libscan_amd.so
cp $CUR /tmp/ab_cur.so
for r in 1 2 3; do
  for which in cur other; do
    if [ $which = cur ]; then cp /tmp/ab_cur.so $CUR; else cp $OTHER $CUR; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$which', d['value'], d['ms_per_step'])
"
  done
done
cp /tmp/ab_cur.so $CUR
