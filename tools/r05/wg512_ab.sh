#!/bin/bash
# 512-lane INLINE workgroups (A/B): lone kernel, pipelined 300 steps, the driver's 20 steps
set -u
OUT=gpurun_out/r05_wg512; mkdir -p $OUT
export TMPDIR=/tmp
for wg in 1024 512 1024 512; do
  python tools/r05/ab_lone.py wg_threads=$wg 2>&1 | grep "slots 2"
  for st in 300 20; do
    python bench.py --gpus 1 --steps $st --warmup 8 --no-cpu-baseline --no-extras --opt wg_threads=$wg 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('wg_threads $wg steps $st:', d['value'], d['ms_per_step'], d.get('verified'), d['config'].get('kernel'))
"
  done
done
