#!/bin/bash
# does a smaller workgroup (fewer waves sharing one ray stream, resources released in smaller pieces) pay
# when launches are pipelined?  binned records (the form that has 256/512-lane instantiations), two rays per lane
for nt in 1024 512 256; do
  for gm in 3 4; do
    for st in "--steps 20 --warmup 5" "--steps 300"; do
      python bench.py --no-cpu-baseline --no-extras --no-verify $st --grid-mult $gm --opt inline_prep=0 --opt slots=2 --opt wg_threads=$nt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wg_threads $nt grid_mult $gm binned slots 2 [$st]', d['value'], d['ms_per_step'], d['config']['kernel'], d['config']['grid'])"
    done
  done
done
python bench.py --no-cpu-baseline --no-extras --no-verify --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['config']['kernel'], d['config']['grid'])"
