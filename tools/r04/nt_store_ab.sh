#!/bin/bash
# A/B: the stream kernels' range stores as NON-TEMPORAL stores (option nt_store 1, the default since round 4:
# scan_device.h range_store) against plain stores (nt_store 0).  The ranges (17.7 MB per cfg2 launch) are write-once;
# plain stores stream them through the XCD's L2, where the band of the step map the launch gathers from is supposed to
# stay.   columns: nt_store value ms_per_step lone_kernel_ms verified min max
OUT=gpurun_out/nt_store_ab.txt
: > $OUT
for args in "--steps 300" "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--pipeline 1" "--method BL --steps 60" \
            "--workload cfg5 --poses 32768 --steps 60" "--workload cfg4 --poses 131072 --steps 40 --warmup 4" \
            "--workload cfg3 --method RMGPU --steps 40" "--poses 200 --pipeline 1" "--gather crash"; do
  echo "== bench.py $args" >> $OUT
  for r in 1 2; do for nt in 1 0; do
    python bench.py --no-cpu-baseline $args --opt nt_store=$nt 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('nt_store $nt', d['value'], d['ms_per_step'], d.get('roofline',{}).get('serial',{}).get('kernel_ms'), d.get('verified'), d.get('value_min'), d.get('value_max'))
" >> $OUT
  done; done
done
cat $OUT
