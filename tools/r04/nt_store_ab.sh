#!/bin/bash
# A/B: the stream kernels' range stores as NON-TEMPORAL stores (the in-tree build since round 4, scan_device.h
# range_store) against plain stores (-DRL_PLAIN_STORE -> tools/ab/libscan_amd_plain_store.so).  The ranges (17.7 MB
# per cfg2 launch) are write-once; plain stores stream them through the XCD's L2, where the band of the step map the
# launch gathers from is supposed to stay.   columns: build value ms_per_step lone_kernel_ms verified min max
OUT=gpurun_out/nt_store_ab.txt
: > $OUT
for args in "--steps 300" "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--pipeline 1" "--gather steer" "--method BL --steps 60" \
            "--workload cfg5 --poses 32768 --steps 60" "--workload cfg4 --poses 131072 --steps 40 --warmup 4" \
            "--workload cfg3 --method RMGPU --steps 40" "--poses 200 --pipeline 1"; do
  echo "== bench.py $args   (cur = non-temporal, other = plain stores)" >> $OUT
  bash tools/ab_env.sh tools/ab/libscan_amd_plain_store.so 2 $args >> $OUT 2>&1
done
cat $OUT
