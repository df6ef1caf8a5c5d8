#!/bin/bash
# round 6, evidence part 2 (after tools/r06/evidence_fold.sh folded part 1's PMC passes into profiles/pmc_traffic.json, so that the
# lines cite this build's traffic): the bench matrix, host-visible latencies, the driver's command twice with its legs
set -u
OUT=gpurun_out/r06_evidence; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/bench_matrix.sh r06_evidence/bench > $OUT/bench_SUMMARY.txt 2>&1
cat $OUT/bench_SUMMARY.txt | cut -c1-150
python tools/r06/tick_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cddt_latency.txt
python tools/gpu_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_latency.txt
python tools/gpu_rollout_latency.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/host_latency.txt
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "driver rc $?"; done
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open("gpurun_out/r06_evidence/driver_cmd_%d.json"%i).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["verified"], {k:(v.get("ms_per_step"),v.get("verified")) for k,v in d["other_configs"].items()})
PY
