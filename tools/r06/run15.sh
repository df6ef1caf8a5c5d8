#!/bin/bash
# round 6, run 15 (final tree): smoke, the whole GPU suite, host-visible latencies, the two-player tick, the driver's command
set -u
OUT=gpurun_out/r06_last; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
python tools/r06/tick_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cddt_latency.txt
python tools/gpu_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_latency.txt
python tools/gpu_rollout_latency.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/host_latency.txt
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "driver rc $?"; done
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open("gpurun_out/r06_last/driver_cmd_%d.json"%i).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["verified"], {k:(v.get("mrays_s"),v.get("verified")) for k,v in d["other_configs"].items()})
PY
