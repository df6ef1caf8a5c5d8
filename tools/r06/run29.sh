#!/bin/bash
# round 6, run 29 (tile_stripe = -1 the default): the whole GPU suite, a 3-minute fuzz, PMC of cfg5 shard / cfg2 with both orders
set -u
OUT=gpurun_out/r06_run29; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 400 python tests/gpu_fuzz.py --seconds 180 --seed 2901 > $OUT/fuzz_180s.log 2>&1; tail -2 $OUT/fuzz_180s.log
P="bash tools/prof_pmc.sh"
$P r06_run29/pmc_cfg5_shard_ts0 --no-other-configs --workload cfg5 --poses 32768 --grid-mult 3 --opt slots=2 --opt tile_stripe=0 > /dev/null 2>&1
$P r06_run29/pmc_cfg5_shard --no-other-configs --workload cfg5 --poses 32768 --grid-mult 3 --opt slots=2 > /dev/null 2>&1
$P r06_run29/pmc_cfg2_serial_ts0 --no-other-configs --pipeline 1 --opt tile_stripe=0 > /dev/null 2>&1
$P r06_run29/pmc_cfg2_serial --no-other-configs --pipeline 1 > /dev/null 2>&1
python - <<'PY'
import json
for t in ("pmc_cfg5_shard_ts0","pmc_cfg5_shard","pmc_cfg2_serial_ts0","pmc_cfg2_serial"):
    p=json.load(open("gpurun_out/r06_run29/%s/pmc_summary.json"%t))
    for k,v in p.items():
        if "rm_fan_stream" in k: print(t, k[-40:], {a:round(v[a]) for a in ("TCC_HIT_sum","TCC_MISS_sum","TCC_EA0_RDREQ_sum","TCP_TCC_READ_REQ_sum","FETCH_SIZE","SQ_WAIT_ANY","SQ_WAVE_CYCLES","GRBM_GUI_ACTIVE") if a in v})
PY
