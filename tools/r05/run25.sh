#!/bin/bash
set -u
export TMPDIR=/tmp
bash tools/prof_pmc.sh r05_run25/pmc_cfg3_cddt112 --no-other-configs --workload cfg3 --method CDDT --theta-disc 112 --pipeline 1 > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_run25/pmc_cfg3_cddt112/pmc_summary.json'))
for k,v in d.items():
    if 'cddt_theta' in k: print(k,{c:round(x) for c,x in v.items() if c in ('SQ_INSTS_VALU','FETCH_SIZE','WRITE_SIZE','_dispatches')})
PY
