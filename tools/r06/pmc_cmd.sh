#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, never combined with tracing) over an arbitrary python script;
# writes gpurun_out/<tag>/pmc_summary.json.  usage: bash tools/r06/pmc_cmd.sh <tag> <script.py> [args...]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.log" 2> "$OUT/p$i.err" || echo "pass $i failed: $P"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
rm -rf "$OUT"/p[0-9]*
