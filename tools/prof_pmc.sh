#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, never combined with tracing) over the bench
# command; writes gpurun_out/<tag>/pmc_summary.json (per-kernel averages of each counter).
# usage: bash tools/prof_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python3 bench.py --no-cpu-baseline --no-extras --no-verify --bursts 2 --steps 5 --warmup 2 "$@" > "$OUT/p$i.json" 2> "$OUT/p$i.err" || echo "pass $i failed: $P"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
rm -rf "$OUT"/p[0-9]* 
cat "$OUT/pmc_summary.json"
