#!/bin/bash
# after tools/r06/evidence.sh: copy its summaries under profiles/r06/ and fold every PMC pass into profiles/pmc_traffic.json
# (run from the repo root, on the commit the evidence was taken on)
set -eu
SRC=gpurun_out/r06_evidence
for d in $SRC/pmc_* $SRC/kt_*; do
  n=$(basename $d); mkdir -p profiles/r06/$n
  for f in pmc_summary.json kernel_stats.csv kernel_trace_tail.csv bench.json; do [ -f $d/$f ] && cp $d/$f profiles/r06/$n/$f; done
done
mkdir -p profiles/r06/final
cp $SRC/smoke.txt profiles/r06/final/smoke.txt
tail -5 $SRC/pytest.txt > profiles/r06/final/pytest_gpu_tail.txt
U="python3 tools/pmc_traffic_update.py"
$U profiles/r06/pmc_cfg2_slots2 cfg2 RMGPU 4096
$U profiles/r06/pmc_cfg2_serial cfg2 RMGPU 4096
$U profiles/r06/pmc_cfg2_literal cfg2 RMGPU 4096 "--variant 3"
$U profiles/r06/pmc_cfg2_rm cfg2 RM 4096
$U profiles/r06/pmc_cfg2_crash cfg2 RMGPU 4096 "--gather crash"
$U profiles/r06/pmc_cfg2_steer cfg2 RMGPU 4096 "--gather steer (nt_store 0)" steer
$U profiles/r06/pmc_cfg2_32k cfg2 RMGPU 32768
$U profiles/r06/pmc_cfg2_bl cfg2 BL 4096
$U profiles/r06/pmc_cfg2_cddt cfg2 CDDT 4096
$U profiles/r06/pmc_cfg3_glt cfg3 GLT 65536
$U profiles/r06/pmc_cfg3_cddt112 cfg3 CDDT 65536 "theta_disc 112"
$U profiles/r06/pmc_cfg3_cddt108 cfg3 CDDT 65536 "theta_disc 108"
$U profiles/r06/pmc_cfg3_rmgpu cfg3 RMGPU 65536
$U profiles/r06/pmc_cfg4_shard cfg4 RMGPU 131072
$U profiles/r06/pmc_cfg4_4096 cfg4 RMGPU 4096
$U profiles/r06/pmc_cfg4_1M cfg4 RMGPU 1048576
$U profiles/r06/pmc_cfg5 cfg5 RMGPU 262144
$U profiles/r06/pmc_cfg5_shard cfg5 RMGPU 32768
$U profiles/r06/pmc_cfg4_rollout cfg4 "RMGPU+rollout" 819200 "rl_car_rollout_check, 4096 roll-outs x 200 steps"
# (the crash / steer passes also hold the plain kernel of their verification launches: the plain passes go in last)
$U profiles/r06/pmc_cfg2_serial cfg2 RMGPU 4096
$U profiles/r06/pmc_cfg2_slots2 cfg2 RMGPU 4096
