#!/bin/bash
# round 6, the evidence of the final tree in one call: smoke, the whole GPU suite, every PMC pass bench.py / bench_legs.py cite,
# kernel traces (part 2, evidence2.sh: the bench matrix, the driver command, latencies).  Summaries land in
# gpurun_out/r06_evidence/; tools/r06/evidence_fold.sh copies them under profiles/r06/ and folds the PMC passes into
# profiles/pmc_traffic.json.
set -u
OUT=gpurun_out/r06_evidence; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
P="bash tools/prof_pmc.sh"
$P r06_evidence/pmc_cfg2_slots2 --no-other-configs --grid-mult 3 --opt slots=2 > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_serial --no-other-configs --pipeline 1 > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_literal --no-other-configs --grid-mult 3 --opt slots=2 --variant 3 > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_rm --no-other-configs --grid-mult 3 --opt slots=2 --method RM > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_crash --no-other-configs --grid-mult 3 --opt slots=2 --gather crash > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_steer --no-other-configs --grid-mult 3 --opt slots=2 --gather steer > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_32k --no-other-configs --poses 32768 --grid-mult 3 --opt slots=2 > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_bl --no-other-configs --method BL --grid-mult 3 > /dev/null 2>&1
$P r06_evidence/pmc_cfg2_cddt --no-other-configs --method CDDT > /dev/null 2>&1
$P r06_evidence/pmc_cfg3_glt --no-other-configs --workload cfg3 > /dev/null 2>&1
$P r06_evidence/pmc_cfg3_cddt112 --no-other-configs --workload cfg3 --method CDDT --theta-disc 112 --pipeline 1 > /dev/null 2>&1
$P r06_evidence/pmc_cfg3_cddt108 --no-other-configs --workload cfg3 --method CDDT --theta-disc 108 --pipeline 1 > /dev/null 2>&1
$P r06_evidence/pmc_cfg3_rmgpu --no-other-configs --workload cfg3 --method RMGPU > /dev/null 2>&1
$P r06_evidence/pmc_cfg4_shard --no-other-configs --workload cfg4 --poses 131072 > /dev/null 2>&1
$P r06_evidence/pmc_cfg4_4096 --no-other-configs --workload cfg4 --poses 4096 --grid-mult 3 --opt slots=2 > /dev/null 2>&1
$P r06_evidence/pmc_cfg4_1M --no-other-configs --workload cfg4 > /dev/null 2>&1
$P r06_evidence/pmc_cfg5 --no-other-configs --workload cfg5 > /dev/null 2>&1
$P r06_evidence/pmc_cfg5_shard --no-other-configs --workload cfg5 --poses 32768 --grid-mult 3 --opt slots=2 > /dev/null 2>&1
bash tools/r06/pmc_cmd.sh r06_evidence/pmc_cfg4_rollout tools/r06/rollout_loop.py 6 > /dev/null 2>&1
ls $OUT | tr '\n' ' '; echo
bash tools/prof_kernel_trace.sh r06_evidence/kt_cfg2_driver_cmd --gpus 1 --steps 20 --warmup 5 --no-other-configs > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r06_evidence/kt_cfg2_serial --pipeline 1 --steps 60 --warmup 5 --no-other-configs > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r06_evidence/kt_cfg2_literal --variant 3 --steps 60 --warmup 5 --no-other-configs > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r06_evidence/kt_cfg3_cddt --no-extras --no-other-configs --workload cfg3 --method CDDT --theta-disc 112 --pipeline 1 --steps 20 --warmup 3 > /dev/null 2>&1
bash tools/prof_kernel_trace.sh r06_evidence/kt_cfg5_shard --no-extras --no-other-configs --workload cfg5 --poses 32768 --steps 40 --warmup 5 > /dev/null 2>&1
for d in kt_cfg2_driver_cmd kt_cfg2_serial kt_cfg2_literal kt_cfg3_cddt kt_cfg5_shard; do echo "== $d"; head -3 $OUT/$d/kernel_stats.csv | cut -c1-60,180-260; done
