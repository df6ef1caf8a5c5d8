#!/bin/bash
# round 6, run 13: lit_sincosf (one reduction, each polynomial once) in the literal stream kernel — parity, then A/B against
# the previous build (tools/probes/bin/libscan_prev.so) on one box
set -u
OUT=gpurun_out/r06_run13; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "literal or audit or golden or cfg1 or beam_counts or code_map" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
B="--no-cpu-baseline --no-extras --no-other-configs"
for rep in 1 2 3; do
for which in cur prev; do
  if [ $which = cur ]; then unset SCANLIB_SO; else export SCANLIB_SO=$PWD/tools/probes/bin/libscan_prev.so; fi
  timeout 300 python bench.py $B --method RM --steps 300 --warmup 20 > $OUT/rm_s300_${which}_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B --method RM --steps 20 --warmup 5 > $OUT/rm_s20_${which}_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B --method RM --pipeline 1 --steps 100 --warmup 10 > $OUT/rm_serial_${which}_$rep.json 2>> $OUT/err.txt
  timeout 300 python bench.py $B --variant 3 --workload cfg5 --poses 4096 --steps 100 --warmup 10 > $OUT/lit_cfg5_4096_${which}_$rep.json 2>> $OUT/err.txt
done
done
unset SCANLIB_SO
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run13/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-30s %10.0f  %.4f ms  verified %s  lone %.4f" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["serial"]["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
bash tools/prof_pmc.sh r06_run13/pmc_cfg2_literal --no-other-configs --grid-mult 3 --opt slots=2 --variant 3 > /dev/null 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r06_run13/pmc_cfg2_literal/pmc_summary.json'))
for k,v in d.items():
    if 'true, 2>' in k: print(k, v.get('SQ_INSTS_VALU'), v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1))
"
