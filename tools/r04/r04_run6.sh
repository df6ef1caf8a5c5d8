#!/bin/bash
# PMC passes of the final round-4 build for the launch shapes bench.py reports traffic for.  Counter collection
# serialises kernels (pipeline.concurrent_streams then finds no two streams that overlap and bench.py falls back to the
# serial schedule), so the PIPELINED shapes (grid_mult 3, two rays per lane) are forced onto the serial schedule here.
set -u
bash tools/prof_pmc.sh r04b_pmc_cfg2_slots2 --pipeline 1 --grid-mult 3 --opt slots=2 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg5_shard_pipe --workload cfg5 --poses 32768 --pipeline 1 --grid-mult 3 --opt slots=2 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg4_4096_pipe --workload cfg4 --poses 4096 --pipeline 1 --grid-mult 3 --opt slots=2 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg2_crash_slots2 --gather crash --pipeline 1 --grid-mult 3 --opt slots=2 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg2_bl_pipe --method BL --pipeline 1 --grid-mult 3 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg2_bl --method BL --pipeline 1 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg5 --workload cfg5 > /dev/null
bash tools/prof_pmc.sh r04b_pmc_cfg4_1M --workload cfg4 > /dev/null
for d in gpurun_out/r04b_pmc_*; do echo $d; python - "$d" <<'PY'
import json, sys
s = json.load(open(sys.argv[1] + "/pmc_summary.json"))
for k, v in s.items():
    if v.get("_dispatches", 0) >= 4 and any(t in k for t in ("rm_fan", "lut_fan", "bl_fan")):
        print("   %-74s disp %3d HBM %.1f MB VALU %.2fM" % (k, v["_dispatches"], (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / 1e6, v.get("SQ_INSTS_VALU", 0) / 1e6))
PY
done
