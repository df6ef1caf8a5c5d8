#!/bin/bash
# round 6, run 5: PMC passes of the code-map kernel next to the float32 map on one box (cfg2 pipelined, cfg5 shard)
set -u
OUT=gpurun_out/r06_run5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "beam_counts or two_player" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
bash tools/prof_pmc.sh r06_run5/pmc_cfg2_code --no-other-configs > /dev/null 2>&1
bash tools/prof_pmc.sh r06_run5/pmc_cfg2_f32 --no-other-configs --opt code_map=0 > /dev/null 2>&1
bash tools/prof_pmc.sh r06_run5/pmc_cfg5_shard_code --no-other-configs --workload cfg5 --poses 32768 > /dev/null 2>&1
bash tools/prof_pmc.sh r06_run5/pmc_cfg5_shard_f32 --no-other-configs --workload cfg5 --poses 32768 --opt code_map=0 > /dev/null 2>&1
python - <<'PY'
import json
for t in ("cfg2_code","cfg2_f32","cfg5_shard_code","cfg5_shard_f32"):
    try: d=json.load(open("gpurun_out/r06_run5/pmc_%s/pmc_summary.json"%t))
    except Exception as e: print(t,"ERR",e); continue
    for k,v in d.items():
        if not k.startswith("rm_fan_stream"): continue
        wl=v.get("SQ_INSTS_VMEM_RD",0); acc=v.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)
        hit=v.get("TCC_HIT_sum",0); miss=v.get("TCC_MISS_sum",0)
        print("%-16s %s\n   VALU %.2fM  VMEM_RD %.3fM  LDS %.3fM  TCP acc %.2fM (%.1f per wave-load)  TCP->TCC rd %.2fM  L2 hit %.3f  wait %.3f  FETCH %.0f KiB WRITE %.0f KiB  ldsconf %.0f  busy %s" % (
          t,k,v.get("SQ_INSTS_VALU",0)/1e6,wl/1e6,v.get("SQ_INSTS_LDS",0)/1e6,acc/1e6,acc/max(wl,1),v.get("TCP_TCC_READ_REQ_sum",0)/1e6,hit/max(hit+miss,1),
          v.get("SQ_WAIT_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),v.get("FETCH_SIZE",0),v.get("WRITE_SIZE",0),v.get("SQ_LDS_BANK_CONFLICT",0),v.get("GRBM_GUI_ACTIVE")))
PY
