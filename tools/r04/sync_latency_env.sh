#!/bin/bash
# does the runtime's wait policy change what scan() / scanMany(200) cost end to end?
for env in "" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=200" "HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=200" "HIP_FORCE_DEV_KERNARG=1" "SPIN=1"; do
  echo "== env: [$env]"
  env $env python tools/r04/host_latency_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -1
done
