#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_run30
python tools/gpu_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_run30/host_latency.txt
python tools/r04/host_latency_breakdown.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_run30/host_latency.txt
