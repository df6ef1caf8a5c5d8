#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_run26
python tools/r05/cfg2_footprint.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_run26/cfg2_footprint.txt
