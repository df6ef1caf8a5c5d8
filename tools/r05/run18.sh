#!/bin/bash
set -u
OUT=gpurun_out/r05_run18; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -s -k "bresenham_vs_upstream" 2>&1 | grep "device Bresenham" | cut -c1-600
