#!/bin/bash
# bounded fuzz of the current build: tests/gpu_fuzz.py for N seconds (default 600), log to gpurun_out/<tag>.log
SECS=${1:-600}; TAG=${2:-fuzz}
mkdir -p gpurun_out
timeout $((SECS + 120)) python tests/gpu_fuzz.py --seconds $SECS > gpurun_out/$TAG.log 2>&1
tail -5 gpurun_out/$TAG.log
