#!/bin/bash
# fused-crash cost across builds of the library (SCANLIB_SO), one ray per lane
for so in "$@"; do echo "== $so"; SCAN_SLOTS=1 SCANLIB_SO=$PWD/tools/ab/$so python tools/gpu_crash_cost.py 2>&1 | grep "^P=\|Error" | head -2; done
