#!/usr/bin/env python3
"""VGPRs / SGPRs / occupancy / LDS of every kernel in libscan_amd.so's source, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (CPU only: cross-compiles gfx950).
usage: python tools/kernel_resources.py [filter substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyracecarsimulator_amd", "csrc")
UNITS = ("abi_map", "abi_fan", "abi_multi", "abi_car")
FLAGS = ("-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "
         "-fhip-fp32-correctly-rounded-divide-sqrt -w -mllvm -amdgpu-atomic-optimizer-strategy=None "
         "-Rpass-analysis=kernel-resource-usage --cuda-device-only -c").split()
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, "-o", "/tmp/_kres_%s.o" % u, os.path.join(CSRC, u + ".hip")],
                          stderr=subprocess.PIPE, text=True) for u in UNITS]
out = "".join(p.communicate()[1] for p in procs)
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True,
                                      text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
        continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
for r in rows:
    if pat in r["name"]:
        print("%-86s VGPR %3d SGPR %3d scratch %3d occupancy %d" % (r["name"][-86:], r.get("VGPRs", -1), r.get("TotalSGPRs", -1),
              r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))
