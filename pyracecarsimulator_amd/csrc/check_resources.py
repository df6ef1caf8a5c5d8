#!/usr/bin/env python3
"""Build gate: the register / occupancy budget of the kernels whose speed depends on it, read from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (the Makefile keeps them per translation unit in .obj/*.remarks).

The march loops pin physical VGPRs in inline assembly and the stream kernels are held at 64 VGPRs / 80 SGPRs so that
eight waves share a SIMD (profiles/r05/sgpr_cap_ab.txt: 92 SGPRs cost the eighth wave and 25 % of the rate).  A compiler
bump that silently spends one register more would not fail a parity test — it fails the build here instead.
usage: check_resources.py <dir with *.remarks> [--print]"""
import os
import re
import subprocess
import sys

# (substring of the demangled kernel name, {limit: value}); every kernel matching a pattern must satisfy it
BOUNDS = [
    ("scan::rm_fan_stream_kernel<", {"vgpr": 64, "sgpr": 80, "occupancy": 8}),
    # the plain-range production shapes (two rays per lane, 1024 lanes, float32 or code map) keep nothing in scratch
    ("scan::rm_fan_stream_kernel<false, false, 1024, true, true, 2, false,", {"scratch": 0}),
    ("scan::rm_fan_stream_kernel<false, false, 1024, false, true, 2, false,", {"scratch": 0}),
    ("scan::lut_fan_lds_kernel<", {"vgpr": 56, "sgpr": 96, "occupancy": 8, "scratch": 0}),
    ("scan::cddt_theta_search2_kernel", {"vgpr": 64, "sgpr": 80, "occupancy": 8, "scratch": 0}),
    ("scan::cddt_theta_fan_kernel", {"vgpr": 32, "sgpr": 96, "occupancy": 8, "scratch": 0}),
    ("scan::cddt_fan_bins_kernel", {"vgpr": 64, "sgpr": 80, "occupancy": 8, "scratch": 0}),
    ("scan::bl_fan_stream_kernel<false, 1024>", {"vgpr": 56, "sgpr": 96, "occupancy": 8, "scratch": 0}),
    ("scan::rm_leftover_kernel<", {"vgpr": 48, "occupancy": 8, "scratch": 0}),
]
KEYS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy"}


def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1)}
            rows.append(cur)
            continue
        for key, short in KEYS.items():
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None and short not in cur:
                cur[short] = int(m.group(1))
    if rows:
        names = subprocess.run(["c++filt"] + [r["mangled"] for r in rows], capture_output=True, text=True).stdout.splitlines()
        for r, n in zip(rows, names):
            r["name"] = n.split("(")[0].replace("void ", "")
    return rows


def check(rows):
    bad, seen = [], {p: 0 for p, _ in BOUNDS}
    for r in rows:
        for pat, lim in BOUNDS:
            if pat not in r.get("name", ""):
                continue
            seen[pat] += 1
            for k, v in lim.items():
                got = r.get(k, -1)
                ok = got >= v if k == "occupancy" else 0 <= got <= v
                if not ok:
                    bad.append("%s: %s = %d, budget %s %d" % (r["name"], k, got, ">=" if k == "occupancy" else "<=", v))
    for pat, n in seen.items():
        if n == 0:
            bad.append("no kernel matches the budget pattern %r (renamed? update check_resources.py)" % pat)
    return bad


def main(argv):
    d = argv[1] if len(argv) > 1 else ".obj"
    text = ""
    for f in sorted(os.listdir(d)):
        if f.endswith(".remarks"):
            text += open(os.path.join(d, f), errors="replace").read()
    rows = parse(text)
    if not rows:
        print("check_resources: no kernel-resource-usage remarks under %s" % d, file=sys.stderr)
        return 2
    if "--print" in argv:
        for r in rows:
            print("%-92s VGPR %3d SGPR %3d scratch %3d occupancy %d" % (r["name"][-92:], r.get("vgpr", -1), r.get("sgpr", -1),
                                                                      r.get("scratch", -1), r.get("occupancy", -1)))
    bad = check(rows)
    for b in bad:
        print("check_resources: " + b, file=sys.stderr)
    if not bad:
        print("check_resources: %d kernels, every register / occupancy budget holds" % len(rows))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
