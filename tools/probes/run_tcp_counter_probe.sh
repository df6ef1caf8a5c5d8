#!/bin/bash
# What does TCP_TOTAL_CACHE_ACCESSES_sum count?  tools/probes/tcp_probe.hip (wave-loads whose 64 lanes touch L
# distinct lines in known lane patterns) under a PMC pass: counter per wave-load next to the measured
# clk / wave-load / CU of the same pattern.
set -u
OUT=$PWD/gpurun_out/${1:-tcp_counter_probe}
mkdir -p "$OUT"
export TMPDIR=/tmp
hipcc -O2 --offload-arch=gfx950 -o /tmp/tcp_probe tools/probes/tcp_probe.hip || exit 1
/tmp/tcp_probe > "$OUT/timing.txt"
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/pmc" -- /tmp/tcp_probe > "$OUT/pmc.txt" 2> "$OUT/pmc.err" || echo "pmc pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
# every configuration: a 10-iteration warm-up launch, then the 2000-iteration one; the counters of a dispatch
# arrive in one row per counter (dimension instances summed above).  Keep the dispatches with the large counts.
big = [i for i in ids if rows[i].get("TCP_TOTAL_READ_sum", 0) > 1e7]
timing = [l.split() for l in open(os.path.join(out, "timing.txt")).read().splitlines()[2:]]
wl = 256 * 32 * 2000 * 8.0
print("dispatches %d, large ones %d, configurations %d" % (len(ids), len(big), len(timing)))
print("%-12s %4s %16s %22s %14s %14s" % ("pattern", "L", "clk/wave-load/CU", "CACHE_ACCESSES/waveload", "TOTAL_READ/wl", "TCC_READ_REQ/wl"))
for t, i in zip(timing, big):
    c = rows[i]
    print("%-12s %4s %16s %22.2f %14.2f %14.3f" % (t[0], t[1], t[3], c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / wl,
                                                 c.get("TCP_TOTAL_READ_sum", 0) / wl, c.get("TCP_TCC_READ_REQ_sum", 0) / wl))
PY
rm -rf "$OUT/pmc"
