#!/bin/bash
# builds and runs tools/probes/fetch_probe.hip under separate rocprofv3 --pmc passes; prints KiB per kernel
set -u
OUT=$PWD/gpurun_out/${1:-fetch_probe}
mkdir -p "$OUT"
export TMPDIR=/tmp
hipcc -O2 --offload-arch=gfx950 -o /tmp/fetch_probe tools/probes/fetch_probe.hip || exit 1
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"; do
  tag=$(echo $P | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/$tag" -- /tmp/fetch_probe > "$OUT/$tag.txt" 2> "$OUT/$tag.err" || echo "pass failed: $P"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(dict)
for f in glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] = acc[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k in ("gather1", "gather2", "quad1", "stream16"):
    for kk, v in acc.items():
        if k in kk:
            print(k, {c: round(x, 1) for c, x in sorted(v.items())})
PY
cat "$OUT/FETCH_SIZE.txt"
