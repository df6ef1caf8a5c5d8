#!/usr/bin/env python3
"""Fold rocprofv3 --pmc csv passes into {"kernel [grid N]": {counter: mean per dispatch}} (last 5 dispatches of
that kernel at that grid: one kernel template serves the pipelined and the lone launch at different grids)."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if not any(t in k for t in ("rm_fan", "pose_bin", "pose_prep", "pose_scatter", "tile_scan", "rm_rays", "bl_", "lut_", "cddt_",
                                        "occ_fan", "rollout_kernel", "crash_", "followgap")):
                continue
            short = k.split("(")[0].replace("void ", "").replace("scan::", "")
            try:
                short += " [grid %d]" % (int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))
            except (KeyError, ValueError):
                pass
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: sum(v[-5:]) / len(v[-5:]) for c, v in cs.items()}
    out[k]["_dispatches"] = max(len(v) for v in cs.values())
json.dump(out, sys.stdout, indent=1, sort_keys=True)
