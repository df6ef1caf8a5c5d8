#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 kernel trace (kernel_trace_tail.csv): for one kernel name pattern,
the average begin-to-end duration of a launch, the spacing between consecutive launch starts (= the
machine time a launch costs when launches overlap), and the average number of launches in flight."""
import csv, sys
import numpy as np
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "rm_fan_stream")
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # optional: only launches of this many workgroups
# (rocprofv3's Grid_Size_X counts work-items: workgroups x Workgroup_Size_X)
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"] and
        (not grid or int(r["Grid_Size_X"]) in (grid, grid * int(r.get("Workgroup_Size_X", 1) or 1)))]
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
o = np.argsort(st); st, en = st[o], en[o]
# steady state: drop the first and last few
k = max(2, len(st) // 10)
s, e = st[k:-k], en[k:-k]
dur = (e - s) / 1e3
spacing = np.diff(s) / 1e3
span = (e.max() - s.min()) / 1e3
print("%s: %d launches analysed" % (pat, len(s)))
print("  begin-to-end duration of a launch: mean %.2f us (min %.2f, max %.2f)" % (dur.mean(), dur.min(), dur.max()))
print("  start-to-start spacing:            mean %.2f us (median %.2f)" % (spacing.mean(), np.median(spacing)))
print("  launches in flight on average:     %.2f  (sum of durations / wall span)" % (dur.sum() / span))
print("  queues used: %s" % sorted({r["Queue_Id"] for r in rows}))
# bench.py times BURSTS of K steps with an idle gap between them: per burst (cut where the spacing exceeds
# 4x its median), launches per burst and the burst's span / launches = the machine time a launch costs
med = np.median(spacing)
cuts = np.where(spacing > 4 * med)[0]
bounds = np.concatenate([[0], cuts + 1, [len(s)]])
per = []
for a, b in zip(bounds[:-1], bounds[1:]):
    if b - a >= 4:
        per.append(((e[a:b].max() - s[a]) / 1e3 / (b - a), b - a, dur[a:b].sum() / ((e[a:b].max() - s[a]) / 1e3)))
if per:
    per = np.array(per)
    print("  bursts: %d of ~%d launches; span / launches per burst: median %.2f us (min %.2f, max %.2f); "
          "launches in flight inside a burst: %.2f" % (len(per), int(np.median(per[:, 1])), np.median(per[:, 0]),
                                                       per[:, 0].min(), per[:, 0].max(), np.median(per[:, 2])))
