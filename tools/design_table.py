#!/usr/bin/env python3
"""Regenerate the rows of DESIGN.md section 5's table from the committed bench lines (profiles/r04/bench/*.json)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def L(n):
    return json.loads(open(os.path.join(ROOT, "profiles/r04/bench/%s.json" % n)).read().strip().splitlines()[-1])
def fmt(n):
    return f"{int(round(n)):,}".replace(",", " ")
rows = [
 ("**cfg2: 2049² maze, 4096×1081** (bench default, 300 steps)", "RMGPU (K1b)", "4 steps in flight, 2 rays per lane", "cfg2_default"),
 ("cfg2, **the driver's command** (`--steps 20 --warmup 5`)", "RMGPU", "same", "driver_cmd"),
 ("cfg2", "RMGPU", "serial (`--pipeline 1`)", "cfg2_serial"),
 ("cfg2", "RM (0.999)", "4 in flight", "cfg2_RM"),
 ("cfg2, `--gather crash` (fused `Car::isCrashed` per 128-pose roll-out)", "RMGPU", "4 in flight", "cfg2_crash"),
 ("cfg2, `--gather crash`, the driver's 20 steps", "RMGPU", "4 in flight", "cfg2_crash_steps20"),
 ("cfg2, `--gather steer` (scan + FollowGap per scan)", "RMGPU", "4 in flight", "cfg2_steer"),
 ("cfg2", "Bresenham, stream (K2b)", "4 in flight", "cfg2_BL"),
 ("cfg2", "CDDT θ=108 (pose-major, blocked table)", "4 in flight", "cfg2_CDDT"),
 ("cfg2, 2048 poses", "RMGPU", "4 in flight", "cfg2_2048"),
 ("cfg2, 2048 poses", "RMGPU", "serial", "cfg2_2048_serial"),
 ("cfg2, 200 poses (the reference's roll-out batch)", "RMGPU", "serial", "cfg2_200"),
 ("cfg2, 32 768 poses", "RMGPU", "4 in flight", "cfg2_32k"),
 ("cfg2, 32 768 poses", "RMGPU", "serial", "cfg2_32k_serial"),
 ("cfg3: 2000² maze, 65 536×1081", "GiantLUT θ=1442 (K3), non-temporal row loads, two generations of workgroups", "serial (bench default for GiantLUT since round 4)", "cfg3_GLT"),
 ("cfg3", "CDDT θ=108 (K3b, θ-major)", "4 in flight", "cfg3_CDDT"),
 ("cfg3", "CDDT", "serial", "cfg3_CDDT_serial"),
 ("cfg3", "RMGPU", "serial", "cfg3_RMGPU"),
 ("cfg4: colombia, 1 048 576 roll-out poses ×1081 in one call", "RMGPU", "serial", "cfg4_1M"),
 ("cfg4, one rank's shard of 8 (131 072 poses)", "RMGPU", "serial", "cfg4_shard131072"),
 ("cfg4 shard, `--gather crash` (200-pose roll-outs)", "RMGPU", "serial", "cfg4_shard131072_crash"),
 ("cfg4, 4096 poses", "RMGPU", "4 in flight", "cfg4_4096"),
 ("cfg5: 4096² maze, 262 144×720 + noise", "RMGPU", "serial", "cfg5"),
 ("cfg5, one rank's shard of 8 (32 768 poses)", "RMGPU", "4 in flight", "cfg5_shard32768"),
]
out = []
for wl, m, sch, f in rows:
    d = L(f); r = d["roofline"]; s = r.get("serial", {})
    fh = r.get("frac_hbm")
    out.append("| %s | %s | %s | %s (%s … %s) | %.4f | %.2f / %s | %.4f ms, %.2f |" % (
        wl, m, sch, fmt(d["value"]), fmt(d["value_min"]), fmt(d["value_max"]), d["ms_per_step"], r["frac"],
        ("%.2f" % fh) if fh is not None else "—", s.get("kernel_ms", 0), s.get("frac", 0)))
path = os.path.join(ROOT, "DESIGN.md")
txt = open(path).read()
a = txt.index("| **cfg2: 2049² maze, 4096×1081** (bench default, 300 steps)")
b = txt.index("| CPU oracle, same cfg2 inputs")
open(path, "w").write(txt[:a] + "\n".join(out) + "\n" + txt[b:])
print("%d rows written" % len(out))
