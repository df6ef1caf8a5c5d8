#!/usr/bin/env python3
"""Regenerate DESIGN.md section 5's table from the committed bench lines (profiles/r06/bench/*.json): the rows between
the two `<!-- bench table -->` markers (or the R05_TABLE placeholder)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = "profiles/r06/bench"
def L(n):
    p = os.path.join(ROOT, DIR, n + ".json")
    return json.loads(open(p).read().strip().splitlines()[-1]) if os.path.exists(p) else None
def fmt(n):
    return f"{int(round(n)):,}".replace(",", " ")
rows = [
 ("**cfg2: 2049² maze, 4096 × 1081**, 300 steps", "RMGPU (K1b), 4 in flight", "cfg2_300steps"),
 ("cfg2, **the driver's command** (20 steps)", "RMGPU, 4 in flight", "driver_cmd"),
 ("cfg2", "RMGPU, serial", "cfg2_serial"),
 ("cfg2, **upstream-literal arithmetic** (`--variant 3`), 300 steps", "RMGPU, 4 in flight", "cfg2_variant3_literal"),
 ("cfg2, upstream-literal", "RMGPU, serial", "cfg2_variant3_literal_serial"),
 ("cfg2, `--method RM` (0.999; upstream-literal arithmetic by default)", "RM, 4 in flight", "cfg2_RM"),
 ("cfg2, `--method RM --variant 1` (canonical)", "RM, 4 in flight", "cfg2_RM_canonical"),
 ("cfg2 on the float32 step map (`--opt code_map=0`: round 5's kernel)", "RMGPU, 4 in flight", "cfg2_f32map"),
 ("cfg2, `--gather crash` (fused `Car::isCrashed`)", "RMGPU, 4 in flight", "cfg2_crash"),
 ("cfg2, `--gather steer` (scan + FollowGap, one bit per beam)", "RMGPU, 4 in flight", "cfg2_steer"),
 ("cfg2", "Bresenham (K2b), 4 in flight", "cfg2_BL"),
 ("cfg2", "CDDT θ 108 (pose-major), 4 in flight", "cfg2_CDDT"),
 ("cfg2, 2048 poses", "RMGPU, 4 in flight / serial", ("cfg2_2048", "cfg2_2048_serial")),
 ("cfg2, 200 poses (the reference's roll-out)", "RMGPU, serial", "cfg2_200"),
 ("cfg2, 32 768 poses", "RMGPU, 4 in flight / serial", ("cfg2_32k", "cfg2_32k_serial")),
 ("cfg3: 2000² maze, 65 536 × 1081", "GiantLUT θ 1442 (K3), serial", "cfg3_GLT_serial"),
 ("cfg3", "CDDT θ 108 (K3b, θ-major), 4 in flight / serial", ("cfg3_CDDT", "cfg3_CDDT_serial")),
 ("cfg3", "CDDT θ 112 (the reference's bin count), 4 in flight", "cfg3_CDDT112"),
 ("cfg3", "RMGPU, serial", "cfg3_RMGPU"),
 ("cfg4: colombia, 1 048 576 roll-out poses × 1081", "RMGPU, serial", "cfg4_1M"),
 ("cfg4, one rank's shard of 8 (131 072 poses)", "RMGPU, serial", "cfg4_shard131072"),
 ("cfg4, 4096 poses", "RMGPU, 4 in flight", "cfg4_4096"),
 ("cfg5: 4096² maze, 262 144 × 720 + noise", "RMGPU, serial", "cfg5"),
 ("cfg5, one rank's shard of 8 (32 768 poses)", "RMGPU, 4 in flight", "cfg5_shard32768"),
]
out = ["| workload | method, schedule | Mrays/s (min … max burst) | ms / step | frac / frac_hbm | lone kernel ms |", "|---|---|---|---|---|---|"]
for wl, m, f in rows:
    fs = f if isinstance(f, tuple) else (f,)
    ds = [L(x) for x in fs]
    if any(d is None for d in ds):
        continue
    def cell(fn, sep=" / "):
        return sep.join(fn(d) for d in ds)
    fh = lambda d: ("%.2f" % d["roofline"]["frac_hbm"]) if d["roofline"].get("frac_hbm") is not None else "—"
    out.append("| %s | %s | %s | %s | %s | %s |" % (
        wl, m, cell(lambda d: "%s (%s … %s)" % (fmt(d["value"]), fmt(d["value_min"]), fmt(d["value_max"]))),
        cell(lambda d: "%.4f" % d["ms_per_step"]), cell(lambda d: "%.2f / %s" % (d["roofline"]["frac"], fh(d)), " ; "),
        cell(lambda d: "%.4f" % d["roofline"].get("serial", {}).get("kernel_ms", 0))))
path = os.path.join(ROOT, "DESIGN.md")
txt = open(path).read()
block = "<!-- bench table -->\n" + "\n".join(out) + "\n<!-- bench table -->"
if "R05_TABLE" in txt:
    txt = txt.replace("R05_TABLE", block)
else:
    a = txt.index("<!-- bench table -->"); b = txt.index("<!-- bench table -->", a + 10) + len("<!-- bench table -->")
    txt = txt[:a] + block + txt[b:]
open(path, "w").write(txt)
print("%d rows written, DESIGN.md %d bytes" % (len(out) - 2, len(txt.encode())))
