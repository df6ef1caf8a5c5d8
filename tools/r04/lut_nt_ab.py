"""GiantLUT fan kernel: non-temporal row loads (default since round 4) against plain loads (lut_debug 16), lone
launches and 2 / 4 steps in flight; non-temporal range stores (lut_debug 8) on top.
prints: args, lut_debug, (Mrays/s, ms per step, verified, lone kernel ms)"""
import subprocess, json, sys
def run(args):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + args, capture_output=True, text=True).stdout
    for l in out.splitlines():
        try: d = json.loads(l)
        except Exception: continue
        return d["value"], d["ms_per_step"], d.get("verified"), d.get("roofline", {}).get("serial", {}).get("kernel_ms")
for pipe in ("1", "2", "4"):
    args = ["--workload", "cfg3", "--steps", "60", "--pipeline", pipe]
    for rep in range(2):
        for dbg in (0, 16, 8):
            print(args, "lut_debug", dbg, run(args + ["--opt", "lut_debug=%d" % dbg]), flush=True)
