#!/usr/bin/env python3
"""Fold committed PMC summaries (tools/prof_pmc.sh -> pmc_summary.json) into profiles/pmc_traffic.json: one entry per
(workload, method, poses per launch, kernel with template arguments, grid) — the key bench.py matches its launch
against.  usage: python tools/pmc_traffic_update.py <profile dir under profiles/> <workload> <method> <poses> [note]
Every entry records the commit the pass was taken on (HEAD when the summary is folded: fold right after the run).
A theta-major CDDT step is three kernels (prep, search, fan): their bytes and VALU counts are summed into the search
kernel's entry; method "RMGPU+rollout" (rl_car_rollout_check) sums every kernel of the chain into the march's entry."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof, workload, method, poses = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
note = sys.argv[5] if len(sys.argv) > 5 else None
mode = sys.argv[6] if len(sys.argv) > 6 else None      # "steer": the same kernel with plain range stores + FollowGap behind it
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
summ = json.load(open(os.path.join(ROOT, prof, "pmc_summary.json")))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
doc = json.load(open(path))
MAIN = ("rm_fan_stream_kernel", "lut_fan_lds_kernel", "bl_fan_stream_kernel", "cddt_fan_bins_kernel", "cddt_theta_search_kernel", "cddt_theta_search2_kernel")
for key, v in summ.items():
    if not any(key.startswith(m) for m in MAIN) or v.get("_dispatches", 0) < 4 or "FETCH_SIZE" not in v:
        continue
    name, grid = key.rsplit(" [grid ", 1)
    grid = int(grid.rstrip("]"))
    e = {"workload": workload, "method": method, "poses": poses, "kernel": "scan::" + name, "grid": grid,
         "bytes": int(round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)), "commit": commit}
    if "SQ_INSTS_VALU" in v:
        e["valu_insts"] = int(round(v["SQ_INSTS_VALU"]))
    # kernels that belong to the same step and are summed into this entry
    extra = []
    if name.startswith("cddt_theta_search"):
        extra = [k for k in summ if k.startswith(("cddt_theta_prep_kernel", "cddt_theta_fan_kernel"))]
    elif method == "RMGPU+rollout":
        extra = [k for k in summ if k != key and k.startswith(("rollout_kernel", "pose_prep", "pose_scatter", "tile_scan", "crash_reduce"))]
    if extra:
        e["main_kernel_bytes"] = e["bytes"]
        for k in extra:
            if "FETCH_SIZE" in summ[k]:
                e["bytes"] += int(round((2 * summ[k]["FETCH_SIZE"] + summ[k]["WRITE_SIZE"]) * 1024))
            if "SQ_INSTS_VALU" in summ[k] and "valu_insts" in e:
                e["valu_insts"] += int(round(summ[k]["SQ_INSTS_VALU"]))
        e["summed_kernels"] = sorted(k.split(" [grid")[0] for k in extra)
    if v.get("SQ_ACTIVE_INST_VALU"):
        e["lanes_per_valu"] = round(v.get("SQ_THREAD_CYCLES_VALU", 0) / v["SQ_ACTIVE_INST_VALU"], 1)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in v:
        e["tcp_line_accesses"] = int(round(v["TCP_TOTAL_CACHE_ACCESSES_sum"]))
    if "SQ_INSTS_VMEM_RD" in v:
        e["wave_loads"] = int(round(v["SQ_INSTS_VMEM_RD"]))
    if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
        e["l2_hit"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3)
    if v.get("SQ_WAVE_CYCLES"):
        e["wait_frac"] = round(v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
    e["profile"] = prof
    if note:
        e["note"] = note
    e["mode"] = mode
    same = lambda o: all(o.get(k) == e[k] for k in ("workload", "method", "poses", "kernel", "grid", "mode"))
    doc["entries"] = [o for o in doc["entries"] if not same(o)] + [e]
    print("entry:", json.dumps(e))
json.dump(doc, open(path, "w"), indent=1)
