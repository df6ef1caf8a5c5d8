#!/bin/bash
# bench.py lines of every workload on the current kernels -> gpurun_out/<tag>/*.json (one JSON line each)
TAG=${1:-bench_matrix}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-other-configs "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"; tail -c 400 $OUT/$name.err | grep -i -E "error|Traceback" ; }
run cfg2_default
run cfg2_300steps --steps 300 --warmup 15
run driver_cmd --gpus 1 --steps 20 --warmup 5
run cfg2_steps20 --steps 20 --warmup 5
run cfg2_variant3_literal --variant 3
run cfg2_variant3_literal_serial --variant 3 --pipeline 1
run cfg2_RM_canonical --method RM --variant 1
run cfg2_crash --gather crash
run cfg2_steer --gather steer
run cfg2_steer_serial --gather steer --pipeline 1
run cfg2_f32map --opt code_map=0
run cfg3_CDDT112 --workload cfg3 --method CDDT --theta-disc 112 --steps 40
run cfg2_serial --pipeline 1
run cfg2_RM --method RM
run cfg2_BL --method BL --steps 60
run cfg2_CDDT --method CDDT
run cfg2_32k --poses 32768 --steps 100
run cfg2_32k_serial --poses 32768 --steps 100 --pipeline 1
run cfg2_2048 --poses 2048
run cfg2_2048_serial --poses 2048 --pipeline 1
run cfg2_200 --poses 200 --pipeline 1
run cfg3_GLT --workload cfg3 --steps 60
run cfg3_GLT_serial --workload cfg3 --steps 60 --pipeline 1
run cfg3_CDDT --workload cfg3 --method CDDT --steps 40
run cfg3_CDDT_serial --workload cfg3 --method CDDT --steps 40 --pipeline 1
run cfg3_RMGPU --workload cfg3 --method RMGPU --steps 40
run cfg4_1M --workload cfg4 --steps 10 --warmup 2
run cfg4_shard131072 --workload cfg4 --poses 131072 --steps 40 --warmup 4
run cfg4_shard131072_steer --workload cfg4 --poses 131072 --steps 10 --warmup 2 --gather steer
run cfg4_4096 --workload cfg4 --poses 4096
run cfg5 --workload cfg5 --steps 20 --warmup 3
run cfg5_shard32768 --workload cfg5 --poses 32768 --steps 60
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print("%-20s %10.0f Mrays/s [%8.0f..%8.0f] %8.4f ms/step  frac %.3f  serial kernel %.4f ms frac %.3f  verified %s | %s | %s" % (os.path.basename(f)[:-5], d["value"], d["value_min"], d["value_max"], d["ms_per_step"], r.get("frac",0), r.get("serial",{}).get("kernel_ms",0), r.get("serial",{}).get("frac",0), d.get("verified"), d["config"]["pipeline"], d["config"]["kernel"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
