#!/bin/bash
# round 6, run 27: FollowGap with the next scan loaded ahead (batches beyond one scan per wave): parity, A/B on cfg4's shard / cfg2 32k
set -u
OUT=gpurun_out/r06_run27; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "followgap" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python - <<'PY' > gpurun_out/r06_run27/big_batch_parity.txt 2>&1
# 20000 scans (> 8192 waves: the look-ahead path) against the one-scan-per-wave kernel and the oracle
import os, sys, numpy as np
sys.path.insert(0, ".")
from pyracecarsimulator_amd.followgap import PyFollowGap
from oracle import oracle as O
rng = np.random.default_rng(5)
for size in (10, 100, 1081, 1280):
    n = 20011
    scans = rng.uniform(0.0, 12.0, (n, size)).astype(np.float32)
    scans[rng.random((n, size)) < 0.4] = 1.0
    a = PyFollowGap(10, 15.0, 1.0e6, 0.004).eval_many(scans)
    os.environ["RL_FOLLOWGAP_PREFETCH"] = "0"
    b = PyFollowGap(10, 15.0, 1.0e6, 0.004).eval_many(scans)
    del os.environ["RL_FOLLOWGAP_PREFETCH"]
    w = np.array([O.followgap_eval(scans[i], 15.0, 1.0e6, 0.004) for i in range(0, n, 7)], np.float32)
    print(size, "ahead == plain:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), " == oracle (every 7th):", np.array_equal(a[::7].view(np.uint32), w.view(np.uint32)))
PY
cat $OUT/big_batch_parity.txt
B="--no-cpu-baseline --no-extras --no-other-configs --gather steer"
for rep in 1 2; do
for pf in 1 0; do
  export RL_FOLLOWGAP_PREFETCH=$pf
  timeout 200 python bench.py $B --workload cfg4 --poses 131072 --steps 10 --warmup 2 > $OUT/cfg4s_steer_pf${pf}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --poses 32768 --steps 40 --warmup 5 > $OUT/cfg2_32k_steer_pf${pf}_$rep.json 2>> $OUT/err.txt
  timeout 200 python bench.py $B --workload cfg5 --poses 32768 --steps 40 --warmup 5 > $OUT/cfg5s_steer_pf${pf}_$rep.json 2>> $OUT/err.txt
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run27/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-26s %10.0f  %.4f ms ver %s" % (f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["verified"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
