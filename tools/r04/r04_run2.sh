#!/bin/bash
set -u
O=gpurun_out/r04_run2; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "gpu rc=$?" >> $O/rc.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "libm" -s > $O/pytest_libm.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_b.json 2> $O/bench_steps20_b.err
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --gather crash > $O/bench_crash.json 2> $O/bench_crash.err
python tools/gpu_noise_cost.py > $O/noise_cost.txt 2>&1
python tools/r04/burst_overhead.py > $O/burst_overhead.txt 2>&1
tail -n 3 $O/pytest_gpu.log; cat $O/rc.txt; grep -v amdgpu $O/noise_cost.txt
