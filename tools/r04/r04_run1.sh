#!/bin/bash
# round 4, GPU call 1: new tests first, the whole GPU suite, the bench lines of the new modes
set -u
O=gpurun_out/r04_run1; mkdir -p $O
python -m pytest tests/test_gpu_multi_device.py -x -q -m gpu > $O/pytest_multi.log 2>&1; echo "multi rc=$?" >> $O/rc.txt
python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $O/pytest_dist.log 2>&1; echo "dist rc=$?" >> $O/rc.txt
python -m pytest tests -q -m gpu --deselect tests/test_gpu_dist.py --deselect tests/test_gpu_multi_device.py > $O/pytest_rest.log 2>&1; echo "rest rc=$?" >> $O/rc.txt
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --gather crash > $O/bench_crash.json 2> $O/bench_crash.err
python bench.py --no-cpu-baseline --gather steer > $O/bench_steer.json 2> $O/bench_steer.err
python bench.py --no-cpu-baseline --method CDDT --workload cfg3 --steps 40 > $O/bench_cfg3_cddt.json 2> $O/bench_cfg3_cddt.err
python bench.py --gpus 2 --same-device --backend gloo --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_2ranks.json 2> $O/bench_2ranks.err
bash tools/prof_pmc.sh r04_pmc_cfg5_shard_pipe --workload cfg5 --poses 32768 --grid-mult 3 --opt slots=2 > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg4_4096_pipe --workload cfg4 --poses 4096 --grid-mult 3 --opt slots=2 > /dev/null
bash tools/prof_pmc.sh r04_pmc_cfg2_bl_pipe --method BL --grid-mult 3 > /dev/null
tail -3 $O/pytest_*.log; cat $O/rc.txt
