#!/bin/bash
# round 4, final validation of the build that is committed: the whole GPU suite, the driver's command, host latencies, a long fuzz
set -u
O=gpurun_out/r04_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu rc=$?" >> $O/rc.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; echo "bench rc=$?" >> $O/rc.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_2.json 2>> $O/driver_cmd.err
python tools/r04/host_latency_breakdown.py 2>&1 | grep -v amdgpu.ids > $O/host_latency.txt
bash tools/r04/sync_latency_env.sh >> $O/host_latency.txt 2>&1
timeout ${FUZZ_SECS:-2820} python tests/gpu_fuzz.py --seconds ${FUZZ_SECS2:-2700} > $O/fuzz_45min.log 2>&1; echo "fuzz rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -2 $O/pytest_gpu.log; tail -1 $O/fuzz_45min.log
