# cfg2 default with 4 vs 8 HIP hardware queues (GPU_MAX_HW_QUEUES) and 4..8 steps in flight (N_LAUNCH_CTX = 8 build)
for q in 4 8; do for p in 4 6 8; do for gm in 2 3; do
echo "== HWQ $q pipeline $p gm $gm"
GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --pipeline $p --grid-mult $gm --steps 600 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['pipeline'])"
done; done; done
