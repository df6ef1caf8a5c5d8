for cfg in "4 8 3" "4 6 3" "3 8 3" "3 6 3" "4 5 3" "4 8 2" "4 3 3"; do
  set -- $cfg
  for st in "--steps 20 --warmup 5" "--steps 300"; do
    python bench.py --no-cpu-baseline --no-extras --no-verify $st --pipeline $1 --grid-mult $2 --opt slots=$3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $1 grid_mult $2 slots $3 [$st]', d['value'], d['ms_per_step'], d['value_min'], d['value_max'])"
  done
done
