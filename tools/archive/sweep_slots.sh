for lw in 8 12 16; do for gm in 2 3 4 5 6; do python tools/gpu_pipeline.py --opt slots=2 --opt low_water=$lw --pipes 3,4 --grid-mults $gm 2>&1 | grep grid_mult | sed "s/^/lw $lw /"; done; done
echo "== serial big workloads slots 1 vs 2"
for wl in "cfg4 131072" "cfg5 32768" "cfg5 262144" "cfg3 65536" "cfg2 8192" "cfg2 16384"; do set -- $wl; for sl in 1 2; do python tools/gpu_pipeline.py --workload $1 --poses $2 --steps 30 --opt slots=$sl --pipes 1 --grid-mults 8 2>&1 | grep grid_mult | sed "s/^/$1 $2 slots $sl /"; done; done
