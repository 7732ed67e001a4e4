#!/bin/bash
# workgroups per CU of the persistent element kernels of the split path (same box, one call)
for g in 2 4 8 16 32 128; do
  python bench.py --mode split3 --steps 20 --no-cpu --opt split_grid_per_cu=$g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid/CU $g', round(d['ms_per_rk_stage'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"
done
