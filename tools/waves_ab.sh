#!/bin/bash
# flux-kernel occupancy A/B on one box
for v in "flux_waves=2" "flux_waves=3"; do
  python bench.py --mode split3 --steps 20 --no-cpu --opt $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_rk_stage'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"
done
