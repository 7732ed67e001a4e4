#!/bin/bash
# one-box sweep of a split3 knob: tools/knob_sweep.sh NAME v1 v2 ...  (each value twice, interleaved)
name=$1; shift
for rep in 1 2; do
for v in "$@"; do
  python bench.py --steps 20 --warmup 3 --no-cpu --reps 3 --opt $name=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('%-24s' % '$name=$v', 'stage %.4f' % d['ms_per_rk_stage'], {k: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"
done
done
