#!/bin/bash
# A/B of one hfx_ctx_set_option knob on the box this runs on: tools/ab_opt.sh NAME=VALUE [bench args]
opt=$1; shift
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'stage %.4f ms' % d['ms_per_rk_stage'], {k: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"; }
for rep in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu "$@" 2>/dev/null | line default
  python bench.py --steps 20 --warmup 3 --no-cpu --opt $opt "$@" 2>/dev/null | line "$opt"
done
