#!/bin/bash
# A/B of the library in hifiles-solver_amd/ against the variant build in hifiles-solver_amd/ab/ (make variant EXTRA=...):
# the split3 bench twice each, interleaved, on the box this runs on.  Extra arguments go to bench.py.
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'stage %.4f ms' % d['ms_per_rk_stage'], {k: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"; }
for rep in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu "$@" 2>/dev/null | line base
  HFX_LIB_DIR=$PWD/hifiles-solver_amd/ab python bench.py --steps 20 --warmup 3 --no-cpu "$@" 2>/dev/null | line variant
done
