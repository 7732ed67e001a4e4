#!/bin/bash
# as tools/ab.sh with several variant directories: tools/ab2.sh "ab ab2" [bench args]
dirs=$1; shift
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'stage %.4f ms' % d['ms_per_rk_stage'], {k: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"; }
for rep in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu "$@" 2>/dev/null | line base
  for v in $dirs; do
    HFX_LIB_DIR=$PWD/hifiles-solver_amd/$v python bench.py --steps 20 --warmup 3 --no-cpu "$@" 2>/dev/null | line $v
  done
done
