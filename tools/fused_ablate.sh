#!/bin/bash
# diagnostic: time the fused kernels with parts of the work replaced by trivial stand-ins
cd $GRAFT_REPO_ROOT/hifiles-solver_amd
for A in ${MASKS:-0 1 2 4 8 16 7 15 31}; do
  rm -f libhfx.so
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DHFX_ABLATE=$A" libhfx.so > /dev/null 2>&1
  echo "== ablate mask $A"
  python ../bench.py --steps 4 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_rk_stage'], d['roofline']['kernels_ms'])" 2>&1 | tail -1
done
rm -f libhfx.so; make libhfx.so > /dev/null 2>&1
