#!/bin/bash
# occupancy sweep of the fused kernels (rebuilds libhfx.so on the GPU box)
# usage: CONFIGS="g,r g,r ..." (waves/SIMD of the gradient and the residual kernel) GRIDS="2 4"
cd $GRAFT_REPO_ROOT/hifiles-solver_amd
for C in ${CONFIGS:-2,2 3,2 3,4 4,4}; do
  WG=${C%,*}; WR=${C#*,}
  rm -f libhfx.so
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DHFX_FUSED_WAVES=$WG -DHFX_FUSED_WAVES_RES=$WR $EXTRA" libhfx.so > /dev/null 2>&1
  for G in ${GRIDS:-2 4}; do
    echo "== waves/SIMD grad $WG res $WR grid/CU $G"
    HFX_FUSED_GRID_PER_CU=$G python ../bench.py --steps 6 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_rk_stage'], d['roofline']['kernels_ms'])"
  done
done
