#!/bin/bash
# sweep of compile-time knobs of the split fused path (rebuilds libhfx.so on the GPU box)
# usage: FLAGSETS="-DA=1;-DA=2 -DB=3;..." [OPTSETS="loader_wave=0;loader_wave=1"] [MODE=split3] bash tools/split_sweep.sh
cd $GRAFT_REPO_ROOT/hifiles-solver_amd
IFS=';' read -ra SETS <<< "${FLAGSETS:--DHFX_SPLIT_WAVES_RES=3;-DHFX_SPLIT_WAVES_RES=4}"
IFS=';' read -ra ENVS <<< "${OPTSETS:-xcd_order=1}"
for F in "${SETS[@]}"; do
  rm -f libhfx.so
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $F" libhfx.so > /dev/null 2>&1
  for E in "${ENVS[@]}"; do
    echo "== $F | $E"
    python ../bench.py --opt $E --steps 20 --warmup 2 --no-cpu --mode ${MODE:-split3} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,2), round(d['ms_per_rk_stage'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"
  done
done
rm -f libhfx.so; make libhfx.so > /dev/null 2>&1
