#!/bin/bash
# A/B of hfx_ctx_set_option knobs on one box: OPTSETS="loader_wave=0;loader_wave=1" [MODE=split3] bash tools/ab_env.sh
cd $GRAFT_REPO_ROOT
IFS=';' read -ra ENVS <<< "${OPTSETS:-xcd_order=1}"
for rep in 1 2; do
for E in "${ENVS[@]}"; do
  echo "== $E"
  python bench.py --opt $E --steps 20 --warmup 2 --no-cpu --mode ${MODE:-split3} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,2), round(d['ms_per_rk_stage'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"
done
done
