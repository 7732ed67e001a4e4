#!/bin/bash
# BASELINE.json configs[4]'s ingredients at bench size on one GPU: 32^3 P4 hexes, HLLC, over-integration with 7
# cubature points per direction, shock capturing after every stage -- sum-factorised kernels vs the dense MFMA form
cd $GRAFT_REPO_ROOT
for E in ${OPTS:-tensor_ops=1 tensor_ops=0}; do
  echo "== $E"
  python bench.py --opt $E --steps 10 --warmup 2 --no-cpu --over-int-order 6 --shock-s0 1e-3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['path'], round(d['value']/1e9,2), 'G DOF-updates/s', round(d['ms_per_rk_stage'],4), 'ms/stage', {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"
done
