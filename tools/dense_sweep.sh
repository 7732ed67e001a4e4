#!/bin/bash
# one-box sweep of the dense MFMA contraction's workgroup shape on the per-method paths
for nw in 0 4 8; do for sp in 0 1 2 4; do
  if [ $nw = 0 ] && [ $sp != 0 ]; then continue; fi
  if [ $nw != 0 ] && [ $sp = 0 ]; then continue; fi
  for w in "--mode dense --steps 4" "--workload tets --mode methods --steps 2" "--workload prisms --mode methods --steps 2" "--workload mixed --mode methods --steps 2"; do
    python bench.py $w --warmup 1 --reps 2 --no-cpu --opt dense_waves=$nw --opt dense_split=$sp 2>gpurun_out/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('nw=$nw spl=$sp', '$w'[:17], round(d['ms_per_rk_stage'],3))"
  done
done; done
