#!/bin/bash
# rocprofv3 kernel stats of config 5's ingredients at bench size (32^3 P4, over-integration + shock capturing)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_config5
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 6 --warmup 1 --no-cpu --over-int-order 6 --shock-s0 1e-3 > $OUT/stats.log 2>&1
python $R/tools/prof_summary.py $OUT/stats/stats_results.db $OUT/kernel_stats.txt | head -14
