#!/bin/bash
# rocprofv3 kernel stats of the P3 tet / prism bench (general dense-MFMA per-method path)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_simplex
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/tools/bench_simplex.py --steps 2 > $OUT/stats.log 2>&1
python $R/tools/prof_summary.py $OUT/stats/stats_results.db $OUT/kernel_stats.txt > /dev/null
head -24 $OUT/kernel_stats.txt | cut -c1-220
