#!/bin/bash
# rocprofv3 kernel stats of the LES (WALE) bench on the split path that keeps the corrected gradients (fused=2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_les
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 6 --warmup 1 --no-cpu --les-cs 0.325 > $OUT/stats.log 2>&1
python $R/tools/prof_summary.py $OUT/stats/stats_results.db $OUT/kernel_stats.txt > /dev/null
head -14 $OUT/kernel_stats.txt | cut -c1-200
