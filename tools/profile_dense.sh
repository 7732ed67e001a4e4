#!/bin/bash
# dense FP64 MFMA contraction path: kernel stats + MFMA counters (separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_dense
rm -rf $OUT; mkdir -p $OUT
python $R/bench.py --mode dense --steps 5 --warmup 1 --no-cpu > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --mode dense --steps 5 --warmup 1 --no-cpu > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/bench.py --mode dense --steps 2 --warmup 1 --no-cpu > $OUT/p1.log 2>&1
python $R/tools/prof_summary.py $OUT/stats/stats_results.db $OUT/kernel_stats.txt > /dev/null
tail -1 $OUT/bench.json | cut -c1-900
tail -3 $OUT/p1.log
