#!/bin/bash
# rocprofv3 evidence for a round: kernel stats of the default bench command + PMC passes (separate runs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p3 -o p3 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p4 -o p4 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p4.log 2>&1
python $R/bench.py --steps 20 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
python $R/tools/pmc_traffic.py $OUT/p3/p3_counter_collection.csv $OUT/p4/p4_counter_collection.csv $OUT/traffic.json > /dev/null
python $R/tools/prof_summary.py $OUT/stats/stats_results.db $OUT/kernel_stats.txt > /dev/null
