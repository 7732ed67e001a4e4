#!/bin/bash
# rocprofv3 evidence of round 2: kernel stats (--kernel-trace --stats) and, in SEPARATE counter-only passes, HBM traffic and
# issue counters, for (a) the default bench command (split3), (b) the general fused stage on the mixed channel and on
# tetrahedra, (c) the partitioned stage with libhfx's RCCL transport on one self-partitioned rank, (d) configs[4]'s
# ingredients (over-integration + shock capturing), (e) the per-method path with dense FP64 MFMA contractions
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
stats() { # name, bench args...
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/$n -o $n -- python $R/bench.py "$@" --no-cpu --reps 1 > $OUT/$n.log 2>&1
  python $R/tools/prof_summary.py $OUT/$n/${n}_results.db $OUT/${n}_kernel_stats.txt > /dev/null 2>> $OUT/$n.log
}
traffic() { # name, bench args...
  n=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${n}_f -o f --output-format csv -- python $R/bench.py "$@" --no-cpu --reps 1 > $OUT/${n}_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${n}_w -o w --output-format csv -- python $R/bench.py "$@" --no-cpu --reps 1 > $OUT/${n}_w.log 2>&1
  python $R/tools/pmc_traffic.py $OUT/${n}_f/f_counter_collection.csv $OUT/${n}_w/w_counter_collection.csv $OUT/${n}_traffic.json > /dev/null 2>> $OUT/${n}_f.log
}
stats split3 --steps 10 --warmup 2
traffic split3 --steps 2 --warmup 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/split3_p -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --reps 1 > $OUT/split3_p.log 2>&1
python $R/tools/pmc_summary.py $OUT/split3_p/p_counter_collection.csv $OUT/split3_pmc.txt > /dev/null 2>&1
stats general_mixed --workload mixed --steps 4 --warmup 1
traffic general_mixed --workload mixed --steps 1 --warmup 1
stats general_tets --workload tets --steps 4 --warmup 1
stats partitioned --self-partition --steps 10 --warmup 2
stats config5 --steps 6 --warmup 2 --over-int-order 6 --shock-s0 1e-3
stats dense --mode dense --steps 3 --warmup 1
stats les --les-cs 0.325 --steps 6 --warmup 2
ls $OUT
