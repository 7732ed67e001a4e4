#!/bin/bash
# PMC passes over the fused kernels (counters only; no tracing domains mixed in)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p3 -o p3 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p4 -o p4 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p4.log 2>&1
ls -R $OUT | head -40
