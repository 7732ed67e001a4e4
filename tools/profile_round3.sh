#!/bin/bash
# rocprofv3 evidence of round 3: kernel stats (--kernel-trace --stats) and, in SEPARATE counter-only passes, HBM traffic and
# issue / MFMA counters, for (a) the default bench command (split3), (b) the general fused stage on the mixed channel and on
# tetrahedra, (c) the partitioned stage with libhfx's RCCL transport on one self-partitioned rank, (d) configs[4]'s
# ingredients (over-integration + shock capturing), (e) LES with the closure in the flux kernel, (f) the per-method path with
# dense FP64 MFMA contractions (with its MFMA-busy counters), (g) the pairwise face kernels' L2 read requests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu --no-also --no-api-path --reps 1"
stats() { # name, bench args...
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/$n -o $n -- python $R/bench.py "$@" $B > $OUT/$n.log 2>&1
  python $R/tools/prof_summary.py $OUT/$n/${n}_results.db $OUT/${n}_kernel_stats.txt > /dev/null 2>> $OUT/$n.log
  echo "stats $n done"
}
traffic() { # name, bench args...
  n=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${n}_f -o f --output-format csv -- python $R/bench.py "$@" $B > $OUT/${n}_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${n}_w -o w --output-format csv -- python $R/bench.py "$@" $B > $OUT/${n}_w.log 2>&1
  python $R/tools/pmc_traffic.py $OUT/${n}_f/f_counter_collection.csv $OUT/${n}_w/w_counter_collection.csv $OUT/${n}_traffic.json > /dev/null 2>> $OUT/${n}_f.log
  echo "traffic $n done"
}
pmc() { # name, "counters", bench args...
  n=$1; c=$2; shift; shift
  rocprofv3 --pmc $c --kernel-trace -d $OUT/${n}_p -o p --output-format csv -- python $R/bench.py "$@" $B > $OUT/${n}_p.log 2>&1
  python $R/tools/pmc_summary.py $OUT/${n}_p/p_counter_collection.csv $OUT/${n}_pmc.txt > /dev/null 2>&1
  echo "pmc $n done"
}
stats split3 --steps 10 --warmup 2
traffic split3 --steps 2 --warmup 1
pmc split3 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" --steps 2 --warmup 1
stats general_mixed --workload mixed --steps 4 --warmup 1
traffic general_mixed --workload mixed --steps 1 --warmup 1
pmc general_mixed "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" --workload mixed --steps 1 --warmup 1
stats general_tets --workload tets --steps 4 --warmup 1
stats general_mixed_les --workload mixed --les-cs 0.325 --steps 4 --warmup 1
stats partitioned --self-partition --steps 10 --warmup 2
stats config5 --steps 6 --warmup 2 --over-int-order 6 --shock-s0 1e-3
stats les --les-cs 0.325 --steps 6 --warmup 2
stats dense --mode dense --steps 3 --warmup 1
pmc dense "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" --mode dense --steps 1 --warmup 1
# the pairwise face kernels' memory-side read requests (the L2's EA requests; 128 bytes each on this part)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $OUT/face_rd -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 $B > $OUT/face_rd.log 2>&1
python - <<PY > $OUT/face_tcc_requests.txt
import csv, collections, glob
f = glob.glob("$OUT/face_rd/*counter_collection.csv")
print("# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (counters only) on python bench.py --steps 2 --warmup 1: L2 read requests to memory per launch")
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void hfx::", "")[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); disp[k] += 1
    for k, v in sorted(acc.items()):
        if k.startswith("__amd") or "at::" in k: continue
        print(k, {c: round(x / disp[k]) for c, x in v.items()}, "launches", disp[k])
PY
ls $OUT
