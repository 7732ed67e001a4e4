#!/bin/bash
# LDS / issue counters of the default bench path (counters only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/p -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/p.log 2>&1
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/p/p_counter_collection.csv")):
    n=r["Kernel_Name"].split("<")[0].replace("void hfx::","")
    if n.startswith("__amd") or "at::" in n: continue
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: "%.3g"%(sum(x)/len(x)) for c,x in v.items()})
PY
