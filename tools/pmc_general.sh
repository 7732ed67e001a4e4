#!/bin/bash
# issue / wait / LDS / MFMA counters of the general fused stage (counters only, two passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_general
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/tools/bench_simplex.py --tiles 2048 --steps 1 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- python $R/tools/bench_simplex.py --tiles 2048 --steps 1 > $OUT/p2.log 2>&1
python - <<PY
import csv, collections
for p in ("p1","p2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        rows=list(csv.DictReader(open("$OUT/%s/%s_counter_collection.csv"%(p,p))))
    except Exception as e:
        print(p, "no csv", e); continue
    for r in rows:
        n=r["Kernel_Name"].split("(")[0].replace("void hfx::","")
        if n.startswith("__amd") or "at::" in n: continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k, {c: "%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
