#!/bin/bash
# true read traffic of the pairwise face kernels: the L2's memory-side read requests by size
# (TCC_EA0_RDREQ = all, _32B = the 32-byte ones; gfx950 also issues 128-byte requests, listed if the counter exists)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_face
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/tcc_names.txt
cat $OUT/tcc_names.txt
pass() { # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$n -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --reps 1 > $OUT/$n.log 2>&1
}
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass hit TCC_HIT_sum TCC_MISS_sum
pass req TCC_REQ_sum TCC_READ_sum
python - <<PY
import csv, collections, glob
for n in ("rd", "hit", "req"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % n)
    if not f:
        print(n, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for r in csv.DictReader(open(f[0])):
        pass
    disp = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        key = (r["Dispatch_Id"])
        if key in seen: continue
        seen.add(key); disp[r["Kernel_Name"].split("(")[0][:60]] += 1
    for k, v in acc.items():
        print(n, k, {c: x / disp[k] for c, x in v.items()}, "dispatches", disp[k])
PY
