#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter)."""
import csv
import re
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(src)):
    name = re.sub(r"^void hfx::", "", r["Kernel_Name"]).split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out, "w") as f:
    f.write("# rocprofv3 --pmc (separate pass, counters only) on `python bench.py --steps 2 --warmup 1 --no-cpu` (default path: split3)\n")
    f.write("# averages per launch; SQ_*_CYCLES / SQ_WAIT_* are in units of 4 cycles\n")
    for k in sorted(acc):
        f.write(k + "\n")
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        for n in sorted(c):
            f.write("    %-22s %.4g\n" % (n, c[n]))
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            f.write("    wait_any/wave_cycles   %.2f   active_any/wave_cycles %.2f\n" %
                    (c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"] / 4))
print(open(out).read()[:3000])
