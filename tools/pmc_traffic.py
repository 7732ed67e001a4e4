#!/usr/bin/env python3
"""HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kB units).

    pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE under-reports by 2x on
this part, WRITE_SIZE does not:  traffic = 2 * FETCH_SIZE + WRITE_SIZE.  Values are averaged over the
launches of each kernel (short name = text before the template arguments)."""
import csv
import json
import re
import sys
from collections import defaultdict


def collect(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void\s+", "", row["Kernel_Name"])
        name = re.sub(r"^hfx::", "", name.split("<")[0].split("(")[0])
        acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


f, nf = collect(sys.argv[1], "FETCH_SIZE")
w, nw = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) & set(w)):
    if k.startswith("__amd") or "at::" in k:
        continue
    out[k] = dict(fetch_kB_raw=f[k], write_kB_raw=w[k], launches=nf[k],
                  traffic_bytes_corrected=(2.0 * f[k] + w[k]) * 1024.0,
                  traffic_bytes_uncorrected=(f[k] + w[k]) * 1024.0)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
