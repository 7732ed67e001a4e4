#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (--kernel-trace --stats) into a small text table under profiles/."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
rows = list(cur.execute("select * from top_kernels"))
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (view top_kernels of %s)\n" % db.split("/")[-1])
    f.write(" | ".join(cols) + "\n")
    for r in rows:
        f.write(" | ".join(str(x) for x in r) + "\n")
print(open(out).read()[:6000])
