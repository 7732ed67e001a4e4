#!/usr/bin/env python3
"""Point tables of the simplex element classes -- container only (reads /root/reference/data).

The reference reads the solution / flux points of triangles and tetrahedra from binary tables at run time
(/root/reference/src/cubature_tri.cpp:50-105 data/tri_inter.bin, src/cubature_tet.cpp:50-105 data/tet_inter.bin: per
order the r, s[, t] coordinates and the weights of the (order+1)(order+2)/2 [x (order+3)/3] points, native doubles).
They are DATA (published point sets), not code; the host mirror's eles_tets / eles_pris need the same numbers, so this
script rewrites the two "internal" tables as one text file, hifiles-solver_amd/data/simplex_points.txt:

    tri_inter <order> <n_pts>      then n_pts lines  r s weight          (17 significant digits)
    tet_inter <order> <n_pts>      then n_pts lines  r s t weight
"""
import os
import struct

REF = os.environ.get("HIFILES_HOME", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hifiles-solver_amd", "data", "simplex_points.txt")


def read(path, ncoord, orders, npts):
    raw = open(path, "rb").read()
    vals = struct.unpack("<%dd" % (len(raw) // 8), raw)
    out, off = {}, 0
    for p in orders:
        n = npts(p)
        cols = [vals[off + c * n: off + (c + 1) * n] for c in range(ncoord + 1)]
        off += (ncoord + 1) * n
        out[p] = list(zip(*cols))
    assert off == len(vals), (off, len(vals))
    return out


tri = read(os.path.join(REF, "data", "tri_inter.bin"), 2, range(0, 8), lambda p: (p + 1) * (p + 2) // 2)
tet = read(os.path.join(REF, "data", "tet_inter.bin"), 3, range(0, 7), lambda p: (p + 1) * (p + 2) * (p + 3) // 6)
with open(OUT, "w") as f:
    f.write("# simplex point tables of the reference's rule 0 (\"internal\"), see tools/export_point_tables.py\n")
    for name, tab in (("tri_inter", tri), ("tet_inter", tet)):
        for p, rows in tab.items():
            f.write("%s %d %d\n" % (name, p, len(rows)))
            for r in rows:
                f.write(" ".join("%.17g" % v for v in r) + "\n")
print(OUT, os.path.getsize(OUT))
