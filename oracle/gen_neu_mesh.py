#!/usr/bin/env python3
"""Periodic box mesh writer in Gambit neutral (.neu) format -- TEST INFRASTRUCTURE.

Writes the input the genuine reference solver (oracle/_ref) reads, following the
format parsed by /root/reference/src/mesh_reader.cpp:105-393 (6-line header,
counts line, NODAL COORDINATES, ELEMENTS/CELLS, BOUNDARY CONDITIONS).

Conventions (re-derived from mesh_reader.cpp:241 and
eles_hexas.cpp:1198-1214): an 8-node brick record lists file nodes f0..f7 which
the reader stores in shape slots 0,2,4,6,1,3,5,7; shape slot s = r + 2 s + 4 t.
We choose (r,s,t) = (x,y,z), so file order is the slots [0,2,4,6,1,3,5,7].
Quads (mesh_reader.cpp:209): file order f0 f1 f2 f3 -> slots 0,1,3,2.

Elements are numbered x-fastest: e = ix + nx*(iy + ny*iz); vertices
v = ix + (nx+1)*(iy + (ny+1)*iz) (1-based in the file).
Optional smooth periodic deformation of the vertices (amp != 0) makes the
Jacobian non-constant and the face normals generic.
"""
import argparse
import math
import numpy as np


def box_vertices(n, dims, length=2.0 * math.pi, amp=0.0):
    """Vertex coordinates of the (possibly deformed) periodic box, shape (nv, dims)."""
    n = list(n)
    ax = [np.linspace(0.0, length, n[d] + 1) for d in range(dims)]
    # exact end point so that opposite faces differ by exactly `length`
    for a in ax:
        a[-1] = length
    if dims == 3:
        Z, Y, X = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        x, y, z = X.ravel().copy(), Y.ravel().copy(), Z.ravel().copy()
        if amp != 0.0:
            k = 2.0 * math.pi / length
            dx = amp * np.sin(k * y + 0.3) * np.cos(k * z + 0.5)
            dy = amp * np.cos(k * x + 0.7) * np.sin(k * z + 0.2)
            dz = amp * np.sin(k * x + 0.1) * np.sin(k * y + 0.9)
            # make the deformation bit-identical on periodic images
            def wrap(v):
                return np.where(np.isclose(v, length), 0.0, v)
            xw, yw, zw = wrap(x), wrap(y), wrap(z)
            dx = amp * np.sin(k * yw + 0.3) * np.cos(k * zw + 0.5)
            dy = amp * np.cos(k * xw + 0.7) * np.sin(k * zw + 0.2)
            dz = amp * np.sin(k * xw + 0.1) * np.sin(k * yw + 0.9)
            x, y, z = x + dx, y + dy, z + dz
        return np.stack([x, y, z], axis=1)
    else:
        Y, X = np.meshgrid(ax[1], ax[0], indexing="ij")
        x, y = X.ravel().copy(), Y.ravel().copy()
        if amp != 0.0:
            k = 2.0 * math.pi / length
            def wrap(v):
                return np.where(np.isclose(v, length), 0.0, v)
            xw, yw = wrap(x), wrap(y)
            dx = amp * np.sin(k * yw + 0.3)
            dy = amp * np.cos(k * xw + 0.7)
            x, y = x + dx, y + dy
        return np.stack([x, y], axis=1)


SIDES3 = ("z-", "y-", "x+", "y+", "x-", "z+")  # code faces 0..5 of a hex
SIDES2 = ("y-", "x+", "y+", "x-")              # code faces 0..3 of a quad


def write_neu(path, n, dims=3, length=2.0 * math.pi, amp=0.0, bcname="Cyclic", bcs=None):
    """bcs: optional {side: boundary-group name} (sides "x-","x+","y-","y+","z-","z+"); sides that are not
    listed belong to the group `bcname`.  One BOUNDARY CONDITIONS section is written per group
    (mesh_reader.cpp:310-372 reads n_bdy of them); the group's type comes from the input key bc_<name>_type."""
    if isinstance(n, int):
        n = [n] * dims
    xv = box_vertices(n, dims, length, amp)
    nv = xv.shape[0]
    nx, ny = n[0], n[1]
    nz = n[2] if dims == 3 else 1
    ne = nx * ny * nz

    def vid(i, j, k=0):
        return 1 + i + (nx + 1) * (j + (ny + 1) * k)

    with open(path, "w") as f:
        f.write("        CONTROL INFO 2.3.16\n** GAMBIT NEUTRAL FILE\nperiodic_box\n")
        f.write("PROGRAM:                Gambit     VERSION:  2.3.16\n\n")
        f.write("     NUMNP     NELEM     NGRPS    NBSETS     NDFCD     NDFVL\n")
        groups = []
        for side in (SIDES3 if dims == 3 else SIDES2):
            g = (bcs or {}).get(side, bcname)
            if g not in groups:
                groups.append(g)
        f.write("%10d%10d%10d%10d%10d%10d\n" % (nv, ne, 1, len(groups), dims, dims))
        f.write("ENDOFSECTION\n   NODAL COORDINATES 2.3.16\n")
        for i in range(nv):
            f.write("%10d" % (i + 1) + "".join(" %.17e" % c for c in xv[i]) + "\n")
        f.write("ENDOFSECTION\n      ELEMENTS/CELLS 2.3.16\n")
        bfaces = []
        e = 0
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    e += 1
                    if dims == 3:
                        slot = {}
                        for t in range(2):
                            for s in range(2):
                                for r in range(2):
                                    slot[r + 2 * s + 4 * t] = vid(i + r, j + s, k + t)
                        order = [0, 2, 4, 6, 1, 3, 5, 7]
                        nodes = [slot[o] for o in order]
                        f.write("%8d %2d %2d " % (e, 4, 8) + "".join("%8d" % v for v in nodes[:7]) + "\n")
                        f.write(" " * 15 + "%8d\n" % nodes[7])
                        # gambit face ids (mesh_reader.cpp:336-350): code face 0<-1, 3<-2, 5<-3, 1<-4, 4<-5, 2<-6
                        if k == 0: bfaces.append((e, 4, 1, "z-"))       # z-min : code face 0
                        if j == 0: bfaces.append((e, 4, 4, "y-"))       # y-min : code face 1
                        if i == nx - 1: bfaces.append((e, 4, 6, "x+"))  # x-max : code face 2
                        if j == ny - 1: bfaces.append((e, 4, 2, "y+"))  # y-max : code face 3
                        if i == 0: bfaces.append((e, 4, 5, "x-"))       # x-min : code face 4
                        if k == nz - 1: bfaces.append((e, 4, 3, "z+"))  # z-max : code face 5
                    else:
                        # quad slots 0:(0,0) 1:(1,0) 2:(0,1) 3:(1,1); file order -> slots 0,1,3,2
                        slot = {r + 2 * s: vid(i + r, j + s) for s in range(2) for r in range(2)}
                        nodes = [slot[0], slot[1], slot[3], slot[2]]
                        f.write("%8d %2d %2d " % (e, 2, 4) + "".join("%8d" % v for v in nodes) + "\n")
                        # quad faces (eles_quads.cpp:209-248): 0: eta=-1, 1: xi=+1, 2: eta=+1, 3: xi=-1; gambit k = face+1
                        if j == 0: bfaces.append((e, 2, 1, "y-"))
                        if i == nx - 1: bfaces.append((e, 2, 2, "x+"))
                        if j == ny - 1: bfaces.append((e, 2, 3, "y+"))
                        if i == 0: bfaces.append((e, 2, 4, "x-"))
        f.write("ENDOFSECTION\n       ELEMENT GROUP 2.3.16\n")
        f.write("GROUP: %10d ELEMENTS: %10d MATERIAL: %10d NFLAGS: %10d\n" % (1, ne, 2, 1))
        f.write("                           fluid\n       0\n")
        ids = list(range(1, ne + 1))
        for s in range(0, ne, 10):
            f.write("".join("%8d" % v for v in ids[s:s + 10]) + "\n")
        f.write("ENDOFSECTION\n")
        for g in groups:
            mine = [b for b in bfaces if (bcs or {}).get(b[3], bcname) == g]
            f.write(" BOUNDARY CONDITIONS 2.3.16\n")
            f.write("%32s%8d%8d%8d%8d\n" % (g, 1, len(mine), 0, 6))
            for (el, ty, fc, _) in mine:
                f.write("%10d%5d%5d\n" % (el, ty, fc))
            f.write("ENDOFSECTION\n")
    return xv


def add_edge_nodes(xv, lat, n, cells, edges, length, curve):
    """Mid-edge nodes of quadratic elements: one node per edge (keyed by its two vertices), at the edge's midpoint moved by
    curve * (cell size) * a smooth vector field that is PERIODIC in the lattice coordinates, so that the periodic images of a
    curved face coincide under the box translation.  -> (all node coordinates, per cell the list of its edge-node ids)."""
    h = length / max(n)
    nodes = [x for x in xv]
    ids = {}
    per_cell = []
    for v in cells:
        mine = []
        for a, b in edges:
            key = (min(v[a], v[b]), max(v[a], v[b]))
            if key not in ids:
                m = 0.5 * (lat[v[a]] + lat[v[b]])
                ph = [2.0 * math.pi * m[d] / n[d] for d in range(3)]
                bump = np.array([math.sin(ph[1] + 1.0) * math.cos(ph[2]), math.sin(ph[2] + 2.0) * math.cos(ph[0]),
                                 math.sin(ph[0] + 3.0) * math.cos(ph[1])])
                ids[key] = len(nodes)
                nodes.append(0.5 * (xv[v[a]] + xv[v[b]]) + curve * h * bump)
            mine.append(ids[key])
        per_cell.append(mine)
    return np.array(nodes), per_cell


def write_neu_tets(path, n, length=2.0 * math.pi, amp=0.0, bcname="Cyclic", curve=None):
    """Periodic box of tetrahedra: every cube of the n^3 grid is cut into the 6 Kuhn tetrahedra (one per ordering
    of the axes, all sharing the cube's main diagonal), which makes the triangulation conforming across cubes and
    across the periodic images.  Gambit tets are element type 6 with 4 nodes; the reference's local faces are
    f0 (1,2,3), f1 (0,3,2), f2 (0,1,3), f3 (0,2,1) (src/mesh.cpp get_corner_vlist_face) and its reader maps the
    Gambit face number k to them as 1->3, 2->2, 3->0, 4->1 (src/mesh_reader.cpp:353-363)."""
    import itertools
    if isinstance(n, int):
        n = [n] * 3
    xv = box_vertices(n, 3, length, amp)
    nx, ny, nz = n
    nv = xv.shape[0]

    def vid(i, j, k):
        return i + (nx + 1) * (j + (ny + 1) * k)

    tets = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for perm in itertools.permutations(range(3)):
                    p = [i, j, k]
                    v = [vid(*p)]
                    for a in perm:
                        p[a] += 1
                        v.append(vid(*p))
                    # positive volume with the reference tet (-1,-1,-1),(1,-1,-1),(-1,1,-1),(-1,-1,1)
                    x = xv[v]
                    if np.linalg.det(np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]])) < 0:
                        v[1], v[2] = v[2], v[1]
                    tets.append(v)
    faces_loc = [(1, 2, 3), (0, 3, 2), (0, 1, 3), (0, 2, 1)]
    to_k = {0: 3, 1: 4, 2: 2, 3: 1}
    # lattice coordinates of every vertex to spot faces on the box boundary
    lat = np.zeros((nv, 3), dtype=int)
    for k in range(nz + 1):
        for j in range(ny + 1):
            for i in range(nx + 1):
                lat[vid(i, j, k)] = (i, j, k)
    bfaces = []
    for e, v in enumerate(tets):
        for f, fl in enumerate(faces_loc):
            c = lat[[v[q] for q in fl]]
            for d in range(3):
                if (c[:, d] == 0).all() or (c[:, d] == n[d]).all():
                    bfaces.append((e + 1, 6, to_k[f]))
    ne = len(tets)
    cells = tets
    if curve is not None:
        # quadratic tetrahedra: 10 nodes in Gambit's order v0 e01 v1 e02 e12 v2 e03 e13 e23 v3 (src/mesh_reader.cpp:219-223 maps
        # them to the shape nodes 0 4 1 5 7 2 6 9 8 3)
        xv, en = add_edge_nodes(xv, lat, n, tets, [(0, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 3)], length, curve)
        nv = xv.shape[0]
        cells = [[v[0], m[0], v[1], m[1], m[2], v[2], m[3], m[4], m[5], v[3]] for v, m in zip(tets, en)]
    with open(path, "w") as f:
        f.write("        CONTROL INFO 2.3.16\n** GAMBIT NEUTRAL FILE\nperiodic_box_tets\n")
        f.write("PROGRAM:                Gambit     VERSION:  2.3.16\n\n")
        f.write("     NUMNP     NELEM     NGRPS    NBSETS     NDFCD     NDFVL\n")
        f.write("%10d%10d%10d%10d%10d%10d\n" % (nv, ne, 1, 1, 3, 3))
        f.write("ENDOFSECTION\n   NODAL COORDINATES 2.3.16\n")
        for i in range(nv):
            f.write("%10d" % (i + 1) + "".join(" %.17e" % c for c in xv[i]) + "\n")
        f.write("ENDOFSECTION\n      ELEMENTS/CELLS 2.3.16\n")
        for e, v in enumerate(cells):
            f.write("%8d %2d %2d " % (e + 1, 6, len(v)) + "".join("%8d" % (q + 1) for q in v) + "\n")
        f.write("ENDOFSECTION\n       ELEMENT GROUP 2.3.16\n")
        f.write("GROUP: %10d ELEMENTS: %10d MATERIAL: %10d NFLAGS: %10d\n" % (1, ne, 2, 1))
        f.write("                           fluid\n       0\n")
        ids = list(range(1, ne + 1))
        for s0 in range(0, ne, 10):
            f.write("".join("%8d" % q for q in ids[s0:s0 + 10]) + "\n")
        f.write("ENDOFSECTION\n BOUNDARY CONDITIONS 2.3.16\n")
        f.write("%32s%8d%8d%8d%8d\n" % (bcname, 1, len(bfaces), 0, 6))
        for (el, ty, fc) in bfaces:
            f.write("%10d%5d%5d\n" % (el, ty, fc))
        f.write("ENDOFSECTION\n")
    return xv


def write_neu_prisms(path, n, length=2.0 * math.pi, amp=0.0, bcname="Cyclic", curve=None):
    """Periodic box of triangular prisms: every cube is cut along the (x,y) diagonal into two prisms extruded in z.
    Gambit prisms are element type 5 with 6 nodes (0,1,2 bottom triangle, 3,4,5 above them); local faces f0 (0,2,1),
    f1 (3,4,5), f2 (0,1,4,3), f3 (1,2,5,4), f4 (2,0,3,5) (src/mesh.cpp get_corner_vlist_face); Gambit face k maps to
    them as 1->2, 2->3, 3->4, 4->0, 5->1 (src/mesh_reader.cpp:364-375)."""
    if isinstance(n, int):
        n = [n] * 3
    xv = box_vertices(n, 3, length, amp)
    nx, ny, nz = n
    nv = xv.shape[0]

    def vid(i, j, k):
        return i + (nx + 1) * (j + (ny + 1) * k)

    pris = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for tri in (((0, 0), (1, 0), (1, 1)), ((0, 0), (1, 1), (0, 1))):  # counter-clockwise seen from +z
                    v = [vid(i + a, j + b, k) for a, b in tri] + [vid(i + a, j + b, k + 1) for a, b in tri]
                    pris.append(v)
    faces_loc = [(0, 2, 1), (3, 4, 5), (0, 1, 4, 3), (1, 2, 5, 4), (2, 0, 3, 5)]
    to_k = {2: 1, 3: 2, 4: 3, 0: 4, 1: 5}
    lat = np.zeros((nv, 3), dtype=int)
    for k in range(nz + 1):
        for j in range(ny + 1):
            for i in range(nx + 1):
                lat[vid(i, j, k)] = (i, j, k)
    bfaces = []
    for e, v in enumerate(pris):
        for f, fl in enumerate(faces_loc):
            c = lat[[v[q] for q in fl]]
            for d in range(3):
                if (c[:, d] == 0).all() or (c[:, d] == n[d]).all():
                    bfaces.append((e + 1, 5, to_k[f]))
    ne = len(pris)
    cells = pris
    if curve is not None:
        # quadratic prisms: 15 nodes in Gambit's order, which src/mesh_reader.cpp:233 maps to the shape nodes
        # 0 6 1 8 7 2 | 9 10 11 | 3 12 4 14 13 5 (6-8 / 12-14: edges (0,1) (1,2) (0,2) below / above, 9-11: the vertical edges)
        xv, en = add_edge_nodes(xv, lat, n, pris, [(0, 1), (1, 2), (0, 2), (0, 3), (1, 4), (2, 5), (3, 4), (4, 5), (3, 5)], length, curve)
        nv = xv.shape[0]
        cells = [[v[0], m[0], v[1], m[2], m[1], v[2], m[3], m[4], m[5], v[3], m[6], v[4], m[8], m[7], v[5]] for v, m in zip(pris, en)]
    with open(path, "w") as f:
        f.write("        CONTROL INFO 2.3.16\n** GAMBIT NEUTRAL FILE\nperiodic_box_prisms\n")
        f.write("PROGRAM:                Gambit     VERSION:  2.3.16\n\n")
        f.write("     NUMNP     NELEM     NGRPS    NBSETS     NDFCD     NDFVL\n")
        f.write("%10d%10d%10d%10d%10d%10d\n" % (nv, ne, 1, 1, 3, 3))
        f.write("ENDOFSECTION\n   NODAL COORDINATES 2.3.16\n")
        for i in range(nv):
            f.write("%10d" % (i + 1) + "".join(" %.17e" % c for c in xv[i]) + "\n")
        f.write("ENDOFSECTION\n      ELEMENTS/CELLS 2.3.16\n")
        for e, v in enumerate(cells):
            f.write("%8d %2d %2d " % (e + 1, 5, len(v)) + "".join("%8d" % (q + 1) for q in v) + "\n")
        f.write("ENDOFSECTION\n       ELEMENT GROUP 2.3.16\n")
        f.write("GROUP: %10d ELEMENTS: %10d MATERIAL: %10d NFLAGS: %10d\n" % (1, ne, 2, 1))
        f.write("                           fluid\n       0\n")
        ids = list(range(1, ne + 1))
        for s0 in range(0, ne, 10):
            f.write("".join("%8d" % q for q in ids[s0:s0 + 10]) + "\n")
        f.write("ENDOFSECTION\n BOUNDARY CONDITIONS 2.3.16\n")
        f.write("%32s%8d%8d%8d%8d\n" % (bcname, 1, len(bfaces), 0, 6))
        for (el, ty, fc) in bfaces:
            f.write("%10d%5d%5d\n" % (el, ty, fc))
        f.write("ENDOFSECTION\n")
    return xv


def write_neu_mixed(path, n, length=2.0 * math.pi, amp=0.0, bcname="Cyclic", wall_lo="WallLo", wall_hi="WallHi"):
    """A channel of MIXED elements (BASELINE.json configs[3]): periodic in x and z, walls at y = 0 and y = length.  The
    cell layers on the two walls (j = 0 and j = ny-1) are triangular prisms extruded in y -- their triangular faces lie
    on the wall and on the interface to the core -- and the core layers are tetrahedra (6 Kuhn tetrahedra per cell).
    Both cut the y = const cell faces along the same diagonal (i,k) -> (i+1,k+1), so the prism / tetrahedron interface
    is conforming (triangle against triangle), as are the periodic images.  Needs ny >= 3.
    Element records: Gambit type 5 (prism, 6 nodes) and 6 (tetrahedron, 4 nodes), see write_neu_prisms / write_neu_tets
    for the node and face conventions.  Three boundary groups: `bcname` (the x and z sides), `wall_lo`, `wall_hi`."""
    import itertools
    if isinstance(n, int):
        n = [n] * 3
    nx, ny, nz = n
    assert ny >= 3, "need a core layer between the two prism layers"
    xv = box_vertices(n, 3, length, amp)
    nv = xv.shape[0]

    def vid(i, j, k):
        return i + (nx + 1) * (j + (ny + 1) * k)

    cells = []  # (gambit type, nodes)
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                if j == 0 or j == ny - 1:
                    # triangles in the (z, x) plane, counter-clockwise seen from +y, extruded in +y: positive volume
                    for tri in (((0, 0), (1, 0), (1, 1)), ((0, 0), (1, 1), (0, 1))):  # (dz, dx)
                        v = [vid(i + b, j, k + a) for a, b in tri] + [vid(i + b, j + 1, k + a) for a, b in tri]
                        cells.append((5, v))
                else:
                    for perm in itertools.permutations(range(3)):
                        p = [i, j, k]
                        v = [vid(*p)]
                        for a in perm:
                            p[a] += 1
                            v.append(vid(*p))
                        x = xv[v]
                        if np.linalg.det(np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]])) < 0:
                            v[1], v[2] = v[2], v[1]
                        cells.append((6, v))
    faces_loc = {6: [(1, 2, 3), (0, 3, 2), (0, 1, 3), (0, 2, 1)],
                 5: [(0, 2, 1), (3, 4, 5), (0, 1, 4, 3), (1, 2, 5, 4), (2, 0, 3, 5)]}
    to_k = {6: {0: 3, 1: 4, 2: 2, 3: 1}, 5: {2: 1, 3: 2, 4: 3, 0: 4, 1: 5}}
    lat = np.zeros((nv, 3), dtype=int)
    for k in range(nz + 1):
        for j in range(ny + 1):
            for i in range(nx + 1):
                lat[vid(i, j, k)] = (i, j, k)
    groups = {bcname: [], wall_lo: [], wall_hi: []}
    for e, (ty, v) in enumerate(cells):
        for f, fl in enumerate(faces_loc[ty]):
            c = lat[[v[q] for q in fl]]
            for d in range(3):
                lo, hi = (c[:, d] == 0).all(), (c[:, d] == n[d]).all()
                if lo or hi:
                    g = bcname if d != 1 else (wall_lo if lo else wall_hi)
                    groups[g].append((e + 1, ty, to_k[ty][f]))
    ne = len(cells)
    with open(path, "w") as f:
        f.write("        CONTROL INFO 2.3.16\n** GAMBIT NEUTRAL FILE\nmixed_channel\n")
        f.write("PROGRAM:                Gambit     VERSION:  2.3.16\n\n")
        f.write("     NUMNP     NELEM     NGRPS    NBSETS     NDFCD     NDFVL\n")
        f.write("%10d%10d%10d%10d%10d%10d\n" % (nv, ne, 1, len(groups), 3, 3))
        f.write("ENDOFSECTION\n   NODAL COORDINATES 2.3.16\n")
        for i in range(nv):
            f.write("%10d" % (i + 1) + "".join(" %.17e" % c for c in xv[i]) + "\n")
        f.write("ENDOFSECTION\n      ELEMENTS/CELLS 2.3.16\n")
        for e, (ty, v) in enumerate(cells):
            f.write("%8d %2d %2d " % (e + 1, ty, len(v)) + "".join("%8d" % (q + 1) for q in v) + "\n")
        f.write("ENDOFSECTION\n       ELEMENT GROUP 2.3.16\n")
        f.write("GROUP: %10d ELEMENTS: %10d MATERIAL: %10d NFLAGS: %10d\n" % (1, ne, 2, 1))
        f.write("                           fluid\n       0\n")
        ids = list(range(1, ne + 1))
        for s0 in range(0, ne, 10):
            f.write("".join("%8d" % q for q in ids[s0:s0 + 10]) + "\n")
        f.write("ENDOFSECTION\n")
        for g, lst in groups.items():
            f.write(" BOUNDARY CONDITIONS 2.3.16\n")
            f.write("%32s%8d%8d%8d%8d\n" % (g, 1, len(lst), 0, 6))
            for (el, ty, fc) in lst:
                f.write("%10d%5d%5d\n" % (el, ty, fc))
            f.write("ENDOFSECTION\n")
    return xv


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("-n", type=int, default=3)
    ap.add_argument("--dims", type=int, default=3)
    ap.add_argument("--amp", type=float, default=0.0)
    a = ap.parse_args()
    write_neu(a.out, a.n, a.dims, amp=a.amp)
