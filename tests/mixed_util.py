"""Mixed-mesh fixtures (several element classes in one mesh): TEST INFRASTRUCTURE.

A fixture of oracle/ref_harness on a mixed mesh carries `classes` (the reference's ele_type numbers: 2 tet, 3 prism),
every class's arrays under the prefix "c<ele_type>_", and per face block the class of the left / right side of every
face (`int<t>_cl`, `int<t>_cr`, `bdy<t>_cl`).  The reference keeps ONE block per face type and wires raw pointers per
(ctype(ic_l), ctype(ic_r)) (/root/reference/src/geometry.cpp:637-706, src/int_inters.cpp:67-121); a library that takes
offsets needs one block per (left class, right class) pair -- the split below, which keeps the faces' order."""
import ctypes as C

import numpy as np

COMMON_PREFIXES = ("c2_", "c3_", "c0_", "c1_", "c4_", "int", "bdy")


def split(d):
    """-> (classes, {cls: per-class dict}, [(cl, cr, L, R)], [(cl, L, ids)])"""
    classes = [int(c) for c in d["classes"]]
    common = {k: v for k, v in d.items() if not k.startswith(COMMON_PREFIXES) and k != "classes"}
    per = {}
    for c in classes:
        p = "c%d_" % c
        per[c] = dict(common)
        per[c].update({k[len(p):]: v for k, v in d.items() if k.startswith(p)})
    faces = []
    for t in range(3):
        if "int%d_L" % t not in d:
            continue
        L, R, cl, cr = d["int%d_L" % t], d["int%d_R" % t], d["int%d_cl" % t], d["int%d_cr" % t]
        for a in classes:
            for b in classes:
                m = (cl == a) & (cr == b)
                if m.any():
                    faces.append((a, b, np.asfortranarray(L[:, m].astype(np.int32)), np.asfortranarray(R[:, m].astype(np.int32))))
    bdy = []
    for t in range(3):
        if "bdy%d_L" % t not in d:
            continue
        L, ids, cl = d["bdy%d_L" % t], np.ravel(d["bdy%d_id" % t]), d["bdy%d_cl" % t]
        for a in classes:
            m = cl == a
            if m.any():
                bdy.append((a, np.asfortranarray(L[:, m].astype(np.int32)), np.ascontiguousarray(ids[m].astype(np.int32))))
    return classes, per, faces, bdy


class MixedOracle:
    """The oracle on a mixed mesh: one oracle_py.Case per class, CalcResidual in the reference's order
    (src/solver.cpp:50-223: every method for all classes before the next one)."""

    def __init__(self, d):
        import oracle_py as O
        self.O, self.o = O, O.load()
        self.classes, per, faces, bdy = split(d)
        self.case = {c: O.Case(per[c]) for c in self.classes}
        self.e = {c: self.case[c].c_eles() for c in self.classes}
        self.params = self.case[self.classes[0]].params
        self.faces = []
        for a, b, L, R in faces:
            f = O.IntInters()
            f.n_fpts_per_inter, f.n_inters = L.shape
            f.L, f.R = O.iptr(L), O.iptr(R)
            self.faces.append((a, b, f, (L, R)))
        self.bdy = []
        if bdy:
            self.bcs = O.bc_records(d)
            for a, L, ids in bdy:
                f = O.BdyInters()
                f.n_fpts_per_inter, f.n_inters = L.shape
                f.L, f.boundary_id, f.bcs, f.n_bcs = O.iptr(L), ids.ctypes.data_as(O.ip), self.bcs, len(self.bcs)
                f.R_ref, f.ramp_counter = float(np.ravel(d["bc_R_ref"])[0]), int(np.ravel(d["ramp_counter"])[0])
                self.bdy.append((a, f, (L, ids)))

    def arr(self, c, name):
        return self.case[c].arr[name]

    def CalcResidual(self, hook=None):
        o, P = self.o, C.byref(self.params)
        E = {c: C.byref(self.e[c]) for c in self.classes}
        visc = self.params.viscous
        hook = hook or (lambda name: None)
        for c in self.classes: o.orc_extrapolate_solution(E[c])
        hook("disu_fpts")
        if visc:
            for c in self.classes: o.orc_calculate_gradient(E[c])
            hook("grad_disu_upts_ref")
        for c in self.classes: o.orc_evaluate_invFlux(E[c], P)
        hook("tdisf_upts_inv")
        for a, b, f, _ in self.faces: o.orc_int_calculate_common_invFlux_lr(C.byref(f), E[a], E[b], P)
        for a, f, _ in self.bdy: o.orc_bdy_evaluate_boundaryConditions_invFlux(C.byref(f), E[a], P)
        hook("norm_tconf_fpts_inv")
        if visc:
            for c in self.classes: o.orc_correct_gradient(E[c])
            hook("grad_disu_fpts")
            for c in self.classes: o.orc_evaluate_viscFlux(E[c], P)
            hook("tdisf_upts")
        for c in self.classes: o.orc_extrapolate_totalFlux(E[c])
        hook("norm_tdisf_fpts")
        for c in self.classes: o.orc_calculate_divergence(E[c])
        hook("div_tconf_upts_disc")
        if visc:
            for a, b, f, _ in self.faces: o.orc_int_calculate_common_viscFlux_lr(C.byref(f), E[a], E[b], P)
            for a, f, _ in self.bdy: o.orc_bdy_evaluate_boundaryConditions_viscFlux(C.byref(f), E[a], P)
            hook("norm_tconf_fpts")
        bad = -1
        for c in self.classes:
            bad = max(bad, o.orc_calculate_corrected_divergence(E[c]))
        hook("div_tconf_upts")
        return bad

    def AdvanceSolution(self, rk):
        for c in self.classes:
            self.o.orc_AdvanceSolution(C.byref(self.e[c]), C.byref(self.params), rk)
