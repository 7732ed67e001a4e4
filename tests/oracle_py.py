"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("gamma", "prandtl", "rt_inf", "mu_inf", "c_sth", "fix_vis", "ldg_beta", "ldg_tau", "dt")] + \
               [(n, C.c_int) for n in
                ("viscous", "riemann_solve_type", "vis_riemann_solve_type", "adv_type", "dt_type", "n_rk")] + \
               [("RK_a", C.c_double * 16), ("RK_b", C.c_double * 16)]


class Eles(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_eles", "n_upts", "n_fpts", "n_fields", "n_dims")] + [
        ("opp_0", dp), ("opp_1", dp * 3), ("opp_2", dp * 3), ("opp_3", dp), ("opp_4", dp * 3),
        ("opp_5", dp * 3), ("opp_6", dp),
        ("detjac_upts", dp), ("JGinv_upts", dp), ("detjac_fpts", dp), ("JGinv_fpts", dp),
        ("tdA_fpts", dp), ("norm_fpts", dp),
        ("disu_upts", dp * 2), ("disu_fpts", dp), ("tdisf_upts", dp), ("norm_tdisf_fpts", dp),
        ("norm_tconf_fpts", dp), ("div_tconf_upts", dp), ("delta_disu_fpts", dp),
        ("grad_disu_upts", dp), ("grad_disu_fpts", dp), ("src_upts", dp), ("dt_local", dp),
        ("sgs_model", C.c_int), ("C_s", C.c_double), ("filter_ratio", C.c_double), ("Kappa", C.c_double),
        ("prandtl_t", C.c_double), ("order_les", C.c_int), ("les_vol_factor", C.c_double), ("wall_distance", dp), ("sgsf_upts", dp), ("sgsf_fpts", dp), ("Jacobian_fpts", dp),
        ("n_cub", C.c_int), ("opp_over_int_cubpts", dp), ("over_int_filter", dp), ("JGinv_over_int_cubpts", dp),
        ("filter_upts", dp), ("disuf_upts", dp), ("uu", dp), ("Lu", dp), ("ue", dp), ("Le", dp)]


class IntInters(C.Structure):
    _fields_ = [("n_inters", C.c_int), ("n_fpts_per_inter", C.c_int), ("L", ip), ("R", ip)]


class Shock(C.Structure):
    _fields_ = [("inv_vandermonde", dp), ("exp_filter", dp), ("norm_basis_persson", dp), ("high_modes", ip),
                ("s0", C.c_double), ("shock_det_field", C.c_int), ("sensor", dp)]


class Bc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("flag", "pressure_ramp", "use_wm", "pad")] + \
               [("rho", C.c_double), ("velocity", C.c_double * 3)] + \
               [(n, C.c_double) for n in ("p_static", "T_static", "p_total", "T_total", "nx", "ny", "nz",
                                          "p_ramp_coeff", "T_ramp_coeff", "p_total_old", "T_total_old")]


class BdyInters(C.Structure):
    _fields_ = [("n_inters", C.c_int), ("n_fpts_per_inter", C.c_int), ("L", ip), ("boundary_id", ip),
                ("bcs", C.POINTER(Bc)), ("n_bcs", C.c_int), ("R_ref", C.c_double), ("ramp_counter", C.c_int)]


def bc_records(data):
    """bc_list of a fixture (`bc_flags` (3,nbc), `bc_params` (15,nbc), written by oracle/ref_harness.cpp) as Bc records."""
    fl = np.asarray(data["bc_flags"]).reshape(3, -1, order="F")
    par = np.asarray(data["bc_params"]).reshape(15, -1, order="F")
    out = (Bc * fl.shape[1])()
    for b in range(fl.shape[1]):
        r = out[b]
        r.flag, r.pressure_ramp, r.use_wm = int(fl[0, b]), int(fl[1, b]), int(fl[2, b])
        q = par[:, b]
        r.rho = q[0]
        for d in range(3):
            r.velocity[d] = q[1 + d]
        (r.p_static, r.T_static, r.p_total, r.T_total, r.nx, r.ny, r.nz,
         r.p_ramp_coeff, r.T_ramp_coeff, r.p_total_old, r.T_total_old) = [float(v) for v in q[4:15]]
    return out


_lib = None


def load():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        _lib = C.CDLL(so)
        _lib.orc_CalcResidual.restype = C.c_long
        _lib.orc_rk_step.restype = C.c_long
        _lib.orc_CalcResidual_bdy.restype = C.c_long
        _lib.orc_rk_step_bdy.restype = C.c_long
        _lib.orc_calculate_corrected_divergence.restype = C.c_long
        _lib.orc_calc_sgs_terms.restype = C.c_long
        _lib.orc_compute_res_upts.restype = C.c_double
        _lib.orc_calc_dt_local.restype = C.c_double
        _lib.orc_calc_dt_local.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
    return _lib


def fptr(a):
    assert a.dtype == np.float64 and a.flags.f_contiguous
    return a.ctypes.data_as(dp)


def iptr(a):
    assert a.dtype == np.int32 and a.flags.f_contiguous
    return a.ctypes.data_as(ip)


def F(shape):
    return np.zeros(shape, dtype=np.float64, order="F")


class Case:
    """Registration data + state of one single-element-type case, as numpy arrays in hf_array (Fortran) order.

    `data` is a dict with the keys of the golden fixtures (opp_*, metrics, int*_L/R, scalars)."""

    def __init__(self, data, u_init=None):
        g = lambda k: np.asfortranarray(np.array(data[k], dtype=np.float64))
        sz = [int(v) for v in data["sizes"]]
        self.n_eles, self.n_upts, self.n_fpts, self.n_fields, self.n_dims, self.order = sz[:6]
        self.ele_type = sz[6] if len(sz) > 6 else (4 if sz[4] == 3 else 1)
        ne, nu, nfp, nf, nd = self.n_eles, self.n_upts, self.n_fpts, self.n_fields, self.n_dims
        self.viscous = int(np.ravel(data["viscous"])[0])
        self.arr = {}
        for k in ("opp_0", "opp_3", "detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts"):
            self.arr[k] = g(k)
        if self.viscous:
            self.arr["opp_6"] = g("opp_6")
        for d in range(nd):
            names = ["opp_1_%d", "opp_2_%d"] + (["opp_4_%d", "opp_5_%d"] if self.viscous else [])
            for n in names:
                self.arr[n % d] = g(n % d)
        u = g("u_init") if u_init is None else np.asfortranarray(np.array(u_init, dtype=np.float64))
        self.arr["u0"] = u.copy(order="F")
        self.arr["u1"] = F((nu, ne, nf))
        self.arr["disu_fpts"] = F((nfp, ne, nf))
        self.arr["tdisf_upts"] = F((nu, ne, nf, nd))
        self.arr["norm_tdisf_fpts"] = F((nfp, ne, nf))
        self.arr["norm_tconf_fpts"] = F((nfp, ne, nf))
        self.arr["div_tconf_upts"] = F((nu, ne, nf))
        self.arr["delta_disu_fpts"] = F((nfp, ne, nf))
        self.arr["grad_disu_upts"] = F((nu, ne, nf, nd))
        self.arr["grad_disu_fpts"] = F((nfp, ne, nf, nd))
        # face blocks
        self.faces = []
        for t in range(3):
            if "int%d_L" % t in data:
                L = np.asfortranarray(np.array(data["int%d_L" % t], dtype=np.int32))
                R = np.asfortranarray(np.array(data["int%d_R" % t], dtype=np.int32))
                self.faces.append((L, R))
        # LES closure
        self.les = None
        if "LES" in data and int(np.ravel(data["LES"])[0]):
            sc = lambda k: float(np.ravel(data[k])[0])
            self.les = dict(sgs_model=int(sc("SGS_model")), C_s=sc("C_s"), filter_ratio=sc("filter_ratio"), Kappa=sc("Kappa"),
                            prandtl_t=sc("prandtl_t"))
            self.arr["sgsf_upts"] = F((nu, ne, nf, nd))
            self.arr["sgsf_fpts"] = F((nfp, ne, nf, nd))
            if self.les["sgs_model"] == 0:
                self.arr["wall_distance"] = g("wall_distance")
            self.arr["Jacobian_fpts"] = g("Jacobian_fpts")
            if self.les["sgs_model"] >= 2:  # similarity-type closures: filter, filtered solution, Leonard terms
                self.arr["filter_upts"] = g("filter_upts")
                self.arr["disuf_upts"] = F((nu, ne, nf))
                for k, n3 in (("uu", 3 if nd == 2 else 6), ("Lu", 3 if nd == 2 else 6), ("ue", nd), ("Le", nd)):
                    self.arr[k] = F((nu, ne, n3))
        # over-integration
        self.n_cub = 0
        if "over_int" in data and int(np.ravel(data["over_int"])[0]):
            for k in ("opp_over_int_cubpts", "over_int_filter", "JGinv_over_int_cubpts"):
                self.arr[k] = g(k)
            self.n_cub = self.arr["opp_over_int_cubpts"].shape[0]
        # shock capturing
        self.shock_cap = int(np.ravel(data["shock_cap"])[0]) if "shock_cap" in data else 0
        if self.shock_cap:
            self.arr["inv_vandermonde"] = g("inv_vandermonde")
            self.arr["exp_filter"] = g("exp_filter")
            self.arr["norm_basis_persson"] = np.ascontiguousarray(np.ravel(data["norm_basis_persson"]).astype(np.float64))
            self.hi = np.ascontiguousarray(np.ravel(data["persson_high_modes"]).astype(np.int32))
            self.arr["sensor"] = np.zeros(ne)
            self.s0 = float(np.ravel(data["s0"])[0])
            self.shock_det_field = int(np.ravel(data["shock_det_field"])[0])
        # boundary-face blocks
        self.bdy = []
        for t in range(3):
            if "bdy%d_L" % t in data:
                L = np.asfortranarray(np.array(data["bdy%d_L" % t], dtype=np.int32))
                ids = np.ascontiguousarray(np.array(data["bdy%d_id" % t], dtype=np.int32).ravel())
                self.bdy.append((L, ids))
        if self.bdy:
            self.bcs = bc_records(data)
            self.bc_R_ref = float(np.ravel(data["bc_R_ref"])[0])
            self.ramp_counter = int(np.ravel(data["ramp_counter"])[0])
        # params
        s = lambda k, dflt=None: float(np.ravel(data[k])[0]) if k in data else dflt
        p = Params()
        for k in ("gamma", "prandtl", "rt_inf", "mu_inf", "c_sth", "fix_vis", "ldg_beta", "ldg_tau", "dt"):
            setattr(p, k, s(k, 0.0))
        for k in ("viscous", "riemann_solve_type", "vis_riemann_solve_type", "adv_type", "dt_type"):
            setattr(p, k, int(s(k, 0)))
        ra, rb = np.ravel(data["RK_a"]), np.ravel(data["RK_b"])
        p.n_rk = len(ra)
        for i in range(len(ra)):
            p.RK_a[i] = float(ra[i])
            p.RK_b[i] = float(rb[i])
        self.params = p

    def c_eles(self):
        a = self.arr
        e = Eles()
        e.n_eles, e.n_upts, e.n_fpts, e.n_fields, e.n_dims = self.n_eles, self.n_upts, self.n_fpts, self.n_fields, self.n_dims
        e.opp_0 = fptr(a["opp_0"]); e.opp_3 = fptr(a["opp_3"])
        if self.viscous:
            e.opp_6 = fptr(a["opp_6"])
        for d in range(self.n_dims):
            e.opp_1[d] = fptr(a["opp_1_%d" % d]); e.opp_2[d] = fptr(a["opp_2_%d" % d])
            if self.viscous:
                e.opp_4[d] = fptr(a["opp_4_%d" % d]); e.opp_5[d] = fptr(a["opp_5_%d" % d])
        for k in ("detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts",
                  "disu_fpts", "tdisf_upts", "norm_tdisf_fpts", "norm_tconf_fpts", "div_tconf_upts",
                  "delta_disu_fpts", "grad_disu_upts", "grad_disu_fpts"):
            setattr(e, k, fptr(a[k]))
        e.disu_upts[0] = fptr(a["u0"]); e.disu_upts[1] = fptr(a["u1"])
        e.sgs_model = -1
        if self.les:
            e.sgs_model = self.les["sgs_model"]
            e.C_s, e.filter_ratio, e.Kappa, e.prandtl_t = (self.les[k] for k in ("C_s", "filter_ratio", "Kappa", "prandtl_t"))
            e.order_les = self.order
            # calc_ele_vol: |J| times the reference element's volume (src/eles_*.cpp)
            e.les_vol_factor = {0: 2.0, 1: 4.0, 2: 8.0 / 6.0, 3: 4.0, 4: 8.0}[self.ele_type]
            e.sgsf_upts, e.sgsf_fpts = fptr(a["sgsf_upts"]), fptr(a["sgsf_fpts"])
            e.Jacobian_fpts = fptr(a["Jacobian_fpts"])
            if e.sgs_model == 0:
                e.wall_distance = fptr(a["wall_distance"])
            if e.sgs_model >= 2:
                for k in ("filter_upts", "disuf_upts", "uu", "Lu", "ue", "Le"):
                    setattr(e, k, fptr(a[k]))
        e.n_cub = self.n_cub
        if self.n_cub:
            for k in ("opp_over_int_cubpts", "over_int_filter", "JGinv_over_int_cubpts"):
                setattr(e, k, fptr(a[k]))
        self._e = e
        return e

    def c_faces(self):
        arr = (IntInters * max(1, len(self.faces)))()
        for i, (L, R) in enumerate(self.faces):
            arr[i].n_fpts_per_inter, arr[i].n_inters = L.shape
            arr[i].L = iptr(L); arr[i].R = iptr(R)
        self._f = arr
        return arr, len(self.faces)


def _c_shock(self):
    sh = Shock()
    sh.inv_vandermonde = fptr(self.arr["inv_vandermonde"])
    sh.exp_filter = fptr(self.arr["exp_filter"])
    sh.norm_basis_persson = self.arr["norm_basis_persson"].ctypes.data_as(dp)
    sh.high_modes = self.hi.ctypes.data_as(ip)
    sh.s0, sh.shock_det_field = self.s0, self.shock_det_field
    sh.sensor = self.arr["sensor"].ctypes.data_as(dp)
    self._sh = sh
    return sh


Case.c_shock = _c_shock


def _c_bdy(self):
    arr = (BdyInters * max(1, len(self.bdy)))()
    for i, (L, ids) in enumerate(self.bdy):
        arr[i].n_fpts_per_inter, arr[i].n_inters = L.shape
        arr[i].L = iptr(L)
        arr[i].boundary_id = ids.ctypes.data_as(ip)
        arr[i].bcs = self.bcs
        arr[i].n_bcs = len(self.bcs)
        arr[i].R_ref = self.bc_R_ref
        arr[i].ramp_counter = self.ramp_counter
    self._b = arr
    return arr, len(self.bdy)


Case.c_bdy = _c_bdy


class MpiInters(C.Structure):
    _fields_ = [("n_inters", C.c_int), ("n_fpts_per_inter", C.c_int), ("L", ip), ("Rlut", ip),
                ("out_disu", dp), ("in_disu", dp), ("out_grad", dp), ("in_grad", dp), ("out_sgsf", dp), ("in_sgsf", dp)]


class PartitionedCase(Case):
    """Case + one partition-face block; CalcResidual in the reference's MPI call order
    (/root/reference/src/solver.cpp:64-211) with the exchange supplied by the caller:
    exchange(kind, phase), kind 0 solution / 1 corrected gradient, phase 0 start / 1 wait."""

    def __init__(self, data, L, Rlut, exchange=None):
        super().__init__(data)
        self.mL = np.asfortranarray(np.array(L, dtype=np.int32))
        self.mR = np.asfortranarray(np.array(Rlut, dtype=np.int32))
        nfpi, ni = self.mL.shape
        nf, nd = self.n_fields, self.n_dims
        # flat buffers: the exchange slices them by face record
        self.buf = {k: np.zeros(nfpi * nf * ni * (1 if "disu" in k else nd)) for k in
                    ("out_disu", "in_disu", "out_grad", "in_grad") + (("out_sgsf", "in_sgsf") if self.les else ())}
        self.exchange = exchange or (lambda kind, phase: None)

    def c_mpi(self):
        m = MpiInters()
        m.n_fpts_per_inter, m.n_inters = self.mL.shape
        m.L, m.Rlut = iptr(self.mL), iptr(self.mR)
        for k, v in self.buf.items():
            setattr(m, k, v.ctypes.data_as(dp))
        self._m = m
        return m

    # kinds of exchange: 0 solution, 1 corrected gradient, 2 SGS flux (LES; src/mpi_inters.cpp:339-397)

    def CalcResidual(self):
        o = load()
        e, (f, nb), m, p = self.c_eles(), self.c_faces(), self.c_mpi(), self.params
        E, M, P = C.byref(e), C.byref(m), C.byref(p)
        have = m.n_inters > 0
        o.orc_extrapolate_solution(E)
        if have:
            o.orc_mpi_pack_solution(M, E)
            self.exchange(0, 0)
        if p.viscous:
            o.orc_calculate_gradient(E)
        o.orc_evaluate_invFlux(E, C.byref(p))
        for b in range(nb):
            o.orc_int_calculate_common_invFlux(C.byref(f[b]), E, P)
        if have:
            self.exchange(0, 1)
            o.orc_mpi_calculate_common_invFlux(M, E, P)
        if p.viscous:
            o.orc_correct_gradient(E)
            if have:
                o.orc_mpi_pack_corrected_gradient(M, E)
                self.exchange(1, 0)
            o.orc_evaluate_viscFlux(E, P)
            if self.les:  # src/solver.cpp:162-178
                o.orc_extrapolate_sgsFlux(E)
                if have:
                    o.orc_mpi_pack_sgsf(M, E)
                    self.exchange(2, 0)
        o.orc_extrapolate_totalFlux(E)
        o.orc_calculate_divergence(E)
        if p.viscous:
            for b in range(nb):
                o.orc_int_calculate_common_viscFlux(C.byref(f[b]), E, P)
            if have:
                self.exchange(1, 1)
                if self.les:  # src/solver.cpp:203-206
                    self.exchange(2, 1)
                o.orc_mpi_calculate_common_viscFlux(M, E, P)
        return o.orc_calculate_corrected_divergence(E)

    def rk_step(self):
        o = load()
        for s in range(self.params.n_rk if self.params.adv_type else 1):
            if s == 0 and self.les and self.les["sgs_model"] >= 2:  # src/solver.cpp:55-62
                assert o.orc_calc_sgs_terms(C.byref(self.c_eles())) < 0
            bad = self.CalcResidual()
            assert bad < 0, "NaN at %d" % bad
            o.orc_AdvanceSolution(C.byref(self._e), C.byref(self.params), C.c_int(s))
