"""ctypes binding of libhfx_host.so (include/hfx_host.h): the host-side mirror of the reference's
eles / int_inters / solver interface.  Setup is host code; every per-stage call ends in libhfx."""
import ctypes as C
import os

import numpy as np

import hfx

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.environ.get("HFX_LIB_DIR", HERE), "libhfx_host.so")  # HFX_LIB_DIR: an A/B build (make variant)

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class CaseDesc(C.Structure):
    _fields_ = [("dims", C.c_int), ("n", C.c_int * 3), ("order", C.c_int), ("length", C.c_double), ("amp", C.c_double),
                ("xv", dp), ("loc_1d_upts", dp),
                ("viscous", C.c_int), ("riemann_solve_type", C.c_int), ("adv_type", C.c_int), ("ic_form", C.c_int),
                ("upts_type", C.c_int), ("vcjh_scheme", C.c_int), ("eta", C.c_double), ("fix_vis", C.c_int),
                ("dt", C.c_double), ("ldg_beta", C.c_double), ("ldg_tau", C.c_double)] + \
               [(k, C.c_double) for k in ("gamma", "prandtl", "S_gas", "T_gas", "R_gas", "mu_gas",
                                          "Mach_free_stream", "rho_free_stream", "L_free_stream", "T_free_stream",
                                          "rho_c_ic", "Mach_c_ic", "T_c_ic", "u_c_ic", "v_c_ic", "w_c_ic", "p_c_ic")] + \
               [("rank", C.c_int), ("nproc", C.c_int), ("pgrid", C.c_int * 3),
                ("n_bcs", C.c_int), ("bcs", C.c_void_p), ("side_bc", C.c_int * 6),
                ("dt_type", C.c_int), ("CFL", C.c_double),
                ("over_int", C.c_int), ("over_int_order", C.c_int), ("shock_cap", C.c_int), ("shock_det_field", C.c_int),
                ("s0", C.c_double), ("expf_fac", C.c_double), ("expf_order", C.c_int), ("expf_cutoff", C.c_int),
                ("LES", C.c_int), ("SGS_model", C.c_int), ("C_s", C.c_double), ("filter_ratio", C.c_double),
                ("prandtl_t", C.c_double), ("p_res", C.c_int), ("self_partition", C.c_int * 3), ("filter_type", C.c_int)]


class BcDesc(C.Structure):
    """hfxh_bc_desc: one boundary group, dimensional inputs as in the reference's input file."""
    _fields_ = [("flag", C.c_int), ("pressure_ramp", C.c_int)] + \
               [(k, C.c_double) for k in ("rho", "u", "v", "w", "p_static", "T_static", "p_total", "T_total", "nx", "ny", "nz",
                                          "mach", "p_ramp_coeff", "T_ramp_coeff", "p_total_old", "T_total_old")]


BC_TYPES = {"sub_in_simp": 0, "sub_out_simp": 1, "sub_in_char": 2, "sub_out_char": 3, "sup_in": 4, "sup_out": 5,
            "slip_wall": 6, "cyclic": 7, "isotherm_wall": 8, "adiabat_wall": 9, "char": 10, "slip_wall_dual": 11}
SIDES3 = ("z-", "y-", "x+", "y+", "x-", "z+")  # element-local face numbers of a hex
SIDES2 = ("y-", "x+", "y+", "x-")

EXCHANGE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)
REDUCE_MIN_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_double)


# the shipped Taylor-Green case (/root/reference/testcases/navier-stokes/Taylor_Green_vortex/input_TGV_SD_hex)
TGV = dict(dims=3, order=4, length=6.2831853071795862, amp=0.0, viscous=1, riemann_solve_type=3, adv_type=3, ic_form=7,
           upts_type=0, vcjh_scheme=1, eta=0.0, fix_vis=1, dt=0.00001440389, ldg_beta=0.5, ldg_tau=0.0,
           gamma=1.4, prandtl=0.72, S_gas=120.0, T_gas=291.15, R_gas=286.9, mu_gas=1.827e-05,
           Mach_free_stream=0.1, rho_free_stream=0.0008421095852102401, L_free_stream=1.0, T_free_stream=300.0,
           rho_c_ic=0.0008421095852102401, Mach_c_ic=0.1, T_c_ic=300.0)

_lib = None


def lib():
    global _lib
    if _lib is None:
        hfx.lib()  # dependency, loaded first so that the rpath-less case also resolves
        if not os.path.exists(LIB_PATH):
            raise hfx.HfxError("libhfx_host.so is not built")
        _lib = C.CDLL(LIB_PATH)
        _lib.hfxh_last_error.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise hfx.HfxError(lib().hfxh_last_error().decode())


class Case:
    def __init__(self, n, xv=None, loc_1d_upts=None, rank=0, pgrid=None, bcs=None, sides=None, self_partition=None, **kw):
        """n: cells per direction of THIS rank's block; pgrid: ranks per direction (None: one rank).
        bcs: list of boundary groups, each a dict with `type` (the reference's bc type name) and the type's
        dimensional parameters (rho,u,v,w,p_static,T_static,p_total,T_total,nx,ny,nz,mach,pressure_ramp,...);
        sides: {"x-": index into bcs, ...}; sides that are not listed are periodic."""
        d = CaseDesc()
        cfg = dict(TGV)
        cfg.update(kw)
        self._bcs = None
        if bcs:
            self._bcs = (BcDesc * len(bcs))()
            for i, b in enumerate(bcs):
                r = self._bcs[i]
                r.nx, r.T_total, r.T_total_old = 1.0, -1.0, -1.0
                for k, v in b.items():
                    if k == "type":
                        r.flag = BC_TYPES[v]
                    else:
                        setattr(r, k, v)
            d.n_bcs = len(bcs)
            d.bcs = C.cast(self._bcs, C.c_void_p)
            names = SIDES3 if cfg.get("dims", 3) == 3 else SIDES2
            for f in range(6):
                d.side_bc[f] = (sides or {}).get(names[f], -1) if f < len(names) else -1
        self.rank, self.pgrid = rank, (list(pgrid) if pgrid is not None else None)
        if pgrid is not None:
            d.rank, d.nproc = rank, int(np.prod(pgrid))
            for i in range(3):
                d.pgrid[i] = pgrid[i] if i < len(pgrid) else 1
        self.nproc = max(1, d.nproc)
        self.cfg = cfg
        for i, v in enumerate(self_partition or ()):
            d.self_partition[i] = int(v)
        for k, v in cfg.items():
            setattr(d, k, v)
        if isinstance(n, int):
            n = [n] * 3
        for i in range(3):
            d.n[i] = n[i] if i < len(n) else 1
        self._xv = None
        if xv is not None:
            self._xv = np.asfortranarray(np.array(xv, dtype=np.float64))
            d.xv = self._xv.ctypes.data_as(dp)
        self._x1 = None
        if loc_1d_upts is not None:
            self._x1 = np.ascontiguousarray(np.array(loc_1d_upts, dtype=np.float64))
            d.loc_1d_upts = self._x1.ctypes.data_as(dp)
        self.h = C.c_void_p()
        check(lib().hfxh_case_create(C.byref(d), C.byref(self.h)))
        sz = (C.c_int * 8)()
        check(lib().hfxh_case_sizes(self.h, sz))
        self.sizes = list(sz)
        self.n_eles, self.n_upts, self.n_fpts, self.n_fields, self.n_dims, self.order, self.ele_type, self.n_stages = self.sizes
        self.on_device = False

    def array(self, name):
        """Copy of a host array, Fortran-ordered with the reference's dims (trailing 1s dropped)."""
        p = dp()
        dims = (C.c_int * 4)()
        check(lib().hfxh_case_get_array(self.h, name.encode(), C.byref(p), dims))
        dims = list(dims)
        while len(dims) > 1 and dims[-1] == 1:
            dims.pop()
        n = int(np.prod(dims))
        a = np.ctypeslib.as_array(p, shape=(n,)).copy()
        return a.reshape(dims, order="F")

    def faces(self):
        L, R = ip(), ip()
        nf, ni = C.c_int(), C.c_int()
        check(lib().hfxh_case_get_faces(self.h, C.byref(L), C.byref(R), C.byref(nf), C.byref(ni)))
        n = nf.value * ni.value
        l = np.ctypeslib.as_array(L, shape=(n,)).copy().reshape((nf.value, ni.value), order="F")
        r = np.ctypeslib.as_array(R, shape=(n,)).copy().reshape((nf.value, ni.value), order="F")
        return l, r

    def bdy_faces(self):
        """(L, boundary_id) of the boundary-face block."""
        L, ids = ip(), ip()
        nf, ni = C.c_int(), C.c_int()
        check(lib().hfxh_case_get_bdy_faces(self.h, C.byref(L), C.byref(ids), C.byref(nf), C.byref(ni)))
        if ni.value == 0:
            return np.zeros((nf.value, 0), dtype=np.int32, order="F"), np.zeros(0, dtype=np.int32)
        n = nf.value * ni.value
        l = np.ctypeslib.as_array(L, shape=(n,)).copy().reshape((nf.value, ni.value), order="F")
        return l, np.ctypeslib.as_array(ids, shape=(ni.value,)).copy()

    def bc_list(self):
        """(flags (3,nbc), params (15,nbc), R_ref, ramp_counter): run_input.bc_list after non-dimensionalisation,
        in the fixtures' array form."""
        p = C.POINTER(hfx.Bc)()
        n, rc = C.c_int(), C.c_int()
        R = C.c_double()
        check(lib().hfxh_case_get_bcs(self.h, C.byref(p), C.byref(n), C.byref(R), C.byref(rc)))
        fl = np.zeros((3, n.value), dtype=np.int32, order="F")
        par = np.zeros((15, n.value), order="F")
        for b in range(n.value):
            r = p[b]
            fl[:, b] = (r.flag, r.pressure_ramp, r.use_wm)
            par[:, b] = [r.rho, r.velocity[0], r.velocity[1], r.velocity[2], r.p_static, r.T_static, r.p_total, r.T_total,
                         r.nx, r.ny, r.nz, r.p_ramp_coeff, r.T_ramp_coeff, r.p_total_old, r.T_total_old]
        return fl, par, R.value, rc.value

    def mpi_faces(self):
        """(L, Rlut, Nout_proc) of the partition-face block."""
        L, R, nout = ip(), ip(), ip()
        nf, ni = C.c_int(), C.c_int()
        check(lib().hfxh_case_get_mpi_faces(self.h, C.byref(L), C.byref(R), C.byref(nf), C.byref(ni), C.byref(nout)))
        n = nf.value * ni.value
        if n == 0:
            z = np.zeros((nf.value, 0), dtype=np.int32, order="F")
            return z, z.copy(), np.zeros(self.nproc, dtype=np.int32)
        l = np.ctypeslib.as_array(L, shape=(n,)).copy().reshape((nf.value, ni.value), order="F")
        r = np.ctypeslib.as_array(R, shape=(n,)).copy().reshape((nf.value, ni.value), order="F")
        return l, r, np.ctypeslib.as_array(nout, shape=(self.nproc,)).copy()

    def mpi_segments(self):
        """[(peer, send_first, recv_first, count)] of the partition-face block."""
        a = [ip(), ip(), ip(), ip()]
        n = C.c_int()
        check(lib().hfxh_case_get_mpi_segments(self.h, *[C.byref(x) for x in a], C.byref(n)))
        return [tuple(int(x[s]) for x in a) for s in range(n.value)]

    def set_reduce_min(self, fn):
        """fn(v) -> min over the ranks: calc_time_step's MPI_Allreduce(MIN) when the transport is the caller's"""
        self._rcb = REDUCE_MIN_CB(lambda user, v: float(fn(v)))
        check(lib().hfxh_case_set_reduce_min(self.h, self._rcb, None))

    def set_comm(self, unique_id):
        """collective: libhfx's own RCCL transport; unique_id = bytes from hfx.comm_unique_id() on rank 0"""
        if unique_id is None:  # back to the exchange / reduce hooks
            check(lib().hfxh_case_set_comm(self.h, None))
            return
        assert len(unique_id) == 128
        self._uid = C.create_string_buffer(bytes(unique_id), 128)
        check(lib().hfxh_case_set_comm(self.h, self._uid))

    def time_partitioned(self, reps):
        ms = (C.c_double * 8)()
        check(lib().hfxh_case_time_partitioned(self.h, C.c_int(reps), ms))
        return dict(zip(("phase1_interior_ldg", "phase2_gradient_flux", "phase3_interior_common_flux", "phase4_update",
                         "exchange_solution", "exchange_flux", "stage", "flux_kernel"), list(ms)[:8]))

    def comm_info(self):
        """what RCCL reports for the case's communicator: {nranks, rank, device, pci_bus_id}"""
        n, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
        bus = C.create_string_buffer(32)
        check(lib().hfxh_case_comm_info(self.h, C.byref(n), C.byref(r), C.byref(d), bus))
        return {"nranks": n.value, "rank": r.value, "device": d.value, "pci_bus_id": bus.value.decode()}

    def set_exchange(self, fn):
        """fn(kind, phase): kind 0 solution / 1 corrected gradient, phase 0 start / 1 wait."""
        self._cb = EXCHANGE_CB(lambda user, kind, phase: fn(kind, phase))
        check(lib().hfxh_case_set_exchange(self.h, self._cb, None))

    def mpi_handle(self):
        f = C.c_void_p()
        check(lib().hfxh_case_mpi_handle(self.h, C.byref(f)))
        return f

    def set_deferred(self, on):
        """deferred execution of the mirrored method calls (default on): whole stages run as fused stages"""
        check(lib().hfxh_case_set_deferred(self.h, C.c_int(1 if on else 0)))

    def run_partitioned(self, n_steps):
        check(lib().hfxh_case_run_partitioned(self.h, C.c_int(n_steps)))

    def params(self):
        p = hfx.Params()
        check(lib().hfxh_case_params(self.h, C.byref(p)))
        return p

    def registration(self):
        """dict with the fixture key names (operators, metrics, faces, params) for the oracle / raw C ABI."""
        d = {"sizes": np.array(self.sizes, dtype=np.int32)}
        names = ["opp_0", "opp_3", "detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts"]
        p = self.params()
        if p.viscous:
            names.append("opp_6")
        for i in range(self.n_dims):
            names += ["opp_1_%d" % i, "opp_2_%d" % i] + (["opp_4_%d" % i, "opp_5_%d" % i] if p.viscous else [])
        for k in names:
            d[k] = self.array(k)
        L, R = self.faces()
        t = 2 if self.n_dims == 3 else 0
        d["int%d_L" % t], d["int%d_R" % t] = L, R
        bL, bid = self.bdy_faces()
        if bid.size:
            d["bdy%d_L" % t], d["bdy%d_id" % t] = bL, bid
            d["bc_flags"], d["bc_params"], R_ref, rc = self.bc_list()
            d["bc_R_ref"], d["ramp_counter"] = np.array([R_ref]), np.array([rc], dtype=np.int32)
        for k in ("gamma", "prandtl", "rt_inf", "mu_inf", "c_sth", "fix_vis", "ldg_beta", "ldg_tau", "dt",
                  "viscous", "riemann_solve_type", "vis_riemann_solve_type", "adv_type", "dt_type"):
            d[k] = np.array([getattr(p, k)], dtype=np.float64)
        d["RK_a"] = np.array(list(p.RK_a)[:p.n_rk])
        d["RK_b"] = np.array(list(p.RK_b)[:p.n_rk])
        d["u_init"] = self.array("disu_upts0")
        if self.cfg.get("LES", 0):
            # the closure's keys in the fixtures' form (oracle_py.Case, hfx.Eles.set_les); Kappa and prandtl_t: the
            # reference's defaults (src/input.cpp:176-180) unless given
            d["LES"] = np.array([1.0])
            d["SGS_model"] = np.array([float(self.cfg.get("SGS_model", 0))])
            d["C_s"] = np.array([float(self.cfg.get("C_s", 0.0))])
            d["filter_ratio"] = np.array([float(self.cfg.get("filter_ratio", 1.0))])
            d["Kappa"] = np.array([0.41])
            d["prandtl_t"] = np.array([float(self.cfg.get("prandtl_t", 0.0)) or 0.9])
            d["Jacobian_fpts"] = self.array("Jacobian_fpts")
            if int(self.cfg.get("SGS_model", 0)) >= 2:
                d["filter_upts"] = self.array("filter_upts")
        return d

    def to_device(self, device=0):
        check(lib().hfxh_case_to_device(self.h, C.c_int(device)))
        self.on_device = True

    def handles(self):
        ctx, e, f = C.c_void_p(), C.c_void_p(), C.POINTER(C.c_void_p)()
        nb = C.c_int()
        check(lib().hfxh_case_handles(self.h, C.byref(ctx), C.byref(e), C.byref(f), C.byref(nb)))
        return ctx, e, f, nb.value

    def CalcResidual(self):
        check(lib().hfxh_case_CalcResidual(self.h))

    def run(self, n_steps):
        check(lib().hfxh_case_run(self.h, C.c_int(n_steps)))

    def run_steps_lib(self, n_steps, fused=False):
        """hfx_run_steps on this case's device blocks (the whole RK loop inside libhfx)."""
        ctx, e, f, nb = self.handles()
        hfx.check(hfx.lib().hfx_run_steps(e, f, C.c_int(nb), C.c_int(n_steps), C.c_int(int(fused))))

    def synchronize(self):
        ctx, e, f, nb = self.handles()
        hfx.check(hfx.lib().hfx_ctx_synchronize(ctx))

    def stream(self):
        ctx, e, f, nb = self.handles()
        return hfx.lib().hfx_ctx_stream(ctx)

    def write_restart(self, directory, file_num):
        check(lib().hfxh_case_write_restart(self.h, str(directory).encode(), C.c_int(file_num)))

    def read_restart(self, directory, file_num, n_files=1):
        check(lib().hfxh_case_read_restart(self.h, str(directory).encode(), C.c_int(file_num), C.c_int(n_files)))

    def calc_time_step(self):
        v = C.c_double(0)
        check(lib().hfxh_case_calc_time_step(self.h, C.byref(v)))
        return v.value

    def calc_disu_ppts(self):
        """eles::calc_disu_ppts for every element: (n_ppts, n_eles, n_fields)"""
        ptr = dp()
        dims = (C.c_int * 3)()
        check(lib().hfxh_case_calc_disu_ppts(self.h, C.byref(ptr), dims))
        n = dims[0] * dims[1] * dims[2]
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(tuple(dims), order="F").copy()

    def sync_host(self):
        check(lib().hfxh_case_sync_host(self.h))

    def close(self):
        if self.h:
            lib().hfxh_case_destroy(self.h)
            self.h = C.c_void_p()


class Simplex:
    """eles_tets (ele_type 2) / eles_pris (3) of the host mirror as producers of operators and metrics for the given
    elements: shape (3, n_spts, n_eles), n_spts 4 / 6 (straight-sided) or 10 / 15 (quadratic)."""

    class Keys(C.Structure):
        _fields_ = [("vcjh_scheme", C.c_int), ("c", C.c_double), ("SGS_model", C.c_int), ("filter_type", C.c_int),
                    ("filter_ratio", C.c_double), ("shock_cap", C.c_int), ("expf_fac", C.c_double), ("expf_order", C.c_int),
                    ("expf_cutoff", C.c_int)]

    def __init__(self, ele_type, order, shape, viscous=1, loc_1d_upts=None, vcjh_scheme=1, c=0.0, SGS_model=-1, filter_type=0,
                 filter_ratio=1.0, shock_cap=0, expf_fac=36.0, expf_order=4, expf_cutoff=0):
        """vcjh_scheme: vcjh_scheme_tet / vcjh_scheme_tri (0: c given, 1 DG, 2 SD-like, 3 Huynh-like, 4 c+); SGS_model >= 0: a
        run with an LES closure (Jacobian_fpts, and filter_upts on tetrahedra for the closures that filter the solution);
        shock_cap 1: the shock-capturing operators (inv_vandermonde, exp_filter, norm_basis_persson, persson_high_modes)"""
        shp = np.asfortranarray(np.array(shape, dtype=np.float64))
        assert shp.shape[0] == 3 and shp.shape[1] in ((4, 10) if ele_type == 2 else (6, 15))
        x1 = None if loc_1d_upts is None else np.ascontiguousarray(np.array(loc_1d_upts, dtype=np.float64))
        self.h = C.c_void_p()
        k = Simplex.Keys(vcjh_scheme, c, SGS_model, filter_type, filter_ratio, shock_cap, expf_fac, expf_order, expf_cutoff)
        check(lib().hfxh_simplex_create_keys(C.c_int(ele_type), C.c_int(order), C.c_int(viscous), C.c_int(shp.shape[2]),
                                             C.c_int(shp.shape[1]), shp.ctypes.data_as(dp), None if x1 is None else x1.ctypes.data_as(dp),
                                             C.byref(k), C.byref(self.h)))

    def array(self, name):
        p = dp()
        dims = (C.c_int * 4)()
        check(lib().hfxh_simplex_get_array(self.h, name.encode(), C.byref(p), dims))
        dims = list(dims)
        while len(dims) > 1 and dims[-1] == 1:
            dims.pop()
        n = int(np.prod(dims))
        return np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(dims, order="F")

    def close(self):
        if self.h:
            lib().hfxh_simplex_destroy(self.h)
            self.h = C.c_void_p()
