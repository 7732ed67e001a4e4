"""ctypes binding of libhfx.so (the C ABI declared in include/hfx.h).

Plumbing only: every call goes straight to the HIP library; there is no CPU
fallback.  If the shared library is missing the import fails loudly.
"""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(os.environ.get("HFX_LIB_DIR", HERE), "libhfx.so")  # HFX_LIB_DIR: an A/B build (make variant)
HEADER = os.path.join(ROOT, "include", "hfx.h")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)

(DISU_UPTS0, DISU_UPTS1, DISU_FPTS, TDISF_UPTS, NORM_TDISF_FPTS, NORM_TCONF_FPTS, DIV_TCONF_UPTS,
 DELTA_DISU_FPTS, GRAD_DISU_UPTS, GRAD_DISU_FPTS, SRC_UPTS, DT_LOCAL, SENSOR, SGSF_UPTS, SGSF_FPTS,
 DISUF_UPTS, LU, LE) = range(18)


class Les(C.Structure):
    _fields_ = [("sgs_model", C.c_int), ("pad", C.c_int), ("C_s", C.c_double), ("filter_ratio", C.c_double),
                ("Kappa", C.c_double), ("prandtl_t", C.c_double)]
CONTRACT_AUTO, CONTRACT_DENSE, CONTRACT_SPARSE = 0, 1, 2


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("gamma", "prandtl", "rt_inf", "mu_inf", "c_sth", "fix_vis", "ldg_beta", "ldg_tau", "dt")] + \
               [(n, C.c_int) for n in
                ("viscous", "riemann_solve_type", "vis_riemann_solve_type", "adv_type", "dt_type", "n_rk")] + \
               [("RK_a", C.c_double * 16), ("RK_b", C.c_double * 16)]


class ElesDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_eles", "n_upts", "n_fpts", "n_fields", "n_dims", "ele_type", "order")] + [
        ("opp_0", dp), ("opp_1", dp * 3), ("opp_2", dp * 3), ("opp_3", dp), ("opp_4", dp * 3),
        ("opp_5", dp * 3), ("opp_6", dp),
        ("detjac_upts", dp), ("JGinv_upts", dp), ("detjac_fpts", dp), ("JGinv_fpts", dp),
        ("tdA_fpts", dp), ("norm_fpts", dp)]


class Bc(C.Structure):
    """hfx_bc: one entry of run_input.bc_list after non-dimensionalisation."""
    _fields_ = [(n, C.c_int) for n in ("flag", "pressure_ramp", "use_wm", "pad")] + \
               [("rho", C.c_double), ("velocity", C.c_double * 3)] + \
               [(n, C.c_double) for n in ("p_static", "T_static", "p_total", "T_total", "nx", "ny", "nz",
                                          "p_ramp_coeff", "T_ramp_coeff", "p_total_old", "T_total_old")]


(BC_SUB_IN_SIMP, BC_SUB_OUT_SIMP, BC_SUB_IN_CHAR, BC_SUB_OUT_CHAR, BC_SUP_IN, BC_SUP_OUT, BC_SLIP_WALL, BC_CYCLIC,
 BC_ISOTHERM_WALL, BC_ADIABAT_WALL, BC_CHAR, BC_SLIP_WALL_DUAL) = range(12)


def bc_records(flags, params):
    """Bc array from the fixture form: flags (3,nbc) = flag, pressure_ramp, use_wm; params (15,nbc) = rho, velocity[3],
    p_static, T_static, p_total, T_total, nx, ny, nz, p_ramp_coeff, T_ramp_coeff, p_total_old, T_total_old."""
    fl = np.asarray(flags).reshape(3, -1, order="F")
    par = np.asarray(params).reshape(15, -1, order="F")
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


def declared_symbols():
    """Names of every function include/hfx.h declares."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hfx_[A-Za-z0-9_]+)\s*\(", txt)))


class HfxError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HfxError("libhfx.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib = C.CDLL(LIB_PATH)
        _lib.hfx_last_error.restype = C.c_char_p
        _lib.hfx_ctx_stream.restype = C.c_void_p
        _lib.hfx_ctx_stream.argtypes = [C.c_void_p]
    return _lib


def check(rc):
    if rc != 0:
        raise HfxError(lib().hfx_last_error().decode())


def _f(a):
    a = np.asfortranarray(np.array(a, dtype=np.float64))
    return a


def deferred_stats(ctx_handle):
    nf, nr, why = C.c_long(0), C.c_long(0), C.c_char_p()
    check(lib().hfx_ctx_deferred_stats(ctx_handle, C.byref(nf), C.byref(nr), C.byref(why)))
    return nf.value, nr.value, (why.value or b"").decode()


class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        check(lib().hfx_ctx_create(C.c_int(device), C.byref(self.h)))

    def set_params(self, p):
        self.params = p
        check(lib().hfx_ctx_set_params(self.h, C.byref(p)))

    def set_fused_mode(self, mode):
        check(lib().hfx_ctx_set_fused_mode(self.h, C.c_int(mode)))

    def set_contract_mode(self, mode):
        check(lib().hfx_ctx_set_contract_mode(self.h, C.c_int(mode)))

    def set_option(self, name, value):
        check(lib().hfx_ctx_set_option(self.h, name.encode(), C.c_int(int(value))))

    def set_CFL(self, CFL):
        check(lib().hfx_ctx_set_CFL(self.h, C.c_double(CFL)))

    def get_dt(self):
        v = C.c_double(0)
        check(lib().hfx_ctx_get_dt(self.h, C.byref(v)))
        return v.value

    def synchronize(self):
        check(lib().hfx_ctx_synchronize(self.h))

    def flush(self):
        """deferred execution: run what has been recorded (no-op otherwise)"""
        check(lib().hfx_ctx_flush(self.h))

    def deferred_stats(self):
        """(stages run fused, records replayed call by call, why the last replay was one)"""
        return deferred_stats(self.h)

    @property
    def stream(self):
        return lib().hfx_ctx_stream(self.h)

    def close(self):
        if self.h:
            lib().hfx_ctx_destroy(self.h)
            self.h = C.c_void_p()


class Eles:
    """One element block; `data` maps the names of hfx_eles_desc to numpy arrays (hf_array / Fortran order)."""

    def __init__(self, ctx, sizes, data, ele_type=4, order=0):
        self.ctx = ctx
        self.n_eles, self.n_upts, self.n_fpts, self.n_fields, self.n_dims = [int(s) for s in sizes]
        d = ElesDesc()
        d.n_eles, d.n_upts, d.n_fpts, d.n_fields, d.n_dims = self.n_eles, self.n_upts, self.n_fpts, self.n_fields, self.n_dims
        d.ele_type, d.order = ele_type, order
        keep = []

        def ptr(name):
            if name not in data:
                return None
            a = _f(data[name])
            keep.append(a)
            return a.ctypes.data_as(dp)

        d.opp_0 = ptr("opp_0"); d.opp_3 = ptr("opp_3"); d.opp_6 = ptr("opp_6")
        for i in range(self.n_dims):
            d.opp_1[i] = ptr("opp_1_%d" % i); d.opp_2[i] = ptr("opp_2_%d" % i)
            d.opp_4[i] = ptr("opp_4_%d" % i); d.opp_5[i] = ptr("opp_5_%d" % i)
        for k in ("detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts"):
            setattr(d, k, ptr(k))
        self.h = C.c_void_p()
        check(lib().hfx_eles_create(ctx.h, C.byref(d), C.byref(self.h)))
        nu, nfp, ne, nf, nd = self.n_upts, self.n_fpts, self.n_eles, self.n_fields, self.n_dims
        self.shapes = {
            DISU_UPTS0: (nu, ne, nf), DISU_UPTS1: (nu, ne, nf), DISU_FPTS: (nfp, ne, nf),
            TDISF_UPTS: (nu, ne, nf, nd), NORM_TDISF_FPTS: (nfp, ne, nf), NORM_TCONF_FPTS: (nfp, ne, nf),
            DIV_TCONF_UPTS: (nu, ne, nf), DELTA_DISU_FPTS: (nfp, ne, nf), GRAD_DISU_UPTS: (nu, ne, nf, nd),
            GRAD_DISU_FPTS: (nfp, ne, nf, nd), SRC_UPTS: (nu, ne, nf), DT_LOCAL: (ne,), SENSOR: (ne,),
            SGSF_UPTS: (nu, ne, nf, nd), SGSF_FPTS: (nfp, ne, nf, nd),
            DISUF_UPTS: (nu, ne, nf), LU: (nu, ne, 3 if nd == 2 else 6), LE: (nu, ne, nd)}

    def upload(self, array_id, a):
        a = _f(a)
        assert a.shape == self.shapes[array_id], (a.shape, self.shapes[array_id])
        check(lib().hfx_eles_upload(self.h, C.c_int(array_id), a.ctypes.data_as(dp)))

    def download(self, array_id):
        a = np.zeros(self.shapes[array_id], dtype=np.float64, order="F")
        check(lib().hfx_eles_download(self.h, C.c_int(array_id), a.ctypes.data_as(dp)))
        return a

    def device_ptr(self, array_id):
        p = dp()
        check(lib().hfx_eles_device_ptr(self.h, C.c_int(array_id), C.byref(p)))
        return C.cast(p, C.c_void_p).value

    def _call(self, name, *args):
        check(getattr(lib(), name)(self.h, *args))

    def set_opp_p(self, opp_p):
        a = _f(opp_p)
        self.n_ppts = a.shape[0]
        check(lib().hfx_eles_set_opp_p(self.h, C.c_int(a.shape[0]), a.ctypes.data_as(dp)))

    def calc_disu_ppts(self):
        out = np.zeros((self.n_ppts, self.n_eles, self.n_fields), dtype=np.float64, order="F")
        check(lib().hfx_eles_calc_disu_ppts(self.h, out.ctypes.data_as(dp)))
        return out

    def extrapolate_solution(self): self._call("hfx_eles_extrapolate_solution")
    def calculate_gradient(self): self._call("hfx_eles_calculate_gradient")
    def evaluate_invFlux(self): self._call("hfx_eles_evaluate_invFlux")
    def correct_gradient(self): self._call("hfx_eles_correct_gradient")
    def evaluate_viscFlux(self): self._call("hfx_eles_evaluate_viscFlux")
    def extrapolate_totalFlux(self): self._call("hfx_eles_extrapolate_totalFlux")
    def calculate_divergence(self): self._call("hfx_eles_calculate_divergence")
    def calculate_corrected_divergence(self): self._call("hfx_eles_calculate_corrected_divergence")

    def AdvanceSolution(self, in_step, adv_type):
        self._call("hfx_eles_AdvanceSolution", C.c_int(in_step), C.c_int(adv_type))

    def set_shock_capture(self, inv_vandermonde, exp_filter, norm_basis_persson, high_modes, s0, shock_det_field):
        a, b = _f(inv_vandermonde), _f(exp_filter)
        n = np.ascontiguousarray(np.ravel(norm_basis_persson).astype(np.float64))
        h = np.ascontiguousarray(np.ravel(high_modes).astype(np.int32))
        check(lib().hfx_eles_set_shock_capture(self.h, a.ctypes.data_as(dp), b.ctypes.data_as(dp), n.ctypes.data_as(dp),
                                               h.ctypes.data_as(ip), C.c_double(s0), C.c_int(shock_det_field)))

    def shock_capture(self): self._call("hfx_eles_shock_capture")

    def set_les(self, sgs_model, C_s, filter_ratio, Kappa, prandtl_t, Jacobian_fpts, wall_distance=None, filter_upts=None):
        les = Les(sgs_model, 0, C_s, filter_ratio, Kappa, prandtl_t)
        J = _f(Jacobian_fpts)
        w = _f(wall_distance) if wall_distance is not None else None
        check(lib().hfx_eles_set_les(self.h, C.byref(les), w.ctypes.data_as(dp) if w is not None else None, J.ctypes.data_as(dp)))
        if filter_upts is not None:
            F = _f(filter_upts)
            assert F.shape == (self.n_upts, self.n_upts)
            check(lib().hfx_eles_set_les_filter(self.h, F.ctypes.data_as(dp)))

    def calc_sgs_terms(self): self._call("hfx_eles_calc_sgs_terms")

    def extrapolate_sgsFlux(self): self._call("hfx_eles_extrapolate_sgsFlux")

    def set_volume_cubpts(self, opp_volume_cubpts, weight_volume_cubpts, vol_detjac_vol_cubpts):
        a, d = _f(opp_volume_cubpts), _f(vol_detjac_vol_cubpts)
        w = np.ascontiguousarray(np.ravel(weight_volume_cubpts).astype(np.float64))
        check(lib().hfx_eles_set_volume_cubpts(self.h, C.c_int(a.shape[0]), a.ctypes.data_as(dp), w.ctypes.data_as(dp),
                                               d.ctypes.data_as(dp)))

    def CalcIntegralQuantities(self, ids):
        ids = np.ascontiguousarray(np.array(ids, dtype=np.int32))
        out = np.zeros(len(ids))
        check(lib().hfx_eles_CalcIntegralQuantities(self.h, C.c_int(len(ids)), ids.ctypes.data_as(ip), out.ctypes.data_as(dp)))
        return out

    def set_h_ref(self, h_ref):
        h = np.ascontiguousarray(np.ravel(h_ref).astype(np.float64))
        check(lib().hfx_eles_set_h_ref(self.h, h.ctypes.data_as(dp)))

    def calc_dt_local(self, CFL):
        v = C.c_double(0)
        check(lib().hfx_eles_calc_dt_local(self.h, C.c_double(CFL), C.byref(v)))
        return v.value

    def set_over_int(self, opp_over_int_cubpts, over_int_filter, JGinv_over_int_cubpts):
        a, b, c = _f(opp_over_int_cubpts), _f(over_int_filter), _f(JGinv_over_int_cubpts)
        assert a.shape == (b.shape[1], self.n_upts) and b.shape[0] == self.n_upts
        check(lib().hfx_eles_set_over_int(self.h, C.c_int(a.shape[0]), a.ctypes.data_as(dp), b.ctypes.data_as(dp),
                                          c.ctypes.data_as(dp)))

    def evaluate_invFlux_over_int(self): self._call("hfx_eles_evaluate_invFlux_over_int")

    def check_nan(self):
        v = C.c_long(0)
        check(lib().hfx_eles_check_nan(self.h, C.byref(v)))
        return v.value

    def compute_res_upts(self, norm_type, field):
        v = C.c_double(0)
        check(lib().hfx_eles_compute_res_upts(self.h, C.c_int(norm_type), C.c_int(field), C.byref(v)))
        return v.value

    def close(self):
        if self.h:
            lib().hfx_eles_destroy(self.h)
            self.h = C.c_void_p()


class IntInters:
    def __init__(self, ctx, left, right, L, R):
        L = np.asfortranarray(np.array(L, dtype=np.int32))
        R = np.asfortranarray(np.array(R, dtype=np.int32))
        self.n_fpts_per_inter, self.n_inters = L.shape
        self.h = C.c_void_p()
        check(lib().hfx_int_inters_create(ctx.h, left.h, right.h, C.c_int(self.n_inters), C.c_int(self.n_fpts_per_inter),
                                          L.ctypes.data_as(ip), R.ctypes.data_as(ip), C.byref(self.h)))

    def calculate_common_invFlux(self): check(lib().hfx_int_inters_calculate_common_invFlux(self.h))
    def calculate_common_viscFlux(self): check(lib().hfx_int_inters_calculate_common_viscFlux(self.h))

    def close(self):
        if self.h:
            lib().hfx_inters_destroy(self.h)
            self.h = C.c_void_p()


class BdyInters:
    """Boundary-face block (bdy_inters): left side only, ghost state from the group's record."""

    def __init__(self, ctx, left, L, boundary_id, bcs, R_ref, ramp_counter=0):
        L = np.asfortranarray(np.array(L, dtype=np.int32))
        ids = np.ascontiguousarray(np.array(boundary_id, dtype=np.int32).ravel())
        self.n_fpts_per_inter, self.n_inters = L.shape
        self.h = C.c_void_p()
        check(lib().hfx_bdy_inters_create(ctx.h, left.h, C.c_int(self.n_inters), C.c_int(self.n_fpts_per_inter),
                                          L.ctypes.data_as(ip), ids.ctypes.data_as(ip), bcs, C.c_int(len(bcs)),
                                          C.c_double(R_ref), C.byref(self.h)))
        if ramp_counter:
            check(lib().hfx_bdy_inters_set_ramp_counter(self.h, C.c_int(ramp_counter)))

    def set_ramp_counter(self, ramp_counter):
        check(lib().hfx_bdy_inters_set_ramp_counter(self.h, C.c_int(ramp_counter)))

    def evaluate_boundaryConditions_invFlux(self, time=0.0):
        check(lib().hfx_bdy_inters_evaluate_boundaryConditions_invFlux(self.h, C.c_double(time)))

    def evaluate_boundaryConditions_viscFlux(self, time=0.0):
        check(lib().hfx_bdy_inters_evaluate_boundaryConditions_viscFlux(self.h, C.c_double(time)))

    def close(self):
        if self.h:
            lib().hfx_inters_destroy(self.h)
            self.h = C.c_void_p()


class MpiInters:
    """Partition-face block (mpi_inters): one-sided faces whose right state arrives in in_buffer_*."""

    def __init__(self, ctx, left, L, Rlut):
        L = np.asfortranarray(np.array(L, dtype=np.int32))
        Rlut = np.asfortranarray(np.array(Rlut, dtype=np.int32))
        self.n_fpts_per_inter, self.n_inters = L.shape
        self.h = C.c_void_p()
        check(lib().hfx_mpi_inters_create(ctx.h, left.h, C.c_int(self.n_inters), C.c_int(self.n_fpts_per_inter),
                                          L.ctypes.data_as(ip), Rlut.ctypes.data_as(ip), C.byref(self.h)))

    def pack_solution(self): check(lib().hfx_mpi_inters_pack_solution(self.h))
    def pack_corrected_gradient(self): check(lib().hfx_mpi_inters_pack_corrected_gradient(self.h))
    def calculate_common_invFlux(self): check(lib().hfx_mpi_inters_calculate_common_invFlux(self.h))
    def calculate_common_viscFlux(self): check(lib().hfx_mpi_inters_calculate_common_viscFlux(self.h))

    def buffer(self, which):
        """(device pointer, doubles) of 0 out_buffer_disu, 1 in_buffer_disu, 2 out_buffer_grad_disu, 3 in_buffer_grad_disu."""
        return mpi_buffer(self.h, which)

    def set_neighbours(self, segments):
        """segments: [(peer, send_first, recv_first, count)] (hfx_mpi_inters_set_neighbours)"""
        a = [np.ascontiguousarray([s[i] for s in segments], dtype=np.int32) for i in range(4)]
        check(lib().hfx_mpi_inters_set_neighbours(self.h, C.c_int(len(segments)), *[x.ctypes.data_as(ip) for x in a]))

    # mpi_inters::send_* / receive_* over libhfx's RCCL transport (comm: hfx.Comm)
    def send_solution(self, comm): check(lib().hfx_mpi_inters_send_solution(self.h, comm.h))
    def receive_solution(self, comm): check(lib().hfx_mpi_inters_receive_solution(self.h, comm.h))
    def send_corrected_gradient(self, comm): check(lib().hfx_mpi_inters_send_corrected_gradient(self.h, comm.h))
    def receive_corrected_gradient(self, comm): check(lib().hfx_mpi_inters_receive_corrected_gradient(self.h, comm.h))

    def close(self):
        if self.h:
            lib().hfx_inters_destroy(self.h)
            self.h = C.c_void_p()


def CalcResidual_blocks(eles, faces):
    """hfx_CalcResidual_blocks: several element blocks (a mixed mesh), face blocks between any two of them"""
    ea = (C.c_void_p * len(eles))(*[e.h for e in eles])
    fa = (C.c_void_p * max(1, len(faces)))(*[f.h for f in faces])
    check(lib().hfx_CalcResidual_blocks(ea, C.c_int(len(eles)), fa, C.c_int(len(faces))))


def run_steps_blocks(eles, faces, n_steps, fused=0):
    ea = (C.c_void_p * len(eles))(*[e.h for e in eles])
    fa = (C.c_void_p * max(1, len(faces)))(*[f.h for f in faces])
    check(lib().hfx_run_steps_blocks(ea, C.c_int(len(eles)), fa, C.c_int(len(faces)), C.c_int(n_steps), C.c_int(int(fused))))


def run_steps_partitioned_blocks(eles, int_faces, mpi_faces, comm, n_steps):
    """hfx_run_steps_partitioned_blocks: the general fused stage on partitioned element blocks"""
    ea = (C.c_void_p * len(eles))(*[e.h for e in eles])
    fa = (C.c_void_p * max(1, len(int_faces)))(*[f.h for f in int_faces])
    ma = (C.c_void_p * max(1, len(mpi_faces)))(*[f.h for f in mpi_faces])
    check(lib().hfx_run_steps_partitioned_blocks(ea, C.c_int(len(eles)), fa, C.c_int(len(int_faces)), ma, C.c_int(len(mpi_faces)), comm.h,
                                                 C.c_int(n_steps)))


def comm_unique_id():
    """128 bytes for hfx_comm_create (ncclGetUniqueId): made on rank 0, distributed by the launcher."""
    b = C.create_string_buffer(128)
    check(lib().hfx_comm_get_unique_id(b))
    return b.raw


class Comm:
    """libhfx's RCCL communicator of one rank (hfx_comm_create is collective over the ranks)."""

    def __init__(self, ctx_handle, unique_id, nranks, rank):
        self.h = C.c_void_p()
        self._uid = C.create_string_buffer(bytes(unique_id), 128)
        check(lib().hfx_comm_create(ctx_handle, self._uid, C.c_int(nranks), C.c_int(rank), C.byref(self.h)))

    def allreduce(self, values, op):
        """op: "min" | "max" | "sum"; returns the reduced list"""
        a = (C.c_double * len(values))(*values)
        check(lib().hfx_comm_allreduce(self.h, a, C.c_int(len(values)), C.c_int({"min": 0, "max": 1, "sum": 2}[op])))
        return list(a)

    def close(self):
        if self.h:
            lib().hfx_comm_destroy(self.h)
            self.h = C.c_void_p()


def mpi_buffer(handle, which):
    p = dp()
    n = C.c_size_t(0)
    check(lib().hfx_mpi_inters_buffer(handle, C.c_int(which), C.byref(p), C.byref(n)))
    return C.cast(p, C.c_void_p).value, n.value


def stage_partitioned(eles_h, int_faces, mpi_faces, phase, in_step, first):
    """hfx_stage_partitioned on raw handles (lists of c_void_p)."""
    fi = (C.c_void_p * max(1, len(int_faces)))(*int_faces)
    fm = (C.c_void_p * max(1, len(mpi_faces)))(*mpi_faces)
    check(lib().hfx_stage_partitioned(eles_h, fi, C.c_int(len(int_faces)), fm, C.c_int(len(mpi_faces)),
                                      C.c_int(phase), C.c_int(in_step), C.c_int(first)))


def _face_array(faces):
    arr = (C.c_void_p * max(1, len(faces)))()
    for i, f in enumerate(faces):
        arr[i] = f.h
    return arr


def CalcResidual(eles, faces):
    check(lib().hfx_CalcResidual(eles.h, _face_array(faces), C.c_int(len(faces))))


def run_steps(eles, faces, n_steps, fused=False):
    check(lib().hfx_run_steps(eles.h, _face_array(faces), C.c_int(len(faces)), C.c_int(n_steps), C.c_int(int(fused))))


def params_from(data):
    """hfx Params from a dict holding the scalar names of the fixtures / host setup."""
    s = lambda k, dflt=0.0: float(np.ravel(data[k])[0]) if k in data else dflt
    p = Params()
    for k in ("gamma", "prandtl", "rt_inf", "mu_inf", "c_sth", "fix_vis", "ldg_beta", "ldg_tau", "dt"):
        setattr(p, k, s(k))
    for k in ("viscous", "riemann_solve_type", "vis_riemann_solve_type", "adv_type", "dt_type"):
        setattr(p, k, int(s(k)))
    ra, rb = np.ravel(data["RK_a"]), np.ravel(data["RK_b"])
    p.n_rk = len(ra)
    for i in range(len(ra)):
        p.RK_a[i] = float(ra[i])
        p.RK_b[i] = float(rb[i])
    return p
