"""GPU parity of the per-method path (libhfx through its C ABI) against the genuine reference's
fixtures and against the oracle, on the same inputs.

Tolerance: FP64; the HIP kernels use FMA contraction and (dense mode) the MFMA's internal 4-term
sums, the reference's CPU branch does neither, so agreement is to rounding: 1e-12 relative to the
array's largest magnitude for one residual (5e-11 for the divergence), 1e-11 after a full RK step.
BASELINE.json's stated bar is 1e-10 on conserved residuals.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import hfx
import oracle_py as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# the restart fixtures hold a state file only (tests/test_restart_io.py), the long run and the plot-point fixtures states
# only (test_gpu_host_mirror.py), the full-size fixtures norms and sample elements (test_fullsize_vs_reference.py), the
# mixed-mesh fixtures several element classes (test_mixed_mesh.py)
ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
             if "restart" not in p and "_long" not in p and "_plot" not in p and "_tgv" not in p and "mixed_" not in p)
BDY = [n for n in ALL if "bdy" in n]
RTOL1 = 1e-12
# div_tconf is a difference of terms ~1e3 times larger than itself (pressure-dominated energy
# flux), so its relative rounding error is correspondingly larger
RTOLD = 5e-11
RTOLS = 1e-11


def relerr(a, b):
    scale = np.abs(b).max()
    return np.abs(a - b).max() / (scale if scale > 0 else 1.0)


@pytest.fixture(scope="module")
def ctx():
    c = hfx.Context(0)
    yield c
    c.close()


def build(ctx, d, mode=hfx.CONTRACT_AUTO):
    ctx.set_params(hfx.params_from(d))
    ctx.set_contract_mode(mode)
    sz = [int(v) for v in d["sizes"]]
    e = hfx.Eles(ctx, sz[:5], d, ele_type=sz[6], order=sz[5])
    faces = []
    for t in range(3):
        if "int%d_L" % t in d:
            faces.append(hfx.IntInters(ctx, e, e, d["int%d_L" % t], d["int%d_R" % t]))
    for t in range(3):
        if "bdy%d_L" % t in d:
            faces.append(hfx.BdyInters(ctx, e, d["bdy%d_L" % t], d["bdy%d_id" % t],
                                       hfx.bc_records(d["bc_flags"], d["bc_params"]),
                                       float(np.ravel(d["bc_R_ref"])[0]), int(np.ravel(d["ramp_counter"])[0])))
    if "LES" in d and int(np.ravel(d["LES"])[0]):
        sc = lambda k: float(np.ravel(d[k])[0])
        e.set_les(int(sc("SGS_model")), sc("C_s"), sc("filter_ratio"), sc("Kappa"), sc("prandtl_t"), d["Jacobian_fpts"],
                  d["wall_distance"] if "wall_distance" in d else None, d["filter_upts"] if "filter_upts" in d else None)
    if "over_int" in d and int(np.ravel(d["over_int"])[0]):
        e.set_over_int(d["opp_over_int_cubpts"], d["over_int_filter"], d["JGinv_over_int_cubpts"])
    if "shock_cap" in d and int(np.ravel(d["shock_cap"])[0]):
        e.set_shock_capture(d["inv_vandermonde"], d["exp_filter"], d["norm_basis_persson"], d["persson_high_modes"],
                            float(np.ravel(d["s0"])[0]), int(np.ravel(d["shock_det_field"])[0]))
    e.upload(hfx.DISU_UPTS0, d["u_init"])
    return e, faces


@pytest.mark.parametrize("mode", [hfx.CONTRACT_SPARSE, hfx.CONTRACT_DENSE])
def test_every_intermediate(ctx, mode):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_n3_deformed.npz")))
    e, faces = build(ctx, d, mode)
    e.extrapolate_solution()
    assert relerr(e.download(hfx.DISU_FPTS), d["s0_disu_fpts"]) < RTOL1
    e.calculate_gradient()
    assert relerr(e.download(hfx.GRAD_DISU_UPTS), d["s0_grad_disu_upts_ref"]) < RTOL1
    e.evaluate_invFlux()
    assert relerr(e.download(hfx.TDISF_UPTS), d["s0_tdisf_upts_inv"]) < RTOL1
    for f in faces:
        f.calculate_common_invFlux()
    assert relerr(e.download(hfx.NORM_TCONF_FPTS), d["s0_norm_tconf_fpts_inv"]) < RTOL1
    assert relerr(e.download(hfx.DELTA_DISU_FPTS), d["s0_delta_disu_fpts"]) < RTOL1
    e.correct_gradient()
    assert relerr(e.download(hfx.GRAD_DISU_UPTS), d["s0_grad_disu_upts"]) < RTOL1
    assert relerr(e.download(hfx.GRAD_DISU_FPTS), d["s0_grad_disu_fpts"]) < RTOL1
    e.evaluate_viscFlux()
    assert relerr(e.download(hfx.TDISF_UPTS), d["s0_tdisf_upts"]) < RTOL1
    e.extrapolate_totalFlux()
    assert relerr(e.download(hfx.NORM_TDISF_FPTS), d["s0_norm_tdisf_fpts"]) < RTOL1
    e.calculate_divergence()
    assert relerr(e.download(hfx.DIV_TCONF_UPTS), d["s0_div_tconf_upts_disc"]) < RTOLD
    for f in faces:
        f.calculate_common_viscFlux()
    assert relerr(e.download(hfx.NORM_TCONF_FPTS), d["s0_norm_tconf_fpts"]) < RTOL1
    e.calculate_corrected_divergence()
    assert relerr(e.download(hfx.DIV_TCONF_UPTS), d["s0_div_tconf_upts"]) < RTOLD
    # norm_tconf is left overwritten with norm_tconf - norm_tdisf (src/eles.cpp:1746)
    assert relerr(e.download(hfx.NORM_TCONF_FPTS), d["s0_norm_tconf_fpts"] - d["s0_norm_tdisf_fpts"]) < RTOL1
    assert e.check_nan() == -1
    for fld in range(e.n_fields):
        for nt in (1, 2):
            got = e.compute_res_upts(nt, fld)
            want = d["s0_res_sums"][fld, nt - 1]
            assert abs(got - want) <= 1e-11 * abs(want)
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("mode", [hfx.CONTRACT_AUTO, hfx.CONTRACT_DENSE])
@pytest.mark.parametrize("name", [n for n in ALL if "overint" in n])
def test_over_integration_flux(ctx, name, mode):
    """eles::evaluate_invFlux_over_int: sum-factorised (one kernel: interpolation, flux at the cubature points,
    projection, all in LDS; 1-D factors recovered from the registered matrices) and as two dense FP64 MFMA
    contractions (n_cub = 343 / 100 columns) around the pointwise flux kernel."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d, mode)
    e.evaluate_invFlux_over_int()
    assert relerr(e.download(hfx.TDISF_UPTS), d["s0_tdisf_upts_inv"]) < RTOL1
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("name", [n for n in ALL if "_les_" in n])
def test_les_intermediates(ctx, name):
    """LES eddy-viscosity closure on the device: SGS flux at solution and flux points, total flux, residual."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d)
    if int(np.ravel(d["SGS_model"])[0]) >= 2:
        e.calc_sgs_terms()
    if "s0_sgsf_upts" not in d:
        pytest.skip("a level-1 fixture: states only")
    hfx.CalcResidual(e, faces)
    assert relerr(e.download(hfx.SGSF_UPTS), d["s0_sgsf_upts"]) < 1e-11
    assert relerr(e.download(hfx.SGSF_FPTS), d["s0_sgsf_fpts"]) < 1e-11
    assert relerr(e.download(hfx.TDISF_UPTS), d["s0_tdisf_upts"]) < RTOL1
    assert relerr(e.download(hfx.DIV_TCONF_UPTS), d["s0_div_tconf_upts"]) < RTOLD
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("name", [n for n in ALL if "integrals" in n])
def test_integral_quantities(ctx, name):
    """eles::CalcIntegralQuantities on the device: kinetic energy, enstrophy, pressure dilatation, strain products."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d)
    e.set_volume_cubpts(d["opp_volume_cubpts"], d["weight_volume_cubpts"], d["vol_detjac_vol_cubpts"])
    hfx.CalcResidual(e, faces)  # corrected gradients of u_init
    got = e.CalcIntegralQuantities(np.ravel(d["integral_quantity_ids"]))
    want = np.ravel(d["s0_integral_quantities"])
    assert np.all(np.abs(got - want) <= 1e-11 * np.abs(want).max())
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("name", BDY)
def test_boundary_intermediates(ctx, name):
    """bdy_inters: the face arrays after the inviscid and the viscous sweep (interior + boundary blocks)."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d)
    inner = [f for f in faces if isinstance(f, hfx.IntInters)]
    bdy = [f for f in faces if isinstance(f, hfx.BdyInters)]
    visc = int(np.ravel(d["viscous"])[0])
    assert bdy
    e.extrapolate_solution()
    if visc:
        e.calculate_gradient()
    e.evaluate_invFlux()
    for f in inner:
        f.calculate_common_invFlux()
    for f in bdy:
        f.evaluate_boundaryConditions_invFlux()
    assert relerr(e.download(hfx.NORM_TCONF_FPTS), d["s0_norm_tconf_fpts_inv"]) < RTOL1
    if visc:
        assert relerr(e.download(hfx.DELTA_DISU_FPTS), d["s0_delta_disu_fpts"]) < RTOL1
        e.correct_gradient()
        e.evaluate_viscFlux()
    e.extrapolate_totalFlux()
    e.calculate_divergence()
    if visc:
        for f in inner:
            f.calculate_common_viscFlux()
        for f in bdy:
            f.evaluate_boundaryConditions_viscFlux()
        assert relerr(e.download(hfx.NORM_TCONF_FPTS), d["s0_norm_tconf_fpts"]) < RTOL1
    e.calculate_corrected_divergence()
    assert relerr(e.download(hfx.DIV_TCONF_UPTS), d["s0_div_tconf_upts"]) < RTOLD
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("name", ALL)
def test_stage_states_vs_reference(ctx, name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d)
    nstage = int(d["sizes"][7])
    adv = int(np.ravel(d["adv_type"])[0])
    steps = sorted({int(k.split("_")[1][4:]) for k in d if k.startswith("u_step")})
    dt_type = int(np.ravel(d["dt_type"])[0])
    if dt_type != 0:
        e.set_h_ref(d["h_ref"])
    for st in steps:
        if dt_type != 0:
            # calc_time_step (src/solver.cpp:484-549) on the device
            dt = e.calc_dt_local(float(np.ravel(d["CFL"])[0]))
            want = float(np.ravel(d["dt_step%d" % st])[0])
            assert abs(dt - want) <= 1e-11 * want
            if dt_type == 2:
                assert relerr(e.download(hfx.DT_LOCAL), np.ravel(d["dt_local_step%d" % st])) < 1e-11
            p = hfx.params_from(d)
            p.dt = dt
            ctx.set_params(p)
        for rk in range(nstage):
            if rk == 0 and "LES" in d and int(np.ravel(d["SGS_model"])[0]) >= 2:
                e.calc_sgs_terms()  # src/solver.cpp:55-62
                if st == 0 and "s0_Lu" in d:
                    assert relerr(e.download(hfx.DISUF_UPTS), d["s0_disuf_upts"]) < 1e-12
                    assert relerr(e.download(hfx.LU), d["s0_Lu"]) < 1e-10
                    assert relerr(e.download(hfx.LE), d["s0_Le"]) < 1e-10
                if st == 0 and "s0_u_after_sgs_terms" in d:
                    assert relerr(e.download(hfx.DISU_UPTS0), d["s0_u_after_sgs_terms"]) < 1e-12
            hfx.CalcResidual(e, faces)
            if st == 0 and rk == 0:
                assert relerr(e.download(hfx.DIV_TCONF_UPTS), d["s0_div_tconf_upts"]) < RTOLD
            e.AdvanceSolution(rk, adv)
            if "shock_cap" in d:  # src/HiFiLES.cpp:214-216
                e.shock_capture()
                if st == 0 and rk == 0:
                    sens = e.download(hfx.SENSOR)
                    assert relerr(sens, np.ravel(d["s0_sensor"])) < 1e-9
                    assert np.array_equal(sens >= float(np.ravel(d["s0"])[0]), np.ravel(d["s0_sensor"]) >= float(np.ravel(d["s0"])[0]))
            key = "u_step%d_stage%d" % (st, rk)
            if key in d:
                assert relerr(e.download(hfx.DISU_UPTS0), d[key]) < RTOLS, key
        if "bc_flags" in d and d["bc_flags"][1].any():
            # `if (run_input.pressure_ramp) run_input.ramp_counter++` after every time step (src/HiFiLES.cpp:224-225):
            # a caller that drives the stages itself passes the counter on (hfx_run_steps does it internally)
            for f in faces:
                if isinstance(f, hfx.BdyInters):
                    f.set_ramp_counter(int(np.ravel(d["ramp_counter"])[0]) + st + 1)
    assert e.check_nan() == -1
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("name", ["hex_p4_n3_deformed", "hex_p3_n3_deformed", "hex_p1_roem", "hex_p4_jet", "hex_p3_shock",
                                  "quad_p3_shock", "quad_p3_overint"])
def test_dense_mfma_path_vs_reference(ctx, name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d, hfx.CONTRACT_DENSE)
    hfx.run_steps(e, faces, 1)
    nstage = int(d["sizes"][7])
    assert relerr(e.download(hfx.DISU_UPTS0), d["u_step0_stage%d" % (nstage - 1)]) < RTOLS
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("waves,split", [(4, 1), (8, 2), (4, 4), (8, 4)])
def test_dense_mfma_workgroup_shapes_give_the_same_bits(ctx, waves, split):
    """options dense_waves / dense_split deal the (16-row tile x column group) units of the dense contraction to 4 or 8
    waves; every output element is still one MFMA accumulator walked over k in the same order, so all shapes -- and the
    shape chosen from the operator's rows -- give identical bits."""
    d = dict(np.load(os.path.join(GOLDEN, "hex_p4_n3_deformed.npz")))

    def run():
        e, faces = build(ctx, d, hfx.CONTRACT_DENSE)
        hfx.run_steps(e, faces, 1)
        u = e.download(hfx.DISU_UPTS0).copy()
        for f in faces:
            f.close()
        e.close()
        return u

    want = run()
    try:
        ctx.set_option("dense_waves", waves)
        ctx.set_option("dense_split", split)
        got = run()
    finally:
        ctx.set_option("dense_waves", 0)
        ctx.set_option("dense_split", 0)
    nstage = int(d["sizes"][7])
    assert relerr(got, d["u_step0_stage%d" % (nstage - 1)]) < RTOLS
    assert np.array_equal(got, want)


def test_run_steps_matches_oracle_two_steps(ctx, oracle):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p3_n3_deformed.npz")))
    e, faces = build(ctx, d)
    hfx.run_steps(e, faces, 2)
    c = O.Case(d)
    ce = c.c_eles()
    cf, nb = c.c_faces()
    for _ in range(2):
        assert oracle.orc_rk_step(C.byref(ce), cf, nb, C.byref(c.params)) == -1
    assert relerr(e.download(hfx.DISU_UPTS0), c.arr["u0"]) < RTOLS
    assert relerr(e.download(hfx.DISU_UPTS1), c.arr["u1"]) < 1e-9  # RK register: small numbers, looser
    for f in faces:
        f.close()
    e.close()


def test_nan_flag_reports_first_index(ctx):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p1_rusanov.npz")))
    e, faces = build(ctx, d)
    u = np.array(d["u_init"], order="F")
    u[3, 5, 0] = np.nan
    e.upload(hfx.DISU_UPTS0, u)
    hfx.CalcResidual(e, faces)
    bad = e.check_nan()
    div = e.download(hfx.DIV_TCONF_UPTS)
    assert bad == int(np.flatnonzero(np.isnan(div.ravel(order="F")))[0])
    assert e.check_nan() == -1  # flag is cleared by the read
    for f in faces:
        f.close()
    e.close()


def test_error_convention(ctx):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p1_rusanov.npz")))
    e, faces = build(ctx, d)
    with pytest.raises(hfx.HfxError):
        e.AdvanceSolution(7, 3)  # stage out of range for RK45
    L = np.array(d["int2_L"], order="F").copy()
    L[0, 0] = 10 ** 8
    with pytest.raises(hfx.HfxError):
        hfx.IntInters(ctx, e, e, L, d["int2_R"])
    p = hfx.params_from(d)
    p.riemann_solve_type = 1
    with pytest.raises(hfx.HfxError):
        ctx.set_params(p)
    for f in faces:
        f.close()
    e.close()
