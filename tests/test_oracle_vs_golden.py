"""Pin the oracle (oracle/oracle.c) against fixtures captured from the GENUINE reference.

The fixtures under tests/golden were produced by oracle/capture_golden.py driving
oracle/_ref/ref_harness (the reference compiled from /root/reference/src).  Inputs
(operators, metrics, face tables, initial state, parameters) and expected outputs are
both the reference's; the oracle must reproduce every intermediate of one residual
evaluation and the state after each RK stage.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle_py as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# the full-size fixtures (hex_p4_n*_tgv) hold norms and sample elements only: tests/test_fullsize_vs_reference.py
# the mixed-mesh fixtures several element classes: tests/test_mixed_mesh.py
ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if "_tgv" not in p and "mixed_" not in p)

# The oracle repeats the reference's operation order; remaining differences are compiler-level
# (x87-free SSE2 both sides, no FMA) so the tolerance is a few ulps of the array's scale.
RTOL = 1e-13


def relerr(a, b):
    scale = np.abs(b).max()
    return np.abs(a - b).max() / (scale if scale > 0 else 1.0)


def needs_metrics(d):
    return "detjac_upts" in d


@pytest.mark.parametrize("name", [n for n in ALL])
def test_stage_states(oracle, name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if not needs_metrics(d):
        pytest.skip("fixture has no metrics (covered by the host-setup tests)")
    c = O.Case(d)
    e = c.c_eles()
    f, nfb = c.c_faces()
    bd, nbd = c.c_bdy()
    sh = c.c_shock() if c.shock_cap else None
    nstage = int(d["sizes"][7])
    steps = sorted({int(k.split("_")[1][4:]) for k in d if k.startswith("u_step")})
    dt_type = int(np.ravel(d["dt_type"])[0])
    dtl = None
    for st in steps:
        if dt_type != 0:
            # calc_time_step (src/solver.cpp:484-549): per-element CFL step, global minimum
            CFL, order = float(np.ravel(d["CFL"])[0]), int(d["sizes"][5])
            h = np.ravel(d["h_ref"])
            dtl = np.array([oracle.orc_calc_dt_local(C.byref(e), C.byref(c.params), i, h[i], CFL, order) for i in range(c.n_eles)])
            assert abs(dtl.min() - float(np.ravel(d["dt_step%d" % st])[0])) <= 1e-13 * dtl.min()
            c.params.dt = dtl.min()
            if dt_type == 2:
                assert relerr(dtl, np.ravel(d["dt_local_step%d" % st])) < 1e-13
                e.dt_local = dtl.ctypes.data_as(O.dp)
        for rk in range(nstage):
            if rk == 0 and c.les and c.les["sgs_model"] >= 2:
                # similarity-type closures: filtered solution and Leonard terms at the first stage of a step (src/solver.cpp:55-62)
                assert oracle.orc_calc_sgs_terms(C.byref(e)) == -1
                if st == 0:
                    assert relerr(c.arr["disuf_upts"], d["s0_disuf_upts"]) < RTOL
                    assert relerr(c.arr["u0"], d["s0_u_after_sgs_terms"]) < RTOL
                    if "s0_Lu" in d:
                        assert relerr(c.arr["Lu"], d["s0_Lu"]) < 1e-11  # differences of nearly equal products
                        assert relerr(c.arr["Le"], d["s0_Le"]) < 1e-11
            bad = oracle.orc_CalcResidual_bdy(C.byref(e), f, nfb, bd, nbd, C.byref(c.params))
            assert bad == -1
            if st == 0 and rk == 0:
                assert relerr(c.arr["div_tconf_upts"], d["s0_div_tconf_upts"]) < RTOL
                for fld in range(c.n_fields):
                    for nt in (1, 2):
                        got = oracle.orc_compute_res_upts(C.byref(e), nt, fld)
                        want = d["s0_res_sums"][fld, nt - 1]
                        assert abs(got - want) <= 1e-12 * abs(want)
            oracle.orc_AdvanceSolution(C.byref(e), C.byref(c.params), rk)
            if sh is not None:  # src/HiFiLES.cpp:214-216
                if st == 0 and rk == 0:
                    assert relerr(c.arr["u0"], d["s0_u_before_shock_capture"]) < RTOL
                oracle.orc_shock_capture(C.byref(e), C.byref(sh))
                if st == 0 and rk == 0:
                    assert relerr(c.arr["sensor"], np.ravel(d["s0_sensor"])) < 1e-11
                    assert 0 < (c.arr["sensor"] >= c.s0).sum() < c.n_eles  # the filter branch is exercised
            key = "u_step%d_stage%d" % (st, rk)
            if key in d:
                assert relerr(c.arr["u0"], d[key]) < RTOL, key
        if "bc_flags" in d and d["bc_flags"][1].any():
            # `if (run_input.pressure_ramp) run_input.ramp_counter++` after every time step (src/HiFiLES.cpp:224-225)
            c.ramp_counter += 1
            bd, nbd = c.c_bdy()


def test_every_intermediate(oracle):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_n3_deformed.npz")))
    c = O.Case(d)
    e = c.c_eles()
    f, nfb = c.c_faces()
    P = C.byref(c.params)
    E = C.byref(e)
    a = c.arr
    oracle.orc_extrapolate_solution(E)
    assert relerr(a["disu_fpts"], d["s0_disu_fpts"]) < RTOL
    oracle.orc_calculate_gradient(E)
    assert relerr(a["grad_disu_upts"], d["s0_grad_disu_upts_ref"]) < RTOL
    oracle.orc_evaluate_invFlux(E, P)
    assert relerr(a["tdisf_upts"], d["s0_tdisf_upts_inv"]) < RTOL
    for b in range(nfb):
        oracle.orc_int_calculate_common_invFlux(C.byref(f[b]), E, P)
    assert relerr(a["norm_tconf_fpts"], d["s0_norm_tconf_fpts_inv"]) < RTOL
    assert relerr(a["delta_disu_fpts"], d["s0_delta_disu_fpts"]) < RTOL
    oracle.orc_correct_gradient(E)
    assert relerr(a["grad_disu_upts"], d["s0_grad_disu_upts"]) < RTOL
    assert relerr(a["grad_disu_fpts"], d["s0_grad_disu_fpts"]) < RTOL
    oracle.orc_evaluate_viscFlux(E, P)
    assert relerr(a["tdisf_upts"], d["s0_tdisf_upts"]) < RTOL
    oracle.orc_extrapolate_totalFlux(E)
    assert relerr(a["norm_tdisf_fpts"], d["s0_norm_tdisf_fpts"]) < RTOL
    oracle.orc_calculate_divergence(E)
    assert relerr(a["div_tconf_upts"], d["s0_div_tconf_upts_disc"]) < RTOL
    for b in range(nfb):
        oracle.orc_int_calculate_common_viscFlux(C.byref(f[b]), E, P)
    assert relerr(a["norm_tconf_fpts"], d["s0_norm_tconf_fpts"]) < RTOL
    assert oracle.orc_calculate_corrected_divergence(E) == -1
    assert relerr(a["div_tconf_upts"], d["s0_div_tconf_upts"]) < RTOL


def test_threads_do_not_change_results(oracle):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_n3_deformed.npz")))
    outs = []
    for nt in (1, 4):
        oracle.orc_set_threads(nt)
        c = O.Case(d)
        e = c.c_eles()
        f, nfb = c.c_faces()
        assert oracle.orc_rk_step(C.byref(e), f, nfb, C.byref(c.params)) == -1
        outs.append(c.arr["u0"].copy())
    oracle.orc_set_threads(1)
    assert np.array_equal(outs[0], outs[1])


BDY = [n for n in ALL if "bdy" in n]


@pytest.mark.parametrize("name", BDY)
def test_boundary_intermediates(oracle, name):
    """bdy_inters (src/bdy_inters.cpp): common flux / common solution after the inviscid sweep and the total
    common flux after the viscous sweep, interior and boundary faces together, against the genuine reference."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c = O.Case(d)
    e = c.c_eles()
    f, nfb = c.c_faces()
    bd, nbd = c.c_bdy()
    assert nbd >= 1
    P, E, a = C.byref(c.params), C.byref(e), c.arr
    oracle.orc_extrapolate_solution(E)
    if c.viscous:
        oracle.orc_calculate_gradient(E)
    oracle.orc_evaluate_invFlux(E, P)
    for b in range(nfb):
        oracle.orc_int_calculate_common_invFlux(C.byref(f[b]), E, P)
    for b in range(nbd):
        oracle.orc_bdy_evaluate_boundaryConditions_invFlux(C.byref(bd[b]), E, P)
    # pow() of libm vs the reference's build: a few ulps
    assert relerr(a["norm_tconf_fpts"], d["s0_norm_tconf_fpts_inv"]) < 1e-12
    if c.viscous:
        assert relerr(a["delta_disu_fpts"], d["s0_delta_disu_fpts"]) < 1e-12
        oracle.orc_correct_gradient(E)
        assert relerr(a["grad_disu_fpts"], d["s0_grad_disu_fpts"]) < 1e-12
        oracle.orc_evaluate_viscFlux(E, P)
    oracle.orc_extrapolate_totalFlux(E)
    oracle.orc_calculate_divergence(E)
    if c.viscous:
        for b in range(nfb):
            oracle.orc_int_calculate_common_viscFlux(C.byref(f[b]), E, P)
        for b in range(nbd):
            oracle.orc_bdy_evaluate_boundaryConditions_viscFlux(C.byref(bd[b]), E, P)
        assert relerr(a["norm_tconf_fpts"], d["s0_norm_tconf_fpts"]) < 1e-12
    assert oracle.orc_calculate_corrected_divergence(E) == -1
    assert relerr(a["div_tconf_upts"], d["s0_div_tconf_upts"]) < 1e-12


@pytest.mark.parametrize("name", [n for n in ALL if "overint" in n])
def test_over_integration_flux(oracle, name):
    """eles::evaluate_invFlux_over_int: the de-aliased transformed inviscid flux at the solution points."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c = O.Case(d)
    assert c.n_cub > c.n_upts
    e = c.c_eles()
    oracle.orc_evaluate_invFlux_over_int(C.byref(e), C.byref(c.params), C.c_int(c.n_cub), e.opp_over_int_cubpts,
                                         e.over_int_filter, e.JGinv_over_int_cubpts)
    assert relerr(c.arr["tdisf_upts"], d["s0_tdisf_upts_inv"]) < RTOL


@pytest.mark.parametrize("name", [n for n in ALL if "integrals" in n])
def test_integral_quantities(oracle, name):
    """eles::CalcIntegralQuantities: kinetic energy, enstrophy, pressure dilatation, strain products."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c = O.Case(d)
    e = c.c_eles()
    f, nfb = c.c_faces()
    assert oracle.orc_CalcResidual(C.byref(e), f, nfb, C.byref(c.params)) == -1  # corrected gradients of u_init
    opp = np.asfortranarray(d["opp_volume_cubpts"])
    w = np.ascontiguousarray(np.ravel(d["weight_volume_cubpts"]))
    dj = np.asfortranarray(d["vol_detjac_vol_cubpts"])
    ids = np.ascontiguousarray(np.ravel(d["integral_quantity_ids"]).astype(np.int32))
    out = np.zeros(len(ids))
    oracle.orc_CalcIntegralQuantities(C.byref(e), C.byref(c.params), C.c_int(opp.shape[0]), O.fptr(opp), w.ctypes.data_as(O.dp),
                                      O.fptr(dj), C.c_int(len(ids)), ids.ctypes.data_as(O.ip), out.ctypes.data_as(O.dp))
    want = np.ravel(d["s0_integral_quantities"])
    assert np.all(np.abs(out - want) <= 1e-12 * np.abs(want).max())


@pytest.mark.parametrize("name", [n for n in ALL if "_les_" in n])
def test_les_intermediates(oracle, name):
    """LES eddy-viscosity closure: SGS flux at the solution points (added to the viscous flux), its extrapolation,
    and the common viscous flux that carries it across the faces."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c = O.Case(d)
    assert c.les is not None
    e = c.c_eles()
    f, nfb = c.c_faces()
    bd, nbd = c.c_bdy()
    if c.les["sgs_model"] >= 2:  # filtered solution and Leonard terms first (src/solver.cpp:55-62)
        assert oracle.orc_calc_sgs_terms(C.byref(e)) == -1
    assert oracle.orc_CalcResidual_bdy(C.byref(e), f, nfb, bd, nbd, C.byref(c.params)) == -1
    if "s0_sgsf_upts" not in d:
        return  # level-1 fixture (the SVV case): stage states only
    # pow() of two libm builds in the WALE formula: a few ulps
    assert relerr(c.arr["sgsf_upts"], d["s0_sgsf_upts"]) < 1e-12
    assert relerr(c.arr["sgsf_fpts"], d["s0_sgsf_fpts"]) < 1e-12
    assert relerr(c.arr["tdisf_upts"], d["s0_tdisf_upts"]) < 1e-13
    assert relerr(c.arr["div_tconf_upts"], d["s0_div_tconf_upts"]) < 1e-12


@pytest.mark.parametrize("name", [n for n in ALL if "_plot" in n])
def test_plot_point_interpolation(oracle, name):
    """eles::calc_disu_ppts: the interpolated state at the plot points of every element, from the reference's opp_p."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e = O.Eles()
    e.n_eles, e.n_upts, e.n_fpts, e.n_fields, e.n_dims = [int(v) for v in d["sizes"][:5]]
    u = np.asfortranarray(d["u_init"])
    e.disu_upts[0] = O.fptr(u)
    opp_p = np.asfortranarray(d["opp_p"])
    out = np.zeros(d["disu_ppts"].shape, order="F")
    oracle.orc_calc_disu_ppts(C.byref(e), C.c_int(opp_p.shape[0]), O.fptr(opp_p), O.fptr(out))
    assert relerr(out, d["disu_ppts"]) < RTOL
