"""The split fused stage (hfx_run_steps(..., fused=2 | 3)) against the genuine reference's fixtures,
against the per-method path, and through the full-size properties."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import hfx
import hfx_host as H
from test_gpu_methods_vs_golden import build, relerr, ALL, GOLDEN
from test_gpu_host_mirror import integrals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hfx.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("name", ALL)
def test_fused_vs_reference(ctx, name, mode):
    """mode 2: split fused kernels (the reference's arrays kept), mode 3: split, fluxes in the gradient kernel"""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d)
    nstage = int(d["sizes"][7])
    steps = sorted({int(k.split("_")[1][4:]) for k in d if k.startswith("u_step")})
    if "cfl" in name:
        # calc_time_step at the top of every step inside hfx_run_steps (src/HiFiLES.cpp:198)
        e.set_h_ref(d["h_ref"])
        ctx.set_CFL(float(np.ravel(d["CFL"])[0]))
    # what a fused path does not do, it refuses (over-integration: mode 3 only -- the de-aliased inviscid flux is
    # evaluated ahead of the flux kernel, which then takes it instead of computing the collocated one)
    over_int = "over_int" in d and int(np.ravel(d["over_int"])[0]) != 0
    # LES closures need the corrected gradients in HBM: a block with a closure runs the kernels of mode 2 whichever is asked for
    if (over_int and mode != 3) or name.startswith(("tet_", "pri_")):
        with pytest.raises(hfx.HfxError):
            hfx.run_steps(e, faces, 1, fused=mode)
        steps = []
    for st in steps:
        hfx.run_steps(e, faces, 1, fused=mode)
        assert relerr(e.download(hfx.DISU_UPTS0), d["u_step%d_stage%d" % (st, nstage - 1)]) < 1e-11, st
    assert e.check_nan() == -1
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_fused_public_arrays_after_a_step(ctx, mode):
    """What the fused paths leave in the public arrays: state, RK register, flux-point solution of the
    NEW state, corrected gradients and the divergence of the step's last stage (for the monitors)."""
    d = dict(np.load(os.path.join(GOLDEN, "hex_p3_n3_deformed.npz")))
    ef, ff = build(ctx, d)
    em, fm = build(ctx, d)
    hfx.run_steps(ef, ff, 1, fused=mode)
    hfx.run_steps(em, fm, 1, fused=False)
    assert relerr(ef.download(hfx.DISU_UPTS0), em.download(hfx.DISU_UPTS0)) < 1e-12
    assert relerr(ef.download(hfx.DISU_UPTS1), em.download(hfx.DISU_UPTS1)) < 1e-9
    assert relerr(ef.download(hfx.DIV_TCONF_UPTS), em.download(hfx.DIV_TCONF_UPTS)) < 5e-11
    if mode != 3:  # mode 3 keeps the corrected gradients in registers (include/hfx.h, hfx_run_steps)
        assert relerr(ef.download(hfx.GRAD_DISU_UPTS), em.download(hfx.GRAD_DISU_UPTS)) < 1e-11
        assert relerr(ef.download(hfx.GRAD_DISU_FPTS), em.download(hfx.GRAD_DISU_FPTS)) < 1e-11
    em.extrapolate_solution()
    assert relerr(ef.download(hfx.DISU_FPTS), em.download(hfx.DISU_FPTS)) < 1e-13
    # mixing the paths: a per-method stage after fused steps sees a consistent state
    hfx.run_steps(ef, ff, 1, fused=False)
    hfx.run_steps(em, fm, 1, fused=mode)
    assert relerr(ef.download(hfx.DISU_UPTS0), em.download(hfx.DISU_UPTS0)) < 1e-12
    for f in ff + fm:
        f.close()
    ef.close(); em.close()


def test_fused_nan_flag(ctx):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p1_rusanov.npz")))
    e, faces = build(ctx, d)
    u = np.array(d["u_init"], order="F")
    u[3, 5, 0] = np.nan
    e.upload(hfx.DISU_UPTS0, u)
    hfx.run_steps(e, faces, 1, fused=3)
    assert e.check_nan() >= 0
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_fused_quads_vs_methods(mode):
    """2-D tensor-product elements (BASELINE.json configs[0]'s element type) through both paths."""
    a = H.Case([6, 5, 1], dims=2, order=3, amp=0.1)
    b = H.Case([6, 5, 1], dims=2, order=3, amp=0.1)
    a.to_device(0); b.to_device(0)
    a.run_steps_lib(2, fused=mode)
    b.run_steps_lib(2, fused=False)
    a.sync_host(); b.sync_host()
    assert relerr(a.array("disu_upts0"), b.array("disu_upts0")) < 1e-12
    assert relerr(a.array("disu_upts0"), H.Case([6, 5, 1], dims=2, order=3, amp=0.1).array("disu_upts0")) > 1e-8
    a.close(); b.close()


@pytest.mark.parametrize("order", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("dims", [2, 3])
def test_split_paths_every_order_vs_methods(dims, order):
    """Every instantiated element size (P1..P5 on hexes, P1..P7 on quads) of the split fused kernels -- with and without the
    loader wave, whichever the size selects -- against the per-method path on a small deformed periodic mesh."""
    if dims == 3 and order > 5:
        pytest.skip("hexes above P5 have more than 256 points per element: they run call by call")
    n = [4, 3, 1] if dims == 2 else [3, 3, 3]
    ref = H.Case(n, dims=dims, order=order, amp=0.1)
    ref.to_device(0)
    ref.run_steps_lib(1, fused=False)
    ref.sync_host()
    want = ref.array("disu_upts0").copy()
    ref.close()
    for mode in (2, 3):
        c = H.Case(n, dims=dims, order=order, amp=0.1)
        c.to_device(0)
        c.run_steps_lib(1, fused=mode)
        c.sync_host()
        assert relerr(c.array("disu_upts0"), want) < 1e-12, (mode, order)
        c.close()


@pytest.mark.parametrize("knob,value,exact", [("simd_roles", 0, True), ("light_wave_short", 0, False), ("gather_delta", 0, False), ("loader_wave", 0, False), ("dictionary_rows", 1, False),
                                              ("buffer_addressing", 0, False), ("xcd_order", 0, True), ("split_grid_per_cu", 2, True), ("split_grid_per_cu", 16, True)])
def test_split3_variant_knobs_agree(knob, value, exact):
    """hfx_ctx_set_option selects between forms of the split3 kernels (wave parts dealt by SIMD or by wave number, loader
    wave or register pipeline, sum-factorised or dictionary rows, buffer or flat addressing, element order, grid size).
    Forms that only move work between waves or change addressing give the SAME bits; the others the same state to rounding."""
    n = [4, 4, 4]
    ref = H.Case(n, order=4, amp=0.1)
    ref.to_device(0)
    ref.run_steps_lib(2, fused=3)
    ref.sync_host()
    want = ref.array("disu_upts0").copy()
    ref.close()
    c = H.Case(n, order=4, amp=0.1)
    c.to_device(0)
    hfx.Context.set_option(_Ctx(c.handles()[0]), knob, value)
    c.run_steps_lib(2, fused=3)
    c.sync_host()
    got = c.array("disu_upts0")
    if exact:
        assert np.array_equal(got, want), knob
    else:
        assert relerr(got, want) < 1e-12, knob
    c.close()


class _Ctx:
    """a bare handle with hfx.Context's methods (the host mirror owns the context)"""
    def __init__(self, h):
        self.h = h


@pytest.mark.parametrize("mode", [2, 3])
def test_fused_full_size_conservation(mode):
    c = H.Case(32, order=4)
    u0 = c.array("disu_upts0")
    i0 = integrals(c, u0)
    c.to_device(0)
    c.run_steps_lib(2, fused=mode)
    c.sync_host()
    u = c.array("disu_upts0")
    assert np.isfinite(u).all()
    i1 = integrals(c, u)
    vol = (2 * np.pi) ** 3
    scale = np.array([1.0, 1.0, 1.0, 1.0, u0[:, :, 4].max()]) * vol
    assert np.all(np.abs(i1 - i0) / scale < 1e-12), (i1 - i0) / scale
    # and the two paths agree at full size
    m = H.Case(32, order=4)
    m.to_device(0)
    m.run_steps_lib(2, fused=False)
    m.sync_host()
    assert relerr(u, m.array("disu_upts0")) < 1e-12
    c.close(); m.close()


@pytest.mark.parametrize("dims,order,viscous", [(3, 4, 1), (3, 2, 1), (2, 3, 1), (2, 3, 0)])
def test_over_integration_folded_into_the_divergence(dims, order, viscous):
    """The sum-factorised over-integration kernel hands the loader-wave flux kernel sum_l Dc[l] tdisf_l (the de-aliased flux's
    contribution to div_tdisf - opp_3 norm_tdisf, n_fields values per point) instead of tdisf_upts (option over_int_fold, default
    1): the same state as with the whole flux handed over, and as the per-method path, to rounding (a re-association)."""
    n = [4] * dims + [1] * (3 - dims)
    kw = dict(dims=dims, order=order, amp=0.1, viscous=viscous, over_int=1, over_int_order=order + 2)
    if not viscous:  # the isentropic vortex of BASELINE.json configs[0] (Euler, Rusanov)
        n = [6, 6, 1]
        kw.update(ic_form=0, riemann_solve_type=0, dt=0.001, rho_c_ic=1.0, u_c_ic=1.0, v_c_ic=1.0, p_c_ic=1.0)
    got = {}
    for fold in (1, 0, None):
        c = H.Case(n, **kw)
        if fold is None:
            c.set_deferred(False)
        c.to_device(0)
        if fold is None:
            c.run(2)
        else:
            hfx.Context.set_option(_Ctx(c.handles()[0]), "over_int_fold", fold)
            c.run_steps_lib(2, fused=3)
        c.sync_host()
        got[fold] = c.array("disu_upts0").copy()
        c.close()
    assert relerr(got[1], got[0]) < 1e-12
    assert relerr(got[1], got[None]) < 1e-12
    assert np.isfinite(got[1]).all()


def test_config5_ingredients_full_size_properties():
    """BASELINE.json configs[4]'s ingredients at bench size (tools/bench_config5.sh, the `config5_overint_shock` leg of bench.py):
    32^3 P4 hexes with over-integration (7 cubature points per direction) and shock capturing after every stage.  The split
    fused stage (sum-factorised over-integration kernel, de-aliased flux into the flux kernel, shock filter + flux-point
    refresh behind the update kernel) equals the per-method path on every element, both conserve the five integrals (the
    projection and the modal filter leave element means alone), and the sensor filters elements in both alike."""
    kw = dict(over_int=1, over_int_order=6, shock_cap=1, s0=1e-3, expf_fac=36.0, expf_order=4, expf_cutoff=1, shock_det_field=0)
    c = H.Case(32, order=4, **kw)
    u0 = c.array("disu_upts0").copy()
    i0 = integrals(c, u0)
    c.to_device(0)
    c.run_steps_lib(2, fused=3)
    c.sync_host()
    u = c.array("disu_upts0").copy()
    assert np.isfinite(u).all()
    vol = (2 * np.pi) ** 3
    scale = np.array([1.0, 1.0, 1.0, 1.0, u0[:, :, 4].max()]) * vol
    assert np.all(np.abs(integrals(c, u) - i0) / scale < 1e-11)
    m = H.Case(32, order=4, **kw)
    m.set_deferred(False)
    m.to_device(0)
    m.run(2)  # the mirrored CalcResidual (evaluate_invFlux_over_int) + AdvanceSolution + shock_capture, call by call
    m.sync_host()
    assert relerr(u, m.array("disu_upts0")) < 1e-11
    sens = [np.zeros(c.n_eles), np.zeros(c.n_eles)]
    for k, case in enumerate((c, m)):
        hfx.check(hfx.lib().hfx_eles_download(case.handles()[1], C.c_int(hfx.SENSOR), sens[k].ctypes.data_as(hfx.dp)))
    assert np.array_equal(sens[0] >= 1e-3, sens[1] >= 1e-3)
    assert relerr(sens[0], sens[1]) < 1e-7
    c.close(); m.close()


def test_les_full_size_closure_in_the_flux_kernel():
    """32^3 P4 hexes, LES with the WALE closure (the `les_wale` leg of bench.py): split variant 3 with the closure evaluated in
    the flux kernel, variant 2 (pointwise closure kernel on the gradient array) and the per-method path agree on every
    element after two steps, through hfx_run_steps and through the deferred mirrored loop; all conserve."""
    kw = dict(LES=1, SGS_model=1, C_s=0.325, filter_ratio=1.0)
    res = {}
    for mode in (3, 2, "calls", "deferred"):
        c = H.Case(32, order=4, **kw)
        if mode == "calls":
            c.set_deferred(False)
        u0 = c.array("disu_upts0").copy()
        i0 = integrals(c, u0)
        c.to_device(0)
        if mode in ("calls", "deferred"):
            c.run(2)
        else:
            c.run_steps_lib(2, fused=mode)
        # (the state alone is read: the stage still pending in the deferred loop runs fused for it)
        u = np.zeros(u0.shape, order="F")
        hfx.check(hfx.lib().hfx_eles_download(c.handles()[1], C.c_int(hfx.DISU_UPTS0), u.ctypes.data_as(hfx.dp)))
        if mode == "deferred":
            nf, nr, why = hfx.deferred_stats(c.handles()[0])
        res[mode] = u
        vol = (2 * np.pi) ** 3
        scale = np.array([1.0, 1.0, 1.0, 1.0, u0[:, :, 4].max()]) * vol
        assert np.all(np.abs(integrals(c, res[mode]) - i0) / scale < 1e-12), mode
        c.close()
    assert (nf, nr) == (10, 0), why
    for mode in (3, 2, "deferred"):
        assert relerr(res[mode], res["calls"]) < 1e-11, mode


@pytest.mark.parametrize("mode", [2, 3])
def test_fused_full_size_residual_norms_vs_reference_stdout(mode):
    """the reference's own iteration-1 row for the 32^3 P4 TGV case."""
    c = H.Case(32, order=4)
    c.to_device(0)
    c.run_steps_lib(1, fused=mode)
    ctx_, e, f, nb = c.handles()
    v = C.c_double()
    r = []
    for fld in range(5):
        hfx.check(hfx.lib().hfx_eles_compute_res_upts(e, 1, fld, C.byref(v)))
        r.append(v.value / (c.n_eles * c.n_upts))
    c.close()
    want = np.array([0.00070019, 0.05031596, 0.05031596, 0.06433493, 0.11799063])
    assert np.all(np.abs(np.array(r) - want) < 6e-9), r


def test_fused_refuses_what_it_cannot_do(ctx):
    d = dict(np.load(os.path.join(GOLDEN, "hex_p1_rusanov.npz")))
    e, faces = build(ctx, d)
    with pytest.raises(hfx.HfxError):
        hfx.run_steps(e, [], 1, fused=3)  # flux points without partner
    with pytest.raises(hfx.HfxError):
        hfx.run_steps(e, faces, 1, fused=1)  # the gather-style variant has been retired
    for f in faces:
        f.close()
    e.close()
