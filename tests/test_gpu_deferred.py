"""Deferred execution (include/hfx.h, option "deferred"): the reference's UNCHANGED call sequence -- the seventeen method
calls of CalcResidual (/root/reference/src/solver.cpp:59-221) followed by AdvanceSolution (src/HiFiLES.cpp:201-217), made
one by one through the C ABI -- must give the reference's results whether libhfx runs a recorded stage as one fused stage
or replays it call by call, and must pick the fused stage whenever the record is a whole stage.

Checked against the genuine reference's fixtures (the same ones the per-method and fused tests use, same tolerances):
  * every intermediate of one residual when each call is followed by a download (every download makes the partial record run
    call by call): the replay is the per-method path;
  * the state after every RK stage of every fixture with the calls made exactly as CalcResidual makes them and nothing read in
    between but the state: whole stages run fused (n_fused counts them), on hexes, quads, boundaries, LES closures,
    over-integration + shock capturing, CFL time steps, tetrahedra / prisms / the mixed channel;
  * monitors (residual norms, integral quantities) asked for after a step get the reference's values;
  * an array the fused stage keeps on chip cannot be read once the next stage has begun: the download fails loudly.
"""
import glob
import os

import numpy as np
import pytest

import hfx
from test_gpu_methods_vs_golden import ALL, GOLDEN, RTOL1, RTOLD, RTOLS, build, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    c = hfx.Context(0)
    c.set_option("deferred", 1)
    yield c
    c.close()


def calc_residual_calls(blocks, faces, viscous, rk=0):
    """the calls of CalcResidual in the reference's order (src/solver.cpp:59-221), one C-ABI call each"""
    ints = [f for f in faces if isinstance(f, hfx.IntInters)]
    bdys = [f for f in faces if isinstance(f, hfx.BdyInters)]
    if rk == 0:
        for e in blocks:
            if e.les_model is not None and e.les_model >= 2:
                e.calc_sgs_terms()
    for e in blocks:
        e.extrapolate_solution()
    if viscous:
        for e in blocks:
            e.calculate_gradient()
    for e in blocks:
        e.evaluate_invFlux_over_int() if e.has_over_int else e.evaluate_invFlux()
    for f in ints:
        f.calculate_common_invFlux()
    for f in bdys:
        f.evaluate_boundaryConditions_invFlux()
    if viscous:
        for e in blocks:
            e.correct_gradient()
        for e in blocks:
            e.evaluate_viscFlux()
        for e in blocks:
            if e.les_model is not None:
                e.extrapolate_sgsFlux()
    for e in blocks:
        e.extrapolate_totalFlux()
    for e in blocks:
        e.calculate_divergence()
    if viscous:
        for f in ints:
            f.calculate_common_viscFlux()
        for f in bdys:
            f.evaluate_boundaryConditions_viscFlux()
    for e in blocks:
        e.calculate_corrected_divergence()


def tag(e, d):
    e.les_model = int(np.ravel(d["SGS_model"])[0]) if ("LES" in d and int(np.ravel(d["LES"])[0])) else None
    e.has_over_int = "over_int" in d and bool(int(np.ravel(d["over_int"])[0]))
    e.has_shock = "shock_cap" in d and bool(int(np.ravel(d["shock_cap"])[0]))
    return e


def test_every_intermediate_through_the_replay(ctx):
    """test_every_intermediate of the per-method suite with the context deferring: each download finds a partial record
    and makes it run call by call"""
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_n3_deformed.npz")))
    e, faces = build(ctx, d)
    e.extrapolate_solution()
    assert relerr(e.download(hfx.DISU_FPTS), d["s0_disu_fpts"]) < RTOL1
    e.calculate_gradient()
    e.evaluate_invFlux()
    assert relerr(e.download(hfx.GRAD_DISU_UPTS), d["s0_grad_disu_upts_ref"]) < RTOL1
    assert relerr(e.download(hfx.TDISF_UPTS), d["s0_tdisf_upts_inv"]) < RTOL1
    for f in faces:
        f.calculate_common_invFlux()
    e.correct_gradient()
    assert relerr(e.download(hfx.DELTA_DISU_FPTS), d["s0_delta_disu_fpts"]) < RTOL1
    assert relerr(e.download(hfx.GRAD_DISU_FPTS), d["s0_grad_disu_fpts"]) < RTOL1
    e.evaluate_viscFlux()
    e.extrapolate_totalFlux()
    e.calculate_divergence()
    for f in faces:
        f.calculate_common_viscFlux()
    assert relerr(e.download(hfx.NORM_TCONF_FPTS), d["s0_norm_tconf_fpts"]) < RTOL1
    e.calculate_corrected_divergence()
    # a complete CalcResidual without AdvanceSolution is not a stage: it runs call by call and leaves the reference's arrays
    assert relerr(e.download(hfx.DIV_TCONF_UPTS), d["s0_div_tconf_upts"]) < RTOLD
    assert relerr(e.download(hfx.NORM_TCONF_FPTS), d["s0_norm_tconf_fpts"] - d["s0_norm_tdisf_fpts"]) < RTOL1
    nf, nr, why = ctx.deferred_stats()
    assert nf == 0 and nr >= 5
    for f in faces:
        f.close()
    e.close()


@pytest.mark.parametrize("name", ALL)
def test_stage_states_through_the_unchanged_call_sequence(ctx, name):
    """every fixture of the per-method suite, the calls made exactly as CalcResidual + the RK loop make them; only the state
    is read between stages, so every stage is a whole record and runs fused"""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    e, faces = build(ctx, d)
    tag(e, d)
    nstage = int(d["sizes"][7])
    adv = int(np.ravel(d["adv_type"])[0])
    viscous = bool(int(np.ravel(d["viscous"])[0]))
    steps = sorted({int(k.split("_")[1][4:]) for k in d if k.startswith("u_step")})
    dt_type = int(np.ravel(d["dt_type"])[0])
    if dt_type != 0:
        e.set_h_ref(d["h_ref"])
    n_stages_run = 0
    for st in steps:
        if dt_type != 0:
            dt = e.calc_dt_local(float(np.ravel(d["CFL"])[0]))  # (needs the state: the pending stage runs first)
            want = float(np.ravel(d["dt_step%d" % st])[0])
            assert abs(dt - want) <= 1e-11 * want
            p = hfx.params_from(d)
            p.dt = dt
            ctx.set_params(p)
        for rk in range(nstage):
            calc_residual_calls([e], faces, viscous, rk)
            e.AdvanceSolution(rk, adv)
            if e.has_shock:
                e.shock_capture()
            n_stages_run += 1
            key = "u_step%d_stage%d" % (st, rk)
            if key in d:
                assert relerr(e.download(hfx.DISU_UPTS0), d[key]) < RTOLS, key
        if "bc_flags" in d and d["bc_flags"][1].any():
            for f in faces:
                if isinstance(f, hfx.BdyInters):
                    f.set_ramp_counter(int(np.ravel(d["ramp_counter"])[0]) + st + 1)
    assert e.check_nan() == -1
    nf, nr, why = ctx.deferred_stats()
    assert nf + nr == n_stages_run
    # tensor-product blocks of orders the split stage is built for run fused; simplex blocks through the general stage unless
    # they carry something it refuses
    sz = [int(v) for v in d["sizes"]]
    tensor = sz[6] in (1, 4)
    # (the general stage evaluates an LES closure in its flux kernel and takes the de-aliased flux of the dense over-integration;
    # shock capturing follows it as its own step)
    general_ok = sz[4] == 3
    if tensor or general_ok:
        assert nr == 0, (name, why)
    else:
        assert nf == 0 and why, name
    for f in faces:
        f.close()
    e.close()


def test_monitors_after_a_step(ctx):
    """residual norms (div_tconf_upts of the last stage, stored by the fused stage) and integral quantities (which read the
    corrected gradient: the pending stage runs call by call for them) against the reference, with the calls deferred"""
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_n3_deformed.npz")))
    e, faces = build(ctx, d)
    tag(e, d)
    nstage = int(d["sizes"][7])
    adv = int(np.ravel(d["adv_type"])[0])
    calc_residual_calls([e], faces, True, 0)
    e.AdvanceSolution(0, adv)
    # asked for before the next stage begins: the pending stage stores div_tconf_upts although it is not the step's last
    for fld in range(e.n_fields):
        for nt in (1, 2):
            got = e.compute_res_upts(nt, fld)
            want = d["s0_res_sums"][fld, nt - 1]
            assert abs(got - want) <= 1e-10 * abs(want)
    assert relerr(e.download(hfx.DISU_UPTS0), d["u_step0_stage0"]) < RTOLS
    nf, nr, _ = ctx.deferred_stats()
    assert (nf, nr) == (1, 0)
    # the gradient of that stage was kept on chip: reading it now fails loudly (and hfx_eles_is_current says so beforehand)
    import ctypes as C
    cur = C.c_int(-1)
    hfx.check(hfx.lib().hfx_eles_is_current(e.h, C.c_int(hfx.GRAD_DISU_UPTS), C.byref(cur)))
    assert cur.value == 0
    hfx.check(hfx.lib().hfx_eles_is_current(e.h, C.c_int(hfx.DISU_UPTS0), C.byref(cur)))
    assert cur.value == 1
    with pytest.raises(hfx.HfxError, match="not materialised"):
        e.download(hfx.GRAD_DISU_UPTS)
    # ... whereas a stage that is still pending when the gradient is asked for runs call by call and leaves it
    calc_residual_calls([e], faces, True, 1)
    e.AdvanceSolution(1, adv)
    hfx.check(hfx.lib().hfx_eles_is_current(e.h, C.c_int(hfx.GRAD_DISU_UPTS), C.byref(cur)))
    assert cur.value == 2  # (a whole stage is pending: asking for the gradient makes it run call by call)
    g = e.download(hfx.GRAD_DISU_UPTS)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    nf, nr, why = ctx.deferred_stats()
    assert (nf, nr) == (1, 1) and "on chip" in why
    assert relerr(e.download(hfx.DISU_UPTS0), d["u_step0_stage1"]) < RTOLS
    for f in faces:
        f.close()
    e.close()


def test_out_of_order_calls_are_replayed(ctx):
    """a caller that does not follow CalcResidual's order (here: the divergence before the flux extrapolation) gets exactly
    what the calls do one by one"""
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_n3_deformed.npz")))
    e, faces = build(ctx, d)
    adv = int(np.ravel(d["adv_type"])[0])
    e.extrapolate_solution()
    e.calculate_gradient()
    e.evaluate_invFlux()
    for f in faces:
        f.calculate_common_invFlux()
    e.correct_gradient()
    e.evaluate_viscFlux()
    e.calculate_divergence()      # swapped with the next call
    e.extrapolate_totalFlux()
    for f in faces:
        f.calculate_common_viscFlux()
    e.calculate_corrected_divergence()
    e.AdvanceSolution(0, adv)
    assert relerr(e.download(hfx.DISU_UPTS0), d["u_step0_stage0"]) < RTOLS
    nf, nr, why = ctx.deferred_stats()
    assert (nf, nr) == (0, 1) and "extrapolate_totalFlux is missing" in why
    for f in faces:
        f.close()
    e.close()


MIXED = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "mixed_*.npz")))


@pytest.mark.parametrize("name", MIXED)
def test_mixed_channel_through_the_unchanged_call_sequence(name):
    """BASELINE.json configs[3]: tetrahedra + prisms + walls, every method over both classes as src/solver.cpp:59-221 loops
    them; whole stages run as the general fused stage"""
    from test_mixed_mesh import build_gpu
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ctx = hfx.Context(0)
    ctx.set_option("deferred", 1)
    classes, E, F = build_gpu(ctx, d)
    blocks = [E[c] for c in classes]
    for e in blocks:
        e.les_model, e.has_over_int, e.has_shock = None, False, False
    nstage = int(d["c2_sizes"][7])
    adv = int(np.ravel(d["adv_type"])[0])
    steps = sorted({int(k.split("_")[2][4:]) for k in d if k.startswith("c2_u_step")})
    n = 0
    for st in steps:
        for rk in range(nstage):
            calc_residual_calls(blocks, F, True, rk)
            for e in blocks:
                e.AdvanceSolution(rk, adv)
            n += 1
            for c in classes:
                k = "c%d_u_step%d_stage%d" % (c, st, rk)
                if k in d:
                    assert relerr(E[c].download(hfx.DISU_UPTS0), d[k]) < 1e-11, k
    nf, nr, why = ctx.deferred_stats()
    assert (nf, nr) == (n, 0), why
    for f in F:
        f.close()
    for c in classes:
        E[c].close()
    ctx.close()


def test_host_mirror_runs_its_unchanged_loop_fused():
    """the mirrored CalcResidual + AdvanceSolution loop (csrc/host/solver.cpp, the reference's calls through the mirrored
    classes) defers by default: 40 steps of the genuine reference (hex_p2_long), every stage but the one pending at each
    sync_host -- which asks for the gradient first, as the reference's CopyGPUCPU does -- runs fused"""
    from test_gpu_host_mirror import fixture_case, rel
    c, d = fixture_case("hex_p2_long")
    c.to_device(0)
    ctx = c.handles()[0]
    for last in (9, 19, 29, 39):
        c.run(10)
        c.sync_host()
        assert rel(c.array("disu_upts0"), d["u_step%d_stage4" % last]) < 1e-11, last
    nf, nr, why = hfx.deferred_stats(ctx)
    assert (nf, nr) == (4 * 49, 4), why
    # and with the option off every call launches: same trajectory
    c2, _ = fixture_case("hex_p2_long")
    c2.set_deferred(False)
    c2.to_device(0)
    c2.run(10)
    c2.sync_host()
    assert rel(c2.array("disu_upts0"), d["u_step9_stage4"]) < 1e-11
    assert hfx.deferred_stats(c2.handles()[0])[:2] == (0, 0)
    c.close()
    c2.close()
