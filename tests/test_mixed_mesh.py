"""BASELINE.json configs[3]: a MIXED mesh -- a channel with a layer of triangular prisms on either wall and tetrahedra in
the core (oracle/gen_neu_mesh.py write_neu_mixed), isothermal wall below, adiabatic wall above, periodic in x and z.
Fixtures mixed_p3_channel (the configuration's order; every stage state) and mixed_p2_channel (every intermediate of
one residual) come from the GENUINE reference.  What is new against the single-class fixtures: interior face blocks
whose left and right sides belong to DIFFERENT element classes (/root/reference/src/int_inters.cpp:67-121, wired per
(ctype(ic_l), ctype(ic_r)) in src/geometry.cpp:637-706) and CalcResidual over several element blocks
(src/solver.cpp:50-223 loops every method over the classes)."""
import json
import os

import numpy as np
import pytest

import mixed_util as MU

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MIXED = ["mixed_p2_channel", "mixed_p3_channel"]
INTERMEDIATES = {  # hook name -> (fixture key, oracle array)
    "disu_fpts": ("s0_disu_fpts", "disu_fpts"), "grad_disu_upts_ref": ("s0_grad_disu_upts_ref", "grad_disu_upts"),
    "tdisf_upts_inv": ("s0_tdisf_upts_inv", "tdisf_upts"), "norm_tconf_fpts_inv": ("s0_norm_tconf_fpts_inv", "norm_tconf_fpts"),
    "grad_disu_fpts": ("s0_grad_disu_fpts", "grad_disu_fpts"), "tdisf_upts": ("s0_tdisf_upts", "tdisf_upts"),
    "norm_tdisf_fpts": ("s0_norm_tdisf_fpts", "norm_tdisf_fpts"), "div_tconf_upts_disc": ("s0_div_tconf_upts_disc", "div_tconf_upts"),
    "norm_tconf_fpts": ("s0_norm_tconf_fpts", "norm_tconf_fpts"), "div_tconf_upts": ("s0_div_tconf_upts", "div_tconf_upts"),
}


def relerr(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def test_fixture_has_faces_between_classes():
    d = dict(np.load(os.path.join(GOLDEN, "mixed_p3_channel.npz")))
    classes, per, faces, bdy = MU.split(d)
    assert classes == [2, 3]
    pairs = {(a, b) for a, b, _, _ in faces}
    assert (2, 3) in pairs and (3, 2) in pairs and (2, 2) in pairs and (3, 3) in pairs  # left != right exists
    assert {a for a, _, _ in bdy} == {3}  # the walls sit on prisms


@pytest.mark.parametrize("name", MIXED)
def test_oracle_on_mixed_mesh_vs_reference(name):
    """the oracle with two element blocks and class-crossing face blocks reproduces the reference: every intermediate of
    the first residual (level-2 fixture) and the state after every RK stage"""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    m = MU.MixedOracle(d)
    nstage = int(d["c2_sizes"][7])
    steps = sorted({int(k.split("_")[2][4:]) for k in d if k.startswith("c2_u_step")})
    first = [True]

    def hook(what):
        key, arr = INTERMEDIATES[what]
        for c in m.classes:
            k = "c%d_%s" % (c, key)
            if first[0] and k in d:
                assert relerr(m.arr(c, arr), d[k]) < 1e-13, k

    for st in steps:
        for rk in range(nstage):
            assert m.CalcResidual(hook) == -1
            first[0] = False
            m.AdvanceSolution(rk)
            for c in m.classes:
                k = "c%d_u_step%d_stage%d" % (c, st, rk)
                if k in d:
                    assert relerr(m.arr(c, "u0"), d[k]) < 1e-13, k


# ---- the device path, through the C ABI -----------------------------------------------------------------------------
def build_gpu(ctx, d):
    """-> ({cls: hfx.Eles}, [face blocks]): one element block per class, one interior block per (left, right) class pair,
    one boundary block per class"""
    import hfx
    classes, per, faces, bdy = MU.split(d)
    ctx.set_params(hfx.params_from(per[classes[0]]))
    ctx.set_contract_mode(hfx.CONTRACT_AUTO)
    E = {}
    for c in classes:
        sz = [int(v) for v in per[c]["sizes"]]
        E[c] = hfx.Eles(ctx, sz[:5], per[c], ele_type=sz[6], order=sz[5])
        E[c].upload(hfx.DISU_UPTS0, per[c]["u_init"])
    F = [hfx.IntInters(ctx, E[a], E[b], L, R) for a, b, L, R in faces]
    for a, L, ids in bdy:
        F.append(hfx.BdyInters(ctx, E[a], L, ids, hfx.bc_records(d["bc_flags"], d["bc_params"]), float(np.ravel(d["bc_R_ref"])[0]),
                               int(np.ravel(d["ramp_counter"])[0])))
    return classes, E, F


GPU_ARRAYS = {"s0_disu_fpts": "DISU_FPTS", "s0_tdisf_upts": "TDISF_UPTS", "s0_norm_tdisf_fpts": "NORM_TDISF_FPTS",
              "s0_delta_disu_fpts": "DELTA_DISU_FPTS", "s0_grad_disu_upts": "GRAD_DISU_UPTS",
              "s0_grad_disu_fpts": "GRAD_DISU_FPTS", "s0_div_tconf_upts": "DIV_TCONF_UPTS"}


@pytest.mark.gpu
@pytest.mark.parametrize("name", MIXED)
def test_stage_states_vs_reference(name):
    """hfx_CalcResidual_blocks + AdvanceSolution per block on the mixed channel: what one residual leaves in the public
    arrays of BOTH classes (level-2 fixture) and the state after every RK stage, against the genuine reference"""
    import hfx
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ctx = hfx.Context(0)
    classes, E, F = build_gpu(ctx, d)
    blocks = [E[c] for c in classes]
    nstage = int(d["c2_sizes"][7])
    adv = int(np.ravel(d["adv_type"])[0])
    steps = sorted({int(k.split("_")[2][4:]) for k in d if k.startswith("c2_u_step")})
    for st in steps:
        for rk in range(nstage):
            hfx.CalcResidual_blocks(blocks, F)
            if st == 0 and rk == 0:
                for key, arr in GPU_ARRAYS.items():
                    for c in classes:
                        k = "c%d_%s" % (c, key)
                        if k in d:
                            tol = 5e-11 if "div" in key else 1e-12
                            assert relerr(E[c].download(getattr(hfx, arr)), d[k]) < tol, k
                    for c in classes:  # calculate_corrected_divergence leaves norm_tconf - norm_tdisf in place (src/eles.cpp:1746)
                        k = "c%d_s0_norm_tconf_fpts" % c
                        if k in d:
                            assert relerr(E[c].download(hfx.NORM_TCONF_FPTS), d[k] - d["c%d_s0_norm_tdisf_fpts" % c]) < 1e-11, k
            for c in classes:
                E[c].AdvanceSolution(rk, adv)
            for c in classes:
                k = "c%d_u_step%d_stage%d" % (c, st, rk)
                if k in d:
                    assert relerr(E[c].download(hfx.DISU_UPTS0), d[k]) < 1e-11, k
    for c in classes:
        assert E[c].check_nan() == -1
    for f in F:
        f.close()
    for c in classes:
        E[c].close()
    ctx.close()


@pytest.mark.gpu
def test_run_steps_blocks_vs_reference():
    """the whole RK loop over both element blocks inside the library (hfx_run_steps_blocks, per-method path)"""
    import hfx
    d = dict(np.load(os.path.join(GOLDEN, "mixed_p3_channel.npz")))
    ctx = hfx.Context(0)
    classes, E, F = build_gpu(ctx, d)
    hfx.run_steps_blocks([E[c] for c in classes], F, 1, fused=0)
    nstage = int(d["c2_sizes"][7])
    for c in classes:
        assert relerr(E[c].download(hfx.DISU_UPTS0), d["c%d_u_step0_stage%d" % (c, nstage - 1)]) < 1e-11
    for f in F:
        f.close()
    for c in classes:
        E[c].close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", MIXED)
def test_general_fused_stage_on_mixed_mesh_vs_reference(name):
    """hfx_run_steps_blocks(..., fused = 4): the fused stage for general element classes (csrc/general.hip: four launches
    per block and stage, FP64-MFMA contractions over batches of 16 elements) on the mixed channel -- prism | tetrahedron
    face pairs, walls on the prisms -- against the genuine reference after every time step"""
    import hfx
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ctx = hfx.Context(0)
    classes, E, F = build_gpu(ctx, d)
    nstage = int(d["c2_sizes"][7])
    steps = sorted({int(k.split("_")[2][4:]) for k in d if k.startswith("c2_u_step")})
    for st in steps:
        hfx.run_steps_blocks([E[c] for c in classes], F, 1, fused=4)
        for c in classes:
            assert relerr(E[c].download(hfx.DISU_UPTS0), d["c%d_u_step%d_stage%d" % (c, st, nstage - 1)]) < 1e-11, (c, st)
    for c in classes:
        assert E[c].check_nan() == -1
    for f in F:
        f.close()
    for c in classes:
        E[c].close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tet_p2_n2_deformed", "tet_p3_n2_deformed", "pri_p2_n2_deformed", "pri_p3_n2_deformed",
                                  "tet_p2_vcjh_sd", "tet_p3_vcjh_cplus", "tet_p2_vcjh_c", "pri_p2_vcjh_hu", "tet_p2_curved", "pri_p2_curved",
                                  "tet_p3_shock", "pri_p2_shock", "tet_p2_les_wale", "pri_p2_les_wale", "tet_p3_les_wsm", "tet_p3_les_sim",
                                  "tet_p3_les_svv", "tet_p2_overint", "pri_p2_overint"])
def test_general_fused_stage_single_class_vs_reference(name):
    """the same on the periodic single-class tetrahedron and prism fixtures (a last batch of fewer than 16 elements
    included), and what it leaves in the public arrays against the per-method path"""
    import hfx
    from test_gpu_methods_vs_golden import build
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ctx = hfx.Context(0)
    e, faces = build(ctx, d)
    m, mfaces = build(ctx, d)
    nstage = int(d["sizes"][7])
    hfx.run_steps(e, faces, 1, fused=4)
    assert relerr(e.download(hfx.DISU_UPTS0), d["u_step0_stage%d" % (nstage - 1)]) < 1e-11
    hfx.run_steps(m, mfaces, 1, fused=0)
    assert relerr(e.download(hfx.DISU_UPTS1), m.download(hfx.DISU_UPTS1)) < 1e-9
    assert relerr(e.download(hfx.DIV_TCONF_UPTS), m.download(hfx.DIV_TCONF_UPTS)) < 5e-10
    m.extrapolate_solution()
    assert relerr(e.download(hfx.DISU_FPTS), m.download(hfx.DISU_FPTS)) < 1e-12
    # a second step from the fused state, against the per-method path
    hfx.run_steps(e, faces, 1, fused=4)
    hfx.run_steps(m, mfaces, 1, fused=0)
    assert relerr(e.download(hfx.DISU_UPTS0), m.download(hfx.DISU_UPTS0)) < 1e-11
    for f in faces + mfaces:
        f.close()
    e.close(); m.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [0, 4])
def test_mixed_channel_from_the_host_mirrors_own_operators(fused):
    """BASELINE.json configs[3] with NOTHING numeric lifted from the reference but the mesh: operators and metrics of both
    classes come from the host mirror's eles_tets / eles_pris (csrc/host/eles_simplex.cpp) built from the shape nodes, the
    face tables from the mesh preprocessor's output (out of scope); the result still equals the genuine
    reference's after every stage of a time step"""
    import hfx
    import hfx_host as H
    d = dict(np.load(os.path.join(GOLDEN, "mixed_p3_channel.npz")))
    classes, per, faces, bdy = MU.split(d)
    for c in classes:
        sz = [int(v) for v in per[c]["sizes"]]
        x1 = per[c]["loc_upts"][2, ::(sz[5] + 1) * (sz[5] + 2) // 2] if sz[6] == 3 else None
        S = H.Simplex(sz[6], sz[5], per[c]["shape"][:, :(4 if sz[6] == 2 else 6), :], viscous=1, loc_1d_upts=x1)
        for k in ["opp_0", "opp_3", "opp_6", "detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts"] + \
                 ["opp_%d_%d" % (w, dd) for w in (1, 2, 4, 5) for dd in range(3)]:
            d["c%d_%s" % (c, k)] = S.array(k)
        S.close()
    ctx = hfx.Context(0)
    classes, E, F = build_gpu(ctx, d)
    nstage = int(d["c2_sizes"][7])
    hfx.run_steps_blocks([E[c] for c in classes], F, 1, fused=fused)
    for c in classes:
        assert relerr(E[c].download(hfx.DISU_UPTS0), d["c%d_u_step0_stage%d" % (c, nstage - 1)]) < 1e-11, c
    for f in F:
        f.close()
    for c in classes:
        E[c].close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tet_p2_les_wale", "pri_p2_les_wale", "tet_p3_les_wsm", "tet_p3_les_sim", "tet_p3_les_svv"])
def test_simplex_les_from_the_host_mirrors_own_operators(name):
    """LES on tetrahedra and prisms (src/eles.cpp:2395 with the class's calc_ele_vol; src/eles_tets.cpp:576 for the filter of the
    closures that filter the solution): operators, metrics, Jacobian_fpts and filter_upts from the host mirror's eles_tets /
    eles_pris, the reference's call sequence through the per-method entry points, call by call and with deferred execution (the
    general fused stage evaluates the closure in its flux kernel), states against the genuine reference after every step"""
    import hfx
    import hfx_host as H
    from test_gpu_methods_vs_golden import build
    from test_gpu_deferred import calc_residual_calls, tag
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    k = json.loads(bytes(d["meta_json"]).decode())["keys"]
    sz = [int(v) for v in d["sizes"]]
    x1 = d["loc_upts"][2, ::(sz[5] + 1) * (sz[5] + 2) // 2] if sz[6] == 3 else None
    S = H.Simplex(sz[6], sz[5], d["shape"][:, :(4 if sz[6] == 2 else 6), :], viscous=1, loc_1d_upts=x1, SGS_model=k["SGS_model"],
                  filter_type=k.get("filter_type", 0), filter_ratio=k["filter_ratio"])
    names = ["opp_0", "opp_3", "opp_6", "detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts", "Jacobian_fpts"] + \
            ["opp_%d_%d" % (w, dd) for w in (1, 2, 4, 5) for dd in range(3)] + (["filter_upts"] if k["SGS_model"] >= 2 else [])
    for n in names:
        a = S.array(n)
        assert a.shape == d[n].shape, n
        d[n] = a
    S.close()
    nstage = int(d["sizes"][7])
    steps = sorted({int(q.split("_")[1][4:]) for q in d if q.startswith("u_step")})
    for deferred in (0, 1):
        ctx = hfx.Context(0)
        ctx.set_option("deferred", deferred)
        e, faces = build(ctx, d)
        tag(e, d)
        adv = int(np.ravel(d["adv_type"])[0])
        for st in steps:
            for rk in range(nstage):
                calc_residual_calls([e], faces, True, rk)
                e.AdvanceSolution(rk, adv)
            assert relerr(e.download(hfx.DISU_UPTS0), d["u_step%d_stage%d" % (st, nstage - 1)]) < 1e-11, (deferred, st)
        nf, nr, why = ctx.deferred_stats()
        assert (nf > 0 and nr == 0) if deferred else (nf == 0), (deferred, nf, nr, why)
        for f in faces:
            f.close()
        e.close()
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tet_p3_shock", "pri_p2_shock"])
def test_simplex_shock_capturing_from_the_host_mirrors_own_operators(name):
    """shock capturing on tetrahedra and prisms (src/eles.cpp:2918 with the class's sensor, src/eles_tets.cpp:748, src/eles_pris.cpp:
    678): every operator and metric, the modal matrices, the filter and the sensor's mode set from the host mirror; the reference's
    call sequence with shock_capture after every AdvanceSolution; sensor and states against the genuine reference"""
    import hfx
    import hfx_host as H
    from test_gpu_methods_vs_golden import build
    from test_gpu_deferred import calc_residual_calls, tag
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    k = json.loads(bytes(d["meta_json"]).decode())["keys"]
    sz = [int(v) for v in d["sizes"]]
    x1 = d["loc_upts"][2, ::(sz[5] + 1) * (sz[5] + 2) // 2] if sz[6] == 3 else None
    S = H.Simplex(sz[6], sz[5], d["shape"], viscous=1, loc_1d_upts=x1, shock_cap=1, expf_fac=k["expf_fac"],
                  expf_order=k["expf_order"], expf_cutoff=k["expf_cutoff"])
    names = ["opp_0", "opp_3", "opp_6", "detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts",
             "inv_vandermonde", "exp_filter"] + ["opp_%d_%d" % (w, dd) for w in (1, 2, 4, 5) for dd in range(3)]
    for n in names:
        a = S.array(n)
        assert a.shape == d[n].shape, n
        d[n] = a
    d["norm_basis_persson"] = S.array("norm_basis_persson")
    d["persson_high_modes"] = S.array("persson_high_modes").astype(np.int32)
    S.close()
    nstage = int(d["sizes"][7])
    adv = int(np.ravel(d["adv_type"])[0])
    steps = sorted({int(q.split("_")[1][4:]) for q in d if q.startswith("u_step")})
    ctx = hfx.Context(0)
    ctx.set_option("deferred", 1)
    e, faces = build(ctx, d)
    tag(e, d)
    for st in steps:
        for rk in range(nstage):
            calc_residual_calls([e], faces, True, rk)
            e.AdvanceSolution(rk, adv)
            e.shock_capture()
            if st == 0 and rk == 0:
                assert relerr(e.download(hfx.SENSOR), np.ravel(d["s0_sensor"])) < 1e-10
            key = "u_step%d_stage%d" % (st, rk)
            if key in d:
                assert relerr(e.download(hfx.DISU_UPTS0), d[key]) < 1e-11, key
    for f in faces:
        f.close()
    e.close()
    ctx.close()


@pytest.mark.gpu
def test_general_fused_stage_at_bench_size_properties():
    """BASELINE.json configs[3] at the size bench.py --workload mixed runs (the reference's mixed channel tiled; here 512
    tiles, every tile with its OWN state: the conserved variables scaled by a tile-dependent factor, which keeps the
    velocity field and scales density and pressure).  Size-independent properties: (i) the general fused stage and the
    per-method path agree on every element after the fixture's steps; (ii) tile 0, whose state is the fixture's, still equals the
    genuine reference; (iii) two tiles with different factors differ (an indexing slip between tiles would show)."""
    import hfx
    import bench
    d = dict(np.load(os.path.join(GOLDEN, "mixed_p3_channel.npz")))
    tiles = 512
    classes, per, faces, bdy = MU.split(d)
    factor = 1.0 + 0.02 * (np.arange(tiles) % 7)

    def make(ctx):
        ctx.set_params(hfx.params_from(per[classes[0]]))
        E, plane = {}, {}
        for c in classes:
            sz = [int(v) for v in per[c]["sizes"]]
            plane[c] = sz[0] * sz[2]
            big = bench.tile_arrays(per[c], tiles)
            u = big["u_init"].copy(order="F")
            u *= np.repeat(factor, sz[0])[None, :, None]
            big["u_init"] = u
            E[c] = hfx.Eles(ctx, [sz[0] * tiles] + sz[1:5], big, ele_type=sz[6], order=sz[5])
            E[c].upload(hfx.DISU_UPTS0, u)
        F = [hfx.IntInters(ctx, E[a], E[b], bench.tile_table(L, plane[a], tiles), bench.tile_table(R, plane[b], tiles)) for a, b, L, R in faces]
        for a, L, ids in bdy:
            F.append(hfx.BdyInters(ctx, E[a], bench.tile_table(L, plane[a], tiles), np.tile(ids, tiles), hfx.bc_records(d["bc_flags"], d["bc_params"]),
                                   float(np.ravel(d["bc_R_ref"])[0]), int(np.ravel(d["ramp_counter"])[0])))
        return E, F

    ctx = hfx.Context(0)
    Ea, Fa = make(ctx)
    Eb, Fb = make(ctx)
    nstage = int(d["c2_sizes"][7])
    last = max(int(k.split("_")[2][4:]) for k in d if k.startswith("c2_u_step"))
    hfx.run_steps_blocks([Ea[c] for c in classes], Fa, last + 1, fused=4)
    hfx.run_steps_blocks([Eb[c] for c in classes], Fb, last + 1, fused=0)
    for c in classes:
        ua, ub = Ea[c].download(hfx.DISU_UPTS0), Eb[c].download(hfx.DISU_UPTS0)
        assert Ea[c].check_nan() == -1
        assert relerr(ua, ub) < 1e-11, c
        ne = int(per[c]["sizes"][0])
        assert relerr(ua[:, :ne, :], d["c%d_u_step%d_stage%d" % (c, last, nstage - 1)]) < 1e-11, c
        assert relerr(ua[:, ne:2 * ne, :], ua[:, :ne, :]) > 1e-3
        assert relerr(ua[:, 7 * ne:8 * ne, :], ua[:, :ne, :]) < 1e-13  # tiles 0 and 7 carry the same factor
    for f in Fa + Fb:
        f.close()
    for c in classes:
        Ea[c].close(); Eb[c].close()
    ctx.close()


@pytest.mark.gpu
def test_general_les_stage_at_bench_size_properties():
    """The mixed channel with the WALE closure at the size of bench.py's `mixed_les_wale` leg (256 tiles here, every tile with its own
    state): the closure inside the general stage's flux kernel -- whose flux-point part re-reads and completes the projected flux the
    workgroup stored a phase earlier -- equals the per-method path on every element of every tile after a time step, on thousands of
    workgroups in flight."""
    import hfx
    import hfx_host as H
    import bench
    d = dict(np.load(os.path.join(GOLDEN, "mixed_p3_channel.npz")))
    tiles = 256
    classes, per, faces, bdy = MU.split(d)
    factor = 1.0 + 0.02 * (np.arange(tiles) % 7)
    J = {}
    for c in classes:
        sz = [int(v) for v in per[c]["sizes"]]
        x1 = per[c]["loc_upts"][2, ::(sz[5] + 1) * (sz[5] + 2) // 2] if sz[6] == 3 else None
        S = H.Simplex(sz[6], sz[5], per[c]["shape"][:, :(4 if sz[6] == 2 else 6), :], viscous=1, loc_1d_upts=x1, SGS_model=1)
        J[c] = np.asfortranarray(np.concatenate([S.array("Jacobian_fpts")] * tiles, axis=3))
        S.close()

    def make(ctx):
        ctx.set_params(hfx.params_from(per[classes[0]]))
        E, plane = {}, {}
        for c in classes:
            sz = [int(v) for v in per[c]["sizes"]]
            plane[c] = sz[0] * sz[2]
            big = bench.tile_arrays(per[c], tiles)
            u = big["u_init"].copy(order="F")
            u *= np.repeat(factor, sz[0])[None, :, None]
            E[c] = hfx.Eles(ctx, [sz[0] * tiles] + sz[1:5], big, ele_type=sz[6], order=sz[5])
            E[c].upload(hfx.DISU_UPTS0, u)
            E[c].set_les(1, 0.325, 1.0, 0.41, 0.9, J[c])
        F = [hfx.IntInters(ctx, E[a], E[b], bench.tile_table(L, plane[a], tiles), bench.tile_table(R, plane[b], tiles)) for a, b, L, R in faces]
        for a, L, ids in bdy:
            F.append(hfx.BdyInters(ctx, E[a], bench.tile_table(L, plane[a], tiles), np.tile(ids, tiles), hfx.bc_records(d["bc_flags"], d["bc_params"]),
                                   float(np.ravel(d["bc_R_ref"])[0]), int(np.ravel(d["ramp_counter"])[0])))
        return E, F

    ctx = hfx.Context(0)
    Ea, Fa = make(ctx)
    Eb, Fb = make(ctx)
    hfx.run_steps_blocks([Ea[c] for c in classes], Fa, 1, fused=4)
    hfx.run_steps_blocks([Eb[c] for c in classes], Fb, 1, fused=0)
    for c in classes:
        ua, ub = Ea[c].download(hfx.DISU_UPTS0), Eb[c].download(hfx.DISU_UPTS0)
        assert Ea[c].check_nan() == -1
        assert relerr(ua, ub) < 1e-11, c
        ne = int(per[c]["sizes"][0])
        assert relerr(ua[:, ne:2 * ne, :], ua[:, :ne, :]) > 1e-3
        assert relerr(ua[:, 7 * ne:8 * ne, :], ua[:, :ne, :]) < 1e-13
    for f in Fa + Fb:
        f.close()
    for c in classes:
        Ea[c].close(); Eb[c].close()
    ctx.close()


# ---- the mixed channel on PARTITIONED element blocks ----------------------------------------------------------------------
def self_partition(ctx, E, faces):
    """Every second face of every interior block that joins a class to ITSELF becomes a pair of one-sided partition faces whose
    neighbour is the rank itself: -> (remaining interior blocks as (a, b, L, R), [hfx.MpiInters]).  The partition-face block
    of a (class, face type) lists the left sides A_0..A_n-1, then the right sides B_0..B_n-1 (in the order of their own
    offsets); Rlut is the slot of the partner's point in the mate's record; the two halves are each other's neighbour segment."""
    import hfx
    rest, mpi = [], []
    for a, b, L, R in faces:
        if a != b or L.shape[1] < 4:
            rest.append((a, b, L, R))
            continue
        pick = np.zeros(L.shape[1], dtype=bool)
        pick[::2] = True
        rest.append((a, b, np.asfortranarray(L[:, ~pick]), np.asfortranarray(R[:, ~pick])))
        La, Ra = L[:, pick], R[:, pick]
        n = La.shape[1]
        order = np.argsort(Ra, axis=0)            # order[j', i]: the A-point j whose partner is B's j'-th point
        rank = np.argsort(order, axis=0)          # rank[j, i]: the slot of A-point j's partner in B's record
        Lb = np.take_along_axis(Ra, order, axis=0)
        Lm = np.asfortranarray(np.concatenate([La, Lb], axis=1).astype(np.int32))
        Rlut = np.asfortranarray(np.concatenate([rank, order], axis=1).astype(np.int32))
        f = hfx.MpiInters(ctx, E[a], Lm, Rlut)
        f.set_neighbours([(0, 0, n, n), (0, n, 0, n)])
        mpi.append(f)
    return rest, mpi


@pytest.mark.gpu
@pytest.mark.parametrize("name,block", [("tet_p2_plot", "tet_p2_n2_deformed"), ("pri_p2_plot", "pri_p2_n2_deformed")])
def test_plot_point_interpolation_on_simplex_classes(name, block):
    """eles::calc_disu_ppts (the VTU writer's input) on tetrahedra and prisms: the reference's opp_p of the class registered on a
    device block of the same mesh, one contraction for all elements, against the reference's interpolated state"""
    import ctypes as C
    import hfx
    from test_gpu_methods_vs_golden import build
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    b = dict(np.load(os.path.join(GOLDEN, block + ".npz")))
    assert [int(v) for v in d["sizes"][:7]] == [int(v) for v in b["sizes"][:7]]
    ctx = hfx.Context(0)
    e, faces = build(ctx, b)
    e.upload(hfx.DISU_UPTS0, d["u_init"])
    opp = np.asfortranarray(d["opp_p"])
    hfx.check(hfx.lib().hfx_eles_set_opp_p(e.h, C.c_int(opp.shape[0]), opp.ctypes.data_as(hfx.dp)))
    out = np.zeros(d["disu_ppts"].shape, order="F")
    hfx.check(hfx.lib().hfx_eles_calc_disu_ppts(e.h, out.ctypes.data_as(hfx.dp)))
    assert relerr(out, d["disu_ppts"]) < 1e-13
    for f in faces:
        f.close()
    e.close()
    ctx.close()


@pytest.mark.gpu
def test_general_stage_follows_a_changed_closure():
    """hfx_eles_set_les after a fused run (another C_s): the general stage's tables -- the closure's length scale among them -- are
    rebuilt, and the next fused step equals the per-method path with the new closure (and differs from the old one)"""
    import hfx
    from test_gpu_methods_vs_golden import build
    d = dict(np.load(os.path.join(GOLDEN, "tet_p2_les_wale.npz")))
    sc = lambda k: float(np.ravel(d[k])[0])
    ctx = hfx.Context(0)
    e, faces = build(ctx, d)
    m, mfaces = build(ctx, d)
    hfx.run_steps(e, faces, 1, fused=4)
    hfx.run_steps(m, mfaces, 1, fused=0)
    assert relerr(e.download(hfx.DISU_UPTS0), m.download(hfx.DISU_UPTS0)) < 1e-12
    old = e.download(hfx.DISU_UPTS0)
    for x in (e, m):
        x.set_les(int(sc("SGS_model")), 4.0 * sc("C_s"), sc("filter_ratio"), sc("Kappa"), sc("prandtl_t"), d["Jacobian_fpts"])
    o, ofaces = build(ctx, d)  # the old closure for comparison
    o.upload(hfx.DISU_UPTS0, old)
    hfx.run_steps(e, faces, 1, fused=4)
    hfx.run_steps(m, mfaces, 1, fused=0)
    hfx.run_steps(o, ofaces, 1, fused=4)
    assert relerr(e.download(hfx.DISU_UPTS0), m.download(hfx.DISU_UPTS0)) < 1e-12
    assert relerr(e.download(hfx.DISU_UPTS0), o.download(hfx.DISU_UPTS0)) > 1e-9
    for f in faces + mfaces + ofaces:
        f.close()
    e.close(); m.close(); o.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tet_p2_les_wale", "tet_p3_les_wsm", "pri_p2_les_wale", "tet_p3_shock", "tet_p2_overint"])
def test_simplex_les_on_partitioned_blocks(name):
    """hfx_run_steps_partitioned_blocks with an LES closure: the projected flux a partition face sends already contains F_sgs . n
    (the closure is evaluated in the general stage's flux kernel), so the self-partitioned block -- half of its faces exchanged
    with itself over RCCL -- equals the genuine reference's undivided run.  The same with shock capturing (the filter and the
    flux-point values of the filtered state before the next solution message is packed) and with over-integration.  (Of the prism box only the triangular faces are
    partitioned: two of its quadrilateral faces have a flux point where the x component of the normal is rounding noise, and the
    reference's LDG switch, src/inters.cpp:568-581, is decided there by the LEFT normal's noise on an interior face and by each
    side's own on a partition face -- the undivided and the partitioned reference differ at those two points by construction.)"""
    import hfx
    from test_gpu_methods_vs_golden import build
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ctx = hfx.Context(0)
    e, whole = build(ctx, d)
    for f in whole:
        f.close()
    c = int(d["sizes"][6])
    faces = [(c, c, d["int%d_L" % t], d["int%d_R" % t]) for t in range(3) if "int%d_L" % t in d]
    rest, M = self_partition(ctx, {c: e}, faces[:1] if c == 3 else faces)
    rest = rest + (faces[1:] if c == 3 else [])
    assert len(M) >= 1
    F = [hfx.IntInters(ctx, e, e, L, R) for a, b, L, R in rest]
    comm = hfx.Comm(ctx.h, hfx.comm_unique_id(), 1, 0)
    nstage = int(d["sizes"][7])
    steps = sorted({int(k.split("_")[1][4:]) for k in d if k.startswith("u_step")})
    for st in steps:
        hfx.run_steps_partitioned_blocks([e], F, M, comm, 1)
        assert relerr(e.download(hfx.DISU_UPTS0), d["u_step%d_stage%d" % (st, nstage - 1)]) < 1e-11, st
    for f in F + M:
        f.close()
    comm.close()
    e.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["run_steps", "deferred"])
def test_mixed_channel_on_partitioned_blocks(how):
    """hfx_run_steps_partitioned_blocks (and the same stage through the deferred call sequence, send_* / receive_* included):
    the mixed channel with half of its tetrahedron | tetrahedron and prism | prism faces turned into partition faces that the
    rank exchanges with itself over libhfx's RCCL transport -- two element blocks, one partition-face block per (class, face
    type) -- equals the genuine reference's undivided run after every step"""
    import hfx
    d = dict(np.load(os.path.join(GOLDEN, "mixed_p3_channel.npz")))
    ctx = hfx.Context(0)
    classes, per, faces, bdy = MU.split(d)
    ctx.set_params(hfx.params_from(per[classes[0]]))
    E = {}
    for c in classes:
        sz = [int(v) for v in per[c]["sizes"]]
        E[c] = hfx.Eles(ctx, sz[:5], per[c], ele_type=sz[6], order=sz[5])
        E[c].upload(hfx.DISU_UPTS0, per[c]["u_init"])
    rest, M = self_partition(ctx, E, faces)
    assert len(M) >= 2
    F = [hfx.IntInters(ctx, E[a], E[b], L, R) for a, b, L, R in rest]
    for a, L, ids in bdy:
        F.append(hfx.BdyInters(ctx, E[a], L, ids, hfx.bc_records(d["bc_flags"], d["bc_params"]), float(np.ravel(d["bc_R_ref"])[0]),
                               int(np.ravel(d["ramp_counter"])[0])))
    comm = hfx.Comm(ctx.h, hfx.comm_unique_id(), 1, 0)
    blocks = [E[c] for c in classes]
    nstage = int(d["c2_sizes"][7])
    adv = int(np.ravel(d["adv_type"])[0])
    steps = sorted({int(k.split("_")[2][4:]) for k in d if k.startswith("c2_u_step")})
    if how == "deferred":
        ctx.set_option("deferred", 1)
    for st in steps:
        if how == "run_steps":
            hfx.run_steps_partitioned_blocks(blocks, F, M, comm, 1)
        else:
            ints = [f for f in F if isinstance(f, hfx.IntInters)]
            bdys = [f for f in F if isinstance(f, hfx.BdyInters)]
            for rk in range(nstage):  # CalcResidual's order with the partition-face calls (src/solver.cpp:59-221)
                for e in blocks: e.extrapolate_solution()
                for f in M: f.send_solution(comm)
                for e in blocks: e.calculate_gradient()
                for e in blocks: e.evaluate_invFlux()
                for f in ints: f.calculate_common_invFlux()
                for f in bdys: f.evaluate_boundaryConditions_invFlux()
                for f in M: f.receive_solution(comm)
                for f in M: f.calculate_common_invFlux()
                for e in blocks: e.correct_gradient()
                for f in M: f.send_corrected_gradient(comm)
                for e in blocks: e.evaluate_viscFlux()
                for e in blocks: e.extrapolate_totalFlux()
                for e in blocks: e.calculate_divergence()
                for f in ints: f.calculate_common_viscFlux()
                for f in bdys: f.evaluate_boundaryConditions_viscFlux()
                for f in M: f.receive_corrected_gradient(comm)
                for f in M: f.calculate_common_viscFlux()
                for e in blocks: e.calculate_corrected_divergence()
                for e in blocks: e.AdvanceSolution(rk, adv)
        for c in classes:
            k = "c%d_u_step%d_stage%d" % (c, st, nstage - 1)
            assert relerr(E[c].download(hfx.DISU_UPTS0), d[k]) < 1e-11, k
    if how == "deferred":
        nf, nr, why = ctx.deferred_stats()
        assert (nf, nr) == (len(steps) * nstage, 0), why
    for c in classes:
        assert E[c].check_nan() == -1
    comm.close()
    for f in F + M:
        f.close()
    for c in classes:
        E[c].close()
    ctx.close()
