"""The host-side mirror's setup (operators, metrics, faces, TGV state) against the genuine
reference's fixtures.  CPU only: nothing here touches the GPU.

The 1-D nodes are computed (Newton) instead of read from the reference's data/*.bin, so the
operators agree to a few ulps of their scale rather than bitwise; the reference's structural
zeros, which the sparse contraction path relies on, must be exact zeros here too.
"""
import glob
import json
import os

import numpy as np
import pytest

import hfx_host as H

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["hex_p2_n3_deformed", "hex_p2_n3_uniform", "hex_p4_n3_deformed", "hex_p3_n3_deformed", "hex_p2_sd",
         "hex_p2_lobatto", "hex_p1_sutherland"]


def case_from_fixture(d):
    meta = json.loads(bytes(d["meta_json"]).decode())
    k = meta["keys"]
    kw = dict(order=k["order"], viscous=k["viscous"], riemann_solve_type=k["riemann_solve_type"], adv_type=k["adv_type"],
              upts_type=k["upts_type_hexa"], vcjh_scheme=k["vcjh_scheme_hexa"], fix_vis=k["fix_vis"], dt=k["dt"],
              ldg_beta=k.get("ldg_beta", 0.5), ldg_tau=k.get("ldg_tau", 0.0), T_c_ic=k["T_c_ic"])
    return H.Case(meta["n"], xv=d["xv"], **kw), meta


def rel(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


@pytest.mark.parametrize("name", NAMES)
def test_setup_matches_reference(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c, meta = case_from_fixture(d)
    assert c.sizes[:7] == [int(v) for v in d["sizes"][:7]]
    ops = ["opp_0", "opp_3", "opp_6"] + ["opp_%d_%d" % (w, i) for w in (1, 2, 4, 5) for i in range(3)]
    for k in ops:
        got, want = c.array(k), d[k]
        assert rel(got, want) < 5e-14, k
        # every structural zero of the reference is an exact zero here; the reference's table nodes
        # are not bit-symmetric (P2 mid node -4.5e-17), which leaves it a few ~1e-16 entries where the
        # symmetric nodes computed here give exact zeros
        assert np.all(got[want == 0.0] == 0.0), "zero pattern of %s" % k
        assert np.all(np.abs(want[got == 0.0]) < 1e-15), "zero pattern of %s" % k
    for k in ("loc_upts", "tloc_fpts", "tnorm_fpts", "shape"):
        assert rel(c.array(k), d[k]) < 1e-15, k
    for k in ("detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts", "tdA_fpts", "norm_fpts"):
        assert rel(c.array(k), d[k]) < 1e-14, k
    # exact zeros of the unit normals decide the LDG switch (inters.cpp:568-581)
    assert np.array_equal(c.array("norm_fpts") == 0.0, d["norm_fpts"] == 0.0)
    assert rel(c.array("disu_upts0"), d["u_init"]) < 1e-14
    p = c.params()
    for k in ("gamma", "prandtl", "rt_inf", "mu_inf", "c_sth", "fix_vis", "dt", "ldg_beta", "ldg_tau"):
        assert abs(getattr(p, k) - float(d[k][0])) <= 1e-15 * max(1.0, abs(float(d[k][0]))), k
    assert np.allclose(list(p.RK_a)[:p.n_rk], np.ravel(d["RK_a"]), rtol=0, atol=0)
    assert np.allclose(list(p.RK_b)[:p.n_rk], np.ravel(d["RK_b"]), rtol=0, atol=0)
    # face tables: identical, including face order and left/right orientation (the LDG switch reads
    # the LEFT normal only, so the orientation is part of the contract)
    L, R = c.faces()
    assert np.array_equal(L, d["int2_L"])
    assert np.array_equal(R, d["int2_R"])
    c.close()


def test_rk414_tableau():
    d = dict(np.load(os.path.join(GOLDEN, "hex_p1_rk414.npz")))
    c = H.Case(3, order=1, adv_type=4, amp=0.1)
    p = c.params()
    assert p.n_rk == 14
    assert np.array_equal(np.array(list(p.RK_a)[:14]), np.ravel(d["RK_a"]))
    assert np.array_equal(np.array(list(p.RK_b)[:14]), np.ravel(d["RK_b"]))
    c.close()


def test_generated_vertices_match_mesh_writer():
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_n3_deformed.npz")))
    c = H.Case(3, order=2, amp=0.15)
    assert rel(c.array("shape"), d["shape"]) < 1e-15
    c.close()


def test_quad_case_builds():
    c = H.Case([4, 4, 1], dims=2, order=3, amp=0.1)
    assert c.sizes[:7] == [16, 16, 16, 4, 2, 3, 1]
    L, R = c.faces()
    assert L.shape == (4, 32)
    assert len(set(L.ravel()) | set(R.ravel())) == 16 * 16  # every flux point on exactly one face side
    assert (c.array("opp_0") != 0).sum(1).max() == 4
    c.close()


def test_error_convention_host():
    import hfx
    with pytest.raises(hfx.HfxError):
        H.Case(2, order=2)  # fewer than 3 cells per direction
    with pytest.raises(hfx.HfxError):
        H.Case(3, order=2, riemann_solve_type=1)  # Lax-Friedrich with NS (input.cpp:546)


@pytest.mark.parametrize("name", ["hex_p2_bdy_walls", "hex_p2_bdy_inout", "hex_p1_bdy_inviscid", "quad_p3_bdy"])
def test_boundary_tables_vs_reference(name):
    """bdy_inters setup of the host mirror: same faces, same order, same boundary ids, same non-dimensional
    bc_list as the genuine reference builds from the mesh file's groups and the input file's bc_* keys."""
    import bdy_util
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c, meta = bdy_util.case_from_fixture(d)
    t = 2 if meta["dims"] == 3 else 0
    L, ids = c.bdy_faces()
    assert np.array_equal(L, d["bdy%d_L" % t])
    assert np.array_equal(ids, np.ravel(d["bdy%d_id" % t]))
    if "int%d_L" % t in d:
        Li, Ri = c.faces()
        assert np.array_equal(Li, d["int%d_L" % t]) and np.array_equal(Ri, d["int%d_R" % t])
    fl, par, R_ref, rc = c.bc_list()
    assert np.array_equal(fl, d["bc_flags"])
    assert np.abs(par - d["bc_params"]).max() <= 1e-15 * np.abs(d["bc_params"]).max()
    assert abs(R_ref - float(np.ravel(d["bc_R_ref"])[0])) <= 1e-15 * abs(R_ref)
    assert rc == int(np.ravel(d["ramp_counter"])[0])
    assert rel(c.array("disu_upts0"), d["u_init"]) < 1e-14
    c.close()


def test_h_ref_vs_reference():
    """eles::h_ref = shortest element edge (calc_h_ref_specific), the length scale of the CFL time step."""
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_cfl_global.npz")))
    c = H.Case(3, xv=d["xv"], order=2, dt_type=1, CFL=0.4)
    assert rel(c.array("h_ref"), np.ravel(d["h_ref"])) < 1e-15
    c.close()


def test_isentropic_vortex_initial_state():
    """BASELINE.json configs[0]: the Euler isentropic vortex on quads (ic_form 0) as the reference initialises it."""
    d = dict(np.load(os.path.join(GOLDEN, "quad_p3_vortex.npz")))
    c = H.Case([6, 6, 1], xv=d["xv"], dims=2, order=3, viscous=0, ic_form=0, riemann_solve_type=0, dt=0.001,
               rho_c_ic=1.0, u_c_ic=1.0, v_c_ic=1.0, p_c_ic=1.0)
    assert rel(c.array("disu_upts0"), d["u_init"]) < 1e-14
    L, R = c.faces()
    assert np.array_equal(L, d["int0_L"]) and np.array_equal(R, d["int0_R"])
    c.close()


MODAL = ["hex_p2_overint", "quad_p3_overint", "hex_p3_shock", "hex_p2_shock_energy", "quad_p3_shock", "hex_p4_jet"]


@pytest.mark.parametrize("name", MODAL)
def test_modal_operators_vs_reference(name):
    """The producers of the shock-capturing and over-integration matrices (csrc/host/eles_modal.cpp: 1-D factors and
    Kronecker products) against the matrices the genuine reference built by dense inversion / multiplication
    (set_vandermonde3D, set_exp_filter, calc_norm_basis, set_over_int; JGinv at the cubature points)."""
    import bdy_util
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c, meta = bdy_util.case_from_fixture(d)
    keys = []
    if "over_int" in d:
        keys += ["opp_over_int_cubpts", "over_int_filter", "JGinv_over_int_cubpts"]
    if "shock_cap" in d:
        keys += ["inv_vandermonde", "exp_filter", "norm_basis_persson"]
    assert keys
    for k in keys:
        got, want = c.array(k), d[k]
        assert got.shape == want.shape, k
        assert rel(got, want) < 5e-14, (k, rel(got, want))
    c.close()


@pytest.mark.parametrize("name", ["hex_p3_plot", "quad_p2_plot"])
def test_plot_points_vs_reference(name):
    """set_loc_ppts / set_opp_p of the host mirror against the genuine reference's plot points and operator."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    meta = json.loads(bytes(d["meta_json"]).decode())
    kk = meta["keys"]
    n = meta["n"] if isinstance(meta["n"], list) else [meta["n"]] * meta["dims"]
    c = H.Case(n + [1] * (3 - len(n)), xv=d["xv"], dims=meta["dims"], order=kk["order"], p_res=kk["p_res"], T_c_ic=kk["T_c_ic"])
    assert np.array_equal(c.array("loc_ppts"), d["loc_ppts"])
    assert rel(c.array("opp_p"), d["opp_p"]) < 5e-14
    c.close()


def test_les_jacobian_vs_reference():
    """Jacobian_fpts of the host mirror (what extrapolate_sgsFlux takes the SGS flux back to physical space with)."""
    d = dict(np.load(os.path.join(GOLDEN, "hex_p2_les_wale.npz")))
    k = json.loads(bytes(d["meta_json"]).decode())["keys"]
    c = H.Case(3, xv=d["xv"], order=2, LES=1, SGS_model=1, C_s=k["C_s"], filter_ratio=k["filter_ratio"], T_c_ic=k["T_c_ic"])
    assert rel(c.array("Jacobian_fpts"), d["Jacobian_fpts"]) < 1e-14
    c.close()


# ---- tetrahedra and triangular prisms: operators and metrics ------------------------------------
SIMPLEX = [("tet_p2_n2_deformed", ""), ("tet_p3_n2_deformed", ""), ("pri_p2_n2_deformed", ""), ("pri_p3_n2_deformed", ""),
           ("mixed_p3_channel", "c2_"), ("mixed_p3_channel", "c3_"), ("mixed_p2_channel", "c2_"), ("mixed_p2_channel", "c3_"),
           # curved: the quadratic tetrahedron (10 shape nodes) and the quadratic prism (15), every mid-edge node off its edge
           ("tet_p2_curved", ""), ("pri_p2_curved", "")]


@pytest.mark.parametrize("name,pre", SIMPLEX)
def test_simplex_operators_and_metrics_vs_reference(name, pre):
    """eles_tets / eles_pris of the host mirror (csrc/host/eles_simplex.cpp) from the mesh's shape nodes alone: point sets,
    the seven operators and the metrics equal the genuine reference's.  The modal basis and the lifting integrals are
    computed differently from the reference (three-term Jacobi recurrence, exact Gauss integration), so equality of the
    matrices checks the DEFINITIONS: 1e-12 of each matrix's scale."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    g = lambda k: d[pre + k]
    sz = [int(v) for v in g("sizes")]
    ele_type, order, nu = sz[6], sz[5], sz[1]
    x1 = None
    if ele_type == 3:  # the prism's line direction: the reference's own Gauss abscissae (data/JacobiGQ.bin)
        x1 = g("loc_upts")[2, ::(order + 1) * (order + 2) // 2]
    shp = g("shape")
    if "curved" in name:
        assert shp.shape[1] == (10 if ele_type == 2 else 15)
        # the elements ARE curved: the Jacobian determinant varies inside an element (constant on straight-sided tetrahedra)
        dj = g("detjac_upts")
        assert (dj.max(axis=0) - dj.min(axis=0)).max() > 1e-3 * np.abs(dj).max()
    else:
        shp = shp[:, :(4 if ele_type == 2 else 6), :]
    S = H.Simplex(ele_type, order, shp, viscous=1, loc_1d_upts=x1)
    names = ["loc_upts", "tloc_fpts", "tnorm_fpts", "opp_0", "opp_3", "opp_6", "detjac_upts", "JGinv_upts", "detjac_fpts", "JGinv_fpts",
             "tdA_fpts", "norm_fpts", "pos_upts", "pos_fpts"]
    for dd in range(3):
        names += ["opp_1_%d" % dd, "opp_2_%d" % dd, "opp_4_%d" % dd, "opp_5_%d" % dd]
    for k in names:
        want, got = g(k), S.array(k)
        assert got.shape == want.shape, (k, got.shape, want.shape)
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), (k, np.abs(got - want).max() / np.abs(want).max())
    S.close()


@pytest.mark.parametrize("name,scheme,c", [("tet_p2_vcjh_sd", 2, 0.0), ("tet_p3_vcjh_cplus", 4, 0.0), ("tet_p2_vcjh_c", 0, 0.02),
                                           ("pri_p2_vcjh_hu", 3, 0.0)])
def test_simplex_vcjh_members_vs_reference(name, scheme, c):
    """The members of the VCJH family other than DG on tetrahedra (vcjh_scheme_tet 0 / 2 / 4) and on the prism's triangle
    (vcjh_scheme_tri 3): opp_3 = Filt . opp_3_dg with the filter matrix of src/eles_tets.cpp:1305-1503 / src/funcs.cpp:717-880,
    built here from the nodal differentiation matrices and V V^T -- equal to the genuine reference's opp_3 to 1e-11 of its
    scale (the filter involves products of up to p differentiation matrices and one inverse), and different from the DG one."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    sz = [int(v) for v in d["sizes"]]
    ele_type, order = sz[6], sz[5]
    x1 = d["loc_upts"][2, ::(order + 1) * (order + 2) // 2] if ele_type == 3 else None
    shp = d["shape"][:, :(4 if ele_type == 2 else 6), :]
    S = H.Simplex(ele_type, order, shp, viscous=1, loc_1d_upts=x1, vcjh_scheme=scheme, c=c)
    got, want = S.array("opp_3"), d["opp_3"]
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()
    for k in ("opp_0", "opp_1_0", "opp_2_1", "opp_5_2", "opp_6"):
        assert np.abs(S.array(k) - d[k]).max() <= 1e-12 * np.abs(d[k]).max(), k
    S.close()
    dg = H.Simplex(ele_type, order, shp, viscous=1, loc_1d_upts=x1)
    assert np.abs(dg.array("opp_3") - want).max() > 1e-3 * np.abs(want).max()
    dg.close()


@pytest.mark.parametrize("name", ["tet_p3_shock", "pri_p2_shock"])
def test_simplex_shock_capture_operators_vs_reference(name):
    """set_vandermonde / set_exp_filter / the Persson sensor's norms and highest modes of eles_tets (src/eles_tets.cpp:705-797) and
    eles_pris (src/eles_pris.cpp:609-730) from the host mirror: exp_filter equals the reference's; inv_vandermonde equals it
    row by row up to the sign of a mode (the sensor squares the modal coefficients, the filter is V diag V^-1)."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    k = json.loads(bytes(d["meta_json"]).decode())["keys"]
    sz = [int(v) for v in d["sizes"]]
    ele_type, order = sz[6], sz[5]
    x1 = d["loc_upts"][2, ::(order + 1) * (order + 2) // 2] if ele_type == 3 else None
    S = H.Simplex(ele_type, order, d["shape"], viscous=1, loc_1d_upts=x1, shock_cap=1, expf_fac=k["expf_fac"],
                  expf_order=k["expf_order"], expf_cutoff=k["expf_cutoff"])
    E, W = S.array("exp_filter"), d["exp_filter"]
    assert np.abs(E - W).max() <= 1e-12 * np.abs(W).max()
    Vi, Wi = S.array("inv_vandermonde"), d["inv_vandermonde"]
    sign = np.sign((Vi * Wi).sum(axis=1))
    assert (sign != 0).all()
    assert np.abs(Vi * sign[:, None] - Wi).max() <= 1e-12 * np.abs(Wi).max()
    assert np.abs(S.array("norm_basis_persson") - np.ravel(d["norm_basis_persson"])).max() <= 1e-15
    assert (S.array("persson_high_modes").astype(int) == np.ravel(d["persson_high_modes"])).all()
    S.close()


@pytest.mark.parametrize("name", ["tet_p2_cfl_local", "pri_p2_cfl_global"])
def test_simplex_reference_length_vs_reference(name):
    """calc_h_ref_specific of tetrahedra (insphere diameter) and prisms (shortest vertical edge / triangle incircle diameter), the
    length scale of the CFL time step, from the host mirror's classes"""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    sz = [int(v) for v in d["sizes"]]
    x1 = d["loc_upts"][2, ::(sz[5] + 1) * (sz[5] + 2) // 2] if sz[6] == 3 else None
    S = H.Simplex(sz[6], sz[5], d["shape"], viscous=1, loc_1d_upts=x1)
    h = np.ravel(S.array("h_ref"))
    assert h.shape == np.ravel(d["h_ref"]).shape
    assert np.abs(h - np.ravel(d["h_ref"])).max() < 1e-13 * np.abs(d["h_ref"]).max()
    S.close()


def test_simplex_classes_refuse_what_they_do_not_build():
    d = dict(np.load(os.path.join(GOLDEN, "tet_p2_n2_deformed.npz")))
    with pytest.raises(Exception):
        H.Simplex(2, 9, d["shape"][:, :4, :])  # no point table for this order


@pytest.mark.parametrize("ftype,order,dims", [(0, 2, 3), (0, 3, 2), (0, 4, 3), (1, 2, 3), (1, 3, 2), (2, 3, 3), (3, 2, 2), (0, 1, 2)])
def test_les_filter_properties(ftype, order, dims):
    """compute_filter_upts (src/eles_hexas.cpp:583, src/eles_quads.cpp:428): every filter preserves constants (rows sum to
    one: the first moment condition / the normalisation / the mean mode) and is the tensor product of its 1-D factor."""
    c = H.Case([3] * dims + [1] * (3 - dims), dims=dims, order=order, LES=1, SGS_model=4, C_s=0.1, filter_ratio=2.0, filter_type=ftype)
    F = c.array("filter_upts")
    N = order + 1
    assert F.shape == (N ** dims, N ** dims)
    if ftype == 0 and order >= 2:
        # the Vasilyev weights are stored by column (the reference's filter_upts_1D(j,i) = B(j)): columns sum to one
        assert np.abs(F.sum(axis=0) - 1).max() < 1e-10
    else:
        assert np.abs(F.sum(axis=1) - 1).max() < 1e-12
    # tensor structure with the first direction fastest: F = F1 (x) F1 [(x) F1]
    f = c.array("filter_upts_1D")
    K = np.kron(f, f) if dims == 2 else np.kron(f, np.kron(f, f))
    assert np.abs(K - F).max() < 1e-13
    c.close()


@pytest.mark.parametrize("name", ["hex_p2_les_wsm", "hex_p2_les_sim", "hex_p2_les_svv"])
def test_les_filter_vs_reference(name):
    """filter_upts of the genuine reference (Vasilyev and Gaussian filters at P2) from the host mirror's producer"""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    meta = json.loads(bytes(d["meta_json"]).decode())
    k = meta["keys"]
    c = H.Case([meta["n"]] * 3 if not isinstance(meta["n"], list) else meta["n"], xv=d["xv"], dims=3, order=k["order"], LES=1,
               SGS_model=k["SGS_model"], C_s=k["C_s"], filter_ratio=k["filter_ratio"], filter_type=k["filter_type"])
    assert np.abs(c.array("filter_upts") - d["filter_upts"]).max() < 1e-12 * np.abs(d["filter_upts"]).max()
    c.close()


@pytest.mark.parametrize("name", ["tet_p3_les_wsm", "tet_p3_les_sim", "tet_p3_les_svv"])
def test_tet_les_filter_vs_reference(name):
    """eles_tets::compute_filter_upts (src/eles_tets.cpp:576-690): the modal filter and the element average, after the
    reference's in-place symmetrisation and normalisation passes, from the host mirror's eles_tets"""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    k = json.loads(bytes(d["meta_json"]).decode())["keys"]
    s = H.Simplex(2, k["order"], d["shape"], SGS_model=k["SGS_model"], filter_ratio=k["filter_ratio"], filter_type=k["filter_type"])
    F = s.array("filter_upts")
    assert F.shape == d["filter_upts"].shape
    assert np.abs(F - d["filter_upts"]).max() < 1e-12 * np.abs(d["filter_upts"]).max()
    s.close()
