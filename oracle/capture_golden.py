#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ -- TEST INFRASTRUCTURE, container only.

Runs the GENUINE reference (oracle/_ref/ref_harness, built by `make -C oracle ref`
from /root/reference/src) on small generated periodic meshes and stores its
inputs and outputs as compressed .npz fixtures.  The fixtures are data (arrays
in, arrays out); no reference source travels.

    python oracle/capture_golden.py            # all cases
    python oracle/capture_golden.py hex_p2_n3_deformed

Each fixture holds: the case description (`meta_json`), the mesh vertices
(`xv`) and the arrays named in oracle/ref_harness.cpp.
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_neu_mesh import write_neu, write_neu_tets, write_neu_prisms, write_neu_mixed  # noqa: E402

REF_HOME = os.environ.get("HIFILES_HOME", "/root/reference")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# fluid / free-stream block of the shipped TGV case
# (/root/reference/testcases/navier-stokes/Taylor_Green_vortex/input_TGV_SD_hex:96-117)
TGV_FLUID = dict(
    gamma=1.4, prandtl=0.72, S_gas=120.0, T_gas=291.15, R_gas=286.9, mu_gas=1.827e-05,
    fix_vis=1, Mach_free_stream=0.1, rho_free_stream=0.0008421095852102401,
    L_free_stream=1.0, T_free_stream=300.0,
    rho_c_ic=0.0008421095852102401, Mach_c_ic=0.1, T_c_ic=300.0,
)

BASE = dict(
    equation=0, viscous=1, riemann_solve_type=3, vis_riemann_solve_type=0,
    ic_form=7, test_case=0, order=2, dt_type=0, dt=0.00001440389, n_steps=1, adv_type=3,
    LES=0, restart_flag=0, mesh_file="mesh.neu",
    dx_cyclic=6.2831853071795862, dy_cyclic=6.2831853071795862, dz_cyclic=6.2831853071795862,
    p_res=2, write_type=0, monitor_res_freq=100000, plot_freq=100000000, restart_dump_freq=100000000,
    res_norm_type=1, error_norm_type=1,
    upts_type_quad=0, vcjh_scheme_quad=1, eta_quad=0.0, sparse_quad=0,
    upts_type_hexa=0, vcjh_scheme_hexa=1, eta_hexa=0.0, sparse_hexa=0,
    sparse_tri=0, sparse_tet=0, sparse_pri=0,
    bc_Cyclic_type="cyclic",
)
BASE.update(TGV_FLUID)


def case(name, n=3, dims=3, amp=0.0, steps=1, level=1, bcs=None, restart=False, tets=False, keep_every=1, ppts=False, curve=None,
         **over):
    d = dict(BASE)
    d.update(over)
    return dict(name=name, n=n, dims=dims, amp=amp, steps=steps, level=level, keys=d, bcs=bcs, restart=restart, tets=tets,
                keep_every=keep_every, ppts=ppts, curve=curve)


# boundary groups for the bdy_inters fixtures: states near the TGV initial state (rho 8.42e-4, T 300, Mach 0.1)
P_TGV = 0.0008421095852102401 * 286.9 * 300.0
BC_KEYS = dict(
    # an inlet group makes InitSolution -> bdy_inters::add_les_inlet read the interface-cubature metrics, which the
    # reference only sets up when forces are requested (src/eles.cpp:4023): without this key it segfaults
    calc_force=1, monitor_cp_freq=100000000, area_ref=1.0,
    bc_In_type="sub_in_char", bc_In_p_total=P_TGV * 1.0070, bc_In_T_total=300.6, bc_In_nx=0.0, bc_In_ny=0.0, bc_In_nz=1.0,
    bc_Out_type="sub_out_char", bc_Out_p_static=P_TGV * 0.999,
    bc_WallT_type="isotherm_wall", bc_WallT_T_static=310.0, bc_WallT_u=3.0,
    bc_WallQ_type="adiabat_wall", bc_WallQ_v=-2.0,
    bc_Far_type="char", bc_Far_p_static=P_TGV, bc_Far_mach=0.12, bc_Far_T_static=295.0, bc_Far_nx=0.8, bc_Far_ny=0.6, bc_Far_nz=0.0,
    bc_Slip_type="slip_wall",
    bc_InS_type="sub_in_simp", bc_InS_rho=0.00085, bc_InS_u=20.0, bc_InS_v=5.0, bc_InS_w=-3.0,
    bc_OutS_type="sub_out_simp", bc_OutS_p_static=P_TGV * 1.001,
    bc_SupI_type="sup_in", bc_SupI_p_static=P_TGV * 1.1, bc_SupI_mach=1.5, bc_SupI_T_static=290.0, bc_SupI_nx=0.0, bc_SupI_ny=0.0,
    bc_SupI_nz=1.0,
    bc_SupO_type="sup_out",
    bc_Dual_type="slip_wall_dual",
    bc_InR_type="sub_in_char", bc_InR_p_total=P_TGV * 1.0070, bc_InR_T_total=300.6, bc_InR_nx=0.0, bc_InR_ny=1.0, bc_InR_nz=0.0,
    bc_InR_pressure_ramp=1, bc_InR_p_ramp_coeff=0.01, bc_InR_T_ramp_coeff=-1.0, bc_InR_p_total_old=P_TGV * 1.002,
)

# the mixed channel: element-class keys of both classes, walls moving against the Taylor-Green field so that the viscous
# wall fluxes are not small
MIXED_KEYS = dict(
    upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0,
    upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0, vcjh_scheme_tri=1, c_tri=0.0,
    bc_WallLo_type="isotherm_wall", bc_WallLo_T_static=310.0, bc_WallLo_u=3.0,
    bc_WallHi_type="adiabat_wall", bc_WallHi_w=-2.0,
)

# threshold inside the widest gap of hex_p4_jet's first-stage sensor values (so that rounding cannot flip an element)
HEX_P4_JET_S0 = 7.0e-7

CASES = [
    # full dump of every intermediate of one residual on a deformed mesh
    case("hex_p2_n3_deformed", amp=0.15, level=2, order=2, steps=2),
    # uniform mesh: exact-zero normal components exercise the LDG tie-breaks (inters.cpp:568-581)
    case("hex_p2_n3_uniform", amp=0.0, level=1, order=2, steps=2),
    # the headline polynomial order
    case("hex_p4_n3_deformed", amp=0.15, level=1, order=4, steps=1),
    # variants on tiny P1 cases: time schemes, Riemann solvers, Sutherland viscosity, correction functions
    case("hex_p1_rk_euler", amp=0.1, order=1, adv_type=0, steps=2),
    case("hex_p1_rk24", amp=0.1, order=1, adv_type=1, steps=2),
    case("hex_p1_rk34", amp=0.1, order=1, adv_type=2, steps=2),
    case("hex_p1_rk414", amp=0.1, order=1, adv_type=4, steps=1),
    case("hex_p1_rusanov", amp=0.1, order=1, riemann_solve_type=0),
    case("hex_p1_roem", amp=0.1, order=1, riemann_solve_type=2),
    case("hex_p1_sutherland", amp=0.1, order=1, fix_vis=0, T_c_ic=350.0),
    case("hex_p1_ldg_tau", amp=0.1, order=1, ldg_tau=0.3, ldg_beta=0.25),
    case("hex_p2_sd", amp=0.1, order=2, vcjh_scheme_hexa=2),
    case("hex_p2_lobatto", amp=0.1, order=2, upts_type_hexa=1),
    case("hex_p3_n3_deformed", amp=0.15, level=1, order=3, steps=1),
    # BASELINE.json configs[0]: Euler isentropic vortex on quads, P3 (the reference's own CPU-runnable case, reduced mesh)
    case("quad_p3_vortex", dims=2, n=6, amp=0.0, level=1, order=3, steps=2, viscous=0, ic_form=0, riemann_solve_type=0,
         dt=0.001, rho_c_ic=1.0, u_c_ic=1.0, v_c_ic=1.0, w_c_ic=0.0, p_c_ic=1.0),
    # quads above P5 (the split fused stage is instantiated to P7 on quads): viscous, deformed
    case("quad_p6_deformed", dims=2, n=3, amp=0.1, level=1, order=6, steps=1),
    case("quad_p7_deformed", dims=2, n=3, amp=0.1, level=1, order=7, steps=1),
    # tetrahedra (BASELINE.json configs[3]'s element family): non-tensor-product operators, triangular faces
    case("tet_p2_n2_deformed", n=2, amp=0.1, level=2, order=2, steps=1, tets=True,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p2_n2_deformed", n=2, amp=0.1, level=2, order=2, steps=1, tets="prisms",
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0,
         vcjh_scheme_tri=1, c_tri=0.0),
    # plot-point interpolation (VTU output's calc_disu_ppts): operator, plot points, interpolated initial state
    case("hex_p3_plot", amp=0.15, level=0, order=3, steps=1, p_res=4, ppts=True),
    case("quad_p2_plot", dims=2, n=4, amp=0.1, level=0, order=2, steps=1, p_res=3, ppts=True),
    case("tet_p2_plot", n=2, amp=0.1, level=0, order=2, steps=1, p_res=3, ppts=True, tets=True,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p2_plot", n=2, amp=0.1, level=0, order=2, steps=1, p_res=3, ppts=True, tets="prisms",
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0, vcjh_scheme_tri=1, c_tri=0.0),
    # a longer run: 40 time steps (200 RK stages) of the genuine reference, the state after every step -- rounding
    # differences between the paths must not grow
    case("hex_p2_long", amp=0.15, level=0, order=2, steps=40, keep_every=10),
    # the other members of the VCJH family on tetrahedra (filter matrix of src/eles_tets.cpp:1305) and on the prism's triangle
    # (src/funcs.cpp:717): SD-like, Huynh-like, c+, and a c given by the user
    case("tet_p2_vcjh_sd", n=2, amp=0.1, level=1, order=2, steps=1, tets=True,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=2, eta_tet=0.0),
    case("tet_p3_vcjh_cplus", n=2, amp=0.1, level=1, order=3, steps=1, tets=True,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=4, eta_tet=0.0),
    case("tet_p2_vcjh_c", n=2, amp=0.1, level=1, order=2, steps=1, tets=True,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=0, c_tet=0.02, eta_tet=0.0),
    case("pri_p2_vcjh_hu", n=2, amp=0.1, level=1, order=2, steps=1, tets="prisms",
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0,
         vcjh_scheme_tri=3, c_tri=0.0),
    # BASELINE.json configs[3]'s order: P3 tetrahedra and prisms (operators of the size the mixed channel case runs with)
    case("tet_p3_n2_deformed", n=2, amp=0.1, level=1, order=3, steps=1, tets=True,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p3_n2_deformed", n=2, amp=0.1, level=1, order=3, steps=1, tets="prisms",
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0,
         vcjh_scheme_tri=1, c_tri=0.0),
    # LES eddy-viscosity closures: WALE on a periodic box, Smagorinsky with wall damping between two walls
    case("hex_p2_les_wale", amp=0.15, level=2, order=2, steps=1, LES=1, SGS_model=1, C_s=0.325, filter_ratio=1.0),
    case("hex_p2_les_smag", amp=0.1, level=2, order=2, steps=1, LES=1, SGS_model=0, C_s=0.17, filter_ratio=1.5,
         bcs={"y-": "WallT", "y+": "WallQ"}, **BC_KEYS),
    case("quad_p3_les_wale", dims=2, n=4, amp=0.1, level=2, order=3, steps=1, LES=1, SGS_model=1, C_s=0.325, filter_ratio=1.0),
    # similarity-type closures: filtered solution and Leonard tensors at the first stage of every step (calc_sgs_terms):
    # 2 WALE + similarity, 4 similarity alone, 3 spectral vanishing viscosity (the state itself is filtered)
    case("hex_p2_les_wsm", amp=0.15, level=2, order=2, steps=2, LES=1, SGS_model=2, C_s=0.325, filter_ratio=2.0, filter_type=0),
    case("hex_p2_les_sim", amp=0.15, level=2, order=2, steps=2, LES=1, SGS_model=4, C_s=0.325, filter_ratio=2.0, filter_type=1),
    case("hex_p2_les_svv", amp=0.15, level=1, order=2, steps=2, LES=1, SGS_model=3, C_s=0.325, filter_ratio=2.0, filter_type=0),
    # LES on the simplex classes (src/eles_tets.cpp:127,576-690: modal and average filters; the prism class builds none,
    # src/eles_pris.cpp:134): WALE on tetrahedra and prisms, WALE + similarity with the modal filter and the similarity closure
    # with the element average on tetrahedra
    case("tet_p2_les_wale", n=2, amp=0.1, level=2, order=2, steps=1, tets=True, LES=1, SGS_model=1, C_s=0.325, filter_ratio=1.0,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p2_les_wale", n=2, amp=0.1, level=2, order=2, steps=1, tets="prisms", LES=1, SGS_model=1, C_s=0.325,
         filter_ratio=1.0, upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0,
         vcjh_scheme_tri=1, c_tri=0.0),
    # (P3: at P2 the genuine reference aborts in its own set-up with "free(): invalid pointer" when a tetrahedral filter is asked for)
    case("tet_p3_les_wsm", n=2, amp=0.1, level=2, order=3, steps=2, tets=True, LES=1, SGS_model=2, C_s=0.325, filter_ratio=2.0,
         filter_type=2, upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("tet_p3_les_sim", n=2, amp=0.1, level=1, order=3, steps=2, tets=True, LES=1, SGS_model=4, C_s=0.325, filter_ratio=2.0,
         filter_type=3, upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    # the spectral vanishing viscosity closure: the state itself is replaced by its modal-filtered form at the first stage of a step
    case("tet_p3_les_svv", n=2, amp=0.1, level=1, order=3, steps=2, tets=True, LES=1, SGS_model=3, C_s=0.325, filter_ratio=2.0,
         filter_type=2, upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    # curved elements: the quadratic tetrahedron (10 shape nodes, src/eles_tets.cpp:1047) and the quadratic prism (15,
    # src/eles_pris.cpp:1115) with every mid-edge node moved off its edge
    case("tet_p2_curved", n=2, amp=0.1, level=1, order=2, steps=1, tets=True, curve=0.06,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p2_curved", n=2, amp=0.1, level=1, order=2, steps=1, tets="prisms", curve=0.06,
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0,
         vcjh_scheme_tri=1, c_tri=0.0),
    # the ASCII restart file of the final state (on-disk format either side of the path)
    case("hex_p2_restart", amp=0.15, level=0, order=2, steps=1, restart=True),
    case("quad_p3_restart", dims=2, n=4, amp=0.1, level=0, order=3, steps=1, restart=True),
    # integral diagnostics of the TGV monitors (kinetic energy, enstrophy, ...)
    case("hex_p2_integrals", amp=0.15, level=1, order=2, steps=1,
         integral_quantities="5 kineticenergy enstropy pressuredilatation straincolonproduct devstraincolonproduct"),
    case("quad_p3_integrals", dims=2, n=4, amp=0.1, level=1, order=3, steps=1,
         integral_quantities="3 enstropy kineticenergy devstraincolonproduct"),
    case("tet_p2_integrals", n=2, amp=0.1, level=1, order=2, steps=1, tets=True,
         integral_quantities="5 kineticenergy enstropy pressuredilatation straincolonproduct devstraincolonproduct",
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p2_integrals", n=2, amp=0.1, level=1, order=2, steps=1, tets="prisms",
         integral_quantities="3 enstropy kineticenergy devstraincolonproduct",
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0, vcjh_scheme_tri=1, c_tri=0.0),
    # CFL time stepping: global minimum (dt_type 1) and local (dt_type 2)
    case("hex_p2_cfl_global", amp=0.15, level=1, order=2, steps=2, dt_type=1, CFL=0.4),
    case("hex_p2_cfl_local", amp=0.15, level=1, order=2, steps=2, dt_type=2, CFL=0.4, adv_type=0),
    # the same on the simplex classes: calc_h_ref_specific is the tetrahedron's insphere diameter (src/eles_tets.cpp:1599-1633) and the
    # smallest of the prism's vertical edges and its triangles' incircle diameters (src/eles_pris.cpp:1535-1557)
    case("tet_p2_cfl_local", n=2, amp=0.1, level=1, order=2, steps=2, tets=True, dt_type=2, CFL=0.4, adv_type=0,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p2_cfl_global", n=2, amp=0.1, level=1, order=2, steps=2, tets="prisms", dt_type=1, CFL=0.4,
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0, vcjh_scheme_tri=1, c_tri=0.0),
    # over-integration of the inviscid flux (polynomial de-aliasing)
    case("hex_p2_overint", amp=0.15, level=2, order=2, steps=1, over_int=1, over_int_order=6),
    case("quad_p3_overint", dims=2, n=4, amp=0.1, level=2, order=3, steps=1, over_int=1, over_int_order=9),
    # shock capturing (Persson sensor + exponential modal filter after every RK stage); s0 sits in a gap of the
    # first stage's sensor distribution so that about a third of the elements are filtered
    case("hex_p3_shock", amp=0.15, level=1, order=3, steps=2, shock_cap=1, shock_det=0, s0=1.4e-7, expf_fac=36.0,
         expf_order=4, expf_cutoff=1, shock_det_field=0),
    case("hex_p2_shock_energy", amp=0.15, level=1, order=2, steps=1, shock_cap=1, shock_det=0, s0=1.11e-6, expf_fac=36.0,
         expf_order=4, expf_cutoff=0, shock_det_field=1),
    case("quad_p3_shock", dims=2, n=4, amp=0.1, level=1, order=3, steps=1, shock_cap=1, shock_det=0, s0=1.8616e-7,
         expf_fac=36.0, expf_order=4, expf_cutoff=1, shock_det_field=0),
    # the same two ingredients on the simplex classes (src/eles_tets.cpp:71-94,718-790,946; src/eles_pris.cpp); s0 inside the
    # widest gap of the first stage's sensor values (14 % wide on the tetrahedra, 1.3e-4 on the prisms, whose values are close)
    case("tet_p3_shock", n=2, amp=0.1, level=1, order=3, steps=2, tets=True, shock_cap=1, shock_det=0, s0=1.23e-7, expf_fac=36.0,
         expf_order=4, expf_cutoff=1, shock_det_field=0, upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("tet_p2_overint", n=2, amp=0.1, level=2, order=2, steps=1, tets=True, over_int=1, over_int_order=6,
         upts_type_tet=0, fpts_type_tet=0, vcjh_scheme_tet=1, eta_tet=0.0),
    case("pri_p2_shock", n=2, amp=0.1, level=1, order=2, steps=2, tets="prisms", shock_cap=1, shock_det=0, s0=4.5661e-6, expf_fac=36.0,
         expf_order=4, expf_cutoff=1, shock_det_field=0, upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0,
         upts_type_tri=0, vcjh_scheme_tri=1, c_tri=0.0),
    case("pri_p2_overint", n=2, amp=0.1, level=2, order=2, steps=1, tets="prisms", over_int=1, over_int_order=6,
         upts_type_pri_tri=0, upts_type_pri_1d=0, vcjh_scheme_pri_1d=1, eta_pri=0.0, upts_type_tri=0, vcjh_scheme_tri=1, c_tri=0.0),
    # BASELINE.json configs[4] in small: P4 hexes, HLLC, shock capturing AND polynomial de-aliasing together, supersonic
    # in- and outflow, characteristic pressure outlets and slip walls around a box (the pieces of the supersonic-jet
    # case).  No "char" far field here: the Taylor-Green field has u.n = 0 on every side of the box, and that boundary
    # type branches on the sign of u.n -- rounding noise would pick the branch.
    case("hex_p4_jet", n=2, amp=0.12, level=1, order=4, steps=1, riemann_solve_type=3, over_int=1, over_int_order=6,
         shock_cap=1, shock_det=0, s0=HEX_P4_JET_S0, expf_fac=36.0, expf_order=4, expf_cutoff=1, shock_det_field=0,
         bcs={"z-": "SupI", "z+": "SupO", "x-": "Out", "x+": "Out", "y-": "Slip", "y+": "Slip"}, **BC_KEYS),
    # boundary faces (bdy_inters): every ghost-state branch that the shipped cases use
    case("hex_p2_bdy_walls", amp=0.1, level=2, order=2, steps=1,
         bcs={"z-": "In", "z+": "Out", "y-": "WallT", "y+": "WallQ", "x-": "Far", "x+": "Slip"}, **BC_KEYS),
    case("hex_p2_bdy_inout", amp=0.1, level=2, order=2, steps=1, riemann_solve_type=0,
         bcs={"y-": "InS", "y+": "OutS", "z-": "SupI", "z+": "SupO"}, **BC_KEYS),
    case("hex_p1_bdy_inviscid", amp=0.1, level=2, order=1, steps=2, viscous=0, ic_form=1, riemann_solve_type=2,
         rho_c_ic=1.2, u_c_ic=30.0, v_c_ic=10.0, w_c_ic=5.0, p_c_ic=101325.0,
         bcs={"y-": "Dual", "y+": "Slip", "z-": "FarI", "z+": "OutI"},
         bc_FarI_type="char", bc_FarI_p_static=101325.0, bc_FarI_mach=0.1, bc_FarI_T_static=294.0, bc_FarI_nx=1.0,
         bc_OutI_type="sub_out_simp", bc_OutI_p_static=101000.0, bc_OutI_T_total=300.0, **BC_KEYS),
    case("quad_p3_bdy", dims=2, n=4, amp=0.1, level=2, order=3, steps=1,
         bcs={"y-": "WallT", "y+": "Far2", "x-": "InR", "x+": "Out"},
         bc_Far2_type="char", bc_Far2_p_static=P_TGV, bc_Far2_mach=0.12, bc_Far2_T_static=295.0, bc_Far2_nx=0.8, bc_Far2_ny=0.6,
         **BC_KEYS),
    # BASELINE.json configs[3]: a MIXED mesh -- a channel with a prism layer on either wall and tetrahedra in the core,
    # isothermal wall below, adiabatic wall above, periodic in x and z; P3 (the configuration's order) with every stage
    # state, P2 with every intermediate of one residual
    case("mixed_p3_channel", n=[2, 3, 2], amp=0.08, level=1, order=3, steps=1, tets="mixed", **MIXED_KEYS),
    case("mixed_p2_channel", n=[2, 3, 2], amp=0.08, level=2, order=2, steps=2, tets="mixed", **MIXED_KEYS),
    # a total-pressure ramp over several time steps: run_input.ramp_counter advances after every step (src/HiFiLES.cpp:224-225)
    case("quad_p3_ramp", dims=2, n=4, amp=0.1, level=1, order=3, steps=3,
         bcs={"y-": "WallT", "y+": "Slip", "x-": "InR", "x+": "Out"}, **BC_KEYS),
]


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    off = 0
    while off < len(data):
        (nl,) = struct.unpack_from("<i", data, off); off += 4
        name = data[off:off + nl].decode(); off += nl
        dtype = chr(data[off]); off += 1
        (nd,) = struct.unpack_from("<i", data, off); off += 4
        dims = struct.unpack_from("<%dq" % nd, data, off); off += 8 * nd
        n = int(np.prod(dims))
        if dtype == "d":
            a = np.frombuffer(data, dtype="<f8", count=n, offset=off); off += 8 * n
        else:
            a = np.frombuffer(data, dtype="<i4", count=n, offset=off); off += 4 * n
        # hf_array is column-major (include/hf_array.h:303-325)
        out[name] = np.array(a).reshape(dims, order="F")
    return out


def run_case(c):
    with tempfile.TemporaryDirectory() as td:
        if c.get("tets") == "mixed":
            xv = write_neu_mixed(os.path.join(td, "mesh.neu"), c["n"], amp=c["amp"])
        elif c.get("tets") == "prisms":
            xv = write_neu_prisms(os.path.join(td, "mesh.neu"), c["n"], amp=c["amp"], curve=c.get("curve"))
        elif c.get("tets"):
            xv = write_neu_tets(os.path.join(td, "mesh.neu"), c["n"], amp=c["amp"], curve=c.get("curve"))
        else:
            xv = write_neu(os.path.join(td, "mesh.neu"), c["n"], c["dims"], amp=c["amp"], bcs=c.get("bcs"))
        keys = dict(c["keys"])
        keys["n_steps"] = c["steps"]
        if c["dims"] == 2:
            keys.pop("dz_cyclic")
        with open(os.path.join(td, "input"), "w") as f:
            for k, v in keys.items():
                f.write("%s %s\n" % (k, repr(v) if isinstance(v, float) else v))
        env = dict(os.environ, HIFILES_HOME=REF_HOME)
        if c.get("restart"):
            env["HFX_DUMP_RESTART"] = "1"
        if c.get("ppts"):
            env["HFX_DUMP_PPTS"] = "1"
        r = subprocess.run([HARNESS, "input", "dump.bin", str(c["steps"]), str(c["level"])],
                           cwd=td, env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout[-3000:] + r.stderr[-3000:])
            raise SystemExit("harness failed for " + c["name"])
        arrs = read_dump(os.path.join(td, "dump.bin"))
    if c.get("keep_every", 1) > 1:
        # long runs: keep the state after every keep_every-th step only
        for k in [k for k in arrs if k.startswith("u_step")]:
            if (int(k.split("_")[1][4:]) + 1) % c["keep_every"] != 0:
                del arrs[k]
    if "restart_ascii" in arrs:
        arrs["restart_ascii"] = arrs["restart_ascii"].astype(np.uint8)
    arrs["xv"] = xv
    meta = dict(name=c["name"], n=c["n"], dims=c["dims"], amp=c["amp"], steps=c["steps"],
                level=c["level"], keys=c["keys"], bcs=c.get("bcs"),
                generator="oracle/capture_golden.py via oracle/_ref/ref_harness (genuine reference)")
    arrs["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(GOLDEN, exist_ok=True)
    out = os.path.join(GOLDEN, c["name"] + ".npz")
    np.savez_compressed(out, **arrs)
    print("%-24s %8.1f kB  %d arrays" % (c["name"], os.path.getsize(out) / 1e3, len(arrs)))


if __name__ == "__main__":
    want = sys.argv[1:]
    for c in CASES:
        if not want or c["name"] in want:
            run_case(c)
