// input.hpp -- the frozen subset of the reference's run_input that the hot path and its
// setup read (/root/reference/include/input.h; keys of src/input.cpp:62-327), plus the
// non-dimensionalisation of input::setup_params (src/input.cpp:527-720).
// Not a global: each `solution` owns one.
#pragma once
#include <string>
#include <vector>

#include "../../../include/hfx.h"
#include "hf_array.hpp"

// one boundary group as the input file gives it: `bc_<name>_type` and the type's parameters, DIMENSIONAL
// (/root/reference/src/input.cpp:328-437)
struct bc_spec
{
  int flag = HFX_BC_CYCLIC; // src/bc.cpp:36-48
  double rho = 0, u = 0, v = 0, w = 0, p_static = 0, T_static = 0, p_total = 0, T_total = 0;
  double nx = 1., ny = 0., nz = 0., mach = 0;
  int pressure_ramp = 0;
  double p_ramp_coeff = 0, T_ramp_coeff = 0, p_total_old = 0, T_total_old = 0;
  bool T_total_given = false, T_total_old_given = false;
};

struct input
{
  // ---- solver
  int equation = 0, viscous = 1, order = 4;
  int riemann_solve_type = 3, vis_riemann_solve_type = 0;
  int ic_form = 7, adv_type = 3, dt_type = 0, n_steps = 1;
  double dt = 0.0, CFL = 0.0;
  double ldg_tau = 0.0, ldg_beta = 0.5;
  // ---- polynomial de-aliasing and shock capturing (src/input.cpp:248-263)
  int over_int = 0, over_int_order = 0;
  int shock_cap = 0, shock_det = 0, shock_det_field = 0;
  double s0 = 0.0, expf_fac = 36.0;
  int expf_order = 4, expf_cutoff = 0;
  // ---- LES closure (src/input.cpp:167-182,666): SGS_model 0 Smagorinsky (needs the wall distance: meshes without walls
  // only here), 1 WALE, 2 WALE + similarity, 3 spectral vanishing viscosity, 4 similarity; the last three filter the
  // solution with filter_upts built from filter_type / filter_ratio
  int LES = 0, SGS_model = 1, filter_type = 0;
  double C_s = 0.0, filter_ratio = 1.0, Kappa = 0.41, prandtl_t = 0.9;
  // ---- plotting: points per edge (src/input.cpp:110; the reference's default is 2)
  int p_res = 2;
  // ---- element parameters
  int upts_type_hexa = 0, vcjh_scheme_hexa = 1;
  double eta_hexa = 0.0;
  // tetrahedra and prisms (src/input.cpp:264-300)
  int upts_type_tet = 0, fpts_type_tet = 0, vcjh_scheme_tet = 1;
  int upts_type_pri_tri = 0, upts_type_pri_1d = 0, vcjh_scheme_pri_1d = 1, vcjh_scheme_tri = 1;
  double eta_pri = 0.0;
  double c_tet = 0.0, c_tri = 0.0; // vcjh_scheme_tet / _tri 0: the filter's c given (src/input.cpp:289,299)
  int upts_type_quad = 0, vcjh_scheme_quad = 1;
  double eta_quad = 0.0;
  // ---- gas
  double gamma = 1.4, prandtl = 0.72, S_gas = 120.0, T_gas = 291.15, R_gas = 286.9, mu_gas = 1.827e-5;
  int fix_vis = 1;
  double Mach_free_stream = 1.0, L_free_stream = 1.0, T_free_stream = 300.0, rho_free_stream = 1.17723946;
  // ---- initial condition (dimensional on input, like the reference)
  double Mach_c_ic = 0.0, nx_c_ic = 1.0, ny_c_ic = 0.0, nz_c_ic = 0.0, T_c_ic = 0.0, rho_c_ic = 0.0;
  double u_c_ic = 0.0, v_c_ic = 0.0, w_c_ic = 0.0, p_c_ic = 0.0; // inviscid inputs
  // optional override of the 1-D solution points (order+1 values); empty: computed
  hf_array<double> loc_1d_upts_override;
  // ---- periodic box
  double dx_cyclic = 0.0, dy_cyclic = 0.0, dz_cyclic = 0.0;

  // ---- derived by setup_params()
  double T_ref = 0, L_ref = 0, rho_ref = 0, uvw_ref = 0, p_ref = 0, mu_ref = 0, time_ref = 0, R_ref = 0;
  double c_sth = 0, mu_inf = 0, rt_inf = 0, uvw_c_ic = 0, mu_c_ic = 0;
  hf_array<double> RK_a, RK_b, RK_c;
  double time = 0.0;

  // ---- boundary groups: bc_specs in, bc_list (non-dimensional records for the device) out
  std::vector<bc_spec> bc_specs;
  std::vector<hfx_bc> bc_list;
  int ramp_counter = 0;
  int pressure_ramp = 0; // 1 when a group ramps its total pressure (src/input.cpp:374-377)
  // input::read_boundary_param (src/input.cpp:328-525) after setup_params()
  int read_boundary_param(std::string &err);
  double bc_R_ref() const { return viscous ? R_ref : R_gas; } // src/bdy_inters.cpp:368-369

  // returns 0 or sets err (the reference's FatalError texts)
  int setup_params(std::string &err);
  int n_rk_stages() const { return adv_type == 0 ? 1 : (adv_type <= 2 ? 4 : (adv_type == 3 ? 5 : 14)); }
  void fill(hfx_params &p) const;
};
