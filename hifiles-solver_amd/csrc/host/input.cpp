#include "input.hpp"

#include <cmath>

// Low-storage RK tableaux.  Values: Carpenter & Kennedy, "Fourth-order 2N-storage
// Runge-Kutta schemes", NASA TM-109112 (1994) for RK45; Niegemann, Diehl & Busch,
// J. Comput. Phys. 231 (2012) for the 14-stage scheme -- the same tableaux the reference
// compiles in from data/RK_coeff.dat:25-79.
static const double RK45_A[5] = {0.0, -567301805773.0 / 1357537059087.0, -2404267990393.0 / 2016746695238.0,
                                 -3550918686646.0 / 2091501179385.0, -1275806237668.0 / 842570457699.0};
static const double RK45_B[5] = {1432997174477.0 / 9575080441755.0, 5161836677717.0 / 13612068292357.0,
                                 1720146321549.0 / 2090206949498.0, 3134564353537.0 / 4481467310338.0,
                                 2277821191437.0 / 14882151754819.0};
static const double RK45_C[5] = {0.0, 1432997174477.0 / 9575080441755.0, 2526269341429.0 / 6820363962896.0,
                                 2006345519317.0 / 3224310063776.0, 2802321613138.0 / 2924317926251.0};
static const double RK414_A[14] = {0.0000000000000000, -0.7188012108672410, -0.7785331173421570, -0.0053282796654044,
                                   -0.8552979934029281, -3.9564138245774565, -1.5780575380587385, -2.0837094552574054,
                                   -0.7483334182761610, -0.7032861106563359, 0.0013917096117681, -0.0932075369637460,
                                   -0.9514200470875948, -7.1151571693922548};
static const double RK414_B[14] = {0.0367762454319673, 0.3136296607553959, 0.1531848691869027, 0.0030097086818182,
                                   0.3326293790646110, 0.2440251405350864, 0.3718879239592277, 0.6204126221582444,
                                   0.1524043173028741, 0.0760894927419266, 0.0077604214040978, 0.0024647284755382,
                                   0.0780348340049386, 5.5059777270269628};
static const double RK414_C[14] = {0.0000000000000000, 0.0367762454319673, 0.1249685262725025, 0.2446177702277698,
                                   0.2476149531070420, 0.2969311120382472, 0.3978149645802642, 0.5270854589440328,
                                   0.6981269994175695, 0.8190890835352128, 0.8527059887098624, 0.8604711817462826,
                                   0.8627060376969976, 0.8734213127600976};

int input::setup_params(std::string &err)
{
  if (equation != 0) { err = "Equation not supported"; return 1; }
  if (riemann_solve_type == 1) { err = "Lax-Friedrich flux not supported with NS/RANS equation"; return 1; }
  if (adv_type < 0 || adv_type > 4) { err = "Time advancement scheme not implemented yet!"; return 1; }

  if (adv_type == 3)
  {
    RK_a.setup(5); RK_b.setup(5); RK_c.setup(5);
    for (int i = 0; i < 5; i++) { RK_a(i) = RK45_A[i]; RK_b(i) = RK45_B[i]; RK_c(i) = RK45_C[i]; }
  }
  else if (adv_type == 4)
  {
    RK_a.setup(14); RK_b.setup(14); RK_c.setup(14);
    for (int i = 0; i < 14; i++) { RK_a(i) = RK414_A[i]; RK_b(i) = RK414_B[i]; RK_c(i) = RK414_C[i]; }
  }
  else
  {
    RK_a.setup(1); RK_b.setup(1); RK_c.setup(adv_type == 0 ? 1 : 4);
  }

  if (viscous && equation == 0)
  {
    // src/input.cpp:596-660
    T_ref = T_free_stream;
    L_ref = L_free_stream;
    rho_ref = rho_free_stream;
    uvw_ref = Mach_free_stream * std::sqrt(gamma * R_gas * T_ref);
    p_ref = rho_ref * uvw_ref * uvw_ref;
    mu_ref = rho_ref * uvw_ref * L_ref;
    time_ref = L_ref / uvw_ref;
    R_ref = (R_gas * T_ref) / (uvw_ref * uvw_ref);
    c_sth = S_gas / T_gas;
    mu_inf = mu_gas / mu_ref;
    rt_inf = T_gas * R_gas / (uvw_ref * uvw_ref);
    if (dt_type == 0) dt /= time_ref;
    dx_cyclic /= L_ref;
    dy_cyclic /= L_ref;
    dz_cyclic /= L_ref;
    uvw_c_ic = Mach_c_ic * std::sqrt(gamma * R_gas * T_c_ic);
    u_c_ic = (uvw_c_ic * nx_c_ic) / uvw_ref;
    v_c_ic = (uvw_c_ic * ny_c_ic) / uvw_ref;
    w_c_ic = (uvw_c_ic * nz_c_ic) / uvw_ref;
    if (fix_vis)
      mu_c_ic = mu_gas;
    else
      mu_c_ic = mu_gas * std::pow(T_c_ic / T_gas, 1.5) * ((T_gas + S_gas) / (T_c_ic + S_gas));
    p_c_ic = rho_c_ic * R_gas * T_c_ic / p_ref;
    mu_c_ic = mu_c_ic / mu_ref;
    rho_c_ic = rho_c_ic / rho_ref;
    T_c_ic = T_c_ic / T_ref;
  }
  else
  {
    T_ref = L_ref = rho_ref = uvw_ref = p_ref = mu_ref = time_ref = R_ref = NAN;
  }
  return 0;
}

void input::fill(hfx_params &p) const
{
  p.gamma = gamma; p.prandtl = prandtl; p.rt_inf = rt_inf; p.mu_inf = mu_inf; p.c_sth = c_sth; p.fix_vis = fix_vis;
  p.ldg_beta = ldg_beta; p.ldg_tau = ldg_tau; p.dt = dt;
  p.viscous = viscous; p.riemann_solve_type = riemann_solve_type; p.vis_riemann_solve_type = vis_riemann_solve_type;
  p.adv_type = adv_type; p.dt_type = dt_type;
  p.n_rk = RK_a.get_dim(0);
  for (int i = 0; i < 16; i++) { p.RK_a[i] = 0; p.RK_b[i] = 0; }
  for (int i = 0; i < p.n_rk; i++) { p.RK_a[i] = RK_a(i); p.RK_b[i] = RK_b(i); }
}

// input::read_boundary_param (src/input.cpp:328-525): defaults, derived values and non-dimensionalisation
int input::read_boundary_param(std::string &err)
{
  bc_list.assign(bc_specs.size(), hfx_bc{});
  ramp_counter = 0;
  pressure_ramp = 0;
  for (size_t i = 0; i < bc_specs.size(); i++)
  {
    const bc_spec &in = bc_specs[i];
    hfx_bc &b = bc_list[i];
    b.flag = in.flag;
    if (in.flag < HFX_BC_SUB_IN_SIMP || in.flag > HFX_BC_SLIP_WALL_DUAL)
    {
      err = "Boundary condition not implemented yet";
      return 1;
    }
    double vel[3] = {in.u, in.v, in.w};
    b.rho = in.rho;
    b.p_static = in.p_static; b.T_static = in.T_static; b.p_total = in.p_total;
    b.T_total = in.T_total;
    b.nx = in.nx; b.ny = in.ny; b.nz = in.nz;
    switch (in.flag)
    {
    case HFX_BC_SUB_IN_SIMP:
      if (viscous)
      {
        b.rho /= rho_ref;
        for (int j = 0; j < 3; j++) vel[j] /= uvw_ref;
      }
      break;
    case HFX_BC_SUB_IN_CHAR:
      b.pressure_ramp = in.pressure_ramp;
      if (in.pressure_ramp)
      {
        pressure_ramp = 1; /* src/input.cpp:376 */
        ramp_counter = 1; /* src/input.cpp:377 */
        b.p_ramp_coeff = in.p_ramp_coeff; b.T_ramp_coeff = in.T_ramp_coeff;
        b.p_total_old = in.p_total_old;
        b.T_total_old = in.T_total_old_given ? in.T_total_old : T_free_stream;
      }
      if (viscous)
      {
        b.T_total /= T_ref; b.p_total /= p_ref;
        if (in.pressure_ramp) { b.p_total_old /= p_ref; b.T_total_old /= T_ref; }
      }
      break;
    case HFX_BC_SUB_OUT_SIMP:
    case HFX_BC_SUB_OUT_CHAR:
      if (!in.T_total_given) b.T_total = T_free_stream; /* src/input.cpp:387 */
      if (viscous) { b.p_static /= p_ref; b.T_total /= T_ref; }
      break;
    case HFX_BC_SUP_IN:
    case HFX_BC_CHAR:
      b.rho = in.p_static / (R_gas * in.T_static);
      vel[0] = in.mach * std::sqrt(gamma * R_gas * in.T_static) * in.nx;
      vel[1] = in.mach * std::sqrt(gamma * R_gas * in.T_static) * in.ny;
      vel[2] = in.mach * std::sqrt(gamma * R_gas * in.T_static) * in.nz;
      if (viscous)
      {
        b.rho /= rho_ref; b.p_static /= p_ref; b.T_static /= T_ref;
        for (int j = 0; j < 3; j++) vel[j] /= uvw_ref;
      }
      break;
    case HFX_BC_ISOTHERM_WALL:
      if (!viscous) { err = "Isothermal wall boundary only available to viscous simulation"; return 1; }
      b.T_static /= T_ref;
      for (int j = 0; j < 3; j++) vel[j] /= uvw_ref;
      break;
    case HFX_BC_ADIABAT_WALL:
      if (!viscous) { err = "Adiabatic wall boundary only available to viscous simulation"; return 1; }
      for (int j = 0; j < 3; j++) vel[j] /= uvw_ref;
      break;
    default:
      break;
    }
    // keep only what the type reads (the reference leaves the other members uninitialised)
    const int f = in.flag;
    const bool has_vel = (f == HFX_BC_SUB_IN_SIMP || f == HFX_BC_SUP_IN || f == HFX_BC_ISOTHERM_WALL || f == HFX_BC_CHAR ||
                          f == HFX_BC_ADIABAT_WALL);
    for (int j = 0; j < 3; j++) b.velocity[j] = has_vel ? vel[j] : 0.0;
    if (!(f == HFX_BC_SUB_IN_SIMP || f == HFX_BC_SUP_IN || f == HFX_BC_CHAR)) b.rho = 0.0;
    if (!(f == HFX_BC_SUB_OUT_SIMP || f == HFX_BC_SUB_OUT_CHAR || f == HFX_BC_SUP_IN || f == HFX_BC_CHAR)) b.p_static = 0.0;
    if (!(f == HFX_BC_SUP_IN || f == HFX_BC_ISOTHERM_WALL || f == HFX_BC_CHAR)) b.T_static = 0.0;
    if (f != HFX_BC_SUB_IN_CHAR) b.p_total = 0.0;
    if (!(f == HFX_BC_SUB_IN_CHAR || f == HFX_BC_SUB_OUT_SIMP || f == HFX_BC_SUB_OUT_CHAR)) b.T_total = 0.0;
    if (!(f == HFX_BC_SUB_IN_CHAR || f == HFX_BC_SUP_IN || f == HFX_BC_CHAR)) b.nx = b.ny = b.nz = 0.0;
  }
  return 0;
}
