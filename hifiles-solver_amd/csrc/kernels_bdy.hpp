// kernels_bdy.hpp -- boundary faces (reference class bdy_inters, /root/reference/src/bdy_inters.cpp).
//
// A boundary face has a left side only; the right ("ghost") state is a function of the left state,
// the unit normal and the boundary group's parameters (bdy_inters::set_boundary_conditions,
// :340-1019), the right gradient a function of the left gradient (set_boundary_gradients,
// :1138-1189).  One thread per boundary flux point; the group's record is read through the face's
// boundary_id.  Wall models, the synthetic-eddy LES inlet and the RANS field are not part of the path
// (LES / RANS off) and are refused at registration.
#pragma once
#include "../../include/hfx.h"
#include "hfx_internal.hpp"
#include "physics.hpp"

namespace hfx
{

// ghost state; sol_spec 0: state for the inviscid Riemann problem, 1: state for the viscous (LDG) terms
template <int ND>
__device__ __forceinline__ void bc_state(const int sol_spec, const hfx_bc &bc, const double (&u_l)[ND + 2],
                                         const double (&norm)[ND], const double gamma, const double R_ref,
                                         const int ramp_counter, double (&u_r)[ND + 2])
{
  double rho_l, rho_r = 0., v_l[ND], v_r[ND], e_l, e_r = 0., p_l, p_r, T_l, T_r, vn_l, v_sq, machn_l;
  const int bc_flag = bc.flag;
#pragma unroll
  for (int i = 0; i < ND; i++) v_r[i] = 0.;
  rho_l = u_l[0];
#pragma unroll
  for (int i = 0; i < ND; i++) v_l[i] = u_l[i + 1] / u_l[0];
  e_l = u_l[ND + 1];
  v_sq = 0.;
#pragma unroll
  for (int i = 0; i < ND; i++) v_sq += (v_l[i] * v_l[i]);
  p_l = (gamma - 1.0) * (e_l - 0.5 * rho_l * v_sq);
  T_l = p_l / (rho_l * R_ref);
  vn_l = 0.;
#pragma unroll
  for (int i = 0; i < ND; i++) vn_l += v_l[i] * norm[i];

  if (bc_flag == HFX_BC_SUB_IN_SIMP)
  {
    rho_r = bc.rho;
    for (int i = 0; i < ND; i++) v_r[i] = bc.velocity[i];
    v_sq = 0.;
    for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = p_l / (gamma - 1.0) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_SUB_OUT_SIMP)
  {
    machn_l = fabs(vn_l) / sqrt(gamma * p_l / rho_l);
    if (vn_l < 0)
    {
      for (int i = 0; i < ND; i++) v_r[i] = vn_l * norm[i];
      v_sq = 0.;
      for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
      T_r = bc.T_total - 0.5 * v_sq * (gamma - 1.0) / (R_ref * gamma);
      p_r = bc.p_static * pow((1.0 + 0.5 * (gamma - 1.0) * (v_sq / (gamma * R_ref * T_r))), -gamma / (gamma - 1.0));
      rho_r = p_r / (R_ref * T_r);
      e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
    }
    else if (vn_l >= 0 && machn_l >= 1)
    {
      rho_r = rho_l;
      for (int i = 0; i < ND; i++) v_r[i] = v_l[i];
      e_r = e_l;
    }
    else
    {
      for (int i = 0; i < ND; i++) v_r[i] = v_l[i];
      rho_r = rho_l;
      p_r = bc.p_static;
      v_sq = 0.;
      for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
      e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
    }
  }
  else if (bc_flag == HFX_BC_SUB_IN_CHAR)
  {
    double V_r, c_l, c_r_sq, c_total_sq, R_plus, aa, bb, cc, dd, Mach_sq, alpha, p_total_temp, T_total_temp;
    if (bc.pressure_ramp)
    {
      if (bc.p_ramp_coeff != 0.0)
      {
        p_total_temp = bc.p_total_old + (bc.p_total - bc.p_total_old) * bc.p_ramp_coeff * ramp_counter;
        if (p_total_temp >= bc.p_total) p_total_temp = bc.p_total;
      }
      else
        p_total_temp = bc.p_total;
      if (bc.T_ramp_coeff > 0)
      {
        T_total_temp = bc.T_total_old + (bc.T_total - bc.T_total_old) * bc.T_ramp_coeff * ramp_counter;
        if (T_total_temp >= bc.T_total) T_total_temp = bc.T_total;
      }
      else if (bc.T_ramp_coeff < 0)
        T_total_temp = T_l * pow(p_total_temp / p_l, (gamma - 1.0) / gamma);
      else
        T_total_temp = bc.T_total;
    }
    else
    {
      p_total_temp = bc.p_total;
      T_total_temp = bc.T_total;
    }
    const double n_free_stream[3] = {bc.nx, bc.ny, bc.nz};
    c_l = sqrt(gamma * p_l / rho_l);
    R_plus = vn_l + 2.0 * c_l / (gamma - 1.0);
    c_total_sq = gamma * R_ref * T_total_temp;
    alpha = 0.;
    for (int i = 0; i < ND; i++) alpha += norm[i] * n_free_stream[i];
    aa = 1.0 + 0.5 * (gamma - 1.0) * alpha * alpha;
    bb = -(gamma - 1.0) * alpha * R_plus;
    cc = 0.5 * (gamma - 1.0) * R_plus * R_plus - 2.0 * c_total_sq / (gamma - 1.0);
    dd = bb * bb - 4.0 * aa * cc;
    dd = sqrt(dd > 0.0 ? dd : 0.0);
    V_r = (-bb + dd) / (2.0 * aa);
    V_r = V_r > 0.0 ? V_r : 0.0;
    v_sq = V_r * V_r;
    c_r_sq = c_total_sq - 0.5 * (gamma - 1.0) * v_sq;
    Mach_sq = v_sq / (c_r_sq);
    Mach_sq = Mach_sq < 1.0 ? Mach_sq : 1.0;
    v_sq = Mach_sq * c_r_sq;
    V_r = sqrt(v_sq);
    c_r_sq = c_total_sq - 0.5 * (gamma - 1.0) * v_sq;
    for (int i = 0; i < ND; i++) v_r[i] = V_r * n_free_stream[i];
    T_r = c_r_sq / (gamma * R_ref);
    p_r = p_total_temp * pow(T_r / T_total_temp, gamma / (gamma - 1.0));
    rho_r = p_r / (R_ref * T_r);
    e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_SUB_OUT_CHAR)
  {
    double c_l, c_r, R_plus, s, vn_r;
    c_l = sqrt(gamma * p_l / rho_l);
    R_plus = vn_l + 2.0 * c_l / (gamma - 1.0);
    s = p_l / pow(rho_l, gamma);
    p_r = bc.p_static;
    rho_r = pow(p_r / s, 1.0 / gamma);
    c_r = sqrt(gamma * p_r / rho_r);
    vn_r = R_plus - 2.0 * c_r / (gamma - 1.0);
    v_sq = 0.;
    for (int i = 0; i < ND; i++)
    {
      v_r[i] = v_l[i] + (vn_r - vn_l) * norm[i];
      v_sq += (v_r[i] * v_r[i]);
    }
    e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_SUP_IN)
  {
    rho_r = bc.rho;
    for (int i = 0; i < ND; i++) v_r[i] = bc.velocity[i];
    p_r = bc.p_static;
    v_sq = 0.;
    for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_SUP_OUT)
  {
    rho_r = rho_l;
    for (int i = 0; i < ND; i++) v_r[i] = v_l[i];
    e_r = e_l;
  }
  else if (bc_flag == HFX_BC_SLIP_WALL)
  {
    rho_r = rho_l;
    if (sol_spec == 0)
      for (int i = 0; i < ND; i++) v_r[i] = v_l[i] - 2 * vn_l * norm[i];
    else
      for (int i = 0; i < ND; i++) v_r[i] = v_l[i] - vn_l * norm[i];
    v_sq = 0.;
    for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = p_l / (gamma - 1.0) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_ISOTHERM_WALL)
  {
    T_r = bc.T_static;
    rho_r = rho_l;
    if (sol_spec == 0)
      for (int i = 0; i < ND; i++) v_r[i] = 2 * bc.velocity[i] - v_l[i];
    else
      for (int i = 0; i < ND; i++) v_r[i] = bc.velocity[i];
    v_sq = 0.;
    for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = rho_r * (R_ref / (gamma - 1.0) * T_r) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_ADIABAT_WALL)
  {
    rho_r = rho_l;
    if (sol_spec == 0)
      for (int i = 0; i < ND; i++) v_r[i] = 2 * bc.velocity[i] - v_l[i];
    else
      for (int i = 0; i < ND; i++) v_r[i] = bc.velocity[i];
    v_sq = 0.;
    for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = p_l / (gamma - 1.0) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_CHAR)
  {
    double c_star, vn_star, vn_r, r_plus, r_minus, c_l, c_r, one_over_s, mach;
    vn_r = 0;
    for (int i = 0; i < ND; i++) vn_r += bc.velocity[i] * norm[i];
    c_l = sqrt(gamma * p_l / rho_l);
    c_r = sqrt(gamma * bc.p_static / bc.rho);
    mach = fabs(vn_l) / c_l;
    if (vn_l < 0)
    {
      if (mach >= 1)
      {
        r_minus = vn_r - 2. / (gamma - 1.) * c_r;
        r_plus = vn_r + 2. / (gamma - 1.) * c_r;
      }
      else
      {
        r_plus = vn_l + 2. / (gamma - 1.) * c_l;
        r_minus = vn_r - 2. / (gamma - 1.) * c_r;
      }
      c_star = 0.25 * (gamma - 1.) * (r_plus - r_minus);
      vn_star = 0.5 * (r_plus + r_minus);
      one_over_s = pow(bc.rho, gamma) / bc.p_static;
      rho_r = pow(1. / gamma * (one_over_s * c_star * c_star), 1. / (gamma - 1.));
      for (int i = 0; i < ND; i++) v_r[i] = vn_star * norm[i] + (bc.velocity[i] - vn_r * norm[i]);
    }
    else
    {
      if (mach >= 1)
      {
        r_minus = vn_l - 2. / (gamma - 1.) * c_l;
        r_plus = vn_l + 2. / (gamma - 1.) * c_l;
      }
      else
      {
        r_plus = vn_l + 2. / (gamma - 1.) * c_l;
        r_minus = vn_r - 2. / (gamma - 1.) * c_r;
      }
      c_star = 0.25 * (gamma - 1.) * (r_plus - r_minus);
      vn_star = 0.5 * (r_plus + r_minus);
      one_over_s = pow(rho_l, gamma) / p_l;
      rho_r = pow(1. / gamma * (one_over_s * c_star * c_star), 1. / (gamma - 1.));
      for (int i = 0; i < ND; i++) v_r[i] = vn_star * norm[i] + (v_l[i] - vn_l * norm[i]);
    }
    v_sq = 0.;
    for (int i = 0; i < ND; i++) v_sq += (v_r[i] * v_r[i]);
    p_r = rho_r / gamma * c_star * c_star;
    e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == HFX_BC_SLIP_WALL_DUAL)
  {
    rho_r = rho_l;
    for (int i = 0; i < ND; i++) v_r[i] = v_l[i] - 2 * vn_l * norm[i];
    e_r = e_l;
  }
  u_r[0] = rho_r;
#pragma unroll
  for (int i = 0; i < ND; i++) u_r[i + 1] = rho_r * v_r[i];
  u_r[ND + 1] = e_r;
}

// set_boundary_gradients (src/bdy_inters.cpp:1138-1189); g(field, dim) = g[field + NF*dim]
template <int ND>
__device__ __forceinline__ void bc_gradients(const int bc_flag, const double (&u_r)[ND + 2], const double (&gl)[(ND + 2) * ND],
                                             const double (&norm)[ND], double (&gr)[(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
  if (bc_flag == HFX_BC_CHAR || bc_flag == HFX_BC_SUP_IN || bc_flag == HFX_BC_SUB_IN_SIMP || bc_flag == HFX_BC_SUB_OUT_SIMP)
  {
#pragma unroll
    for (int q = 0; q < NF * ND; q++) gr[q] = 0.;
  }
  else
  {
#pragma unroll
    for (int q = 0; q < NF * ND; q++) gr[q] = gl[q];
  }
  if (bc_flag == HFX_BC_ADIABAT_WALL)
  {
    double v_sq = 0., inte, grad_vel[ND * ND], grad_inte[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) v_sq += (u_r[i + 1] * u_r[i + 1]);
    inte = (u_r[ND + 1] - 0.5 * v_sq / u_r[0]) / u_r[0];
#pragma unroll
    for (int j = 0; j < ND; j++)
#pragma unroll
      for (int i = 0; i < ND; i++) grad_vel[i + ND * j] = (gr[(i + 1) + NF * j] - gr[0 + NF * j] * u_r[i + 1] / u_r[0]) / u_r[0];
#pragma unroll
    for (int i = 0; i < ND; i++)
    {
      double s = inte * gr[0 + NF * i] + 0.5 * v_sq / (u_r[0] * u_r[0]) * gr[0 + NF * i] + u_r[1] * grad_vel[0 + ND * i] +
                 u_r[2] * grad_vel[1 + ND * i];
      if (ND == 3) s = s + u_r[ND] * grad_vel[(ND - 1) + ND * i];
      grad_inte[i] = gr[(ND + 1) + NF * i] - (s);
    }
    double dn = grad_inte[0] * norm[0] + grad_inte[1] * norm[1];
    if (ND == 3) dn = dn + grad_inte[ND - 1] * norm[ND - 1];
#pragma unroll
    for (int i = 0; i < ND; i++) gr[(ND + 1) + NF * i] -= dn * norm[i];
  }
}

struct BdyArgs
{
  long npts; // n_fpts_per_inter * n_inters
  int nfpi;
  const int *L, *boundary_id;
  const hfx_bc *bcs;
  long plane;
  const double *disu, *grad, *norm, *tdA;
  double *tconf, *delta;
  Phys P;
  double R_ref;
  int ramp_counter;
};

__device__ __forceinline__ bool bc_is_wall(int f)
{
  return f == HFX_BC_SLIP_WALL || f == HFX_BC_ISOTHERM_WALL || f == HFX_BC_ADIABAT_WALL || f == HFX_BC_SLIP_WALL_DUAL;
}

// bdy_inters::evaluate_boundaryConditions_invFlux (src/bdy_inters.cpp:213-338)
template <int ND, bool FAST>
__global__ __launch_bounds__(256) void bdy_invflux_kernel(const BdyArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npts) return;
  const long i = q / a.nfpi;
  const long il = a.L[q];
  const hfx_bc bc = a.bcs[a.boundary_id[i]];
  double ul[NF], ur[NF], n[ND], fn[NF];
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm[il + m * a.plane];
#pragma unroll
  for (int k = 0; k < NF; k++) ul[k] = a.disu[il + k * a.plane];
  bc_state<ND>(0, bc, ul, n, a.P.gamma, a.R_ref, a.ramp_counter, ur);
  if (bc.flag == HFX_BC_SLIP_WALL_DUAL)
  {
    double fl[NF * ND];
    calc_invf<ND, FAST>(a.P.gamma, ul, fl);
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      fn[k] = 0.;
#pragma unroll
      for (int l = 0; l < ND; l++) fn[k] += fl[k + NF * l] * n[l];
    }
  }
  else if (FAST)
  {
    if (a.P.riemann == 0)
      riemann_flux_t<ND, 0, true>(a.P, ul, ur, n, fn);
    else if (a.P.riemann == 2)
      riemann_flux_t<ND, 2, true>(a.P, ul, ur, n, fn);
    else
      riemann_flux_t<ND, 3, true>(a.P, ul, ur, n, fn);
  }
  else
    riemann_flux<ND>(a.P, ul, ur, n, fn);
  const double tl = a.tdA[il];
#pragma unroll
  for (int k = 0; k < NF; k++) a.tconf[il + k * a.plane] = fn[k] * tl;
  if (a.P.viscous)
  {
    if (bc_is_wall(bc.flag)) bc_state<ND>(1, bc, ul, n, a.P.gamma, a.R_ref, a.ramp_counter, ur);
    // ldg_solution(flux_spec 1): u_c = u_r
#pragma unroll
    for (int k = 0; k < NF; k++) a.delta[il + k * a.plane] = (ur[k] - ul[k]);
  }
}

// bdy_inters::evaluate_boundaryConditions_viscFlux (src/bdy_inters.cpp:1024-1136), faces without wall model
template <int ND, bool FAST>
__global__ __launch_bounds__(256) void bdy_viscflux_kernel(const BdyArgs a)
{
  constexpr int NF = ND + 2, NG = NF * ND;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npts) return;
  const long i = q / a.nfpi;
  const long il = a.L[q];
  const hfx_bc bc = a.bcs[a.boundary_id[i]];
  if (bc.flag == HFX_BC_SLIP_WALL) return;
  double ul[NF], ur[NF], n[ND], gl[NG], gr[NG], fr[NG];
#pragma unroll
  for (int k = 0; k < NF; k++) ul[k] = a.disu[il + k * a.plane];
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm[il + m * a.plane];
#pragma unroll
  for (int s = 0; s < NG; s++) gl[s] = a.grad[il + s * a.plane];
  bc_state<ND>(1, bc, ul, n, a.P.gamma, a.R_ref, a.ramp_counter, ur);
  bc_gradients<ND>(bc.flag, ur, gl, n, gr);
  calc_visf<ND, FAST>(a.P, ur, gr, fr);
  const double tl = a.tdA[il];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    // ldg_flux(flux_spec 1): f_c = f_r
    double fn = 0.0;
#pragma unroll
    for (int l = 0; l < ND; l++) fn += fr[k + NF * l] * n[l];
    fn -= a.P.ldg_tau * (ur[k] - ul[k]);
    a.tconf[il + k * a.plane] += fn * tl;
  }
}

} // namespace hfx
