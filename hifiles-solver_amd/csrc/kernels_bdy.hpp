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

// ---- ghost state -----------------------------------------------------------------------------
// sol_spec 0: the state the inviscid Riemann problem sees, 1: the state the viscous (LDG) terms see.
// Every boundary type is a function of the left primitives (rho, v, e, p, T, v.n, c), computed once, the
// unit normal and the group's record; the arithmetic of each type follows the cited lines operation by
// operation (the parity tests hold the boundary sweeps to 1e-12 of the reference).
template <int ND>
struct LeftPrim
{
  double rho, v[ND], e, p, T, vn;
};

template <int ND>
__device__ __forceinline__ double sum_sq(const double (&w)[ND])
{
  double s = 0.;
#pragma unroll
  for (int i = 0; i < ND; i++) s += (w[i] * w[i]);
  return s;
}

template <int ND>
__device__ __forceinline__ void bc_state(const int sol_spec, const hfx_bc &bc, const double (&u_l)[ND + 2],
                                         const double (&norm)[ND], const double gamma, const double R_ref,
                                         const int ramp_counter, double (&u_r)[ND + 2])
{
  const double gm1 = gamma - 1.0;
  LeftPrim<ND> L;
  L.rho = u_l[0];
#pragma unroll
  for (int i = 0; i < ND; i++) L.v[i] = u_l[i + 1] / u_l[0];
  L.e = u_l[ND + 1];
  L.p = gm1 * (L.e - 0.5 * L.rho * sum_sq<ND>(L.v));
  L.T = L.p / (L.rho * R_ref);
  L.vn = 0.;
#pragma unroll
  for (int i = 0; i < ND; i++) L.vn += L.v[i] * norm[i];

  double rho_g = 0., e_g = 0., w[ND]; // ghost density, total energy, velocity
#pragma unroll
  for (int i = 0; i < ND; i++) w[i] = 0.;
  // total energy of a ghost state given by pressure, density and its velocity w
  auto energy = [&](double p, double rho) { return (p / gm1) + 0.5 * rho * sum_sq<ND>(w); };
  auto copy_left_velocity = [&]() {
#pragma unroll
    for (int i = 0; i < ND; i++) w[i] = L.v[i];
  };
  auto prescribed_velocity = [&]() {
#pragma unroll
    for (int i = 0; i < ND; i++) w[i] = bc.velocity[i];
  };
  // reflect (factor 2) or remove (factor 1) the wall-normal velocity
  auto wall_normal = [&](double factor) {
#pragma unroll
    for (int i = 0; i < ND; i++) w[i] = L.v[i] - factor * L.vn * norm[i];
  };

  switch (bc.flag)
  {
  case HFX_BC_SUB_IN_SIMP: // :378-400 density and velocity fixed, pressure from inside
    rho_g = bc.rho;
    prescribed_velocity();
    e_g = L.p / gm1 + 0.5 * rho_g * sum_sq<ND>(w);
    break;

  case HFX_BC_SUB_OUT_SIMP: // :404-468 back pressure; reverse flow and supersonic outflow handled apart
  {
    const double machn = fabs(L.vn) / sqrt(gamma * L.p / L.rho);
    if (L.vn < 0)
    {
#pragma unroll
      for (int i = 0; i < ND; i++) w[i] = L.vn * norm[i];
      const double q2 = sum_sq<ND>(w);
      const double T_g = bc.T_total - 0.5 * q2 * gm1 / (R_ref * gamma);
      const double p_g = bc.p_static * pow((1.0 + 0.5 * gm1 * (q2 / (gamma * R_ref * T_g))), -gamma / gm1);
      rho_g = p_g / (R_ref * T_g);
      e_g = (p_g / gm1) + 0.5 * rho_g * q2;
    }
    else if (machn >= 1)
    {
      rho_g = L.rho;
      copy_left_velocity();
      e_g = L.e;
    }
    else
    {
      copy_left_velocity();
      rho_g = L.rho;
      e_g = energy(bc.p_static, rho_g);
    }
    break;
  }

  case HFX_BC_SUB_IN_CHAR: // :475-590 total conditions + outgoing Riemann invariant (with the optional ramps)
  {
    double p0 = bc.p_total, T0 = bc.T_total;
    if (bc.pressure_ramp)
    {
      if (bc.p_ramp_coeff != 0.0)
      {
        p0 = bc.p_total_old + (bc.p_total - bc.p_total_old) * bc.p_ramp_coeff * ramp_counter;
        if (p0 >= bc.p_total) p0 = bc.p_total;
      }
      if (bc.T_ramp_coeff > 0)
      {
        T0 = bc.T_total_old + (bc.T_total - bc.T_total_old) * bc.T_ramp_coeff * ramp_counter;
        if (T0 >= bc.T_total) T0 = bc.T_total;
      }
      else if (bc.T_ramp_coeff < 0)
        T0 = L.T * pow(p0 / L.p, gm1 / gamma);
    }
    const double dir[3] = {bc.nx, bc.ny, bc.nz};
    const double c_l = sqrt(gamma * L.p / L.rho);
    const double R_plus = L.vn + 2.0 * c_l / gm1;
    const double c0sq = gamma * R_ref * T0;
    double alpha = 0.;
#pragma unroll
    for (int i = 0; i < ND; i++) alpha += norm[i] * dir[i];
    const double qa = 1.0 + 0.5 * gm1 * alpha * alpha;
    const double qb = -gm1 * alpha * R_plus;
    const double qc = 0.5 * gm1 * R_plus * R_plus - 2.0 * c0sq / gm1;
    double disc = qb * qb - 4.0 * qa * qc;
    disc = sqrt(disc > 0.0 ? disc : 0.0);
    double speed = (-qb + disc) / (2.0 * qa);
    speed = speed > 0.0 ? speed : 0.0;
    double q2 = speed * speed;
    double csq = c0sq - 0.5 * gm1 * q2;
    double M2 = q2 / (csq);
    M2 = M2 < 1.0 ? M2 : 1.0;
    q2 = M2 * csq;
    speed = sqrt(q2);
    csq = c0sq - 0.5 * gm1 * q2;
#pragma unroll
    for (int i = 0; i < ND; i++) w[i] = speed * dir[i];
    const double T_g = csq / (gamma * R_ref);
    const double p_g = p0 * pow(T_g / T0, gamma / gm1);
    rho_g = p_g / (R_ref * T_g);
    e_g = (p_g / gm1) + 0.5 * rho_g * q2;
    break;
  }

  case HFX_BC_SUB_OUT_CHAR: // :598-640 entropy and the outgoing invariant extrapolated, pressure fixed
  {
    const double c_l = sqrt(gamma * L.p / L.rho);
    const double R_plus = L.vn + 2.0 * c_l / gm1;
    const double entropy = L.p / pow(L.rho, gamma);
    const double p_g = bc.p_static;
    rho_g = pow(p_g / entropy, 1.0 / gamma);
    const double c_g = sqrt(gamma * p_g / rho_g);
    const double vn_g = R_plus - 2.0 * c_g / gm1;
    double q2 = 0.;
#pragma unroll
    for (int i = 0; i < ND; i++)
    {
      w[i] = L.v[i] + (vn_g - L.vn) * norm[i];
      q2 += (w[i] * w[i]);
    }
    e_g = (p_g / gm1) + 0.5 * rho_g * q2;
    break;
  }

  case HFX_BC_SUP_IN: // :643-659 everything prescribed
    rho_g = bc.rho;
    prescribed_velocity();
    e_g = energy(bc.p_static, rho_g);
    break;

  case HFX_BC_SUP_OUT: // :662-669 everything extrapolated
    rho_g = L.rho;
    copy_left_velocity();
    e_g = L.e;
    break;

  case HFX_BC_SLIP_WALL: // :672-701
    rho_g = L.rho;
    wall_normal(sol_spec == 0 ? 2.0 : 1.0);
    e_g = L.p / gm1 + 0.5 * rho_g * sum_sq<ND>(w);
    break;

  case HFX_BC_ISOTHERM_WALL: // :704-792 (wall model off): wall velocity, wall temperature
  case HFX_BC_ADIABAT_WALL:  // :795-860 (wall model off): wall velocity, interior pressure
    rho_g = L.rho;
    if (sol_spec == 0)
    {
#pragma unroll
      for (int i = 0; i < ND; i++) w[i] = 2 * bc.velocity[i] - L.v[i];
    }
    else
      prescribed_velocity();
    if (bc.flag == HFX_BC_ISOTHERM_WALL)
      e_g = rho_g * (R_ref / gm1 * bc.T_static) + 0.5 * rho_g * sum_sq<ND>(w);
    else
      e_g = L.p / gm1 + 0.5 * rho_g * sum_sq<ND>(w);
    break;

  case HFX_BC_CHAR: // :863-960 Riemann invariants against the far-field state
  {
    double vn_far = 0;
#pragma unroll
    for (int i = 0; i < ND; i++) vn_far += bc.velocity[i] * norm[i];
    const double c_l = sqrt(gamma * L.p / L.rho);
    const double c_far = sqrt(gamma * bc.p_static / bc.rho);
    const bool supersonic = fabs(L.vn) / c_l >= 1;
    const bool inflow = L.vn < 0;
    double r_plus, r_minus;
    if (supersonic && inflow)
    {
      r_minus = vn_far - 2. / gm1 * c_far;
      r_plus = vn_far + 2. / gm1 * c_far;
    }
    else if (supersonic)
    {
      r_minus = L.vn - 2. / gm1 * c_l;
      r_plus = L.vn + 2. / gm1 * c_l;
    }
    else
    {
      r_plus = L.vn + 2. / gm1 * c_l;
      r_minus = vn_far - 2. / gm1 * c_far;
    }
    const double c_star = 0.25 * gm1 * (r_plus - r_minus);
    const double vn_star = 0.5 * (r_plus + r_minus);
    // entropy from the far field on inflow, from the interior on outflow; the tangential velocity likewise
    const double inv_entropy = inflow ? pow(bc.rho, gamma) / bc.p_static : pow(L.rho, gamma) / L.p;
    rho_g = pow(1. / gamma * (inv_entropy * c_star * c_star), 1. / gm1);
#pragma unroll
    for (int i = 0; i < ND; i++)
      w[i] = inflow ? vn_star * norm[i] + (bc.velocity[i] - vn_far * norm[i]) : vn_star * norm[i] + (L.v[i] - L.vn * norm[i]);
    const double p_g = rho_g / gamma * c_star * c_star;
    e_g = energy(p_g, rho_g);
    break;
  }

  case HFX_BC_SLIP_WALL_DUAL: // :963-980
    rho_g = L.rho;
    wall_normal(2.0);
    e_g = L.e;
    break;

  default:
    break;
  }
  u_r[0] = rho_g;
#pragma unroll
  for (int i = 0; i < ND; i++) u_r[i + 1] = rho_g * w[i];
  u_r[ND + 1] = e_g;
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
