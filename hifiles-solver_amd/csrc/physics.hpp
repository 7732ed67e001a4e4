// physics.hpp -- point physics of the HiFiLES hot path as gfx950 device functions.
//
// Restated from the behaviour of the reference's CPU branch:
//   inviscid / viscous point fluxes   /root/reference/src/flux.cpp:33,74,129,257
//   Rusanov / RoeM / HLLC             /root/reference/src/inters.cpp:277,327,439
//   LDG common solution / flux        /root/reference/src/inters.cpp:561,615
// All values live in registers (fully unrolled small arrays); ND is a template
// parameter so the 2-D and 3-D variants are separate straight-line code.
#pragma once
#include <hip/hip_runtime.h>

namespace hfx
{

// physics scalars handed to kernels by value (a snapshot of the reference's
// global `run_input`, include/input.h)
struct Phys
{
  double gamma, prandtl, rt_inf, mu_inf, c_sth, fix_vis, ldg_beta, ldg_tau;
  int riemann, viscous;
};

// f(k,m) = f[k + NF*m]
// FAST (fused path only): divisions by the density become multiplications by its reciprocal
// (one v_rcp + Newton step instead of ND full IEEE divisions) -- a <= 1 ulp change per quotient.
template <int ND, bool FAST = false>
__device__ __forceinline__ void calc_invf(const double gamma, const double (&u)[ND + 2], double (&f)[(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
  const double rho = u[0];
  double v[ND];
  double vsq = 0.0;
  if (FAST)
  {
    const double ir = 1.0 / rho;
#pragma unroll
    for (int d = 0; d < ND; d++) v[d] = u[d + 1] * ir;
  }
  else
  {
#pragma unroll
    for (int d = 0; d < ND; d++) v[d] = u[d + 1] / rho;
  }
  // ((vx*vx)+(vy*vy))+(vz*vz), the reference's association (flux.cpp:43,87)
  vsq = v[0] * v[0] + v[1] * v[1];
  if (ND == 3) vsq = vsq + v[ND - 1] * v[ND - 1];
  const double p = (gamma - 1.0) * (u[ND + 1] - (0.5 * u[0] * vsq));
#pragma unroll
  for (int m = 0; m < ND; m++)
  {
    f[0 + NF * m] = u[m + 1];
#pragma unroll
    for (int d = 0; d < ND; d++)
    {
      const double mom = u[d + 1] * v[m];
      f[(d + 1) + NF * m] = (d == m) ? (p + mom) : mom;
    }
    f[(ND + 1) + NF * m] = v[m] * (u[ND + 1] + p);
  }
}

// viscosity law shared by flux and time-step code (flux.cpp:319-321)
template <bool FAST = false>
__device__ __forceinline__ double viscosity(const Phys &P, const double inte)
{
  // fix_vis == 1: mu + 1*(mu_inf - mu) is mu_inf up to one rounding; the fused path skips the
  // Sutherland evaluation (sqrt + 2 divisions) altogether in that case
  if (FAST && P.fix_vis == 1.0) return P.mu_inf;
  const double rt_ratio = (P.gamma - 1.0) * inte / P.rt_inf;
  // rt_ratio^1.5 as x*sqrt(x): within 1 ulp of the reference's pow(x,1.5) at a fraction of its
  // instruction and register cost (the library pow expands to ~100 instructions per call)
  double mu = P.mu_inf * (rt_ratio * sqrt(rt_ratio)) * (1.0 + P.c_sth) / (rt_ratio + P.c_sth);
  mu = mu + P.fix_vis * (P.mu_inf - mu);
  return mu;
}

// grad_u(k,m) = g[k + NF*m]; RANS off (mu_t = 0)
template <int ND, bool FAST = false>
__device__ __forceinline__ void calc_visf(const Phys &P, const double (&u)[ND + 2], const double (&g)[(ND + 2) * ND],
                                          double (&f)[(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
  const double rho = u[0];
  const double ir = FAST ? 1.0 / rho : 0.0;
  double v[ND];
  double ke2 = 0.0; // u*u+v*v+w*w
#pragma unroll
  for (int d = 0; d < ND; d++) v[d] = FAST ? u[d + 1] * ir : u[d + 1] / rho;
  ke2 = v[0] * v[0] + v[1] * v[1];
  if (ND == 3) ke2 = ke2 + v[ND - 1] * v[ND - 1];
  const double inte = (FAST ? u[ND + 1] * ir : u[ND + 1] / rho) - 0.5 * ke2;
  const double mu = viscosity<FAST>(P, inte);

  // velocity gradients dv[d][m] = d v_d / d x_m
  double dv[ND][ND];
#pragma unroll
  for (int d = 0; d < ND; d++)
#pragma unroll
    for (int m = 0; m < ND; m++)
    {
      const double num = g[(d + 1) + NF * m] - g[0 + NF * m] * v[d];
      dv[d][m] = FAST ? num * ir : num / rho;
    }

  double de[ND];
#pragma unroll
  for (int m = 0; m < ND; m++)
  {
    double conv = v[0] * dv[0][m] + v[1] * dv[1][m];
    if (ND == 3) conv = conv + v[ND - 1] * dv[ND - 1][m];
    const double dke = 0.5 * ke2 * g[0 + NF * m] + rho * conv;
    const double num = g[(ND + 1) + NF * m] - dke - g[0 + NF * m] * inte;
    de[m] = FAST ? num * ir : num / rho;
  }

  double divv = dv[0][0] + dv[1][1];
  if (ND == 3) divv = divv + dv[ND - 1][ND - 1];
  const double diag = FAST ? divv * (1.0 / 3.0) : divv / 3.0;

  double tau[ND][ND];
#pragma unroll
  for (int a = 0; a < ND; a++)
#pragma unroll
    for (int b = 0; b < ND; b++)
      tau[a][b] = (a == b) ? 2.0 * mu * (dv[a][a] - diag) : mu * ((a < b) ? (dv[a][b] + dv[b][a]) : (dv[b][a] + dv[a][b]));

  const double kap = (mu / P.prandtl) * P.gamma;
#pragma unroll
  for (int m = 0; m < ND; m++)
  {
    f[0 + NF * m] = 0.0;
#pragma unroll
    for (int d = 0; d < ND; d++) f[(d + 1) + NF * m] = -tau[d][m];
    double work = v[0] * tau[0][m] + v[1] * tau[1][m];
    if (ND == 3) work = work + v[ND - 1] * tau[ND - 1][m];
    f[(ND + 1) + NF * m] = -(work + kap * de[m]);
  }
}

// calc_visf<ND, true> for TWO independent states at once, every statement issued for both before the next one: a wave
// that has nothing else to run beside it (the fused flux kernel: one heavy wave per SIMD) then finds an independent
// instruction behind every dependent one.  Same operations and order per state as calc_visf<ND, true>.
template <int ND>
__device__ __forceinline__ void calc_visf_pair(const Phys &P, const double (&u)[2][ND + 2], const double (&g)[2][(ND + 2) * ND],
                                               double (&f)[2][(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
#define HFX_P2 _Pragma("unroll") for (int p = 0; p < 2; p++)
  double ir[2], v[2][ND], ke2[2], inte[2], mu[2];
  HFX_P2 ir[p] = 1.0 / u[p][0];
#pragma unroll
  for (int d = 0; d < ND; d++) HFX_P2 v[p][d] = u[p][d + 1] * ir[p];
  HFX_P2 ke2[p] = v[p][0] * v[p][0] + v[p][1] * v[p][1];
  if (ND == 3) HFX_P2 ke2[p] = ke2[p] + v[p][ND - 1] * v[p][ND - 1];
  HFX_P2 inte[p] = u[p][ND + 1] * ir[p] - 0.5 * ke2[p];
  if (P.fix_vis == 1.0)
  {
    HFX_P2 mu[p] = P.mu_inf;
  }
  else
  {
    HFX_P2 mu[p] = viscosity<true>(P, inte[p]);
  }
  double dv[2][ND][ND];
#pragma unroll
  for (int d = 0; d < ND; d++)
#pragma unroll
    for (int m = 0; m < ND; m++) HFX_P2 dv[p][d][m] = (g[p][(d + 1) + NF * m] - g[p][0 + NF * m] * v[p][d]) * ir[p];
  double de[2][ND];
#pragma unroll
  for (int m = 0; m < ND; m++)
  {
    double conv[2], dke[2];
    HFX_P2 conv[p] = v[p][0] * dv[p][0][m] + v[p][1] * dv[p][1][m];
    if (ND == 3) HFX_P2 conv[p] = conv[p] + v[p][ND - 1] * dv[p][ND - 1][m];
    HFX_P2 dke[p] = 0.5 * ke2[p] * g[p][0 + NF * m] + u[p][0] * conv[p];
    HFX_P2 de[p][m] = (g[p][(ND + 1) + NF * m] - dke[p] - g[p][0 + NF * m] * inte[p]) * ir[p];
  }
  double diag[2];
  HFX_P2 diag[p] = dv[p][0][0] + dv[p][1][1];
  if (ND == 3) HFX_P2 diag[p] = diag[p] + dv[p][ND - 1][ND - 1];
  HFX_P2 diag[p] = diag[p] * (1.0 / 3.0);
  double tau[2][ND][ND];
#pragma unroll
  for (int a = 0; a < ND; a++)
#pragma unroll
    for (int b = 0; b < ND; b++)
      HFX_P2 tau[p][a][b] = (a == b) ? 2.0 * mu[p] * (dv[p][a][a] - diag[p])
                                     : mu[p] * ((a < b) ? (dv[p][a][b] + dv[p][b][a]) : (dv[p][b][a] + dv[p][a][b]));
  double kap[2];
  HFX_P2 kap[p] = (mu[p] / P.prandtl) * P.gamma;
#pragma unroll
  for (int m = 0; m < ND; m++)
  {
    HFX_P2 f[p][0 + NF * m] = 0.0;
#pragma unroll
    for (int d = 0; d < ND; d++) HFX_P2 f[p][(d + 1) + NF * m] = -tau[p][d][m];
    double work[2];
    HFX_P2 work[p] = v[p][0] * tau[p][0][m] + v[p][1] * tau[p][1][m];
    if (ND == 3) HFX_P2 work[p] = work[p] + v[p][ND - 1] * tau[p][ND - 1][m];
    HFX_P2 f[p][(ND + 1) + NF * m] = -(work[p] + kap[p] * de[p][m]);
  }
#undef HFX_P2
}

template <int ND>
__device__ __forceinline__ void normal_flux(const double (&f)[(ND + 2) * ND], const double (&n)[ND], double (&fn)[ND + 2])
{
  constexpr int NF = ND + 2;
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < ND; l++) s += f[k + NF * l] * n[l];
    fn[k] = s;
  }
}

// left/right primitive helpers
template <int ND>
struct Side
{
  double v[ND], vn, vsq, p, h;
};

template <int ND, bool FAST = false>
__device__ __forceinline__ Side<ND> side_state(const double gamma, const double (&u)[ND + 2], const double (&n)[ND])
{
  Side<ND> s;
  s.vn = 0.0;
  s.vsq = 0.0;
  const double ir = FAST ? 1.0 / u[0] : 0.0;
#pragma unroll
  for (int i = 0; i < ND; i++)
  {
    s.v[i] = FAST ? u[i + 1] * ir : u[i + 1] / u[0];
    s.vn += s.v[i] * n[i];
    s.vsq += s.v[i] * s.v[i];
  }
  s.p = (gamma - 1.0) * (u[ND + 1] - 0.5 * u[0] * s.vsq);
  s.h = FAST ? (u[ND + 1] + s.p) * ir : (u[ND + 1] + s.p) / u[0];
  return s;
}

// inters.cpp:277-324
template <int ND, bool FAST = false>
__device__ __forceinline__ void rusanov_flux(const double gamma, const double (&ul)[ND + 2], const double (&ur)[ND + 2],
                                             const double (&fnl)[ND + 2], const double (&fnr)[ND + 2],
                                             const double (&n)[ND], double (&fn)[ND + 2])
{
  const Side<ND> L = side_state<ND, FAST>(gamma, ul, n), R = side_state<ND, FAST>(gamma, ur, n);
  const double eig = sqrt(gamma * (L.p + R.p) / (ul[0] + ur[0])) + 0.5 * fabs(L.vn + R.vn);
#pragma unroll
  for (int k = 0; k < ND + 2; k++) fn[k] = 0.5 * ((fnl[k] + fnr[k]) - eig * (ur[k] - ul[k]));
}

// inters.cpp:327-437
template <int ND, bool FAST = false>
__device__ __forceinline__ void roeM_flux(const double gamma, const double (&ul)[ND + 2], const double (&ur)[ND + 2],
                                          const double (&fnl)[ND + 2], const double (&fnr)[ND + 2],
                                          const double (&n)[ND], double (&fn)[ND + 2])
{
  constexpr int NF = ND + 2;
  const Side<ND> L = side_state<ND, FAST>(gamma, ul, n), R = side_state<ND, FAST>(gamma, ur, n);
  const double drho = ur[0] - ul[0], dp = R.p - L.p, dh = R.h - L.h, dvn = R.vn - L.vn;
  const double sq_rho = sqrt(ur[0] / ul[0]);
  const double rrho = 1.0 / (1.0 + sq_rho);
  const double ratr = sq_rho * rrho;
  const double ra = sq_rho * ul[0];
  const double ha = L.h * rrho + R.h * ratr;
  double va[ND], qq = 0.0, va_n = 0.0;
#pragma unroll
  for (int i = 0; i < ND; i++)
  {
    va[i] = L.v[i] * rrho + R.v[i] * ratr;
    qq += va[i] * va[i];
    va_n += n[i] * va[i];
  }
  const double aa = sqrt((gamma - 1.0) * (ha - 0.5 * qq));
  const double rcp_aa = 1.0 / aa;
  const double abs_ma = fabs(va_n * rcp_aa);
  double b1 = fmax(0.0, fmax(va_n + aa, R.vn + aa));
  double b2 = fmin(0.0, fmin(va_n - aa, L.vn - aa));
  double b1b2 = b1 * b2;
  const double rcp = 1.0 / (b1 - b2);
  b1 = b1 * rcp;
  b2 = b2 * rcp;
  b1b2 = b1b2 * rcp;
  const double hh = 1.0 - ((L.p < R.p) ? (L.p / R.p) : (R.p / L.p));
  const double ff = (abs_ma != 0.0) ? pow(abs_ma, hh) : 1.0;
  const double gg = ff / (1.0 + abs_ma);
  double du[NF], bdq[NF];
#pragma unroll
  for (int i = 0; i < NF - 1; i++) du[i] = ur[i] - ul[i];
  du[ND + 1] = ur[0] * R.h - ul[0] * L.h;
  bdq[0] = drho - ff * dp * rcp_aa * rcp_aa;
  bdq[ND + 1] = bdq[0] * ha + ra * dh;
#pragma unroll
  for (int i = 0; i < ND; i++) bdq[i + 1] = bdq[0] * va[i] + ra * ((R.v[i] - L.v[i]) - n[i] * dvn);
#pragma unroll
  for (int i = 0; i < NF; i++) fn[i] = (b1 * fnl[i] - b2 * fnr[i]) + b1b2 * (du[i] - gg * bdq[i]);
}

// inters.cpp:439-532.  NOTE a_m uses the NORMAL Roe velocity only (inters.cpp:497) -- kept as is.
template <int ND, bool FAST = false>
__device__ __forceinline__ void hllc_flux(const double gamma, const double (&ul)[ND + 2], const double (&ur)[ND + 2],
                                          const double (&fnl)[ND + 2], const double (&fnr)[ND + 2],
                                          const double (&n)[ND], double (&fn)[ND + 2])
{
  const Side<ND> L = side_state<ND, FAST>(gamma, ul, n), R = side_state<ND, FAST>(gamma, ur, n);
  const double sq_rho = sqrt(ur[0] / ul[0]);
  const double rrho = 1.0 / (sq_rho + 1.0);
  const double vn_m = rrho * (L.vn + sq_rho * R.vn);
  const double h_m = rrho * (L.h + sq_rho * R.h);
  const double a_m = sqrt((gamma - 1.0) * (h_m - 0.5 * vn_m * vn_m));
  const double S_R = vn_m + a_m;
  const double S_L = vn_m - a_m;
  const double S_star = (R.p - L.p + ul[0] * L.vn * (S_L - L.vn) - ur[0] * R.vn * (S_R - R.vn)) /
                        (ul[0] * (S_L - L.vn) - ur[0] * (S_R - R.vn));
  if (S_L >= 0)
  {
#pragma unroll
    for (int k = 0; k < ND + 2; k++) fn[k] = fnl[k];
  }
  else if (S_star >= 0)
  {
    const double rcp_star = S_L - S_star;
    const double pst = S_L * (L.p + ul[0] * (S_L - L.vn) * (S_star - L.vn));
    if (FAST)
    {
      const double is = 1.0 / rcp_star;
      fn[0] = S_star * (S_L * ul[0] - fnl[0]) * is;
#pragma unroll
      for (int i = 0; i < ND; i++) fn[i + 1] = (S_star * (S_L * ul[i + 1] - fnl[i + 1]) + pst * n[i]) * is;
      fn[ND + 1] = (S_star * (S_L * ul[ND + 1] - fnl[ND + 1]) + pst * S_star) * is;
    }
    else
    {
      fn[0] = S_star * (S_L * ul[0] - fnl[0]) / rcp_star;
#pragma unroll
      for (int i = 0; i < ND; i++) fn[i + 1] = (S_star * (S_L * ul[i + 1] - fnl[i + 1]) + pst * n[i]) / rcp_star;
      fn[ND + 1] = (S_star * (S_L * ul[ND + 1] - fnl[ND + 1]) + pst * S_star) / rcp_star;
    }
  }
  else if (S_R >= 0)
  {
    const double rcp_star = S_R - S_star;
    const double pst = S_R * (R.p + ur[0] * (S_R - R.vn) * (S_star - R.vn));
    if (FAST)
    {
      const double is = 1.0 / rcp_star;
      fn[0] = S_star * (S_R * ur[0] - fnr[0]) * is;
#pragma unroll
      for (int i = 0; i < ND; i++) fn[i + 1] = (S_star * (S_R * ur[i + 1] - fnr[i + 1]) + pst * n[i]) * is;
      fn[ND + 1] = (S_star * (S_R * ur[ND + 1] - fnr[ND + 1]) + pst * S_star) * is;
    }
    else
    {
      fn[0] = S_star * (S_R * ur[0] - fnr[0]) / rcp_star;
#pragma unroll
      for (int i = 0; i < ND; i++) fn[i + 1] = (S_star * (S_R * ur[i + 1] - fnr[i + 1]) + pst * n[i]) / rcp_star;
      fn[ND + 1] = (S_star * (S_R * ur[ND + 1] - fnr[ND + 1]) + pst * S_star) / rcp_star;
    }
  }
  else
  {
#pragma unroll
    for (int k = 0; k < ND + 2; k++) fn[k] = fnr[k];
  }
}

// Riemann dispatch (int_inters.cpp:185-205); fnl/fnr are the normal point fluxes of the two states
template <int ND>
__device__ __forceinline__ void riemann_flux(const Phys &P, const double (&ul)[ND + 2], const double (&ur)[ND + 2],
                                             const double (&n)[ND], double (&fn)[ND + 2])
{
  constexpr int NF = ND + 2;
  double fl[NF * ND], fr[NF * ND], fnl[NF], fnr[NF];
  calc_invf<ND>(P.gamma, ul, fl);
  calc_invf<ND>(P.gamma, ur, fr);
  normal_flux<ND>(fl, n, fnl);
  normal_flux<ND>(fr, n, fnr);
  if (P.riemann == 0)
    rusanov_flux<ND>(P.gamma, ul, ur, fnl, fnr, n, fn);
  else if (P.riemann == 2)
    roeM_flux<ND>(P.gamma, ul, ur, fnl, fnr, n, fn);
  else
    hllc_flux<ND>(P.gamma, ul, ur, fnl, fnr, n, fn);
}

// the same with the solver fixed at compile time (RS = riemann_solve_type): only one solver's
// code and registers end up in the kernel
template <int ND, int RS, bool FAST = false>
__device__ __forceinline__ void riemann_flux_t(const Phys &P, const double (&ul)[ND + 2], const double (&ur)[ND + 2],
                                               const double (&n)[ND], double (&fn)[ND + 2])
{
  constexpr int NF = ND + 2;
  double fnl[NF], fnr[NF];
  {
    double fq[NF * ND];
    calc_invf<ND, FAST>(P.gamma, ul, fq);
    normal_flux<ND>(fq, n, fnl);
  }
  {
    double fq[NF * ND];
    calc_invf<ND, FAST>(P.gamma, ur, fq);
    normal_flux<ND>(fq, n, fnr);
  }
  if (RS == 0)
    rusanov_flux<ND, FAST>(P.gamma, ul, ur, fnl, fnr, n, fn);
  else if (RS == 2)
    roeM_flux<ND, FAST>(P.gamma, ul, ur, fnl, fnr, n, fn);
  else
    hllc_flux<ND, FAST>(P.gamma, ul, ur, fnl, fnr, n, fn);
}

// the "consistent switch" (inters.cpp:568-581, :620-633): exact floating-point
// zero tests on the LEFT element's unit normal.
template <int ND>
__device__ __forceinline__ double ldg_switch(double beta, const double (&n)[ND])
{
  if (beta != 0.0)
  {
    if (n[0] < 0.0)
      beta = -beta;
    else if (n[0] == 0.0)
    {
      if ((n[0] + n[1]) < 0.0)
        beta = -beta;
      else if ((n[0] + n[1]) == 0.0)
      {
        if (ND == 3 && (n[0] + n[ND - 1]) < 0.0) beta = -beta;
      }
    }
  }
  return beta;
}

// LES eddy-viscosity closure, eles::calc_sgsf_upts (src/eles.cpp:2395-2650): sgs_model 0 Smagorinsky with
// near-wall damping (y = distance to the nearest no-slip wall), 1 WALE.  f(k,m) = f[k + NF*m].
// 2 WALE + similarity, 4 similarity: the Leonard terms Lu (n_upts,n_eles,3|6), Le (n_upts,n_eles,n_dims) of the step's
// calc_sgs_terms join the flux; 3 (spectral vanishing viscosity): no SGS flux, the solution itself is filtered.
struct LesParams
{
  int sgs_model, order;
  double C_s, filter_ratio, Kappa, prandtl_t;
  double vol_factor; // calc_ele_vol / detjac: the reference element's volume (hexes 8, quads 4, prisms 4, tetrahedra 8/6, triangles 2)
  const double *Lu, *Le;
};

// p, plane: the point's offset and the (point, element) plane size of Lu / Le (similarity terms only)
template <int ND>
__device__ __forceinline__ void calc_sgsf(const Phys &P, const LesParams &Lp, const double (&tu)[ND + 2],
                                          const double (&g)[(ND + 2) * ND], const double detjac, const double y, const long p,
                                          const long plane, double (&sg)[(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
  double u[ND], drho[ND], dene[ND], dke[ND], de[ND], dmom[ND][ND], du[ND][ND], S[ND][ND];
  const double rho = tu[0];
  double ke = 0.;
#pragma unroll
  for (int i = 0; i < ND; i++)
  {
    u[i] = tu[i + 1] / rho;
    ke += 0.5 * (u[i] * u[i]);
  }
  const double inte = tu[NF - 1] / rho - ke;
  const double vol = detjac * Lp.vol_factor; // calc_ele_vol of the class (src/eles_hexas.cpp:1542, eles_pris.cpp:1525, eles_tets.cpp:1589)
  const double delta = Lp.filter_ratio * pow(vol, 1. / ND) / (Lp.order + 1.);
#pragma unroll
  for (int i = 0; i < ND; i++)
  {
    drho[i] = g[0 + NF * i];
    dene[i] = g[(NF - 1) + NF * i];
#pragma unroll
    for (int j = 1; j < NF - 1; j++) dmom[i][j - 1] = g[j + NF * i];
  }
#pragma unroll
  for (int i = 0; i < ND; i++)
  {
    dke[i] = ke * drho[i];
#pragma unroll
    for (int j = 0; j < ND; j++)
    {
      du[i][j] = (dmom[i][j] - u[j] * drho[i]) / rho;
      dke[i] += rho * u[j] * du[i][j];
    }
    de[i] = (dene[i] - dke[i] - drho[i] * inte) / rho;
  }
#pragma unroll
  for (int i = 0; i < ND; i++)
#pragma unroll
    for (int j = 0; j < ND; j++) S[i][j] = (du[i][j] + du[j][i]) / 2.0;
  const bool eddy = Lp.sgs_model <= 2, sim = Lp.sgs_model == 2 || Lp.sgs_model == 4; /* src/eles.cpp:2436-2465 */
  double mu_t = 0.0;
  if (!eddy)
    ;
  else if (Lp.sgs_model == 0)
  {
    double Smod = 0.0;
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++) Smod += 2.0 * S[i][j] * S[i][j];
    Smod = sqrt(Smod);
    mu_t = rho * fmin(y * y * Lp.Kappa * Lp.Kappa, Lp.C_s * Lp.C_s * delta * delta) * Smod;
  }
  else
  {
    double num = 0.0, denom = 0.0, Sq[ND][ND], gT[ND][ND];
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++)
      {
        double s = 0.;
#pragma unroll
        for (int l = 0; l < ND; l++) s += du[i][l] * du[l][j];
        gT[i][j] = s;
      }
    double diag = 0.0;
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++)
      {
        Sq[i][j] = 0.;
        Sq[i][j] += 0.5 * gT[j][i];
        Sq[i][j] += 0.5 * gT[i][j];
      }
#pragma unroll
    for (int i = 0; i < ND; i++) diag += gT[i][i] / 3.0;
#pragma unroll
    for (int i = 0; i < ND; i++) Sq[i][i] -= diag;
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++)
      {
        num += Sq[i][j] * Sq[i][j];
        denom += S[i][j] * S[i][j];
      }
    denom = pow(denom, 2.5) + pow(num, 1.25);
    num = pow(num, 1.5);
    mu_t = rho * Lp.C_s * Lp.C_s * delta * delta * num / (denom + 1.e-12);
  }
#pragma unroll
  for (int q = 0; q < NF * ND; q++) sg[q] = 0.0;
  if (eddy)
  {
    double diag = 0.;
#pragma unroll
    for (int i = 0; i < ND; i++) diag += S[i][i] / 3.0;
#pragma unroll
    for (int i = 0; i < ND; i++) S[i][i] -= diag;
#pragma unroll
    for (int j = 0; j < ND; j++)
    {
      sg[0 + NF * j] = 0.0;
      double ef = -1.0 * P.gamma * mu_t / Lp.prandtl_t * de[j];
#pragma unroll
      for (int k = 0; k < ND; k++) ef -= u[k] * 2.0 * mu_t * S[k][j];
      sg[(NF - 1) + NF * j] = ef;
#pragma unroll
      for (int i = 1; i < NF - 1; i++) sg[i + NF * j] = -2.0 * mu_t * S[i - 1][j];
    }
  }
  if (sim)
  {
    // src/eles.cpp:2602-2634; the off-diagonal momentum entries are filled from their transposes AFTER those received
    // their own term (and the eddy part), as the reference does
#pragma unroll
    for (int j = 0; j < ND; j++) sg[(NF - 1) + NF * j] += P.gamma * rho * Lp.Le[p + j * plane];
    if (ND == 2)
    {
      sg[1 + NF * 0] += rho * Lp.Lu[p + 0 * plane];
      sg[1 + NF * 1] += rho * Lp.Lu[p + 2 * plane];
      sg[2 + NF * 0] += sg[1 + NF * 1];
      sg[2 + NF * 1] += rho * Lp.Lu[p + 1 * plane];
    }
    else
    {
      sg[1 + NF * 0] += rho * Lp.Lu[p + 0 * plane];
      sg[1 + NF * 1] += rho * Lp.Lu[p + 3 * plane];
      sg[1 + NF * (ND - 1)] += rho * Lp.Lu[p + 4 * plane];
      sg[2 + NF * 0] += sg[1 + NF * 1];
      sg[2 + NF * 1] += rho * Lp.Lu[p + 1 * plane];
      sg[2 + NF * (ND - 1)] += rho * Lp.Lu[p + 5 * plane];
      sg[(NF - 2) + NF * 0] += sg[1 + NF * (ND - 1)];
      sg[(NF - 2) + NF * 1] += sg[2 + NF * (ND - 1)];
      sg[(NF - 2) + NF * (ND - 1)] += rho * Lp.Lu[p + 2 * plane];
    }
  }
}

// The same closure for the fused flux kernel (split_flux_tensor_kernel<..., LES>), where it runs as ONE dependency chain per
// thread with nothing beside it: the squared length scale of the point -- min(y^2 Kappa^2, C_s^2 Delta^2) of the damped
// Smagorinsky model, C_s^2 Delta^2 of WALE; it depends on the metrics and the wall distance only -- comes precomputed
// (len2, evaluated once on the host exactly as above), the divisions by rho, 3 and Pr_t are multiplications by reciprocals and
// the reference's pow() calls with the exponents 5/2, 5/4, 3/2 are sqrt products: the same quantities to rounding.
template <int ND>
__device__ __forceinline__ void calc_sgsf_fast(const Phys &P, const LesParams &Lp, const double (&tu)[ND + 2],
                                               const double (&g)[(ND + 2) * ND], const double len2, const long p, const long plane,
                                               double (&sg)[(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
  double u[ND], de[ND], du[ND][ND], S[ND][ND];
  const double rho = tu[0], ir = 1.0 / rho, third = 1.0 / 3.0;
  double ke = 0.;
#pragma unroll
  for (int i = 0; i < ND; i++)
  {
    u[i] = tu[i + 1] * ir;
    ke += 0.5 * (u[i] * u[i]);
  }
  const double inte = tu[NF - 1] * ir - ke;
#pragma unroll
  for (int i = 0; i < ND; i++)
  {
    const double drho = g[0 + NF * i];
    double dke = ke * drho;
#pragma unroll
    for (int j = 0; j < ND; j++)
    {
      du[i][j] = (g[(j + 1) + NF * i] - u[j] * drho) * ir;
      dke += rho * u[j] * du[i][j];
    }
    de[i] = (g[(NF - 1) + NF * i] - dke - drho * inte) * ir;
  }
#pragma unroll
  for (int i = 0; i < ND; i++)
#pragma unroll
    for (int j = 0; j < ND; j++) S[i][j] = (du[i][j] + du[j][i]) * 0.5;
  const bool eddy = Lp.sgs_model <= 2, sim = Lp.sgs_model == 2 || Lp.sgs_model == 4;
  double mu_t = 0.0;
  if (!eddy)
    ;
  else if (Lp.sgs_model == 0)
  {
    double Smod = 0.0;
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++) Smod += 2.0 * S[i][j] * S[i][j];
    mu_t = rho * len2 * sqrt(Smod);
  }
  else
  {
    double num = 0.0, den = 0.0, gT[ND][ND];
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++)
      {
        double s = 0.;
#pragma unroll
        for (int l = 0; l < ND; l++) s += du[i][l] * du[l][j];
        gT[i][j] = s;
      }
    double diag = 0.0;
#pragma unroll
    for (int i = 0; i < ND; i++) diag += gT[i][i] * third;
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++)
      {
        double sq = 0.5 * gT[j][i] + 0.5 * gT[i][j];
        if (i == j) sq -= diag;
        num += sq * sq;
        den += S[i][j] * S[i][j];
      }
    const double sn = sqrt(num);
    const double denom = den * den * sqrt(den) + num * sqrt(sn);
    mu_t = rho * len2 * (num * sn) / (denom + 1.e-12);
  }
#pragma unroll
  for (int q = 0; q < NF * ND; q++) sg[q] = 0.0;
  if (eddy)
  {
    double diag = 0.;
#pragma unroll
    for (int i = 0; i < ND; i++) diag += S[i][i] * third;
#pragma unroll
    for (int i = 0; i < ND; i++) S[i][i] -= diag;
    const double kt = -1.0 * P.gamma * mu_t / Lp.prandtl_t, m2 = 2.0 * mu_t;
#pragma unroll
    for (int j = 0; j < ND; j++)
    {
      double ef = kt * de[j];
#pragma unroll
      for (int k = 0; k < ND; k++) ef -= u[k] * m2 * S[k][j];
      sg[(NF - 1) + NF * j] = ef;
#pragma unroll
      for (int i = 1; i < NF - 1; i++) sg[i + NF * j] = -m2 * S[i - 1][j];
    }
  }
  if (sim)
  {
    // (src/eles.cpp:2602-2634, as in calc_sgsf: the off-diagonal momentum entries from their transposes AFTER those got their own)
#pragma unroll
    for (int j = 0; j < ND; j++) sg[(NF - 1) + NF * j] += P.gamma * rho * Lp.Le[p + j * plane];
    if (ND == 2)
    {
      sg[1 + NF * 0] += rho * Lp.Lu[p + 0 * plane];
      sg[1 + NF * 1] += rho * Lp.Lu[p + 2 * plane];
      sg[2 + NF * 0] += sg[1 + NF * 1];
      sg[2 + NF * 1] += rho * Lp.Lu[p + 1 * plane];
    }
    else
    {
      sg[1 + NF * 0] += rho * Lp.Lu[p + 0 * plane];
      sg[1 + NF * 1] += rho * Lp.Lu[p + 3 * plane];
      sg[1 + NF * (ND - 1)] += rho * Lp.Lu[p + 4 * plane];
      sg[2 + NF * 0] += sg[1 + NF * 1];
      sg[2 + NF * 1] += rho * Lp.Lu[p + 1 * plane];
      sg[2 + NF * (ND - 1)] += rho * Lp.Lu[p + 5 * plane];
      sg[(NF - 2) + NF * 0] += sg[1 + NF * (ND - 1)];
      sg[(NF - 2) + NF * 1] += sg[2 + NF * (ND - 1)];
      sg[(NF - 2) + NF * (ND - 1)] += rho * Lp.Lu[p + 2 * plane];
    }
  }
}

// f(k, i) += sum_l (|J|^-1 ts(k, l)) J(i, l): the reference-space SGS flux of flux point `o`, taken to physical space in
// the operation order of the reference's dgemm (alpha = 1/detjac, l outer; src/funcs.cpp:110-117)
template <int ND>
__device__ __forceinline__ void add_sgs_flux(const double *sgsf, const double *jac, const double *detjac, long o, long plane,
                                             double (&f)[(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
  double J[ND * ND];
#pragma unroll
  for (int q = 0; q < ND * ND; q++) J[q] = jac[o * (ND * ND) + q];
  const double inv_detjac = 1.0 / detjac[o];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    double ps[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) ps[i] = 0.0;
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      const double temp = inv_detjac * sgsf[o + (k + NF * l) * plane];
#pragma unroll
      for (int i = 0; i < ND; i++) ps[i] += temp * J[i + ND * l];
    }
#pragma unroll
    for (int i = 0; i < ND; i++) f[k + NF * i] += ps[i];
  }
}


} // namespace hfx
