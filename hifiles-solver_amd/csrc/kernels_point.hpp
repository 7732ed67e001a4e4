// kernels_point.hpp -- per-point, per-face-point and per-DOF kernels of the hot path.
//
//   evaluate_invFlux      /root/reference/src/eles.cpp:1415-1478
//   gradient transform    /root/reference/src/eles.cpp:1958-2011 (second half of correct_gradient)
//   evaluate_viscFlux     /root/reference/src/eles.cpp:2285-2392
//   common inviscid flux  /root/reference/src/int_inters.cpp:160-249
//   common viscous flux   /root/reference/src/int_inters.cpp:254-343
//   AdvanceSolution       /root/reference/src/eles.cpp:1080-1265
//   compute_res_upts      /root/reference/src/eles.cpp:5045-5074
//
// Layout notes (hf_array, first index fastest): fields of one point are a
// whole (pt,ele) plane apart, so consecutive lanes = consecutive points give
// coalesced loads/stores per field.  The 3x3 JGinv block of a point is AoS
// (9 consecutive doubles); a wave reads 64 consecutive blocks = one contiguous
// 4.6 kB run.
#pragma once
#include "hfx_internal.hpp"

namespace hfx
{

constexpr int PT_BLOCK = 256;

// ---- eles::evaluate_invFlux ------------------------------------------------
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void invflux_kernel(long plane, const double gamma, const double *__restrict__ U,
                                                           const double *__restrict__ JGinv, double *__restrict__ tdisf)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double u[NF], f[NF * ND], JG[ND * ND];
#pragma unroll
  for (int k = 0; k < NF; k++) u[k] = U[p + k * plane];
#pragma unroll
  for (int q = 0; q < ND * ND; q++) JG[q] = JGinv[p * (ND * ND) + q];
  calc_invf<ND>(gamma, u, f);
#pragma unroll
  for (int k = 0; k < NF; k++)
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < ND; m++) t += JG[l + ND * m] * f[k + NF * m];
      tdisf[p + (k + NF * l) * plane] = t;
    }
}

// ---- second half of eles::correct_gradient: reference -> physical gradient, in place
// g_phys(d,k) = sum_l (inv_detjac * g_ref(l,k)) * JGinv(l,d)
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void grad_transform_kernel(long plane, const double *__restrict__ detjac,
                                                                  const double *__restrict__ JGinv, double *G)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double JG[ND * ND];
#pragma unroll
  for (int q = 0; q < ND * ND; q++) JG[q] = JGinv[p * (ND * ND) + q];
  const double inv_detjac = 1.0 / detjac[p];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    double tg[ND], cg[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) tg[d] = G[p + (k + NF * d) * plane];
#pragma unroll
    for (int d = 0; d < ND; d++) cg[d] = 0.0;
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      const double temp = inv_detjac * tg[l];
#pragma unroll
      for (int d = 0; d < ND; d++) cg[d] += temp * JG[l + ND * d];
    }
#pragma unroll
    for (int d = 0; d < ND; d++) G[p + (k + NF * d) * plane] = cg[d];
  }
}

// ---- eles::evaluate_viscFlux (LES off): tdisf += JGinv * F_v -------------------
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void viscflux_kernel(long plane, const Phys P, const double *__restrict__ U,
                                                            const double *__restrict__ G,
                                                            const double *__restrict__ JGinv, double *tdisf)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double u[NF], g[NF * ND], f[NF * ND], JG[ND * ND];
#pragma unroll
  for (int k = 0; k < NF; k++) u[k] = U[p + k * plane];
#pragma unroll
  for (int q = 0; q < NF * ND; q++) g[q] = G[p + q * plane];
#pragma unroll
  for (int q = 0; q < ND * ND; q++) JG[q] = JGinv[p * (ND * ND) + q];
  calc_visf<ND>(P, u, g, f);
#pragma unroll
  for (int k = 0; k < NF; k++)
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      double t = tdisf[p + (k + NF * l) * plane];
#pragma unroll
      for (int m = 0; m < ND; m++) t += JG[l + ND * m] * f[k + NF * m];
      tdisf[p + (k + NF * l) * plane] = t;
    }
}

// ---- eles::evaluate_viscFlux with LES (src/eles.cpp:2322-2348): F_v + F_sgs, and sgsf_upts = JGinv * F_sgs
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void viscflux_les_kernel(long plane, const Phys P, const LesParams Lp,
                                                                const double *__restrict__ U, const double *__restrict__ G,
                                                                const double *__restrict__ JGinv,
                                                                const double *__restrict__ detjac,
                                                                const double *__restrict__ wall_distance, double *tdisf,
                                                                double *__restrict__ sgsf_upts)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double u[NF], g[NF * ND], f[NF * ND], sg[NF * ND], JG[ND * ND];
#pragma unroll
  for (int k = 0; k < NF; k++) u[k] = U[p + k * plane];
#pragma unroll
  for (int q = 0; q < NF * ND; q++) g[q] = G[p + q * plane];
#pragma unroll
  for (int q = 0; q < ND * ND; q++) JG[q] = JGinv[p * (ND * ND) + q];
  calc_visf<ND>(P, u, g, f);
  double y = 0.0;
  if (Lp.sgs_model == 0)
  {
#pragma unroll
    for (int i = 0; i < ND; i++)
    {
      const double w = wall_distance[p + i * plane];
      y += w * w;
    }
    y = sqrt(y);
  }
  calc_sgsf<ND>(P, Lp, u, g, detjac[p], y, p, plane, sg);
#pragma unroll
  for (int q = 0; q < NF * ND; q++) f[q] += 1.0 * sg[q];
#pragma unroll
  for (int k = 0; k < NF; k++)
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      double ts = 0.0;
#pragma unroll
      for (int m = 0; m < ND; m++) ts += JG[l + ND * m] * sg[k + NF * m];
      sgsf_upts[p + (k + NF * l) * plane] = ts;
      double t = tdisf[p + (k + NF * l) * plane];
#pragma unroll
      for (int m = 0; m < ND; m++) t += JG[l + ND * m] * f[k + NF * m];
      tdisf[p + (k + NF * l) * plane] = t;
    }
}

// the SGS part alone: sgsf_upts = JGinv * F_sgs(u, grad u) (the split fused path adds it to its own total flux)
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void sgsf_upts_kernel(long plane, const Phys P, const LesParams Lp, const double *__restrict__ U,
                                                             const double *__restrict__ G, const double *__restrict__ JGinv,
                                                             const double *__restrict__ detjac,
                                                             const double *__restrict__ wall_distance, double *__restrict__ sgsf_upts)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double u[NF], g[NF * ND], sg[NF * ND], JG[ND * ND];
#pragma unroll
  for (int k = 0; k < NF; k++) u[k] = U[p + k * plane];
#pragma unroll
  for (int q = 0; q < NF * ND; q++) g[q] = G[p + q * plane];
#pragma unroll
  for (int q = 0; q < ND * ND; q++) JG[q] = JGinv[p * (ND * ND) + q];
  double y = 0.0;
  if (Lp.sgs_model == 0)
  {
#pragma unroll
    for (int i = 0; i < ND; i++)
    {
      const double w = wall_distance[p + i * plane];
      y += w * w;
    }
    y = sqrt(y);
  }
  calc_sgsf<ND>(P, Lp, u, g, detjac[p], y, p, plane, sg);
#pragma unroll
  for (int k = 0; k < NF; k++)
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      double ts = 0.0;
#pragma unroll
      for (int m = 0; m < ND; m++) ts += JG[l + ND * m] * sg[k + NF * m];
      sgsf_upts[p + (k + NF * l) * plane] = ts;
    }
}

// second half of eles::extrapolate_sgsFlux (src/eles.cpp:2862-2893): sgsf_fpts <- |J|^-1 J sgsf_fpts, in place
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void sgsf_to_physical_kernel(long plane, const double *__restrict__ detjac,
                                                                    const double *__restrict__ Jac, double *S)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double J[ND * ND];
#pragma unroll
  for (int q = 0; q < ND * ND; q++) J[q] = Jac[p * (ND * ND) + q];
  const double inv_detjac = 1.0 / detjac[p];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    double ts[ND], ps[ND];
#pragma unroll
    for (int d = 0; d < ND; d++)
    {
      ts[d] = S[p + (k + NF * d) * plane];
      ps[d] = 0.0;
    }
    // dgemm(alpha = inv_detjac): c(i) += (alpha * b(l)) * a(i,l), l outer (src/funcs.cpp:110-117)
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      const double temp = inv_detjac * ts[l];
#pragma unroll
      for (int i = 0; i < ND; i++) ps[i] += temp * J[i + ND * l];
    }
#pragma unroll
    for (int d = 0; d < ND; d++) S[p + (k + NF * d) * plane] = ps[d];
  }
}

// ---- int_inters::calculate_common_invFlux ------------------------------------
struct FaceArgs
{
  long npairs; // n_fpts_per_inter * n_inters
  const int *L, *R;
  long plane_l, plane_r; // n_fpts*n_eles of the left / right element block
  const double *disu_l, *disu_r;
  const double *norm_l; // left block norm_fpts (fpt,ele,dim)
  const double *tdA_l, *tdA_r;
  double *tconf_l, *tconf_r;
  double *delta_l, *delta_r;
  const double *grad_l, *grad_r;
  const double *sgsf_l, *sgsf_r; // LES: physical SGS flux at the flux points (NULL: off)
};

template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void common_invflux_kernel(const FaceArgs a, const Phys P)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (q >= a.npairs) return;
  const long il = a.L[q], ir = a.R[q];
  double ul[NF], ur[NF], n[ND], fn[NF];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu_l[il + k * a.plane_l];
    ur[k] = a.disu_r[ir + k * a.plane_r];
  }
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm_l[il + m * a.plane_l];
  riemann_flux<ND>(P, ul, ur, n, fn);
  const double tl = a.tdA_l[il], tr = a.tdA_r[ir];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    a.tconf_l[il + k * a.plane_l] = fn[k] * tl;
    a.tconf_r[ir + k * a.plane_r] = -fn[k] * tr;
  }
  if (P.viscous)
  {
    const double beta = ldg_switch<ND>(P.ldg_beta, n);
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      const double uc = 0.5 * (ul[k] + ur[k]) - beta * (ul[k] - ur[k]);
      a.delta_l[il + k * a.plane_l] = uc - ul[k];
      a.delta_r[ir + k * a.plane_r] = uc - ur[k];
    }
  }
}

// ---- int_inters::calculate_common_viscFlux (LES off) ---------------------------
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void common_viscflux_kernel(const FaceArgs a, const Phys P)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (q >= a.npairs) return;
  const long il = a.L[q], ir = a.R[q];
  double ul[NF], ur[NF], gl[NF * ND], gr[NF * ND], fl[NF * ND], fr[NF * ND], n[ND];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu_l[il + k * a.plane_l];
    ur[k] = a.disu_r[ir + k * a.plane_r];
  }
#pragma unroll
  for (int s = 0; s < NF * ND; s++)
  {
    gl[s] = a.grad_l[il + s * a.plane_l];
    gr[s] = a.grad_r[ir + s * a.plane_r];
  }
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm_l[il + m * a.plane_l];
  calc_visf<ND>(P, ul, gl, fl);
  calc_visf<ND>(P, ur, gr, fr);
  if (a.sgsf_l != nullptr)
  {
    // src/int_inters.cpp:302-318
#pragma unroll
    for (int s = 0; s < NF * ND; s++)
    {
      fl[s] += a.sgsf_l[il + s * a.plane_l];
      fr[s] += a.sgsf_r[ir + s * a.plane_r];
    }
  }
  const double beta = ldg_switch<ND>(P.ldg_beta, n);
  const double tl = a.tdA_l[il], tr = a.tdA_r[ir];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    double fn = 0.0;
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
      const double fc = (0.5 + beta) * fl[k + NF * l] + (0.5 - beta) * fr[k + NF * l];
      fn += fc * n[l];
    }
    fn -= P.ldg_tau * (ur[k] - ul[k]);
    a.tconf_l[il + k * a.plane_l] += fn * tl;
    a.tconf_r[ir + k * a.plane_r] += -fn * tr;
  }
}

// ---- eles::calc_sgs_terms (src/eles.cpp:2058-2283), the two point-wise parts ----------------------------
// products of the UNFILTERED solution: uu (3|6 components), ue (n_dims); and the NaN scan of the filtered solution
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void sgs_products_kernel(long plane, const double *__restrict__ U, const double *__restrict__ Uf,
                                                                double *__restrict__ uu, double *__restrict__ ue,
                                                                unsigned long long *nan_flag)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double ut[NF];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ut[k] = U[p + k * plane];
    const double f = Uf[p + k * plane];
    if (f != f) atomicMin(nan_flag, (unsigned long long)(p + k * plane)); // "nan in filtered solution"
  }
  if (uu == nullptr) return;
  const double rsq = ut[0] * ut[0];
  if (ND == 2)
  {
    uu[p + 0 * plane] = ut[1] * ut[1] / rsq;
    uu[p + 1 * plane] = ut[2] * ut[2] / rsq;
    uu[p + 2 * plane] = ut[1] * ut[2] / rsq;
    ut[3] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2]) / ut[0];
    ue[p + 0 * plane] = ut[1] * ut[3] / rsq;
    ue[p + 1 * plane] = ut[2] * ut[3] / rsq;
  }
  else
  {
    uu[p + 0 * plane] = ut[1] * ut[1] / rsq;
    uu[p + 1 * plane] = ut[2] * ut[2] / rsq;
    uu[p + 2 * plane] = ut[ND] * ut[ND] / rsq;
    uu[p + 3 * plane] = ut[1] * ut[2] / rsq;
    uu[p + 4 * plane] = ut[1] * ut[ND] / rsq;
    uu[p + 5 * plane] = ut[2] * ut[ND] / rsq;
    ut[NF - 1] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2] + ut[ND] * ut[ND]) / ut[0];
    ue[p + 0 * plane] = ut[1] * ut[NF - 1] / rsq;
    ue[p + 1 * plane] = ut[2] * ut[NF - 1] / rsq;
    ue[p + (ND - 1) * plane] = ut[ND] * ut[NF - 1] / rsq;
  }
}

// Leonard terms: the filtered products minus the products of the FILTERED solution, Lu made traceless
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void sgs_leonard_kernel(long plane, const double *__restrict__ Uf, double *Lu, double *Le)
{
  constexpr int NF = ND + 2;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  double ut[NF], diag;
#pragma unroll
  for (int k = 0; k < NF; k++) ut[k] = Uf[p + k * plane];
  const double rsq = ut[0] * ut[0];
  if (ND == 2)
  {
    Lu[p + 0 * plane] -= (ut[1] * ut[1]) / rsq;
    Lu[p + 1 * plane] -= (ut[2] * ut[2]) / rsq;
    Lu[p + 2 * plane] -= (ut[1] * ut[2]) / rsq;
    diag = (Lu[p + 0 * plane] + Lu[p + 1 * plane]) / 3.0;
    ut[3] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2]) / ut[0];
    Le[p + 0 * plane] = (Le[p + 0 * plane] - ut[1] * ut[3]) / rsq;
    Le[p + 1 * plane] = (Le[p + 1 * plane] - ut[2] * ut[3]) / rsq;
  }
  else
  {
    Lu[p + 0 * plane] -= (ut[1] * ut[1]) / rsq;
    Lu[p + 1 * plane] -= (ut[2] * ut[2]) / rsq;
    Lu[p + 2 * plane] -= (ut[ND] * ut[ND]) / rsq;
    Lu[p + 3 * plane] -= (ut[1] * ut[2]) / rsq;
    Lu[p + 4 * plane] -= (ut[1] * ut[ND]) / rsq;
    Lu[p + 5 * plane] -= (ut[2] * ut[ND]) / rsq;
    diag = (Lu[p + 0 * plane] + Lu[p + 1 * plane] + Lu[p + 2 * plane]) / 3.0;
    ut[NF - 1] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2] + ut[ND] * ut[ND]) / ut[0];
    Le[p + 0 * plane] = (Le[p + 0 * plane] - ut[1] * ut[NF - 1]) / rsq;
    Le[p + 1 * plane] = (Le[p + 1 * plane] - ut[2] * ut[NF - 1]) / rsq;
    Le[p + (ND - 1) * plane] = (Le[p + (ND - 1) * plane] - ut[ND] * ut[NF - 1]) / rsq;
  }
#pragma unroll
  for (int k = 0; k < ND; ++k) Lu[p + k * plane] -= diag;
}

// ---- eles::AdvanceSolution ------------------------------------------------------
struct AdvArgs
{
  long n;     // n_upts*n_eles*n_fields
  long plane; // n_upts*n_eles
  int n_upts;
  int adv_type, in_step, dt_local_on;
  double dt, rk_a, rk_b;
  double *u0, *u1;
  const double *div, *detjac, *src, *dt_local;
};

__global__ __launch_bounds__(PT_BLOCK) void advance_kernel(const AdvArgs a)
{
  const long q = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (q >= a.n) return;
  const long p = q % a.plane;
  const double s = a.src ? a.src[q] : 0.0;
  const double dt = a.dt_local_on ? a.dt_local[p / a.n_upts] : a.dt;
  const double dd = a.div[q] / a.detjac[p];
  double u = a.u0[q];
  if (a.adv_type == 0)
    u -= dt * (dd - s);
  else if (a.adv_type == 1)
  {
    if (a.in_step == 0) a.u1[q] = u; // disu_upts(1) = disu_upts(0)
    if (a.in_step < 3)
      u -= dt / 3.0 * (dd - s);
    else
    {
      const double rhs = -dd + s;
      const double r1 = (a.in_step == 0) ? u : a.u1[q];
      u = 3.0 / 4.0 * u + 1.0 / 4.0 * r1 + dt / 4.0 * rhs;
    }
  }
  else if (a.adv_type == 2)
  {
    if (a.in_step == 0) a.u1[q] = u;
    if (a.in_step < 2 || a.in_step == 3)
      u -= dt / 2.0 * (dd - s);
    else if (a.in_step == 2)
    {
      const double rhs = -dd + s;
      u = 1.0 / 3.0 * u + 2.0 / 3.0 * a.u1[q] + dt / 6.0 * rhs;
    }
  }
  else
  {
    const double rhs = -dd + s;
    const double r1 = a.rk_a * a.u1[q] + dt * rhs;
    a.u1[q] = r1;
    u += a.rk_b * r1;
  }
  a.u0[q] = u;
}

// ---- eles::compute_res_upts: deterministic two-level reduction -----------------------
// level 1: one partial per workgroup (fixed tree), level 2 on the host in block order
__global__ __launch_bounds__(PT_BLOCK) void res_partial_kernel(long plane, int norm_type, const double *__restrict__ div,
                                                               const double *__restrict__ detjac,
                                                               const double *__restrict__ src, double *partial)
{
  __shared__ double sm[PT_BLOCK];
  double acc = 0.0;
  for (long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x; p < plane; p += (long)gridDim.x * PT_BLOCK)
  {
    const double r = div[p] / detjac[p] - (src ? src[p] : 0.0);
    if (norm_type == 0)
      acc = fmax(acc, fabs(r));
    else if (norm_type == 1)
      acc += fabs(r);
    else
      acc += r * r;
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = PT_BLOCK / 2; s > 0; s >>= 1)
  {
    if (threadIdx.x < s)
      sm[threadIdx.x] = (norm_type == 0) ? fmax(sm[threadIdx.x], sm[threadIdx.x + s]) : sm[threadIdx.x] + sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

// ---- Persson sensor from the modal coefficients (src/eles_hexas.cpp:1033-1056) -------------
// sensor(e) = sum_j num(j) modal(j,e)^2 / sum_j den(j) modal(j,e)^2 ; one workgroup (64 lanes) per element
__global__ __launch_bounds__(64) void persson_sensor_kernel(int n_upts, long n_eles, const double *__restrict__ modal,
                                                            const double *__restrict__ num, const double *__restrict__ den,
                                                            double *__restrict__ sensor)
{
  const long e = blockIdx.x;
  if (e >= n_eles) return;
  double a = 0.0, b = 0.0;
  for (int j = threadIdx.x; j < n_upts; j += 64)
  {
    const double m = modal[j + (long)n_upts * e];
    const double m2 = m * m;
    a += m2 * num[j];
    b += den[j] * m2;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  if (threadIdx.x == 0) sensor[e] = a / b;
}

// disu_upts(0)(:, e, :) = filt(:, e, :) where sensor(e) >= s0 (src/eles.cpp:2936-2954)
__global__ __launch_bounds__(PT_BLOCK) void shock_select_kernel(int n_upts, long n_eles, int n_fields, double s0,
                                                                const double *__restrict__ sensor,
                                                                const double *__restrict__ filt, double *__restrict__ u)
{
  const long plane = (long)n_upts * n_eles;
  const long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= plane) return;
  if (!(sensor[p / n_upts] >= s0)) return;
  for (int k = 0; k < n_fields; k++) u[p + k * plane] = filt[p + k * plane];
}

// ---- eles::calc_dt_local (src/eles.cpp:1267-1356) for every element + the minimum -----------
// one 64-lane workgroup per element; dt_min accumulates through an atomic min on the bit pattern
// (positive doubles order like unsigned integers)
template <int ND>
__global__ __launch_bounds__(64) void dt_local_kernel(int n_upts, long n_eles, const double *__restrict__ U,
                                                      const double *__restrict__ h_ref, const Phys P, double CFL, int order,
                                                      double *__restrict__ dt_local, unsigned long long *dt_min_bits)
{
  const long e = blockIdx.x;
  if (e >= n_eles) return;
  const long plane = (long)n_upts * n_eles;
  double lam_inv = 0.0, lam_visc = 0.0;
  for (int i = threadIdx.x; i < n_upts; i += 64)
  {
    const long p = i + (long)n_upts * e;
    const double rho = U[p];
    double vsq = 0.0;
#pragma unroll
    for (int d = 0; d < ND; d++)
    {
      const double v = U[p + (d + 1) * plane] / rho;
      vsq += v * v;
    }
    const double pr = (P.gamma - 1.0) * (U[p + (ND + 1) * plane] - 0.5 * rho * vsq);
    const double c = sqrt(P.gamma * pr / rho);
    const double inte = pr / ((P.gamma - 1.0) * rho);
    const double rt_ratio = (P.gamma - 1.0) * inte / (P.rt_inf);
    double mu = (P.mu_inf) * pow(rt_ratio, 1.5) * (1. + (P.c_sth)) / (rt_ratio + (P.c_sth));
    mu = mu + P.fix_vis * (P.mu_inf - mu);
    lam_inv = fmax(lam_inv, sqrt(vsq) + c);
    lam_visc = fmax(lam_visc, fmax(4.0 / 3.0, P.gamma / P.prandtl) * mu / rho);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
    lam_inv = fmax(lam_inv, __shfl_down(lam_inv, off, 64));
    lam_visc = fmax(lam_visc, __shfl_down(lam_visc, off, 64));
  }
  if (threadIdx.x == 0)
  {
    const double h = h_ref[e];
    const double dt_visc = P.viscous ? (CFL * 0.25 * h * h) / (lam_visc) * 1.0 / (2.0 * order + 1.0) : 1e16;
    const double dt_inv = CFL * h / lam_inv * 1.0 / (2.0 * order + 1.0);
    const double dt = fmin(dt_visc, dt_inv);
    dt_local[e] = dt;
    atomicMin(dt_min_bits, (unsigned long long)__double_as_longlong(dt));
  }
}

// ---- eles::CalcIntegralQuantities (src/eles.cpp:5485-5627) -----------------------------------
// pointwise diagnostics at the volume cubature points (state and gradient already interpolated
// there), weighted and reduced to one partial per workgroup per quantity (fixed tree; the host
// adds the partials in block order)
constexpr int IQ_MAX = 8;
template <int ND>
__global__ __launch_bounds__(PT_BLOCK) void integral_quantities_kernel(int n_cub, long n_eles, const double *__restrict__ Uc,
                                                                       const double *__restrict__ Gc,
                                                                       const double *__restrict__ wgt,
                                                                       const double *__restrict__ vdj, double gamma, int nq,
                                                                       const int *__restrict__ ids, double *partial)
{
  constexpr int NF = ND + 2;
  __shared__ double sm[PT_BLOCK];
  const long plane = (long)n_cub * n_eles;
  double acc[IQ_MAX];
#pragma unroll
  for (int m = 0; m < IQ_MAX; m++) acc[m] = 0.0;
  for (long p = (long)blockIdx.x * PT_BLOCK + threadIdx.x; p < plane; p += (long)gridDim.x * PT_BLOCK)
  {
    double u[NF], g[NF * ND];
#pragma unroll
    for (int k = 0; k < NF; k++) u[k] = Uc[p + k * plane];
#pragma unroll
    for (int q = 0; q < NF * ND; q++) g[q] = Gc[p + q * plane];
    const double irho = 1. / u[0];
    double dv[ND][ND]; // dv[i][j] = d u_i / d x_j
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++) dv[i][j] = irho * (g[(i + 1) + NF * j] - u[i + 1] * irho * g[0 + NF * j]);
    double tke = 0.0;
#pragma unroll
    for (int n = 1; n < ND + 1; n++) tke += 0.5 * u[n] * u[n];
    const double w = wgt[p % n_cub] * vdj[p];
    for (int m = 0; m < nq; m++)
    {
      double diagnostic = 0.0;
      const int id = ids[m];
      if (id == 0)
        diagnostic = irho * tke;
      else if (id == 1)
      {
        const double wz = dv[1][0] - dv[0][1];
        diagnostic = wz * wz;
        if (ND == 3)
        {
          const double wx = dv[ND - 1][1] - dv[1][ND - 1], wy = dv[0][ND - 1] - dv[ND - 1][0];
          diagnostic += wx * wx + wy * wy;
        }
        diagnostic *= 0.5 / irho;
      }
      else if (id == 2)
      {
        const double pressure = (gamma - 1.0) * (u[ND + 1] - irho * tke);
        double dil = dv[0][0] + dv[1][1];
        if (ND == 3) dil = dil + dv[ND - 1][ND - 1];
        diagnostic = pressure * dil;
      }
      else
      {
        double S[ND][ND];
#pragma unroll
        for (int a = 0; a < ND; a++)
#pragma unroll
          for (int b = 0; b < ND; b++) S[a][b] = (a == b) ? dv[a][a] : (dv[a][b] + dv[b][a]) / 2.0;
        double diag = (S[0][0] + S[1][1]) / 3.0;
        if (ND == 3) diag += S[ND - 1][ND - 1] / 3.0;
        if (id == 4)
#pragma unroll
          for (int a = 0; a < ND; a++) S[a][a] -= diag;
#pragma unroll
        for (int a = 0; a < ND; a++)
#pragma unroll
          for (int b = 0; b < ND; b++) diagnostic += S[a][b] * S[a][b];
      }
      acc[m] += diagnostic * w;
    }
  }
  for (int m = 0; m < nq; m++)
  {
    sm[threadIdx.x] = acc[m];
    __syncthreads();
    for (int s = PT_BLOCK / 2; s > 0; s >>= 1)
    {
      if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x + (long)gridDim.x * m] = sm[0];
    __syncthreads();
  }
}

} // namespace hfx
