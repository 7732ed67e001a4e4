// split2_kernels.hpp -- split fused stage, variant 2 (hfx_run_steps fused = 2): the reference's gradient arrays kept in HBM; carries the LES closures
// (device code of the split fused stage; included by fused_hex.hip only -- one translation unit, so that every kernel is
// instantiated once)
#pragma once
#include "split_common.hpp"

namespace hfx
{

// =======================================================================================
// SPLIT fused path (mode 2): four launches per stage, every one a simple high-occupancy kernel
//
//   face_delta_kernel    (thread per flux-point pair) LDG common solution -> delta_disu_fpts L,R
//   split_gradient_kernel(workgroup per element)      u, delta -> grad_disu_upts, grad_disu_fpts
//   face_flux_kernel     (thread per pair)            Riemann + LDG viscous flux -> norm_tconf_fpts L,R
//   split_residual_kernel(workgroup per element)      u, grad, norm_tconf -> RK update, new disu_fpts
//
// Same HBM traffic as the gather-style pair (~21 000 doubles per P4 hex and stage) because the
// pairwise face kernels read every flux-point datum once instead of twice, and every element
// kernel is a thread-per-point kernel small enough to keep 3-4 workgroups resident per CU.
// =======================================================================================

struct SplitFaceArgs
{
  // LES: the SGS flux at the flux points in REFERENCE space (n_fpts,n_eles,n_fields,n_dims), NULL: off; the kernel takes
  // it to physical space with |J|^-1 J (second half of eles::extrapolate_sgsFlux, src/eles.cpp:2862-2893)
  const double *sgsf, *jac_fpts, *detjac_fpts;

  long npairs;
  const int *L, *R;
  const unsigned char *meta; // bit1 of the LEFT point: beta sign flipped
  long plane_f;
  const double *disu, *grad, *fnorm, *tdA;
  double *delta, *tconf;
  Phys P;
};

template <int ND>
__global__ __launch_bounds__(256) void face_delta_kernel(const SplitFaceArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long il = a.L[q], ir = a.R[q];
  const double beta = (a.meta[il] & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
  // every load before the first store (the compiler must assume that delta and disu overlap: a load behind a store waits)
  double ul[NF], ur[NF];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu[il + k * a.plane_f];
    ur[k] = a.disu[ir + k * a.plane_f];
  }
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    const double uc = 0.5 * (ul[k] + ur[k]) - beta * (ul[k] - ur[k]); // src/inters.cpp:637
    a.delta[il + k * a.plane_f] = uc - ul[k];
    a.delta[ir + k * a.plane_f] = uc - ur[k];
  }
}

template <int ND, int RS>
__global__ __launch_bounds__(256) void face_flux_kernel(const SplitFaceArgs a)
{
  constexpr int NF = ND + 2, NG = NF * ND;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long il = a.L[q], ir = a.R[q];
  double ul[NF], ur[NF], n[ND], fn[NF];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu[il + k * a.plane_f];
    ur[k] = a.disu[ir + k * a.plane_f];
  }
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.fnorm[il + m * a.plane_f];
  const double tl = a.tdA[il], tr = a.tdA[ir];
  riemann_flux_t<ND, RS, true>(a.P, ul, ur, n, fn);
  if (a.P.viscous)
  {
    const double beta = (a.meta[il] & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
    double pl[NF];
    {
      double gq[NG], fq[NG];
#pragma unroll
      for (int s = 0; s < NG; s++) gq[s] = a.grad[il + s * a.plane_f];
      calc_visf<ND, true>(a.P, ul, gq, fq);
      if (a.sgsf) add_sgs_flux<ND>(a.sgsf, a.jac_fpts, a.detjac_fpts, il, a.plane_f, fq); // src/int_inters.cpp:302-318
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < ND; l++) s += ((0.5 + beta) * fq[k + NF * l]) * n[l];
        pl[k] = s;
      }
    }
    {
      double gq[NG], fq[NG];
#pragma unroll
      for (int s = 0; s < NG; s++) gq[s] = a.grad[ir + s * a.plane_f];
      calc_visf<ND, true>(a.P, ur, gq, fq);
      if (a.sgsf) add_sgs_flux<ND>(a.sgsf, a.jac_fpts, a.detjac_fpts, ir, a.plane_f, fq);
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < ND; l++) s += ((0.5 - beta) * fq[k + NF * l]) * n[l];
        double fv = pl[k] + s;
        fv -= a.P.ldg_tau * (ur[k] - ul[k]);
        // norm_tconf_l = fn*tdA_l + fv*tdA_l ; norm_tconf_r = -fn*tdA_r + -fv*tdA_r   (int_inters.cpp:217-220,329-332)
        a.tconf[il + k * a.plane_f] = fn[k] * tl + fv * tl;
        a.tconf[ir + k * a.plane_f] = -fn[k] * tr + -fv * tr;
      }
    }
  }
  else
  {
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      a.tconf[il + k * a.plane_f] = fn[k] * tl;
      a.tconf[ir + k * a.plane_f] = -fn[k] * tr;
    }
  }
}

struct SplitEleArgs
{
  int n_eles;
  const unsigned *pk;
  const double *tab;
  const int *o1m_dim;
  const double *detjac_upts, *JGinv_upts, *detjac_fpts, *JGinv_fpts;
  double *u0, *u1;
  const double *delta, *tconf;
  double *disu_next;
  double *grad_upts, *grad_fpts, *div_out;
  const double *sgsf_upts; // LES: JGinv * F_sgs at the solution points, added to the total flux (NULL: off)
  const double *src, *dt_local;
  unsigned long long *nan_flag;
  Phys P;
  int adv_type, in_step, dt_local_on, write_div, need_u1;
  double dt, rk_a, rk_b;
};

template <int ND, int N>
struct SGeo
{
  using G = Geo<ND, N>;
  static constexpr int TB = 64 * (G::WU > G::WF ? G::WU : G::WF); // thread t: solution point t and flux point t
};

// ---- u, delta -> corrected gradient at solution points (physical) and flux points (physical)
template <int ND, int N>
__global__ __launch_bounds__((SGeo<ND, N>::TB)) void split_gradient_kernel(const SplitEleArgs a)
{
  using G = Geo<ND, N>;
  constexpr int NF = G::NF, NU = G::NU, NFP = G::NFP, WN = G::WN, TB = SGeo<ND, N>::TB;
  constexpr int PW = G::G_WU + WN; // opp_4[d] | opp_5[d] | opp_6
  __shared__ double tab[MAX_TAB];
  __shared__ double su[NF][NU];
  __shared__ double sd[NF][NFP];
  __shared__ double sg[NF * ND][NU];
  const int t = threadIdx.x;
  const int tu = t < NU ? t : NU - 1, tf = t < NFP ? t : NFP - 1;
  const bool is_u = t < NU, is_f = t < NFP;
  const long ne = a.n_eles, plane_u = (long)NU * ne, plane_f = (long)NFP * ne;
  for (int q = t; q < MAX_TAB; q += TB) tab[q] = a.tab[q];
  unsigned pw[PW];
#pragma unroll
  for (int d = 0; d < ND; d++)
  {
#pragma unroll
    for (int i = 0; i < WN; i++) pw[d * WN + i] = a.pk[G::G_O4 + (d * WN + i) * NU + tu];
    pw[ND * WN + d] = a.pk[G::G_O5 + d * NU + tu];
  }
#pragma unroll
  for (int i = 0; i < WN; i++) pw[G::G_WU + i] = a.pk[G::G_O6 + i * NFP + tf];

  for (long e = blockIdx.x; e < ne; e += gridDim.x)
  {
    for (int q = t; q < NF * NU; q += TB)
    {
      const int f = q / NU, p = q - f * NU;
      su[f][p] = a.u0[p + NU * e + f * plane_u];
    }
    for (int q = t; q < NF * NFP; q += TB)
    {
      const int f = q / NFP, p = q - f * NFP;
      sd[f][p] = a.delta[p + NFP * e + f * plane_f];
    }
    double JG[ND * ND];
    double inv_detjac;
    {
      const long p = tu + NU * e;
#pragma unroll
      for (int q = 0; q < ND * ND; q++) JG[q] = a.JGinv_upts[p * (ND * ND) + q];
      inv_detjac = 1.0 / a.detjac_upts[p];
    }
    __syncthreads();
    if (is_u)
    {
      const long p = tu + NU * e;
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double tg[ND], cg[ND];
        tg[0] = row_dot<N, 0, PW>(pw, tab, &su[k][0], 0.0);
        tg[1] = row_dot<N, WN, PW>(pw, tab, &su[k][0], 0.0);
        if (ND == 3) tg[ND - 1] = row_dot<N, (ND - 1) * WN, PW>(pw, tab, &su[k][0], 0.0);
        tg[0] = row_dot<2, ND * WN + 0, PW>(pw, tab, &sd[k][0], tg[0]);
        tg[1] = row_dot<2, ND * WN + 1, PW>(pw, tab, &sd[k][0], tg[1]);
        if (ND == 3) tg[ND - 1] = row_dot<2, ND * WN + ND - 1, PW>(pw, tab, &sd[k][0], tg[ND - 1]);
#pragma unroll
        for (int d = 0; d < ND; d++) sg[k + NF * d][tu] = tg[d];
        to_physical<ND>(inv_detjac, JG, tg, cg);
#pragma unroll
        for (int d = 0; d < ND; d++) a.grad_upts[p + (k + NF * d) * plane_u] = cg[d];
      }
    }
    {
      const long o = tf + NFP * e;
#pragma unroll
      for (int q = 0; q < ND * ND; q++) JG[q] = a.JGinv_fpts[o * (ND * ND) + q];
      inv_detjac = 1.0 / a.detjac_fpts[o];
    }
    __syncthreads();
    if (is_f)
    {
      const long o = tf + NFP * e;
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double tg[ND], cg[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) tg[d] = row_dot<N, G::G_WU, PW>(pw, tab, &sg[k + NF * d][0], 0.0);
        to_physical<ND>(inv_detjac, JG, tg, cg);
#pragma unroll
        for (int d = 0; d < ND; d++) a.grad_fpts[o + (k + NF * d) * plane_f] = cg[d];
      }
    }
    __syncthreads();
  }
}

// ---- u, grad, norm_tconf -> divergence, correction, RK update, disu_fpts of the new state.
// Loads are placed right before their use: several workgroups are resident per CU and cover each
// other's latency, and short live ranges keep the register count (= the occupancy) in check.
template <int ND, int N>
__global__ __launch_bounds__((SGeo<ND, N>::TB), HFX_SPLIT_WAVES_RES) void split_residual_kernel(const SplitEleArgs a)
{
  using G = Geo<ND, N>;
  constexpr int NF = G::NF, NU = G::NU, NFP = G::NFP, WN = G::WN, TB = SGeo<ND, N>::TB;
  constexpr int N3 = 2 * ND, NG = NF * ND;
  constexpr int PW = G::R_WU + G::R_WF; // opp_2[d] | opp_3 | opp_0 | merged opp_1
  __shared__ double tab[MAX_TAB];
  __shared__ double su[NF][NU];
  __shared__ double st[NF * ND][NU];
  __shared__ double sc[NF][NFP];
  const int t = threadIdx.x;
  const int tu = t < NU ? t : NU - 1, tf = t < NFP ? t : NFP - 1;
  const bool is_u = t < NU, is_f = t < NFP;
  const long ne = a.n_eles, plane_u = (long)NU * ne, plane_f = (long)NFP * ne;
  const bool viscous = a.P.viscous;
  for (int q = t; q < MAX_TAB; q += TB) tab[q] = a.tab[q];
  unsigned pw[PW];
#pragma unroll
  for (int i = 0; i < ND * WN; i++) pw[i] = a.pk[G::R_O2 + i * NU + tu];
#pragma unroll
  for (int i = 0; i < words_of(N3); i++) pw[ND * WN + i] = a.pk[G::R_O3 + i * NU + tu];
#pragma unroll
  for (int i = 0; i < WN; i++)
  {
    pw[G::R_WU + i] = a.pk[G::R_O0 + i * NFP + tf];
    pw[G::R_WU + WN + i] = a.pk[G::R_O1 + i * NFP + tf];
  }
  const int d1 = a.o1m_dim[tf];

  for (long e = blockIdx.x; e < ne; e += gridDim.x)
  {
    const long p = tu + NU * e, o = tf + NFP * e;
    for (int q = t; q < NF * NU; q += TB)
    {
      const int f = q / NU, p2 = q - f * NU;
      su[f][p2] = a.u0[p2 + NU * e + f * plane_u];
    }
    __syncthreads();
    if (is_u)
    {
      double u[NF], f[NG], JG[ND * ND];
#pragma unroll
      for (int q = 0; q < ND * ND; q++) JG[q] = a.JGinv_upts[p * (ND * ND) + q];
#pragma unroll
      for (int k = 0; k < NF; k++) u[k] = su[k][tu];
      calc_invf<ND, true>(a.P.gamma, u, f);
#pragma unroll
      for (int k = 0; k < NF; k++)
#pragma unroll
        for (int l = 0; l < ND; l++)
        {
          double s = 0.0;
#pragma unroll
          for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
          st[k + NF * l][tu] = s;
        }
      if (viscous)
      {
        {
          double gr[NG];
#pragma unroll
          for (int q = 0; q < NG; q++) gr[q] = a.grad_upts[p + q * plane_u];
          calc_visf<ND, true>(a.P, u, gr, f);
        }
        // tdisf += JGinv * F_v : read-modify-write of this thread's own LDS column
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = st[k + NF * l][tu];
#pragma unroll
            for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
            if (a.sgsf_upts) s += a.sgsf_upts[p + (k + NF * l) * plane_u]; // src/eles.cpp:2322-2348
            st[k + NF * l][tu] = s;
          }
      }
    }
    __syncthreads();
    double div[NF];
    if (is_u)
    {
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = row_dot<N, 0, PW>(pw, tab, &st[k][0], 0.0);
        s = row_dot<N, WN, PW>(pw, tab, &st[k + NF][0], s);
        if (ND == 3) s = row_dot<N, (ND - 1) * WN, PW>(pw, tab, &st[k + NF * (ND - 1)][0], s);
        div[k] = s;
      }
    }
    if (is_f)
    {
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        const double ntd = row_dot<N, G::R_WU + WN, PW>(pw, tab, &st[k + NF * d1][0], 0.0);
        sc[k][tf] = a.tconf[o + k * plane_f] + -1.0 * ntd; // norm_tconf -= norm_tdisf (src/eles.cpp:1746)
      }
    }
    __syncthreads();
    if (is_u)
    {
      const double dt = a.dt_local_on ? a.dt_local[e] : a.dt;
      const double dj = a.detjac_upts[p];
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        const double dv = row_dot<N3, ND * WN, PW>(pw, tab, &sc[k][0], div[k]);
        const long q = p + k * plane_u;
        if (dv != dv) atomicMin(a.nan_flag, (unsigned long long)q);
        if (a.write_div) a.div_out[q] = dv;
        const double s = a.src ? a.src[q] : 0.0;
        const double dd = dv / dj;
        const double u1v = a.need_u1 ? a.u1[q] : 0.0;
        double u = su[k][tu];
        if (a.adv_type == 0)
          u -= dt * (dd - s);
        else if (a.adv_type == 1)
        {
          if (a.in_step == 0) a.u1[q] = u;
          if (a.in_step < 3)
            u -= dt / 3.0 * (dd - s);
          else
          {
            const double rhs = -dd + s;
            u = 3.0 / 4.0 * u + 1.0 / 4.0 * u1v + dt / 4.0 * rhs;
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
            u = 1.0 / 3.0 * u + 2.0 / 3.0 * u1v + dt / 6.0 * rhs;
          }
        }
        else
        {
          const double rhs = -dd + s;
          const double r1 = a.rk_a * u1v + dt * rhs;
          a.u1[q] = r1;
          u += a.rk_b * r1;
        }
        a.u0[q] = u;
        su[k][tu] = u;
      }
    }
    __syncthreads();
    if (is_f)
    {
#pragma unroll
      for (int k = 0; k < NF; k++) a.disu_next[o + k * plane_f] = row_dot<N, G::R_WU, PW>(pw, tab, &su[k][0], 0.0);
    }
    __syncthreads();
  }
}


} // namespace hfx
