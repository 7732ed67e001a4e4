// kernels_mpi.hpp -- partition faces (reference class mpi_inters, /root/reference/src/mpi_inters.cpp).
//
// A partition face has only its LEFT side on this rank.  The right state is read from the
// received buffer: record (fpt, field[, dim], inter) of the neighbour's packed face, at the
// neighbour's flux-point slot lut(j) (mpi_inters::set_mpi, src/mpi_inters.cpp:165-172).  Both
// ranks evaluate the common flux from their own side as "left"; the results agree to rounding.
#pragma once
#include "hfx_internal.hpp"

namespace hfx
{

struct MpiArgs
{
  long npairs; // n_fpts_per_inter * n_inters
  int nfpi;
  const int *L, *Rlut;
  long plane; // n_fpts*n_eles of the left block
  const double *disu, *grad, *norm, *tdA;
  const double *fn; // split variant 3: viscous flux of the left side projected on its own normal
  double *tconf, *delta;
  double *out_disu, *out_grad;
  const double *in_disu, *in_grad;
  Phys P;
  int accumulate; // split path: 0 write the total common flux, per-method viscous call: 1 (+=)
  // LES (mpi_inters::send_sgsf_fpts, src/mpi_inters.cpp:339-397): the left block's SGS flux at the flux points -- physical
  // (per-method path: extrapolate_sgsFlux has taken it back) or, sgs_ref != 0, still in reference space (split path,
  // fused mode 2: the consumer applies |J|^-1 J) -- and the exchanged PHYSICAL records; NULL: no closure
  const double *sgsf, *jac_fpts, *detjac_fpts, *in_sgsf;
  double *out_sgsf;
  int sgs_ref;
};

// the left side's physical SGS flux at flux point `o`: f(k, m) = f[k + NF*m]
template <int ND>
__device__ __forceinline__ void mpi_own_sgs(const MpiArgs &a, long o, double (&f)[(ND + 2) * ND])
{
  constexpr int NF = ND + 2;
#pragma unroll
  for (int s = 0; s < NF * ND; s++) f[s] = 0.0;
  if (a.sgs_ref)
    add_sgs_flux<ND>(a.sgsf, a.jac_fpts, a.detjac_fpts, o, a.plane, f);
  else
#pragma unroll
    for (int s = 0; s < NF * ND; s++) f[s] = a.sgsf[o + s * a.plane];
}

// out_buffer_sgsf(fpt, field, dim, inter) = sgsf_fpts_l   (src/mpi_inters.cpp:344-350)
template <int ND>
__global__ __launch_bounds__(256) void mpi_pack_sgsf_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const int j = (int)(q - i * a.nfpi);
  double f[NF * ND];
  mpi_own_sgs<ND>(a, a.L[q], f);
#pragma unroll
  for (int s = 0; s < NF * ND; s++) a.out_sgsf[j + (long)a.nfpi * (s + NF * ND * i)] = f[s];
}

// out_buffer_disu(fpt, field, inter) = disu_fpts_l   (src/mpi_inters.cpp:225-229)
template <int ND>
__global__ __launch_bounds__(256) void mpi_pack_disu_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const int j = (int)(q - i * a.nfpi);
  const long il = a.L[q];
#pragma unroll
  for (int k = 0; k < NF; k++) a.out_disu[j + (long)a.nfpi * (k + NF * i)] = a.disu[il + k * a.plane];
}

// out_buffer_grad_disu(fpt, field, dim, inter) = grad_disu_fpts_l   (src/mpi_inters.cpp:284-289)
template <int ND>
__global__ __launch_bounds__(256) void mpi_pack_grad_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const int j = (int)(q - i * a.nfpi);
  const long il = a.L[q];
#pragma unroll
  for (int s = 0; s < NF * ND; s++) a.out_grad[j + (long)a.nfpi * (s + NF * ND * i)] = a.grad[il + s * a.plane];
}

// mpi_inters::calculate_common_invFlux (src/mpi_inters.cpp:400-483): writes the LEFT side only
template <int ND, bool FAST>
__global__ __launch_bounds__(256) void mpi_common_invflux_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const long il = a.L[q];
  const int jr = a.Rlut[q];
  double ul[NF], ur[NF], n[ND], fn[NF];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu[il + k * a.plane];
    ur[k] = a.in_disu[jr + (long)a.nfpi * (k + NF * i)];
  }
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm[il + m * a.plane];
  if (FAST)
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
  if (a.P.viscous && a.delta != nullptr)
  {
    const double beta = ldg_switch<ND>(a.P.ldg_beta, n);
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      const double uc = 0.5 * (ul[k] + ur[k]) - beta * (ul[k] - ur[k]);
      a.delta[il + k * a.plane] = uc - ul[k];
    }
  }
}

// LDG common solution only (split path, phase 1)
template <int ND>
__global__ __launch_bounds__(256) void mpi_delta_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const long il = a.L[q];
  const int jr = a.Rlut[q];
  double n[ND];
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm[il + m * a.plane];
  const double beta = ldg_switch<ND>(a.P.ldg_beta, n);
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    const double ul = a.disu[il + k * a.plane];
    const double ur = a.in_disu[jr + (long)a.nfpi * (k + NF * i)];
    const double uc = 0.5 * (ul + ur) - beta * (ul - ur);
    a.delta[il + k * a.plane] = uc - ul;
  }
}

// mpi_inters::calculate_common_viscFlux (src/mpi_inters.cpp:485-576): norm_tconf_l += fn_v * tdA_l
template <int ND, bool FAST>
__global__ __launch_bounds__(256) void mpi_common_viscflux_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2, NG = NF * ND;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const long il = a.L[q];
  const int jr = a.Rlut[q];
  double ul[NF], ur[NF], gl[NG], gr[NG], fl[NG], fr[NG], n[ND];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu[il + k * a.plane];
    ur[k] = a.in_disu[jr + (long)a.nfpi * (k + NF * i)];
  }
#pragma unroll
  for (int s = 0; s < NG; s++)
  {
    gl[s] = a.grad[il + s * a.plane];
    gr[s] = a.in_grad[jr + (long)a.nfpi * (s + NG * i)];
  }
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm[il + m * a.plane];
  calc_visf<ND, FAST>(a.P, ul, gl, fl);
  calc_visf<ND, FAST>(a.P, ur, gr, fr);
  if (a.sgsf != nullptr) // src/mpi_inters.cpp:536-551: the physical SGS flux of both sides joins the viscous flux
  {
    double sl[NG];
    mpi_own_sgs<ND>(a, il, sl);
#pragma unroll
    for (int s = 0; s < NG; s++)
    {
      fl[s] += sl[s];
      fr[s] += a.in_sgsf[jr + (long)a.nfpi * (s + NG * i)];
    }
  }
  const double beta = ldg_switch<ND>(a.P.ldg_beta, n);
  const double tl = a.tdA[il];
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
    fn -= a.P.ldg_tau * (ur[k] - ul[k]);
    a.tconf[il + k * a.plane] += fn * tl;
  }
}

// ---- split variant 3: the neighbour sends its projected viscous flux Fn = F_v(u,grad).n_own (n_fields
// doubles per flux point) instead of the gradient (n_fields*n_dims): a third of the bytes on the wire.
// The record layout is out_buffer_disu's; the gradient buffers carry it.
template <int ND>
__global__ __launch_bounds__(256) void mpi_pack_fn_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const int j = (int)(q - i * a.nfpi);
  const long il = a.L[q];
#pragma unroll
  for (int k = 0; k < NF; k++) a.out_grad[j + (long)a.nfpi * (k + NF * i)] = a.fn[il + k * a.plane];
}

// total common flux of a partition face from the exchanged solution and projected fluxes (left side only)
template <int ND>
__global__ __launch_bounds__(256) void mpi_common_flux2_kernel(const MpiArgs a)
{
  constexpr int NF = ND + 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long i = q / a.nfpi;
  const long il = a.L[q];
  const int jr = a.Rlut[q];
  double ul[NF], ur[NF], n[ND], fn[NF];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu[il + k * a.plane];
    ur[k] = a.in_disu[jr + (long)a.nfpi * (k + NF * i)];
  }
#pragma unroll
  for (int m = 0; m < ND; m++) n[m] = a.norm[il + m * a.plane];
  if (a.P.riemann == 0)
    riemann_flux_t<ND, 0, true>(a.P, ul, ur, n, fn);
  else if (a.P.riemann == 2)
    riemann_flux_t<ND, 2, true>(a.P, ul, ur, n, fn);
  else
    riemann_flux_t<ND, 3, true>(a.P, ul, ur, n, fn);
  const double tl = a.tdA[il];
  if (a.P.viscous)
  {
    const double beta = ldg_switch<ND>(a.P.ldg_beta, n);
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      // the neighbour projected on ITS normal = -n
      double fv = (0.5 + beta) * a.fn[il + k * a.plane] - (0.5 - beta) * a.in_grad[jr + (long)a.nfpi * (k + NF * i)];
      fv -= a.P.ldg_tau * (ur[k] - ul[k]);
      a.tconf[il + k * a.plane] = fn[k] * tl + fv * tl;
    }
  }
  else
  {
#pragma unroll
    for (int k = 0; k < NF; k++) a.tconf[il + k * a.plane] = fn[k] * tl;
  }
}

} // namespace hfx
