// fused_hex.hip -- fused per-stage kernels for tensor-product elements (hexes, quads).
//
// One RK stage = the 17 calls of CalcResidual + AdvanceSolution
// (/root/reference/src/solver.cpp:50-223, src/HiFiLES.cpp:201-217).  Executed call by
// call they stream ~64 000 doubles per P4 hex through HBM.  Here a stage is
// TWO persistent kernels that keep everything element-local in LDS / registers and touch
// HBM only for what must cross a kernel boundary:
//
//   gradient kernel  : u -> own + neighbour disu_fpts -> LDG common solution (delta) ->
//                      corrected gradient at upts (written) and at fpts (written: the
//                      neighbours need it)                       [steps 1,3,5(LDG),8]
//   residual kernel  : u, grad -> inviscid + viscous flux at upts -> divergence and
//                      normal flux at fpts; Riemann + LDG common fluxes from own and
//                      neighbour fpt data; correction; RK update; disu_fpts of the NEW
//                      state (double-buffered)                    [steps 4,10-13,16,17,1']
//
// Face coupling is GATHER style: every flux point reads its partner's data through a
// per-point neighbour index and evaluates the common flux itself, in the reference's
// left/right argument order with the LEFT normal, so both sides compute bit-identical
// numbers and nothing is scattered (no atomics, no norm_tconf / delta round trip).
//
// The operators are applied in their registered sparse (ELL) form, every thread owning one
// operator row whose non-zeros live in registers for the whole persistent loop; summation
// orders are the reference's (column ascending, dimension slabs in order).
#include "fused_hex.hpp"

#include <algorithm>
#include <cstring>

#include "physics.hpp"

namespace hfx
{

struct FusedData
{
  int *nbr = nullptr;           // (n_fpts, n_eles) partner offset in the (fpt,ele) plane, -1: none
  unsigned char *meta = nullptr; // bit0: this point is the RIGHT side, bit1: beta sign flipped
  double *fnorm = nullptr;      // (n_fpts, n_eles, n_dims) the LEFT element's unit normal of the pair
  double *disu_alt = nullptr;   // second disu_fpts buffer
  bool built = false;
  int grid = 0;
};

void fused_invalidate(hfx_eles *e)
{
  if (e && e->fused) e->fused->built = false;
}

void fused_destroy(hfx_eles *e)
{
  if (!e || !e->fused) return;
  FusedData *f = e->fused;
  if (f->nbr) (void)hipFree(f->nbr);
  if (f->meta) (void)hipFree(f->meta);
  if (f->fnorm) (void)hipFree(f->fnorm);
  if (f->disu_alt) (void)hipFree(f->disu_alt);
  delete f;
  e->fused = nullptr;
}

struct EllRef
{
  const double *val;
  const int *idx;
  int w; // stored width (nnz_max, >= 1)
};

struct FusedArgs
{
  int n_eles;
  EllRef o0, o1[3], o2[3], o3, o4[3], o5[3], o6;
  const double *detjac_upts, *JGinv_upts, *detjac_fpts, *JGinv_fpts, *tdA_fpts, *fnorm;
  const int *nbr;
  const unsigned char *meta;
  double *u0, *u1;
  const double *disu_cur; // disu_fpts of the current state (read)
  double *disu_next;      // disu_fpts of the new state (written by the residual kernel)
  double *grad_upts, *grad_fpts, *div_out;
  const double *src, *dt_local;
  unsigned long long *nan_flag;
  Phys P;
  // time stepping
  int adv_type, in_step, dt_local_on, write_div;
  double dt, rk_a, rk_b;
};

constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }

template <int W>
__device__ __forceinline__ void load_row(const EllRef &o, int m, int r, double (&v)[W], int (&ix)[W])
{
#pragma unroll
  for (int q = 0; q < W; q++)
  {
    const bool in = q < o.w;
    v[q] = in ? o.val[r + m * q] : 0.0;
    ix[q] = in ? o.idx[r + m * q] : o.idx[r];
  }
}

// ---------------------------------------------------------------------------------------
// gradient kernel
// ---------------------------------------------------------------------------------------
template <int ND, int N>
__global__ __launch_bounds__(((ipow(N, ND) > 2 * ND * ipow(N, ND - 1) ? ipow(N, ND) : 2 * ND * ipow(N, ND - 1)) + 63) / 64 * 64)
void fused_gradient_kernel(const FusedArgs a)
{
  constexpr int NF = ND + 2;
  constexpr int NU = ipow(N, ND);
  constexpr int NFP = 2 * ND * ipow(N, ND - 1);
  constexpr int TB = ((NU > NFP ? NU : NFP) + 63) / 64 * 64;
  __shared__ double su[NF][NU];
  __shared__ double sd[NF][NFP];
  __shared__ double sg[NF * ND][NU];
  const int t = threadIdx.x;
  const bool is_u = t < NU, is_f = t < NFP;
  const long ne = a.n_eles;
  const long plane_u = (long)NU * ne, plane_f = (long)NFP * ne;

  double v0[N], v6[N], v4[ND][N], v5[ND][2];
  int i0[N], i6[N], i4[ND][N], i5[ND][2];
  if (is_f)
  {
    load_row<N>(a.o0, NFP, t, v0, i0);
    load_row<N>(a.o6, NFP, t, v6, i6);
  }
  if (is_u)
  {
#pragma unroll
    for (int d = 0; d < ND; d++)
    {
      load_row<N>(a.o4[d], NU, t, v4[d], i4[d]);
      load_row<2>(a.o5[d], NU, t, v5[d], i5[d]);
    }
  }

  for (long e = blockIdx.x; e < ne; e += gridDim.x)
  {
    // ---- stage u of this element through LDS (5 contiguous runs of NU doubles)
    for (int q = t; q < NF * NU; q += TB)
    {
      const int f = q / NU, p = q - f * NU;
      su[f][p] = a.u0[p + NU * e + f * plane_u];
    }
    __syncthreads();

    // ---- flux points: own value (opp_0 row), partner value, LDG common solution -> delta
    if (is_f)
    {
      const long o = t + NFP * e;
      const long nb = a.nbr[o];
      const unsigned char mt = a.meta[o];
      const bool right = mt & 1;
      const double beta = (mt & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double own = 0.0;
#pragma unroll
        for (int q = 0; q < N; q++) own += v0[q] * su[k][i0[q]];
        const double oth = a.disu_cur[nb + k * plane_f];
        const double ul = right ? oth : own, ur = right ? own : oth;
        // u_c = 1/2 (u_l + u_r) - beta (u_l - u_r)   (src/inters.cpp:637)
        const double uc = 0.5 * (ul + ur) - beta * (ul - ur);
        sd[k][t] = uc - own;
      }
    }
    __syncthreads();

    // ---- solution points: grad_ref = opp_4 u + opp_5 delta ; physical transform ; store
    if (is_u)
    {
      const long p = t + NU * e;
      double JG[ND * ND];
#pragma unroll
      for (int q = 0; q < ND * ND; q++) JG[q] = a.JGinv_upts[p * (ND * ND) + q];
      const double inv_detjac = 1.0 / a.detjac_upts[p];
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double tg[ND], cg[ND];
#pragma unroll
        for (int d = 0; d < ND; d++)
        {
          double g = 0.0;
#pragma unroll
          for (int q = 0; q < N; q++) g += v4[d][q] * su[k][i4[d][q]];
#pragma unroll
          for (int q = 0; q < 2; q++) g += v5[d][q] * sd[k][i5[d][q]];
          tg[d] = g;
          sg[k + NF * d][t] = g; // opp_6 acts on the reference-space corrected gradient
        }
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
        for (int d = 0; d < ND; d++) a.grad_upts[p + (k + NF * d) * plane_u] = cg[d];
      }
    }
    __syncthreads();

    // ---- flux points: grad_fpts = opp_6 grad_ref ; physical transform ; store for the neighbours
    if (is_f)
    {
      const long o = t + NFP * e;
      double JG[ND * ND];
#pragma unroll
      for (int q = 0; q < ND * ND; q++) JG[q] = a.JGinv_fpts[o * (ND * ND) + q];
      const double inv_detjac = 1.0 / a.detjac_fpts[o];
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double tg[ND], cg[ND];
#pragma unroll
        for (int d = 0; d < ND; d++)
        {
          double g = 0.0;
#pragma unroll
          for (int q = 0; q < N; q++) g += v6[q] * sg[k + NF * d][i6[q]];
          tg[d] = g;
        }
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
        for (int d = 0; d < ND; d++) a.grad_fpts[o + (k + NF * d) * plane_f] = cg[d];
      }
    }
    // no barrier needed here: the next iteration's writes to su / sd / sg are each separated from
    // this iteration's last reads of them by the barriers above
  }
}

// ---------------------------------------------------------------------------------------
// residual kernel
// ---------------------------------------------------------------------------------------
template <int ND, int N>
__global__ __launch_bounds__(((ipow(N, ND) > 2 * ND * ipow(N, ND - 1) ? ipow(N, ND) : 2 * ND * ipow(N, ND - 1)) + 63) / 64 * 64)
void fused_residual_kernel(const FusedArgs a)
{
  constexpr int NF = ND + 2;
  constexpr int NU = ipow(N, ND);
  constexpr int NFP = 2 * ND * ipow(N, ND - 1);
  constexpr int TB = ((NU > NFP ? NU : NFP) + 63) / 64 * 64;
  constexpr int N3 = 2 * ND;
  __shared__ double su[NF][NU];
  __shared__ double st[NF * ND][NU];
  __shared__ double sc[NF][NFP];
  const int t = threadIdx.x;
  const bool is_u = t < NU, is_f = t < NFP;
  const long ne = a.n_eles;
  const long plane_u = (long)NU * ne, plane_f = (long)NFP * ne;
  const bool viscous = a.P.viscous;

  double v0[N], v1[ND][N], v2[ND][N], v3[N3];
  int i0[N], i1[ND][N], i2[ND][N], i3[N3];
  if (is_f)
  {
    load_row<N>(a.o0, NFP, t, v0, i0);
#pragma unroll
    for (int d = 0; d < ND; d++) load_row<N>(a.o1[d], NFP, t, v1[d], i1[d]);
  }
  if (is_u)
  {
#pragma unroll
    for (int d = 0; d < ND; d++) load_row<N>(a.o2[d], NU, t, v2[d], i2[d]);
    load_row<N3>(a.o3, NU, t, v3, i3);
  }

  for (long e = blockIdx.x; e < ne; e += gridDim.x)
  {
    for (int q = t; q < NF * NU; q += TB)
    {
      const int f = q / NU, p = q - f * NU;
      su[f][p] = a.u0[p + NU * e + f * plane_u];
    }
    __syncthreads();

    // ---- solution points: total transformed flux, discontinuous divergence
    double div[NF];
    if (is_u)
    {
      const long p = t + NU * e;
      double u[NF], f[NF * ND], JG[ND * ND], td[NF * ND];
#pragma unroll
      for (int k = 0; k < NF; k++) u[k] = su[k][t];
#pragma unroll
      for (int q = 0; q < ND * ND; q++) JG[q] = a.JGinv_upts[p * (ND * ND) + q];
      calc_invf<ND>(a.P.gamma, u, f);
#pragma unroll
      for (int k = 0; k < NF; k++)
#pragma unroll
        for (int l = 0; l < ND; l++)
        {
          double s = 0.0;
#pragma unroll
          for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
          td[k + NF * l] = s;
        }
      if (viscous)
      {
        double g[NF * ND];
#pragma unroll
        for (int q = 0; q < NF * ND; q++) g[q] = a.grad_upts[p + q * plane_u];
        calc_visf<ND>(a.P, u, g, f);
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = td[k + NF * l];
#pragma unroll
            for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
            td[k + NF * l] = s;
          }
      }
#pragma unroll
      for (int q = 0; q < NF * ND; q++) st[q][t] = td[q];
    }
    __syncthreads();

    if (is_u)
    {
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < ND; d++)
#pragma unroll
          for (int q = 0; q < N; q++) s += v2[d][q] * st[k + NF * d][i2[d][q]];
        div[k] = s;
      }
    }

    // ---- flux points: normal discontinuous flux, common fluxes from own + partner data
    if (is_f)
    {
      const long o = t + NFP * e;
      const long nb = a.nbr[o];
      const unsigned char mt = a.meta[o];
      const bool right = mt & 1;
      const double beta = (mt & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
      double own[NF], oth[NF], n[ND], fn[NF];
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < N; q++) s += v0[q] * su[k][i0[q]];
        own[k] = s;
        oth[k] = a.disu_cur[nb + k * plane_f];
      }
#pragma unroll
      for (int m = 0; m < ND; m++) n[m] = a.fnorm[o + m * plane_f];
      const double tdA = a.tdA_fpts[o];
      const double sgn_tdA = right ? -tdA : tdA;
      double ul[NF], ur[NF];
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        ul[k] = right ? oth[k] : own[k];
        ur[k] = right ? own[k] : oth[k];
      }
      riemann_flux<ND>(a.P, ul, ur, n, fn);
      double tconf[NF];
#pragma unroll
      for (int k = 0; k < NF; k++) tconf[k] = fn[k] * sgn_tdA;
      if (viscous)
      {
        double gl[NF * ND], gr[NF * ND], fl[NF * ND], fr[NF * ND];
#pragma unroll
        for (int s = 0; s < NF * ND; s++)
        {
          const double go = a.grad_fpts[o + s * plane_f];
          const double gn = a.grad_fpts[nb + s * plane_f];
          gl[s] = right ? gn : go;
          gr[s] = right ? go : gn;
        }
        calc_visf<ND>(a.P, ul, gl, fl);
        calc_visf<ND>(a.P, ur, gr, fr);
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double fv = 0.0;
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            const double fc = (0.5 + beta) * fl[k + NF * l] + (0.5 - beta) * fr[k + NF * l];
            fv += fc * n[l];
          }
          fv -= a.P.ldg_tau * (ur[k] - ul[k]);
          tconf[k] += fv * sgn_tdA;
        }
      }
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double ntd = 0.0;
#pragma unroll
        for (int d = 0; d < ND; d++)
#pragma unroll
          for (int q = 0; q < N; q++) ntd += v1[d][q] * st[k + NF * d][i1[d][q]];
        sc[k][t] = tconf[k] + -1.0 * ntd; // norm_tconf -= norm_tdisf (src/eles.cpp:1746)
      }
    }
    __syncthreads();

    // ---- solution points: correction, RK update
    if (is_u)
    {
      const long p = t + NU * e;
      const double dj = a.detjac_upts[p];
      const double dt = a.dt_local_on ? a.dt_local[e] : a.dt;
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double dv = div[k];
#pragma unroll
        for (int q = 0; q < N3; q++) dv += v3[q] * sc[k][i3[q]];
        const long q = p + k * plane_u;
        if (dv != dv) atomicMin(a.nan_flag, (unsigned long long)q);
        if (a.write_div) a.div_out[q] = dv;
        const double s = a.src ? a.src[q] : 0.0;
        const double dd = dv / dj;
        double u = su[k][t];
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
            u = 3.0 / 4.0 * u + 1.0 / 4.0 * a.u1[q] + dt / 4.0 * rhs;
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
        su[k][t] = u; // each thread overwrites only its own point, read again after the barrier
      }
    }
    __syncthreads();

    // ---- disu_fpts of the NEW state into the other buffer (the partners still read the old one)
    if (is_f)
    {
      const long o = t + NFP * e;
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < N; q++) s += v0[q] * su[k][i0[q]];
        a.disu_next[o + k * plane_f] = s;
      }
    }
    __syncthreads(); // su is rewritten at the top of the loop
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static EllRef ellref(const Operator &op)
{
  EllRef r;
  r.val = op.ell_val;
  r.idx = op.ell_idx;
  r.w = std::max(op.nnz_max, 1);
  return r;
}

static int tensor_n(const hfx_eles *e)
{
  // N with N^ND = n_upts and 2 ND N^(ND-1) = n_fpts, or 0
  for (int n = 2; n <= 8; n++)
    if (ipow(n, e->n_dims) == e->n_upts && 2 * e->n_dims * ipow(n, e->n_dims - 1) == e->n_fpts) return n;
  return 0;
}

static int fused_build(hfx_eles *e, hfx_inters *const *faces, int nfb)
{
  HFX_CHECK(e->ele_type == 4 || e->ele_type == 1, "fused path: tensor-product elements only (hexes, quads)");
  const int N = tensor_n(e);
  HFX_CHECK(N >= 2 && N <= 6, "fused path: built for orders 1..5 (n_upts %d, n_fpts %d)", e->n_upts, e->n_fpts);
  const int nd = e->n_dims;
  // the registered operators must have the collocated tensor-product sparsity the kernels are sized for
  HFX_CHECK(e->opp_0.nnz_max <= N && e->opp_3.nnz_max <= 2 * nd, "fused path: opp_0 / opp_3 are not tensor-product sparse");
  for (int d = 0; d < nd; d++)
  {
    HFX_CHECK(e->opp_1[d].nnz_max <= N && e->opp_2[d].nnz_max <= N, "fused path: opp_1 / opp_2 are not tensor-product sparse");
    if (e->viscous_ops)
      HFX_CHECK(e->opp_4[d].nnz_max <= N && e->opp_5[d].nnz_max <= 2, "fused path: opp_4 / opp_5 are not tensor-product sparse");
  }
  if (e->viscous_ops) HFX_CHECK(e->opp_6.nnz_max <= N, "fused path: opp_6 is not tensor-product sparse");
  HFX_CHECK(!e->ctx->params.viscous || e->viscous_ops, "fused path: viscous run but the block has no opp_4/5/6");

  if (!e->fused) e->fused = new FusedData();
  FusedData *F = e->fused;
  const long plane_f = (long)e->n_fpts * e->n_eles;
  std::vector<int> nbr(plane_f, -1);
  std::vector<unsigned char> meta(plane_f, 0);
  std::vector<double> norm((size_t)plane_f * nd), fnorm((size_t)plane_f * nd);
  HFX_HIP(hipMemcpy(norm.data(), e->norm_fpts, sizeof(double) * norm.size(), hipMemcpyDeviceToHost));
  fnorm = norm;
  const double beta0 = 1.0; // only the sign decision is stored
  for (int b = 0; b < nfb; b++)
  {
    hfx_inters *f = faces[b];
    HFX_CHECK(f->left == e && f->right == e, "fused path: face blocks must connect the element block to itself");
    const long np = (long)f->n_inters * f->n_fpts_per_inter;
    for (long q = 0; q < np; q++)
    {
      const int il = f->hL[q], ir = f->hR[q];
      nbr[il] = ir;
      nbr[ir] = il;
      // the consistent switch of src/inters.cpp:568-581 on the LEFT normal (exact zero tests)
      double n[3] = {norm[il], norm[il + plane_f], nd == 3 ? norm[il + 2 * plane_f] : 0.0};
      double bt = beta0;
      if (n[0] < 0.)
        bt = -bt;
      else if (n[0] == 0.)
      {
        if ((n[0] + n[1]) < 0.)
          bt = -bt;
        else if ((n[0] + n[1]) == 0)
        {
          if (nd == 3 && (n[0] + n[2]) < 0.) bt = -bt;
        }
      }
      const unsigned char flip = (bt < 0) ? 2 : 0;
      meta[il] = flip;
      meta[ir] = flip | 1;
      for (int d = 0; d < nd; d++) fnorm[ir + d * plane_f] = norm[il + d * plane_f];
    }
  }
  for (long o = 0; o < plane_f; o++)
    HFX_CHECK(nbr[o] >= 0, "fused path: flux point %ld has no partner (boundary / partition faces are not fused yet)", o);
  if (!F->nbr) HFX_HIP(hipMalloc((void **)&F->nbr, sizeof(int) * plane_f));
  if (!F->meta) HFX_HIP(hipMalloc((void **)&F->meta, plane_f));
  if (!F->fnorm) HFX_HIP(hipMalloc((void **)&F->fnorm, sizeof(double) * plane_f * nd));
  if (!F->disu_alt) HFX_HIP(hipMalloc((void **)&F->disu_alt, sizeof(double) * plane_f * e->n_fields));
  HFX_HIP(hipMemcpy(F->nbr, nbr.data(), sizeof(int) * plane_f, hipMemcpyHostToDevice));
  HFX_HIP(hipMemcpy(F->meta, meta.data(), plane_f, hipMemcpyHostToDevice));
  HFX_HIP(hipMemcpy(F->fnorm, fnorm.data(), sizeof(double) * plane_f * nd, hipMemcpyHostToDevice));
  F->grid = (int)std::min<long>(e->n_eles, (long)e->ctx->n_cu * 6);
  F->built = true;
  return 0;
}

static FusedArgs fused_args(hfx_eles *e)
{
  FusedArgs a{};
  FusedData *F = e->fused;
  a.n_eles = e->n_eles;
  a.o0 = ellref(e->opp_0);
  a.o3 = ellref(e->opp_3);
  for (int d = 0; d < e->n_dims; d++)
  {
    a.o1[d] = ellref(e->opp_1[d]);
    a.o2[d] = ellref(e->opp_2[d]);
    if (e->viscous_ops)
    {
      a.o4[d] = ellref(e->opp_4[d]);
      a.o5[d] = ellref(e->opp_5[d]);
    }
  }
  if (e->viscous_ops) a.o6 = ellref(e->opp_6);
  a.detjac_upts = e->detjac_upts; a.JGinv_upts = e->JGinv_upts;
  a.detjac_fpts = e->detjac_fpts; a.JGinv_fpts = e->JGinv_fpts;
  a.tdA_fpts = e->tdA_fpts; a.fnorm = F->fnorm; a.nbr = F->nbr; a.meta = F->meta;
  a.u0 = e->arr[HFX_DISU_UPTS0]; a.u1 = e->arr[HFX_DISU_UPTS1];
  a.disu_cur = e->arr[HFX_DISU_FPTS]; a.disu_next = F->disu_alt;
  a.grad_upts = e->arr[HFX_GRAD_DISU_UPTS]; a.grad_fpts = e->arr[HFX_GRAD_DISU_FPTS];
  a.div_out = e->arr[HFX_DIV_TCONF_UPTS];
  a.src = e->src_nonzero ? e->arr[HFX_SRC_UPTS] : nullptr;
  a.dt_local = e->arr[HFX_DT_LOCAL];
  a.nan_flag = e->nan_flag;
  a.P = e->ctx->phys();
  a.adv_type = e->ctx->params.adv_type;
  a.dt_local_on = e->ctx->params.dt_type == 2;
  a.dt = e->ctx->params.dt;
  return a;
}

template <int ND, int N>
static int launch_stage(hfx_eles *e, FusedArgs &a, bool do_grad, bool do_res)
{
  constexpr int NU = ipow(N, ND), NFP = 2 * ND * ipow(N, ND - 1);
  constexpr int TB = ((NU > NFP ? NU : NFP) + 63) / 64 * 64;
  const int grid = e->fused->grid;
  if (do_grad) hipLaunchKernelGGL((fused_gradient_kernel<ND, N>), dim3(grid), dim3(TB), 0, e->ctx->stream, a);
  if (do_res) hipLaunchKernelGGL((fused_residual_kernel<ND, N>), dim3(grid), dim3(TB), 0, e->ctx->stream, a);
  HFX_HIP(hipGetLastError());
  return 0;
}

static int dispatch_stage(hfx_eles *e, FusedArgs &a, bool do_grad, bool do_res)
{
  const int N = tensor_n(e);
  if (e->n_dims == 3)
  {
    switch (N)
    {
    case 2: return launch_stage<3, 2>(e, a, do_grad, do_res);
    case 3: return launch_stage<3, 3>(e, a, do_grad, do_res);
    case 4: return launch_stage<3, 4>(e, a, do_grad, do_res);
    case 5: return launch_stage<3, 5>(e, a, do_grad, do_res);
    case 6: return launch_stage<3, 6>(e, a, do_grad, do_res);
    }
  }
  else
  {
    switch (N)
    {
    case 2: return launch_stage<2, 2>(e, a, do_grad, do_res);
    case 3: return launch_stage<2, 3>(e, a, do_grad, do_res);
    case 4: return launch_stage<2, 4>(e, a, do_grad, do_res);
    case 5: return launch_stage<2, 5>(e, a, do_grad, do_res);
    case 6: return launch_stage<2, 6>(e, a, do_grad, do_res);
    }
  }
  set_error("fused path: no kernel for N = %d, n_dims = %d", N, e->n_dims);
  return 1;
}

// one RK stage; which != 0 restricts to the gradient (1) or residual (2) kernel (for timing)
static int fused_stage(hfx_eles *e, int in_step, bool last_stage, int which = 0)
{
  FusedArgs a = fused_args(e);
  const hfx_params &p = e->ctx->params;
  a.in_step = in_step;
  a.rk_a = (p.adv_type >= 3) ? p.RK_a[in_step] : 0.0;
  a.rk_b = (p.adv_type >= 3) ? p.RK_b[in_step] : 0.0;
  a.write_div = last_stage ? 1 : 0; // the monitors read the divergence of a step's last stage
  const bool do_grad = p.viscous && which != 2, do_res = which != 1;
  if (dispatch_stage(e, a, do_grad, do_res)) return 1;
  if (do_res)
  {
    // the new state's disu_fpts is in the other buffer now
    std::swap(e->arr[HFX_DISU_FPTS], e->fused->disu_alt);
  }
  return 0;
}

int fused_run_steps(hfx_eles *e, hfx_inters *const *faces, int nfb, int n_steps)
{
  HFX_CHECK(e->n_eles > 0, "fused path: empty element block");
  if (!e->fused || !e->fused->built)
    if (fused_build(e, faces, nfb)) return 1;
  if (n_steps <= 0) return 0;
  const int adv = e->ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  HFX_CHECK(e->ctx->params.dt_type != 2 || e->arr[HFX_DT_LOCAL], "dt_type 2 needs HFX_DT_LOCAL uploaded");
  // disu_fpts of the current state (the caller may have changed disu_upts since the last call)
  if (hfx_eles_extrapolate_solution(e)) return 1;
  for (int s = 0; s < n_steps; s++)
    for (int rk = 0; rk < nst; rk++)
      if (fused_stage(e, rk, rk == nst - 1)) return 1;
  return 0;
}

int fused_time_kernels(hfx_eles *e, hfx_inters *const *faces, int nfb, int reps, double *ms, char *names, int names_len)
{
  if (!e->fused || !e->fused->built)
    if (fused_build(e, faces, nfb)) return 1;
  const int adv = e->ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  hipStream_t st = e->ctx->stream;
  hipEvent_t ev[3];
  for (auto &x : ev) HFX_HIP(hipEventCreate(&x));
  if (hfx_eles_extrapolate_solution(e)) return 1;
  double acc[2] = {0, 0};
  for (int r = 0; r < reps; r++)
  {
    const int rk = r % nst;
    HFX_HIP(hipEventRecord(ev[0], st));
    if (e->ctx->params.viscous && fused_stage(e, rk, false, 1)) return 1;
    HFX_HIP(hipEventRecord(ev[1], st));
    if (fused_stage(e, rk, rk == nst - 1, 2)) return 1;
    HFX_HIP(hipEventRecord(ev[2], st));
    HFX_HIP(hipStreamSynchronize(st));
    float t0 = 0, t1 = 0;
    HFX_HIP(hipEventElapsedTime(&t0, ev[0], ev[1]));
    HFX_HIP(hipEventElapsedTime(&t1, ev[1], ev[2]));
    acc[0] += t0;
    acc[1] += t1;
  }
  for (auto &x : ev) (void)hipEventDestroy(x);
  for (int i = 0; i < 8; i++) ms[i] = 0.0;
  ms[0] = acc[0] / reps;
  ms[1] = acc[1] / reps;
  snprintf(names, names_len, "fused_gradient_kernel,fused_residual_kernel");
  return 0;
}

void fused_kernel_bytes(const hfx_eles *e, double *bytes)
{
  // ALGORITHMIC HBM bytes per launch of the two kernels (DESIGN.md "fused path"): what each kernel
  // must read / write given that gradients and flux-point data cross a kernel boundary; the
  // grad_disu_upts round trip between the two kernels is an implementation choice and not counted.
  const double nu = e->n_upts, nfp = e->n_fpts, nf = e->n_fields, nd = e->n_dims, ne = e->n_eles;
  const double grad = nu * nf + nu * (nd * nd + 1) + nfp * nf /*partner disu*/ + nfp * (nd * nd + 1) + nfp * nf * nd /*write*/;
  const double res = nu * nf + nu * (nd * nd + 1) + nfp * nf + 2 * nfp * nf * nd + nfp * (nd + 1) + 3 * nu * nf /*u1 r, u0 u1 w*/ +
                     nfp * nf /*new disu*/;
  for (int i = 0; i < 8; i++) bytes[i] = 0.0;
  bytes[0] = 8.0 * grad * ne;
  bytes[1] = 8.0 * res * ne;
}

} // namespace hfx
