// split3_kernels.hpp -- split fused stage, variant 3 (the default): fluxes evaluated by the kernel that has the corrected gradient in registers
// (device code of the split fused stage; included by fused_hex.hip only -- one translation unit, so that every kernel is
// instantiated once)
#pragma once
#include "split2_kernels.hpp"

namespace hfx
{

// =======================================================================================
// SPLIT path, variant 3 ("flux in the gradient kernel"): the element kernel that has the corrected
// gradient in registers goes on to the fluxes, so that neither grad_disu_upts nor grad_disu_fpts
// crosses HBM.  What leaves it is what the face and update kernels need, 5 instead of 15 doubles
// per point:
//
//   face_delta_kernel   unchanged
//   split_flux_kernel   u, delta -> corrected gradient (upts: registers, fpts: registers) ->
//                       tdisf = JGinv (F_inv + F_visc) in LDS -> div_tdisf (upts), norm_tdisf (fpts),
//                       and the viscous flux of this side projected on its OWN normal, Fn = F_v(u,grad).n
//   face_flux2_kernel   Riemann flux + LDG combination (1/2+b) Fn_L - (1/2-b) Fn_R - tau (u_R-u_L)
//   split_update_kernel div_tdisf + opp_3 (norm_tconf - norm_tdisf), RK update, disu_fpts of the new state
//
// The right side's flux is projected on the right element's own normal (= -left normal up to
// rounding on a conforming mesh) instead of on the left normal as src/inters.cpp:616-633 does:
// a last-bit difference, inside the fused paths' documented rounding tolerance.
// =======================================================================================
#ifndef HFX_SPLIT2_WAVES
#define HFX_SPLIT2_WAVES 3
#endif

struct Split2Args
{
  int n_eles;
  const unsigned *pk_g, *pk_r;
  const double *tab_g, *tab_r;
  const int *o1m_dim;
  const double *detjac_upts, *JGinv_upts, *detjac_fpts, *JGinv_fpts, *norm_fpts;
  double *u0, *u1;
  const double *delta, *tconf;
  double *fn_fpts;  // (n_fpts,n_eles,n_fields) projected viscous flux of this side
  double *ntd_fpts; // norm_tdisf_fpts
  int folded;       // 1: div holds div_tdisf - opp_3 norm_tdisf (sum-factorised flux kernel), norm_tdisf_fpts is not written
  double *div;      // div_tdisf (flux kernel) -> read by the update kernel, which may overwrite it with div_tconf
  double *disu_next;
  double *grad_upts, *grad_fpts; // optional outputs (NULL: not written)
  int xcd_order;                 // EleOrder: contiguous element ranges per XCD
  const double *tdisf_in;        // over-integration: the transformed inviscid flux, already evaluated (NULL: computed here); the loader-wave
                                 // form takes the over-integration kernel's FOLDED result there instead: sum_l Dc[l] tdisf_l, (n_upts, n_eles, n_fields)
  const unsigned char *meta;     // with grad_fpts: only flux points whose bit2 is set are written (NULL: all)
  int simd_roles;    // 1: the waves' parts are dealt by SIMD (split_flux_tensor_kernel)
  int light_short;   // 1: a wave without solution points runs the flux-point physics alone instead of the paired form on dummies
  // split_flux_tensor_kernel, loader-wave form: the LDG correction of a flux point is formed IN the kernel from the partner's
  // flux-point solution (nbr: (partner offset << 2) | (beta sign flipped) << 1 | (this point is the right side); -1: a boundary
  // or partition-face point, whose correction its one-sided kernel has left in `delta`).  NULL: `delta` holds all of them.
  const int *nbr;
  const double *disu;
  int stamp_it;      // which iteration of workgroup 0 is stamped
  long long *stamps; // diagnostics (tools/flux_phase_stamps.py): cycle counter of wave w of workgroup 0 at the phase boundaries
  // update kernel: opp_3 and opp_0 in ELL form (values, columns, width), rows held in registers
  const double *o3v, *o0v;
  const int *o3i, *o0i;
  int o3w, o0w;
  const double *src, *dt_local;
  unsigned long long *nan_flag;
  Phys P;
  int adv_type, in_step, dt_local_on, write_div, need_u1;
  double dt, rk_a, rk_b;
  // LES closure evaluated in the flux kernel (split_flux_tensor_kernel<..., LES = true>): eles::calc_sgsf_upts at the solution
  // points; wall_distance (n_upts,n_eles,n_dims) for the Smagorinsky damping; tdA_fpts to take the extrapolated SGS flux of a
  // flux point from F~ . n~ to F . n
  // flux / update kernel on a LIST of elements (partitioned blocks: the elements that own partition-face points in a launch of
  // their own, so that the exchanges of their data run beside the work on the others); NULL: all elements
  const int *ele_list;
  long n_list;
  LesParams les;
  const double *les_len2, *tdA_fpts; // les_len2 (n_upts,n_eles): the closure's squared length scale (calc_sgsf_fast)
};

template <int ND, int N>
__global__ __launch_bounds__((SGeo<ND, N>::TB), HFX_SPLIT2_WAVES) void split_flux_kernel(const Split2Args a)
{
  using G = Geo<ND, N>;
  constexpr int NF = G::NF, NU = G::NU, NFP = G::NFP, WN = G::WN, TB = SGeo<ND, N>::TB;
  constexpr int NG = NF * ND;
  // packed words of this thread: opp_4[d] | opp_5[d] | opp_2[d] (upt row) | opp_0 | opp_6 | merged opp_1 (fpt row)
  constexpr int O2 = G::G_WU, O0 = O2 + ND * WN, O6 = O0 + WN, O1 = O6 + WN, PW = O1 + WN;
  constexpr int UNION = cmax(NF * (NU + NFP), NG * NU);
  __shared__ double tabg[MAX_TAB];
  __shared__ double tabr[MAX_TAB];
  __shared__ double sA[UNION]; // su | sd, later st
  __shared__ double sg[NG * NU];
  double *const su = sA, *const sd = sA + NF * NU, *const st = sA;
  const int t = threadIdx.x;
  const int tu = t < NU ? t : NU - 1, tf = t < NFP ? t : NFP - 1;
  const bool is_u = t < NU, is_f = t < NFP;
  const long ne = a.n_eles, plane_u = (long)NU * ne, plane_f = (long)NFP * ne;
  const bool viscous = a.P.viscous;
  for (int q = t; q < MAX_TAB; q += TB)
  {
    tabg[q] = a.tab_g[q];
    tabr[q] = a.tab_r[q];
  }
  unsigned pw[PW];
#pragma unroll
  for (int d = 0; d < ND; d++)
  {
#pragma unroll
    for (int i = 0; i < WN; i++)
    {
      pw[d * WN + i] = a.pk_g[G::G_O4 + (d * WN + i) * NU + tu];
      pw[O2 + d * WN + i] = a.pk_r[G::R_O2 + (d * WN + i) * NU + tu];
    }
    pw[ND * WN + d] = a.pk_g[G::G_O5 + d * NU + tu];
  }
#pragma unroll
  for (int i = 0; i < WN; i++)
  {
    pw[O0 + i] = a.pk_g[G::G_O0 + i * NFP + tf];
    pw[O6 + i] = a.pk_g[G::G_O6 + i * NFP + tf];
    pw[O1 + i] = a.pk_r[G::R_O1 + i * NFP + tf];
  }
  const int d1 = a.o1m_dim[tf];

  for (long e = blockIdx.x; e < ne; e += gridDim.x)
  {
    const long p = tu + NU * e, o = tf + NFP * e;
    for (int q = t; q < NF * NU; q += TB)
    {
      const int f = q / NU, p2 = q - f * NU;
      su[q] = a.u0[p2 + NU * e + f * plane_u];
    }
    if (viscous)
      for (int q = t; q < NF * NFP; q += TB)
      {
        const int f = q / NFP, p2 = q - f * NFP;
        sd[q] = a.delta[p2 + NFP * e + f * plane_f];
      }
    double JG[ND * ND];
#pragma unroll
    for (int q = 0; q < ND * ND; q++) JG[q] = a.JGinv_upts[p * (ND * ND) + q];
    const double inv_detjac = viscous ? 1.0 / a.detjac_upts[p] : 0.0;
    __syncthreads();
    double tfl[NG]; // transformed total flux at this solution point
    double uf[NF];  // solution at this flux point
    if (is_u)
    {
      double u[NF], f[NG];
#pragma unroll
      for (int k = 0; k < NF; k++) u[k] = su[k * NU + tu];
      if (a.tdisf_in)
      {
#pragma unroll
        for (int q = 0; q < NG; q++) tfl[q] = a.tdisf_in[p + q * plane_u];
      }
      else
      {
        calc_invf<ND, true>(a.P.gamma, u, f);
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
            tfl[k + NF * l] = s;
          }
      }
      if (viscous)
      {
        double gr[NG];
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double tg[ND], cg[ND];
          tg[0] = row_dot<N, 0, PW>(pw, tabg, &su[k * NU], 0.0);
          tg[1] = row_dot<N, WN, PW>(pw, tabg, &su[k * NU], 0.0);
          if (ND == 3) tg[ND - 1] = row_dot<N, (ND - 1) * WN, PW>(pw, tabg, &su[k * NU], 0.0);
          tg[0] = row_dot<2, ND * WN + 0, PW>(pw, tabg, &sd[k * NFP], tg[0]);
          tg[1] = row_dot<2, ND * WN + 1, PW>(pw, tabg, &sd[k * NFP], tg[1]);
          if (ND == 3) tg[ND - 1] = row_dot<2, ND * WN + ND - 1, PW>(pw, tabg, &sd[k * NFP], tg[ND - 1]);
#pragma unroll
          for (int d = 0; d < ND; d++) sg[(k + NF * d) * NU + tu] = tg[d];
          to_physical<ND>(inv_detjac, JG, tg, cg);
#pragma unroll
          for (int d = 0; d < ND; d++) gr[k + NF * d] = cg[d];
        }
        if (a.grad_upts)
#pragma unroll
          for (int q = 0; q < NG; q++) a.grad_upts[p + q * plane_u] = gr[q];
        calc_visf<ND, true>(a.P, u, gr, f);
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = tfl[k + NF * l];
#pragma unroll
            for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
            tfl[k + NF * l] = s;
          }
      }
    }
    if (viscous && is_f)
    {
#pragma unroll
      for (int k = 0; k < NF; k++) uf[k] = row_dot<N, O0, PW>(pw, tabg, &su[k * NU], 0.0);
    }
    __syncthreads(); // every reader of su / sd is done: the region becomes st
    if (is_u)
    {
#pragma unroll
      for (int q = 0; q < NG; q++) st[q * NU + tu] = tfl[q];
    }
    __syncthreads();
    if (is_u)
    {
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = row_dot<N, O2, PW>(pw, tabr, &st[k * NU], 0.0);
        s = row_dot<N, O2 + WN, PW>(pw, tabr, &st[(k + NF) * NU], s);
        if (ND == 3) s = row_dot<N, O2 + (ND - 1) * WN, PW>(pw, tabr, &st[(k + NF * (ND - 1)) * NU], s);
        a.div[p + k * plane_u] = s;
      }
    }
    if (is_f)
    {
#pragma unroll
      for (int k = 0; k < NF; k++)
        a.ntd_fpts[o + k * plane_f] = row_dot<N, O1, PW>(pw, tabr, &st[(k + NF * d1) * NU], 0.0);
      if (viscous)
      {
        double JF[ND * ND], n[ND], grf[NG], fq[NG];
#pragma unroll
        for (int q = 0; q < ND * ND; q++) JF[q] = a.JGinv_fpts[o * (ND * ND) + q];
        const double inv_df = 1.0 / a.detjac_fpts[o];
#pragma unroll
        for (int m = 0; m < ND; m++) n[m] = a.norm_fpts[o + m * plane_f];
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double tg[ND], cg[ND];
#pragma unroll
          for (int d = 0; d < ND; d++) tg[d] = row_dot<N, O6, PW>(pw, tabg, &sg[(k + NF * d) * NU], 0.0);
          to_physical<ND>(inv_df, JF, tg, cg);
#pragma unroll
          for (int d = 0; d < ND; d++) grf[k + NF * d] = cg[d];
        }
        if (a.grad_fpts && (a.meta == nullptr || (a.meta[o] & 4)))
#pragma unroll
          for (int q = 0; q < NG; q++) a.grad_fpts[o + q * plane_f] = grf[q];
        calc_visf<ND, true>(a.P, uf, grf, fq);
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < ND; l++) s += fq[k + NF * l] * n[l];
          a.fn_fpts[o + k * plane_f] = s;
        }
      }
    }
    __syncthreads();
  }
}


// =======================================================================================
// Sum-factorised ("tensor") form of split_flux_kernel.  On a tensor-product element every operator
// row is a 1-D stencil along one pencil of N points, with coefficients that depend only on the
// position along the pencil.  The contractions are therefore done PENCIL-wise: a work item reads
// its N inputs from LDS once and produces N outputs (N^2 FMAs against the 1-D matrix, which is
// wave-uniform and lives in scalar registers), instead of one thread per output row reading N
// inputs and N dictionary values (2N LDS reads + unpacking per output).  ~3x fewer LDS and VALU
// instructions; the FMAs of an output run over ascending column exactly as in the row form.
// =======================================================================================
#ifndef HFX_SPLIT2T_WAVES
#define HFX_SPLIT2T_WAVES 2
#endif

// constant address space: loads with a wave-uniform address are selected as scalar loads
typedef const double __attribute__((address_space(4))) *cdptr;

template <int ND, int N>
struct TGeo
{
  using G = Geo<ND, N>;
  static constexpr int L = ipow(N, ND - 1);           // pencils per direction
  static constexpr int ITEMS_D = G::NF * L;           // (field, pencil) items per direction
  static constexpr int SP = ((ITEMS_D + 63) / 64) * 64; // padded so that a wave works on one direction
  static constexpr int TB = SGeo<ND, N>::TB;
  static constexpr int ROUNDS = (ND * SP + TB - 1) / TB;
  static constexpr int C_D = 0, C_5 = N * N, C_LF = C_5 + ND * 2 * N, C_L1 = C_LF + ND * 2 * N, C_3 = C_L1 + ND * 2 * N;
  static constexpr int I_PF = 0, I_FDQ = ND * L * 2, I_FB = I_FDQ + G::NFP;
};

// LDS read that the load/store optimiser leaves alone: merged pairs become ds_read2_b64, which runs at
// half the rate of two ds_read_b64 and banks modulo 32 dwords instead of 64 (MI355X_MICROARCH.md, LDS)
__device__ __forceinline__ double ldsv(const double *p)
{
  return *(const volatile __attribute__((address_space(3))) double *)p;
}

// Element order of a persistent workgroup.  Workgroups whose ids are equal modulo 8 share an XCD and its L2
// (MI355X_MICROARCH.md, workgroup dispatch): each such group walks ONE contiguous eighth of the elements, so that the
// 128-byte lines two neighbouring elements share (an element's run per field is 1000 / 1200 bytes, not a multiple of
// a line) are fetched into one L2 once instead of into two.  HFX_NO_XCD_ORDER=1 (or a grid that is no multiple of 8):
// element = workgroup id + k * grid.
struct EleOrder
{
  long ne, chunk;
  int per, slot, xcd;
  bool remap;
  const int *list; // optional: the launch works on list[0 .. n_list) instead of all elements (partitioned blocks)
  long n_list;
  __device__ __forceinline__ EleOrder(long n_eles, bool want, const int *ele_list = nullptr, long n_ele_list = 0)
      : ne(n_eles), list(ele_list), n_list(n_ele_list)
  {
    remap = want && (gridDim.x % 8 == 0);
    per = gridDim.x / 8;
    slot = blockIdx.x / 8;
    xcd = blockIdx.x % 8;
    chunk = (ne + 7) / 8;
  }
  // k-th element of this workgroup, -1 past the end
  __device__ __forceinline__ long at(long k) const
  {
    if (list != nullptr)
    {
      const long q = blockIdx.x + k * gridDim.x; // (uniform: a scalar load)
      return q < n_list ? (long)list[q] : -1;
    }
    if (!remap)
    {
      const long e = blockIdx.x + k * gridDim.x;
      return e < ne ? e : -1;
    }
    const long l = slot + k * per, e = xcd * chunk + l;
    return (l < chunk && e < ne) ? e : -1;
  }
};

// A global array of doubles addressed as [wave-uniform offset + lane offset].  BUF: through a buffer descriptor
// (4 SGPRs per array), the uniform part in the instruction's scalar offset and the lane part in ONE 32-bit VGPR that
// is loop invariant -- instead of a 64-bit VGPR address per (array, field), which the compiler otherwise keeps live
// across the element loop (a third of the registers of these kernels) and recomputes with 64-bit VALU adds.
// Offsets are 32-bit byte counts: the launcher selects BUF only when every array is smaller than 4 GiB.
typedef unsigned hfx_v2u __attribute__((ext_vector_type(2)));
template <bool BUF>
struct GArr
{
  double *p;
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ GArr(const double *q, long n) : p(const_cast<double *>(q))
  {
    if constexpr (BUF) r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, q ? (int)(unsigned)(n * 8) : 0, 0x00020000);
  }
  // uni: element offset common to the wave, lane: this lane's element offset (small, loop invariant)
  __device__ __forceinline__ double ld(long uni, unsigned lane) const
  {
    if constexpr (BUF)
      return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8u, (unsigned)uni * 8u, 0));
    else
      return p[uni + lane];
  }
  __device__ __forceinline__ void st(long uni, unsigned lane, double v) const
  {
    if constexpr (BUF)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(hfx_v2u, v), r, lane * 8u, (unsigned)uni * 8u, 0);
    else
      p[uni + lane] = v;
  }
};

// transformed gradient along direction D on one pencil: sg = Dm x + c5[.][0] delta_a + c5[.][1] delta_b
template <int ND, int N, int D>
__device__ __forceinline__ void pencil_grad(cdptr coef, const double *su_p, const double *sda, const double *sdb, double *sg_p)
{
  using T = TGeo<ND, N>;
  constexpr int S = ipow(N, D);
  double x[N];
#pragma unroll
  for (int m = 0; m < N; m++) x[m] = ldsv(su_p + m * S);
  const double da = ldsv(sda), db = ldsv(sdb);
#pragma unroll
  for (int mp = 0; mp < N; mp++)
  {
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < N; m++) acc += coef[T::C_D + mp * N + m] * x[m];
    acc += coef[T::C_5 + (D * 2 + 0) * N + mp] * da;
    acc += coef[T::C_5 + (D * 2 + 1) * N + mp] * db;
    sg_p[mp * S] = acc;
  }
}

// d/dxi_D of one pencil of the transformed flux
template <int ND, int N, int D>
__device__ __forceinline__ void pencil_div(cdptr coef, const double *st_p, double *sp_p)
{
  using T = TGeo<ND, N>;
  constexpr int S = ipow(N, D);
  double x[N];
#pragma unroll
  for (int m = 0; m < N; m++) x[m] = ldsv(st_p + m * S);
#pragma unroll
  for (int mp = 0; mp < N; mp++)
  {
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < N; m++) acc += coef[T::C_D + mp * N + m] * x[m];
    sp_p[mp * S] = acc;
  }
}

// LDS-DMA of one contiguous run: `lanes` lanes of 16 bytes, lane l of wave-instruction j takes bytes
// [16 (l + 64 j), +16) of the run to the same offset of the LDS region; the partial last instruction is masked (an
// inactive lane writes nothing).
typedef __attribute__((address_space(3))) double *lds_dp;
template <bool BUF>
__device__ __forceinline__ void dma16_region(const GArr<BUF> &g, lds_dp lds_dst, int lanes, int lane, unsigned soff)
{
  typedef __attribute__((address_space(3))) void *lds_vp [[maybe_unused]];
  [[maybe_unused]] __attribute__((address_space(3))) char *base = (__attribute__((address_space(3))) char *)lds_dst;
#pragma unroll
  for (int j = 0; j < 16; j++) // `lanes` is a compile-time constant at every call: the loop folds to ceil(lanes / 64) instructions
    if (64 * j < lanes && lane + 64 * j < lanes)
    {
#if __HIP_DEVICE_COMPILE__ // the 16-byte form exists on gfx950 only: the host pass of this file must not see it
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g.r, (lds_vp)(base + 1024 * j), 16, (unsigned)(lane + 64 * j) * 16u, soff, 0, 0);
#endif
    }
}

// does the loader-wave form fit this element size?  (vmcnt counts at most 63 DMA instructions in flight; two workgroups
// of input slots + metric slot + work regions must fit the CU's 160 KiB of LDS)
template <int ND, int N>
constexpr bool loader_wave_fits()
{
  using G = Geo<ND, N>;
  constexpr int NF = G::NF, NU = G::NU, NFP = G::NFP, NQ = ND * ND, NG = NF * ND;
  constexpr int UJ = NF * (((NU + 1) / 2 + 63) / 64), DJ = NF * (((NFP + 1) / 2 + 63) / 64);
  constexpr int L_JGU = (NQ * NU + 1) / 2, L_DJU = (NU + 1) / 2, L_JGF = (NQ * NFP + 1) / 2, L_DJF = (NFP + 1) / 2;
  constexpr int N_M = (L_JGU + 63) / 64 + (L_DJU + 63) / 64 + (L_JGF + 63) / 64 + (1 + ND) * ((L_DJF + 63) / 64);
  constexpr long lds = 8L * (cmax(NF * (NU + NFP), NG * NU) + NG * NU + 2 * NF * (NU + 1 + NFP) + NQ * (NU + NFP) + NU + (1 + ND) * NFP + 16);
  return UJ + DJ <= 63 && N_M <= 63 && 2 * lds <= 160 * 1024;
}

// LW: a LOADER WAVE (one extra wave per workgroup) brings the next element's state and LDG corrections straight into
// an LDS slot by buffer_load ... lds -- no registers, requested a whole element ahead and counted on the loader's own
// vmcnt, so the compute waves never wait for them (the register prefetch of the LW = false form is issued in phase C
// and still needs ~2 500 cycles at the top of the next iteration: the kernel is bound by bytes in flight per CU).
// LES: the eddy-viscosity / similarity closure of an LES run (eles::calc_sgsf_upts, src/eles.cpp:2395-2650) is evaluated HERE,
// on the corrected gradient the solution-point threads hold, instead of by a pointwise kernel on a gradient array in HBM:
//   * solution points: F_sgs joins the total flux (evaluate_viscFlux adds it, src/eles.cpp:2360-2392);
//   * flux points: the reference extrapolates the TRANSFORMED SGS flux (sgsf_fpts = opp_0 sgsf_upts), takes it back to
//     physical space and adds it to each side's viscous flux in the common-flux sweep (src/eles.cpp:2817-2893,
//     src/int_inters.cpp:299-313).  What the face kernel needs of it is F_sgs . n = (F~_sgs . n~) / tdA, and on a
//     tensor-product element F~_sgs . n~ at a flux point is +- the 1-D extrapolation of ONE component along the point's pencil:
//     a second, short pencil pass (two more barriers) over the transformed SGS flux the solution-point threads kept in
//     registers, whose result joins Fn before it is stored.
// So an LES stage moves the bytes of a plain one (+ tdA, the Leonard terms of the similarity models) in the same three launches.
template <int ND, int N, int WV, bool BUF, bool OI, bool LW, bool GA = false, bool LES = false>
__global__ __launch_bounds__((SGeo<ND, N>::TB + (LW ? 64 : 0)), WV) void split_flux_tensor_kernel(const Split2Args a,
                                                                                               const double *coef_g,
                                                                                               const int *tidx)
{
  using G = Geo<ND, N>;
  using T = TGeo<ND, N>;
  constexpr int NF = G::NF, NU = G::NU, NFP = G::NFP, TB = T::TB, NG = NF * ND, L = T::L, ROUNDS = T::ROUNDS;
  constexpr int R1 = cmax(NF * (NU + NFP), NG * NU);
  // LW: two input slots; a slot holds u and delta, field after field (an LDS-DMA writes lane-linear: base + 16 * lane)
  // (LW) field stride of the state in the slot: whole 16-byte lanes (P4: 125 -> 126 doubles); NFP is even already
  constexpr int NUS = LW ? ((NU + 1) & ~1) : NU;
  constexpr int U_DW = 2 * NF * NUS, D_DW = 2 * NF * NFP, SLOT = (U_DW + D_DW) / 2;
  static_assert(!LW || BUF, "the loader wave addresses through buffer descriptors");
  static_assert(!LES || (LW && !OI), "the in-kernel LES closure belongs to the loader-wave form without over-integration");
  __shared__ double sA[R1];      // su | sd, later st
  __shared__ double sB[NG * NU]; // sg, later the per-direction parts of the divergence
  __shared__ double s_in[LW ? 2 * SLOT : 1];
  // LW: ONE metric slot (volume metrics, flux-point metrics, own normals of the element); it is free from the end of
  // phase B (barrier 3) on, which is when the loader refills it for the next element.  Regions start on 16-byte
  // boundaries (the loader moves 16 bytes per lane).
  constexpr int NQ = ND * ND;
  constexpr int O_JGU = 0, O_DJU = O_JGU + ((NQ * NU + 1) & ~1), O_JGF = O_DJU + ((NU + 1) & ~1), O_DJF = O_JGF + ((NQ * NFP + 1) & ~1),
                NFPP = (NFP + 1) & ~1, O_NRM = O_DJF + NFPP, MET = O_NRM + ND * NFPP;
  __shared__ double s_met[LW ? MET : 1];
  double *su = sA, *sd = sA + NF * NU;
  double *const st = sA, *const sg = sB, *const sp = sB;
  const cdptr coef = (cdptr)(uintptr_t)coef_g;
  // ---- which wave plays which part.  With 4 waves (P4 hexes: two HEAVY waves that own solution points and flux points,
  // a LIGHT one with the remaining flux points, the loader) the hardware puts the 4 waves of a workgroup on the 4 SIMDs
  // of the CU, rotated from one workgroup to the next -- but not so that the heavy waves of the two co-resident
  // workgroups avoid each other (tools/probes/hwid_probe.hip: one SIMD gets two heavy waves, another the two idle
  // ones).  The parts are therefore dealt by SIMD: the workgroup in the even wave slot plays loader / heavy / heavy /
  // light on SIMDs 0..3, the one in the odd slot heavy / light / loader / heavy, so that every SIMD carries exactly one
  // heavy wave.  `t` below is the VIRTUAL thread number 64 * part + lane.
  // LW: the 1-D tables of the pencil phases in LDS, read as 16-byte broadcasts into registers at the top of a phase (through
  // the scalar cache every two or three FMAs waited for their own s_load: the kernel is out of scalar registers).  For
  // phase C the correction is folded into the matrix: Dc[d] = D - c3[d][0] (sgn L)[d][0]^T - c3[d][1] (sgn L)[d][1]^T, so
  // that div_tdisf - opp_3 norm_tdisf along a pencil is ONE 5x5 product (rounded once here, in every workgroup alike).
  constexpr int NN2 = (N * N + 1) & ~1, NP = (N + 1) & ~1;
  constexpr int S_D = 0, S_DC = NN2, S_C5 = S_DC + ND * NN2, S_TOT = S_C5 + ND * 2 * NP;
  __shared__ __attribute__((aligned(16))) double s_coef[LW ? S_TOT : 2];
  if constexpr (LW)
  {
    for (int i = threadIdx.x; i < S_TOT; i += blockDim.x)
    {
      double v = 0.0;
      if (i < S_DC)
        v = (i < N * N) ? coef_g[T::C_D + i] : 0.0;
      else if (i < S_C5)
      {
        const int d = (i - S_DC) / NN2, q = (i - S_DC) - d * NN2;
        if (q < N * N)
        {
          const int mp = q / N, m = q - mp * N;
          const double ta = coef_g[T::C_3 + (d * 2 + 0) * N + mp] * (coef_g[T::C_L1 + (d * 2 + 0) * N] * coef_g[T::C_LF + (d * 2 + 0) * N + m]);
          const double tb = coef_g[T::C_3 + (d * 2 + 1) * N + mp] * (coef_g[T::C_L1 + (d * 2 + 1) * N] * coef_g[T::C_LF + (d * 2 + 1) * N + m]);
          v = coef_g[T::C_D + q] - ta - tb;
        }
      }
      else
      {
        const int row = (i - S_C5) / NP, mp = (i - S_C5) - row * NP;
        v = (mp < N) ? coef_g[T::C_5 + row * N + mp] : 0.0;
      }
      s_coef[i] = v;
    }
    if constexpr ((TB + 64) / 64 != 4) __syncthreads();
  }
  // CNT doubles (even count, 16-byte aligned) from the table, two per LDS instruction
  auto table = [&](int off, auto &out) {
    typedef double hfx_d2 __attribute__((ext_vector_type(2)));
    constexpr int CNT = sizeof(out) / sizeof(double);
#pragma unroll
    for (int i = 0; i < CNT / 2; i++)
    {
      const hfx_d2 v = *(const volatile __attribute__((address_space(3))) hfx_d2 *)(&s_coef[off + 2 * i]);
      out[2 * i] = v.x;
      out[2 * i + 1] = v.y;
    }
  };
  int t = threadIdx.x;
  if constexpr (LW && (TB + 64) / 64 == 4)
  {
    __shared__ unsigned s_hw[4];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); // wave slot [3:0], SIMD [5:4]
    if ((threadIdx.x & 63) == 0) s_hw[threadIdx.x >> 6] = hw;
    __syncthreads();
    unsigned seen = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) seen |= 1u << ((s_hw[w] >> 4) & 3);
    if (a.simd_roles && seen == 0xfu) // (4 waves on 4 SIMDs: otherwise the parts stay with the hardware wave numbers)
    {
      const unsigned odd = s_hw[0] & 1u, simd = (hw >> 4) & 3;
      const unsigned part = odd ? ((0x1320u >> (4 * simd)) & 3u) : ((0x2103u >> (4 * simd)) & 3u);
      t = (int)(part * 64 + (threadIdx.x & 63));
    }
  }
  const int tu = t < NU ? t : NU - 1, tf = t < NFP ? t : NFP - 1;
  const bool is_u = t < NU, is_f = t < NFP;
  const long ne = a.n_eles, plane_u = (long)NU * ne, plane_f = (long)NFP * ne;
  const bool viscous = a.P.viscous;
  // GA: the LDG corrections are formed here from the partner's flux-point solution (a.nbr; viscous runs of the loader-wave form)
  constexpr bool gather = GA;
  static_assert(!GA || LW, "the in-kernel LDG corrections belong to the loader-wave form");
  const bool dma_delta = viscous && !gather;

  // flux-point role: the 1-D extrapolation rows of this point and its pencil
  const int dq = tidx[T::I_FDQ + tf], d_f = dq >> 1;
  const int bf = tidx[T::I_FB + tf], sf = (d_f == 0) ? 1 : (d_f == 1 ? N : N * N);
  double Lrow[N];
#pragma unroll
  for (int m = 0; m < N; m++) Lrow[m] = coef_g[T::C_LF + dq * N + m];
  // pencil role: ROUNDS work items (field, direction, pencil); a wave's items share the direction
  int it_d[ROUNDS], it_o[ROUNDS], it_fa[ROUNDS], it_fb[ROUNDS], it_k[ROUNDS];
  int it_dq[ROUNDS]; // the round's direction, wave-uniform (inactive lanes included)
#pragma unroll
  for (int r = 0; r < ROUNDS; r++)
  {
    const int slot = t + TB * r;
    int d = slot / T::SP;
    const int w = slot - d * T::SP;
    const bool on = d < ND && w < T::ITEMS_D;
    const int k = on ? w / L : 0, line = on ? w - (w / L) * L : 0;
    if (d >= ND) d = ND - 1;
    int base;
    if (ND == 2)
      base = d == 0 ? N * line : line;
    else
      base = d == 0 ? N * line : (d == 1 ? (line % N) + N * N * (line / N) : line);
    it_d[r] = on ? d : -1;
    it_dq[r] = __builtin_amdgcn_readfirstlane(d);
    it_o[r] = (k + NF * d) * NU + base;
    it_k[r] = k;
    it_fa[r] = k * NFP + tidx[T::I_PF + (d * L + line) * 2 + 0];
    it_fb[r] = k * NFP + tidx[T::I_PF + (d * L + line) * 2 + 1];
  }

  // ---- software pipeline: the inputs of element e+1 (state, delta, volume metrics) are fetched into
  // registers during phases C/D of element e -- phases with few live registers -- and land in LDS at the
  // top of the next iteration.  All barriers order LDS traffic only, so these loads and the result
  // stores stay in flight across them.
  constexpr int UNP = (NF * NU + TB - 1) / TB, DNP = (NF * NFP + TB - 1) / TB;
  double pf_u[UNP], pf_d[DNP], JG[ND * ND], detjac_raw = 1.0;
  // global arrays as [wave-uniform element offset + loop-invariant lane offset] (GArr)
  const long tot_u = plane_u * NF, tot_f = plane_f * NF;
  const GArr<BUF> g_u0(a.u0, tot_u), g_delta(a.delta, tot_f), g_JGu(a.JGinv_upts, plane_u * (ND * ND)), g_dju(a.detjac_upts, plane_u);
  const GArr<BUF> g_JGf(a.JGinv_fpts, plane_f * (ND * ND)), g_djf(a.detjac_fpts, plane_f), g_nrm(a.norm_fpts, plane_f * ND);
  const GArr<BUF> g_gu(a.grad_upts, plane_u * NG), g_gf(a.grad_fpts, plane_f * NG), g_fn(a.fn_fpts, tot_f),
      g_div(a.div, tot_u), g_td(a.tdisf_in, plane_u * NG);
  const unsigned lu = tu, lf = tf;
  unsigned lo_u[UNP], lo_d[DNP]; // lane offsets of the state / delta prefetch: (field plane + point)
#pragma unroll
  for (int i = 0; i < UNP; i++)
  {
    const int q = t + TB * i, f = q / NU;
    lo_u[i] = (unsigned)(q - f * NU) + (unsigned)f * (unsigned)plane_u;
  }
#pragma unroll
  for (int i = 0; i < DNP; i++)
  {
    const int q = t + TB * i, f = q / NFP;
    lo_d[i] = (unsigned)(q - f * NFP) + (unsigned)f * (unsigned)plane_f;
  }
  auto fetch_state = [&](long e) {
#pragma unroll
    for (int i = 0; i < UNP; i++)
      if (t + TB * i < NF * NU) pf_u[i] = g_u0.ld((long)NU * e, lo_u[i]);
    if (viscous)
    {
#pragma unroll
      for (int i = 0; i < DNP; i++)
        if (t + TB * i < NF * NFP) pf_d[i] = g_delta.ld((long)NFP * e, lo_d[i]);
    }
  };
  auto fetch_metrics = [&](long e) {
#pragma unroll
    for (int q = 0; q < ND * ND; q++) JG[q] = g_JGu.ld((long)NU * e * (ND * ND), lu * (ND * ND) + q);
    // only the load here: the reciprocal is taken at the top of the next iteration, so that nothing in phase C
    // waits for this (last issued) load and with it for the whole prefetch
    detjac_raw = g_dju.ld((long)NU * e, lu);
  };
  // flux-point metrics of the current element, requested at their use in phase B, behind the solution-point block (requested at
  // the top of phase B or of phase A they cost registers and gained nothing: DESIGN.md 5.1)
  double JF[ND * ND], nrm[ND], djf_raw = 1.0;
  long ef_cur = 0;
  auto fetch_fmetrics = [&]() {
    if constexpr (LW)
    {
      if (viscous && is_f)
      {
#pragma unroll
        for (int q = 0; q < ND * ND; q++) JF[q] = ldsv(&s_met[O_JGF + tf * (ND * ND) + q]);
        djf_raw = ldsv(&s_met[O_DJF + tf]);
#pragma unroll
        for (int m = 0; m < ND; m++) nrm[m] = ldsv(&s_met[O_NRM + m * NFPP + tf]);
      }
    }
    else if (viscous && is_f)
    {
#pragma unroll
      for (int q = 0; q < ND * ND; q++) JF[q] = g_JGf.ld(ef_cur * (ND * ND), lf * (ND * ND) + q);
      djf_raw = g_djf.ld(ef_cur, lf);
#pragma unroll
      for (int m = 0; m < ND; m++) nrm[m] = g_nrm.ld(ef_cur + m * plane_f, lf);
    }
  };
  const EleOrder order(ne, a.xcd_order != 0, a.ele_list, a.n_list);
  if constexpr (LW)
  {
    if (t >= TB)
    {
      // ---- the loader wave: 4 barriers per element like the compute waves; the DMA of element k+1 is issued after
      // barrier 1 of element k (slot (k+1)&1 was last read in phase A of element k-1) and waited for on this wave's
      // own counter before barrier 1 of element k+1
      const int lane = t - TB;
      // state and LDG corrections: one run per field, 16 bytes per lane (the last lane of a state run carries 8 bytes of
      // the next run into the padding of the slot)
      constexpr int L_U = (NU + 1) / 2, L_D = (NFP + 1) / 2; // lanes per field run
      constexpr int UJ = NF * ((L_U + 63) / 64), DJ = NF * ((L_D + 63) / 64);
      auto issue = [&](long e, int which) {
        const lds_dp base = (lds_dp)s_in + which * SLOT;
#pragma unroll
        for (int k = 0; k < NF; k++) dma16_region(g_u0, base + k * NUS, L_U, lane, (unsigned)((long)NU * e + k * plane_u) * 8u);
        if (dma_delta)
        {
#pragma unroll
          for (int k = 0; k < NF; k++)
            dma16_region(g_delta, base + (U_DW / 2 + k * NFP), L_D, lane, (unsigned)((long)NFP * e + k * plane_f) * 8u);
        }
      };
      // metrics: 16 bytes per lane, partial last wave-instruction masked (an inactive lane writes nothing)
      constexpr int L_JGU = (NQ * NU + 1) / 2, L_DJU = (NU + 1) / 2, L_JGF = (NQ * NFP + 1) / 2, L_DJF = (NFP + 1) / 2; // lanes
      constexpr int N_MU = (L_JGU + 63) / 64 + (L_DJU + 63) / 64;                          // volume metrics
      constexpr int N_MF = (L_JGF + 63) / 64 + (L_DJF + 63) / 64 + ND * ((L_DJF + 63) / 64); // flux-point metrics, normals
      constexpr int N_UDV = UJ + DJ, N_UDI = UJ, N_MV = N_MU + N_MF, N_MI = N_MU;
      static_assert(N_UDV <= 63 && N_MV <= 63, "loader wave: more DMA instructions in flight than vmcnt can count");
      auto issue_met = [&](long e) {
        dma16_region(g_JGu, (lds_dp)s_met + O_JGU, L_JGU, lane, (unsigned)((long)NU * NQ * e) * 8u);
        dma16_region(g_dju, (lds_dp)s_met + O_DJU, L_DJU, lane, (unsigned)((long)NU * e) * 8u);
        if (viscous)
        {
          dma16_region(g_JGf, (lds_dp)s_met + O_JGF, L_JGF, lane, (unsigned)((long)NFP * NQ * e) * 8u);
          dma16_region(g_djf, (lds_dp)s_met + O_DJF, L_DJF, lane, (unsigned)((long)NFP * e) * 8u);
#pragma unroll
          for (int m = 0; m < ND; m++)
            dma16_region(g_nrm, (lds_dp)s_met + (O_NRM + m * NFPP), L_DJF, lane, (unsigned)((long)NFP * e + m * plane_f) * 8u);
        }
      };
#define HFX_VMCNT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
      if (order.at(0) >= 0)
      {
        issue(order.at(0), 0);
        issue_met(order.at(0));
      }
      auto lstamp = [&](long kk, int slot) {
        if (a.stamps != nullptr && blockIdx.x == 0 && kk == a.stamp_it && lane == 0) a.stamps[3 * 16 + slot] = clock64();
      };
      for (long kk = 0, e = order.at(0), e_next; e >= 0; kk++, e = e_next)
      {
        e_next = order.at(kk + 1);
        lstamp(kk, 0);
        // outstanding, oldest first: state/delta of this element, metrics of this element
        if (viscous) HFX_VMCNT(N_MV); else HFX_VMCNT(N_MI);
        lstamp(kk, 1);
        lds_barrier(); // 1: the compute waves may read the input slot
        if (gather) lds_barrier(); // 1b: (the flux-point threads have written the LDG corrections)
        lstamp(kk, 2);
        if (e_next >= 0)
        {
          issue(e_next, (int)((kk + 1) & 1));
          if (dma_delta) HFX_VMCNT(N_UDV); else HFX_VMCNT(N_UDI); // the metrics of this element have landed
        }
        else
          HFX_VMCNT(0);
        lstamp(kk, 3);
        lds_barrier(); // 2: the compute waves may read the metric slot
        lstamp(kk, 4);
        if (viscous)
        {
          // paired physics: the compute waves take their metrics into registers at the top of phase B and say so (2b):
          // the slot is refilled most of an iteration ahead of its next use
          lds_barrier(); // 2b
          lstamp(kk, 5);
          if (e_next >= 0) issue_met(e_next);
          lstamp(kk, 6);
          lds_barrier(); // 3
        }
        else
        {
          lds_barrier(); // 3: ... and have finished with it
          lstamp(kk, 5);
          if (e_next >= 0) issue_met(e_next);
          lstamp(kk, 6);
        }
        lds_barrier(); // 4
        if constexpr (LES)
        {
          lds_barrier(); // 5: (the transformed SGS flux is in the work region)
          lds_barrier(); // 6: (its pencil ends are in the correction region of the input slot)
        }
        lstamp(kk, 7);
      }
#undef HFX_VMCNT
      return;
    }
  }
  if (order.at(0) >= 0)
  {
    if (!LW)
    {
      fetch_state(order.at(0));
      fetch_metrics(order.at(0));
    }
  }

  int it_no = 0;
  auto stamp = [&](int slot) {
    if (a.stamps != nullptr && blockIdx.x == 0 && it_no == a.stamp_it && (t & 63) == 0) a.stamps[(t >> 6) * 16 + slot] = clock64();
  };
  // gather: what this thread's flux point needs for its LDG correction -- the partner's flux-point solution, or at a
  // boundary / partition-face point the correction its one-sided kernel left in `delta` -- is requested a whole element
  // ahead into NF registers, the partner words another element ahead.  (Through the loader wave -- 4-byte LDS-DMA with
  // per-lane offsets, or its registers -- the 25 scattered requests of an element took 6 000 - 8 000 cycles to ISSUE and
  // held up the wave's other duties; spread over the 150 flux-point threads they cost 2 700.)
  double pv[NF];
  int nb_cur = -1, nb_next = -1, nb_nn = -1;
  auto request_partner = [&](long e_of, int nb) {
    const double *src = (nb < 0) ? a.delta + ((long)NFP * e_of + tf) : a.disu + (nb >> 2);
#pragma unroll
    for (int k = 0; k < NF; k++) pv[k] = src[k * plane_f];
  };
  if (gather && order.at(0) >= 0)
  {
    nb_cur = a.nbr[(long)NFP * order.at(0) + tf];
    request_partner(order.at(0), nb_cur);
    if (order.at(1) >= 0) nb_next = a.nbr[(long)NFP * order.at(1) + tf];
  }
  for (long kk = 0, e = order.at(0), e_next; e >= 0; kk++, e = e_next, it_no++)
  {
    e_next = order.at(kk + 1);
    const long eu = (long)NU * e, ef = (long)NFP * e;
    stamp(0);
    double inv_detjac = (viscous && !LW) ? 1.0 / detjac_raw : 0.0;
    // the pencil addresses of the flux-point role are loop invariant; left alone the compiler hoists
    // one address register per (plane, m) out of the element loop.  Rebuild the N of them here from an
    // opaque copy and let the plane offsets be immediates.
    int am[N];
    {
      int bfo = bf;
      asm volatile("" : "+v"(bfo));
#pragma unroll
      for (int m = 0; m < N; m++) am[m] = bfo + m * sf;
    }
    if constexpr (LW)
    {
      su = s_in + (kk & 1) * SLOT;
      sd = su + U_DW / 2;
    }
    else
    {
#pragma unroll
      for (int i = 0; i < UNP; i++)
        if (t + TB * i < NF * NU) su[t + TB * i] = pf_u[i];
      if (viscous)
      {
#pragma unroll
        for (int i = 0; i < DNP; i++)
          if (t + TB * i < NF * NFP) sd[t + TB * i] = pf_d[i];
      }
    }
    stamp(1);
    lds_barrier();
    stamp(2);
    ef_cur = ef;
    double u[NF], uf[NF];
    // LES: what outlives phase B -- the transformed SGS flux of this solution point, the projected viscous flux and the face
    // Jacobian of this flux point
    [[maybe_unused]] double tsg[LES ? NG : 1], tdA_f = 1.0, len2 = 0.0;
    double accg[GA ? ROUNDS : 1][N]; // GA: the pencils' D . u, formed in A0 while the partner values are waited for
    if constexpr (GA)
    {
      // first the part of phase A that needs no correction -- the 1-D derivative of the pencils' state: the wait for the
      // partner values below also drains this wave's last result stores, and this work fills it
      double xa[ROUNDS][N], Dm[NN2];
      table(S_D, Dm);
#pragma unroll
      for (int r = 0; r < ROUNDS; r++)
      {
        const int d = it_dq[r];
        const int srr = (d == 0) ? 1 : (d == 1 ? N : N * N);
        const double *su_p = su + (it_o[r] - NF * d * NU) + it_k[r] * (NUS - NU);
#pragma unroll
        for (int m = 0; m < N; m++) xa[r][m] = ldsv(su_p + m * srr);
      }
#pragma unroll
      for (int r = 0; r < ROUNDS; r++)
      {
#pragma unroll
        for (int mp = 0; mp < N; mp++) accg[r][mp] = 0.0;
#pragma unroll
        for (int m = 0; m < N; m++)
#pragma unroll
          for (int mp = 0; mp < N; mp++) accg[r][mp] += Dm[mp * N + m] * xa[r][m];
      }
    }
    if (gather)
    {
      // ---- A0: this flux point's solution (the extrapolation phase B needs anyway) and its LDG correction
      // delta = u_common - u_own, u_common = (u_L + u_R)/2 - beta (u_L - u_R) on the pair's left / right orientation
      // (src/inters.cpp:637) -> the slot's correction region, from where the pencils of phase A take it
#pragma unroll
      for (int k = 0; k < NF; k++) uf[k] = 0.0;
#pragma unroll
      for (int m = 0; m < N; m++)
      {
        double x[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) x[k] = ldsv(&su[k * NUS + am[m]]);
#pragma unroll
        for (int k = 0; k < NF; k++) uf[k] += Lrow[m] * x[k];
      }
      const int nb = nb_cur;
      const double beta = (nb & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        const double ul = (nb & 1) ? pv[k] : uf[k], ur = (nb & 1) ? uf[k] : pv[k];
        const double uc = 0.5 * (ul + ur) - beta * (ul - ur);
        const double dl = (nb < 0) ? pv[k] : uc - uf[k];
        if (is_f) sd[k * NFP + tf] = dl;
      }
      stamp(11);
      // the next element's partner values, and the word of the one after (unconditional, into a register of its own: the
      // rotation nb_cur <- nb_next <- nb_nn happens at the END of the iteration -- placed here the compiler waited for the
      // word, and with it for the five values just requested, right behind the barrier)
      if (e_next >= 0) request_partner(e_next, nb_next);
      const long e_nn = order.at(kk + 2);
      nb_nn = a.nbr[(long)NFP * (e_nn >= 0 ? e_nn : e) + tf];
      stamp(12);
      lds_barrier(); // 1b
      stamp(13);
    }
    // over-integration: the de-aliased inviscid flux of this point is requested here, a phase ahead of its use (the loader-wave
    // form takes its folded contribution to the divergence instead, requested at the top of phase C)
    double td[(OI && !LW) ? NG : 1];
    if (OI && !LW && is_u)
    {
#pragma unroll
      for (int q = 0; q < NG; q++) td[q] = g_td.ld(eu + q * plane_u, lu);
    }

    // ---- A: transformed gradient, pencil-wise; flux-point solution, point-wise
    if (viscous)
    {
      // all rounds' pencils at once, branch-free (run-time stride): the LDS reads of every round are in flight
      // together, then the N^2 FMAs per pencil against the wave-uniform 1-D matrix
      double xa[ROUNDS][N], da[ROUNDS], db[ROUNDS];
      int sr[ROUNDS];
#pragma unroll
      for (int r = 0; r < ROUNDS; r++)
      {
        const int d = it_dq[r];
        sr[r] = (d == 0) ? 1 : (d == 1 ? N : N * N);
        const double *su_p = su + (it_o[r] - NF * d * NU) + it_k[r] * (NUS - NU);
        if constexpr (!GA)
        {
#pragma unroll
          for (int m = 0; m < N; m++) xa[r][m] = ldsv(su_p + m * sr[r]);
        }
        da[r] = ldsv(sd + it_fa[r]);
        db[r] = ldsv(sd + it_fb[r]);
      }
      if constexpr (LW)
      {
        double Dm[GA ? 2 : NN2], c5a[ROUNDS][NP], c5b[ROUNDS][NP];
        if constexpr (!GA) table(S_D, Dm);
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
        {
          table(S_C5 + (it_dq[r] * 2 + 0) * NP, c5a[r]);
          table(S_C5 + (it_dq[r] * 2 + 1) * NP, c5b[r]);
        }
        // the N outputs of a pencil side by side (column outermost: consecutive FMAs are independent; every output still
        // adds its terms in ascending column order), one predicated block of stores per round
        double acc[ROUNDS][N];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
        {
          if constexpr (GA)
          {
#pragma unroll
            for (int mp = 0; mp < N; mp++) acc[r][mp] = accg[r][mp];
          }
          else
          {
#pragma unroll
            for (int mp = 0; mp < N; mp++) acc[r][mp] = 0.0;
#pragma unroll
            for (int m = 0; m < N; m++)
#pragma unroll
              for (int mp = 0; mp < N; mp++) acc[r][mp] += Dm[mp * N + m] * xa[r][m];
          }
#pragma unroll
          for (int mp = 0; mp < N; mp++) acc[r][mp] += c5a[r][mp] * da[r];
#pragma unroll
          for (int mp = 0; mp < N; mp++) acc[r][mp] += c5b[r][mp] * db[r];
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
          if (it_d[r] >= 0)
          {
            double *sg_p = sg + it_o[r];
#pragma unroll
            for (int mp = 0; mp < N; mp++) sg_p[mp * sr[r]] = acc[r][mp];
          }
      }
      else
      {
#pragma unroll
      for (int r = 0; r < ROUNDS; r++)
      {
        const int d = it_dq[r];
        double *sg_p = sg + it_o[r];
#pragma unroll
        for (int mp = 0; mp < N; mp++)
        {
          double acc = 0.0;
#pragma unroll
          for (int m = 0; m < N; m++) acc += coef[T::C_D + mp * N + m] * xa[r][m];
          acc += coef[T::C_5 + (d * 2 + 0) * N + mp] * da[r];
          acc += coef[T::C_5 + (d * 2 + 1) * N + mp] * db[r];
          if (it_d[r] >= 0) sg_p[mp * sr[r]] = acc;
        }
      }
      }
    }
    if (is_u || LW) // (LW: every lane, on clamped point numbers -- the paired physics of phase B)
    {
#pragma unroll
      for (int k = 0; k < NF; k++) u[k] = ldsv(&su[k * NUS + tu]);
    }
    if (viscous && (is_f || LW) && !gather)
    {
      // pencil position outermost: NF independent accumulators per batch of LDS reads (one wait per batch
      // instead of one per field); each output still sums over ascending m
#pragma unroll
      for (int k = 0; k < NF; k++) uf[k] = 0.0;
#pragma unroll
      for (int m = 0; m < N; m++)
      {
        double x[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) x[k] = ldsv(&su[k * NUS + am[m]]);
#pragma unroll
        for (int k = 0; k < NF; k++) uf[k] += Lrow[m] * x[k];
      }
    }
    stamp(3);
    lds_barrier(); // sg complete; su / sd are dead: their region becomes st
    stamp(4);



    __builtin_amdgcn_sched_barrier(0);
    // ---- B: gradient and projected viscous flux at the flux points; fluxes at the solution points
    // (a wave that owns no solution point -- the third compute wave of a P4 hex: flux points 128..149 -- would run the paired
    // form with a dummy solution-point chain: twice the FP64 instructions it needs, on a SIMD it shares with the other
    // workgroup's heavy wave.  It takes the flux-point block of the unpaired form below instead.)
    const bool wave_u = __builtin_amdgcn_readfirstlane((int)((t & ~63) < NU || !a.light_short)) != 0;
    if (LW && viscous && wave_u)
    {
      // PAIRED form (loader-wave kernel: registers to spare).  A thread's solution point and its flux point go through
      // metric transform and viscous flux together, statement by statement -- two independent dependency chains, so the
      // wave issues back to back where one chain alone waits for its previous result (one heavy wave per SIMD: nobody
      // else would fill the gaps).  Lanes beyond the last point of a kind repeat that point's arithmetic on clamped
      // numbers and do not store.  The inviscid and the viscous flux are summed BEFORE the one transform to reference space
      // (the reference transforms them separately, src/eles.cpp:1439-1470 and :2360-2387: a re-association).
      double jg2[2][NQ], inv2[2], u2[2][NF], g2[2][NG], f2[2][NG];
#pragma unroll
      for (int q = 0; q < NQ; q++)
      {
        jg2[0][q] = ldsv(&s_met[O_JGU + tu * NQ + q]);
        jg2[1][q] = ldsv(&s_met[O_JGF + tf * NQ + q]);
      }
      double nr2[ND];
#pragma unroll
      for (int l = 0; l < ND; l++) nr2[l] = ldsv(&s_met[O_NRM + l * NFPP + tf]);
      inv2[0] = ldsv(&s_met[O_DJU + tu]);
      inv2[1] = ldsv(&s_met[O_DJF + tf]);
      if constexpr (LES)
      {
        // requested here, used behind the paired physics
        tdA_f = a.tdA_fpts[ef + tf];
        len2 = a.les_len2[eu + tu];
      }
      lds_barrier(); // 2b: the metric slot is free, the loader requests the next element's metrics
      inv2[0] = 1.0 / inv2[0];
      inv2[1] = 1.0 / inv2[1];
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        u2[0][k] = u[k];
        u2[1][k] = uf[k];
      }
      // transformed gradients: the solution point's own, the flux point's extrapolated along its pencil
#pragma unroll
      for (int q = 0; q < NG; q++) g2[0][q] = ldsv(&sg[q * NU + tu]);
#pragma unroll
      for (int q = 0; q < NG; q++) g2[1][q] = 0.0;
#pragma unroll
      for (int m = 0; m < N; m++)
      {
        double x[NG];
#pragma unroll
        for (int q = 0; q < NG; q++) x[q] = ldsv(&sg[q * NU + am[m]]);
#pragma unroll
        for (int q = 0; q < NG; q++) g2[1][q] += Lrow[m] * x[q];
      }
      // to physical space (to_physical, both points per statement)
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double cg[2][ND], tmp[2];
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
          for (int d = 0; d < ND; d++) cg[p][d] = 0.0;
#pragma unroll
        for (int l = 0; l < ND; l++)
        {
#pragma unroll
          for (int p = 0; p < 2; p++) tmp[p] = inv2[p] * g2[p][k + NF * l];
#pragma unroll
          for (int d = 0; d < ND; d++)
#pragma unroll
            for (int p = 0; p < 2; p++) cg[p][d] += tmp[p] * jg2[p][l + ND * d];
        }
#pragma unroll
        for (int d = 0; d < ND; d++)
#pragma unroll
          for (int p = 0; p < 2; p++) g2[p][k + NF * d] = cg[p][d];
      }
      if (a.grad_fpts && is_f && (a.meta == nullptr || (a.meta[ef + tf] & 4)))
#pragma unroll
        for (int q = 0; q < NG; q++) g_gf.st(ef + q * plane_f, lf, g2[1][q]);
      calc_visf_pair<ND>(a.P, u2, g2, f2);
      // flux point: this side's viscous flux on its own normal
      if constexpr (LES)
      {
        // (parked in the correction region of this element's input slot, dead since phase A: the extrapolated SGS flux joins it
        // behind phase C)
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < ND; l++) s += f2[1][k + NF * l] * nr2[l];
          if (is_f) sd[k * NFP + tf] = s;
        }
      }
      else if (is_f)
      {
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < ND; l++) s += f2[1][k + NF * l] * nr2[l];
          g_fn.st(ef + k * plane_f, lf, s);
        }
      }
      // solution point: total flux to reference space
      if (is_u)
      {
        double ft[NG];
        if (OI)
        {
          // over-integration: the de-aliased inviscid flux arrives transformed; only the viscous part is transformed here
#pragma unroll
          for (int q = 0; q < NG; q++) ft[q] = f2[0][q];
        }
        else
        {
          calc_invf<ND, true>(a.P.gamma, u, ft);
#pragma unroll
          for (int q = 0; q < NG; q++) ft[q] += f2[0][q];
        }
        if constexpr (LES)
        {
          // the closure on this point's corrected gradient (physical space); its flux, transformed, is kept for the pencil
          // pass behind phase C and joins the total flux here.  (The physical flux waits in this thread's own st column
          // meanwhile: the closure needs the registers.)
#pragma unroll
          for (int q = 0; q < NG; q++) st[q * NU + tu] = ft[q];
          asm volatile("" ::: "memory");
          double sgq[NG];
          calc_sgsf_fast<ND>(a.P, a.les, u, g2[0], len2, eu + tu, plane_u, sgq);
#pragma unroll
          for (int k = 0; k < NF; k++)
#pragma unroll
            for (int l = 0; l < ND; l++)
            {
              double ts = 0.0;
#pragma unroll
              for (int m = 0; m < ND; m++) ts += jg2[0][l + ND * m] * sgq[k + NF * m];
              tsg[k + NF * l] = ts;
            }
          // (its first direction waits in the state region of this element's input slot, dead since phase A -- five registers
          // fewer across phase C)
#pragma unroll
          for (int k = 0; k < NF; k++) su[k * NUS + tu] = tsg[k];
#pragma unroll
          for (int q = 0; q < NG; q++) ft[q] = ldsv(&st[q * NU + tu]);
        }
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < ND; m++) s += jg2[0][l + ND * m] * ft[k + NF * m];
            if constexpr (LES) s += tsg[k + NF * l];
            st[(k + NF * l) * NU + tu] = s;
          }
      }
    }
    else if (LW && viscous)
    {
      // ---- a wave without solution points: the flux-point physics alone, metrics from the slot into registers first (2b)
      double JFl[NQ], nl[ND], grf[NG], fq[NG];
#pragma unroll
      for (int q = 0; q < NQ; q++) JFl[q] = ldsv(&s_met[O_JGF + tf * NQ + q]);
#pragma unroll
      for (int l = 0; l < ND; l++) nl[l] = ldsv(&s_met[O_NRM + l * NFPP + tf]);
      double inv_df = ldsv(&s_met[O_DJF + tf]);
      if constexpr (LES) tdA_f = a.tdA_fpts[ef + tf];
      lds_barrier(); // 2b
      inv_df = 1.0 / inv_df;
#pragma unroll
      for (int q = 0; q < NG; q++) grf[q] = 0.0;
#pragma unroll
      for (int m = 0; m < N; m++)
      {
        double x[NG];
#pragma unroll
        for (int q = 0; q < NG; q++) x[q] = ldsv(&sg[q * NU + am[m]]);
#pragma unroll
        for (int q = 0; q < NG; q++) grf[q] += Lrow[m] * x[q];
      }
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double tg[ND], cg[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) tg[d] = grf[k + NF * d];
        to_physical<ND>(inv_df, JFl, tg, cg);
#pragma unroll
        for (int d = 0; d < ND; d++) grf[k + NF * d] = cg[d];
      }
      if (a.grad_fpts && is_f && (a.meta == nullptr || (a.meta[ef + tf] & 4)))
#pragma unroll
        for (int q = 0; q < NG; q++) g_gf.st(ef + q * plane_f, lf, grf[q]);
      calc_visf<ND, true>(a.P, uf, grf, fq);
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < ND; l++) s += fq[k + NF * l] * nl[l];
        if constexpr (LES)
        {
          if (is_f) sd[k * NFP + tf] = s; // (parked: the extrapolated SGS flux joins it behind phase C)
        }
        else if (is_f)
          g_fn.st(ef + k * plane_f, lf, s);
      }
    }
    else
    {
    if (is_u)
    {
      if constexpr (LW)
      {
        // volume metrics of this point from the slot the loader wave filled
#pragma unroll
        for (int q = 0; q < NQ; q++) JG[q] = ldsv(&s_met[O_JGU + tu * NQ + q]);
        if (viscous) inv_detjac = 1.0 / ldsv(&s_met[O_DJU + tu]);
      }
      // the transformed flux is accumulated in this thread's own st column (LDS) instead of 15 registers
      if (OI)
      {
        // over-integration: the de-aliased inviscid flux was evaluated at the cubature points and projected back (loader-wave
        // form: its contribution joins the divergence in phase D)
#pragma unroll
        for (int q = 0; q < NG; q++) st[q * NU + tu] = LW ? 0.0 : td[q];
      }
      else
      {
        double f[NG];
        calc_invf<ND, true>(a.P.gamma, u, f);
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
            st[(k + NF * l) * NU + tu] = s;
          }
      }
      if (viscous)
      {
        double gr[NG], f[NG];
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double tg[ND], cg[ND];
#pragma unroll
          for (int d = 0; d < ND; d++) tg[d] = ldsv(&sg[(k + NF * d) * NU + tu]);
          to_physical<ND>(inv_detjac, JG, tg, cg);
#pragma unroll
          for (int d = 0; d < ND; d++) gr[k + NF * d] = cg[d];
        }
        if (a.grad_upts)
#pragma unroll
          for (int q = 0; q < NG; q++) g_gu.st(eu + q * plane_u, lu, gr[q]);
        calc_visf<ND, true>(a.P, u, gr, f);
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = ldsv(&st[(k + NF * l) * NU + tu]);
#pragma unroll
            for (int m = 0; m < ND; m++) s += JG[l + ND * m] * f[k + NF * m];
            st[(k + NF * l) * NU + tu] = s;
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0); // keep the two independent physics blocks apart: interleaving them doubles the live registers
    if (viscous && is_f)
    {
      double grf[NG], fq[NG];
      // flux-point metrics are fetched here, after the solution-point block has released its registers
      fetch_fmetrics();
      const double inv_df = 1.0 / djf_raw;
#pragma unroll
      for (int q = 0; q < NG; q++) grf[q] = 0.0;
#pragma unroll
      for (int m = 0; m < N; m++)
      {
        double x[NG];
#pragma unroll
        for (int q = 0; q < NG; q++) x[q] = ldsv(&sg[q * NU + am[m]]);
#pragma unroll
        for (int q = 0; q < NG; q++) grf[q] += Lrow[m] * x[q];
      }
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double tg[ND], cg[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) tg[d] = grf[k + NF * d];
        to_physical<ND>(inv_df, JF, tg, cg);
#pragma unroll
        for (int d = 0; d < ND; d++) grf[k + NF * d] = cg[d];
      }
      if (a.grad_fpts && (a.meta == nullptr || (a.meta[ef + tf] & 4)))
#pragma unroll
        for (int q = 0; q < NG; q++) g_gf.st(ef + q * plane_f, lf, grf[q]);
      calc_visf<ND, true>(a.P, uf, grf, fq);
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < ND; l++) s += fq[k + NF * l] * nrm[l];
        g_fn.st(ef + k * plane_f, lf, s);
      }
    }
    } // (unpaired form)
    stamp(5);
    lds_barrier(); // st complete; sg is dead: its region takes the divergence parts
    stamp(6);

    __builtin_amdgcn_sched_barrier(0);
    // ---- C: next element's inputs on their way; divergence parts pencil-wise, normal flux at the flux points
    // (over-integration, loader-wave form: the de-aliased flux's folded contribution sum_l Dc[l] tdisf_l of this solution point
    // is requested here and joins the sum of phase D)
    double oi_res[(OI && LW) ? NF : 1];
    if constexpr (OI && LW)
    {
      if (is_u)
      {
#pragma unroll
        for (int k = 0; k < NF; k++) oi_res[k] = g_td.ld(eu + k * plane_u, lu);
      }
    }
    if (e_next >= 0)
    {
      if (!LW)
      {
        fetch_state(e_next);
        fetch_metrics(e_next);
      }
    }
    {
      double xa[ROUNDS][N];
      int sr[ROUNDS];
#pragma unroll
      for (int r = 0; r < ROUNDS; r++)
      {
        const int d = it_dq[r];
        sr[r] = (d == 0) ? 1 : (d == 1 ? N : N * N);
#pragma unroll
        for (int m = 0; m < N; m++) xa[r][m] = ldsv(st + it_o[r] + m * sr[r]);
      }
      if (a.stamps != nullptr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(10); }
      if constexpr (LW)
      {
        // the folded matrix of the round's direction: div_tdisf - opp_3 norm_tdisf of the pencil in N^2 FMAs
        double acc[ROUNDS][N], Dc[ROUNDS][NN2];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) table(S_DC + it_dq[r] * NN2, Dc[r]); // all rounds' tables in one batch of reads
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
        {
#pragma unroll
          for (int mp = 0; mp < N; mp++) acc[r][mp] = 0.0;
#pragma unroll
          for (int m = 0; m < N; m++)
#pragma unroll
            for (int mp = 0; mp < N; mp++) acc[r][mp] += Dc[r][mp * N + m] * xa[r][m];
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
          if (it_d[r] >= 0)
          {
#pragma unroll
            for (int mp = 0; mp < N; mp++) sp[it_o[r] + mp * sr[r]] = acc[r][mp];
          }
      }
      else
      {
#pragma unroll
      for (int r = 0; r < ROUNDS; r++)
      {
        // folded correction: the normal transformed flux at the two ends of this pencil (norm_tdisf_fpts of those flux
        // points, which therefore never goes to HBM) and its share -opp_3 . norm_tdisf of the corrected divergence
        const int d = it_dq[r];
        double na = 0.0, nb = 0.0;
#pragma unroll
        for (int m = 0; m < N; m++)
        {
          na += coef[T::C_LF + (d * 2 + 0) * N + m] * xa[r][m];
          nb += coef[T::C_LF + (d * 2 + 1) * N + m] * xa[r][m];
        }
        na *= -coef[T::C_L1 + (d * 2 + 0) * N];
        nb *= -coef[T::C_L1 + (d * 2 + 1) * N];
#pragma unroll
        for (int mp = 0; mp < N; mp++)
        {
          double acc = 0.0;
#pragma unroll
          for (int m = 0; m < N; m++) acc += coef[T::C_D + mp * N + m] * xa[r][m];
          acc += coef[T::C_3 + (d * 2 + 0) * N + mp] * na;
          acc += coef[T::C_3 + (d * 2 + 1) * N + mp] * nb;
          if (it_d[r] >= 0) sp[it_o[r] + mp * sr[r]] = acc;
        }
      }
      }
    }
    stamp(7);
    lds_barrier();
    stamp(8);
    if (is_u)
    {
      // all NF * ND parts first (one wait), then the sums in the old order
      double part[ND][NF];
#pragma unroll
      for (int d = 0; d < ND; d++)
#pragma unroll
        for (int k = 0; k < NF; k++) part[d][k] = ldsv(&sp[(k + NF * d) * NU + tu]);
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double s = part[0][k];
        s += part[1][k];
        if (ND == 3) s += part[ND - 1][k];
        if constexpr (OI && LW) s += oi_res[k];
        g_div.st(eu + k * plane_u, lu, s);
      }
    }
    if constexpr (LES)
    {
      // ---- E: F_sgs . n at the flux points.  The transformed SGS flux goes where the total flux was (dead since barrier 4) ...
      if (is_u)
      {
#pragma unroll
        for (int q = 0; q < NG; q++) st[q * NU + tu] = (q < NF) ? ldsv(&su[q * NUS + tu]) : tsg[q];
      }
      lds_barrier(); // 5
      // ... every pencil item extrapolates its component to the pencil's two ends: +- L . F~_sgs,d = (F~_sgs . n~) there,
      // left where the divergence parts were (read by phase D, before barrier 5) under the ends' flux-point numbers ...
      {
        double xs[ROUNDS][N];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
        {
          const int d = it_dq[r];
          const int srr = (d == 0) ? 1 : (d == 1 ? N : N * N);
#pragma unroll
          for (int m = 0; m < N; m++) xs[r][m] = ldsv(st + it_o[r] + m * srr);
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
        {
          const int d = it_dq[r];
          double na = 0.0, nb = 0.0;
#pragma unroll
          for (int m = 0; m < N; m++)
          {
            na += coef[T::C_LF + (d * 2 + 0) * N + m] * xs[r][m];
            nb += coef[T::C_LF + (d * 2 + 1) * N + m] * xs[r][m];
          }
          na *= coef[T::C_L1 + (d * 2 + 0) * N];
          nb *= coef[T::C_L1 + (d * 2 + 1) * N];
          if (it_d[r] >= 0)
          {
            sp[it_fa[r]] = na;
            sp[it_fb[r]] = nb;
          }
        }
      }
      lds_barrier(); // 6
      // ... and joins this side's projected viscous flux: Fn = (F_v + F_sgs) . n
      if (is_f)
      {
        const double itd = 1.0 / tdA_f;
#pragma unroll
        for (int k = 0; k < NF; k++) g_fn.st(ef + k * plane_f, lf, ldsv(&sd[k * NFP + tf]) + ldsv(&sp[k * NFP + tf]) * itd);
      }
    }
    stamp(9);
    if (gather)
    {
      asm volatile("" : "+v"(nb_nn)); // (the word is needed here, not earlier)
      nb_cur = nb_next;
      nb_next = nb_nn;
    }
    // no barrier: the next iteration's writes to sA (dead since the last barrier) do not touch sB, and
    // its writes to sB come after its first barrier
  }
}


struct Split2FaceArgs
{
  long npairs;
  const int *L, *R;
  const unsigned char *meta;
  long plane_f;
  const double *disu, *fn, *fnorm, *tdA;
  double *tconf;
  Phys P;
};

template <int ND, int RS>
__global__ __launch_bounds__(256) void face_flux2_kernel(const Split2FaceArgs a)
{
  constexpr int NF = ND + 2;
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
  // (every load before the first store: tconf may overlap the inputs as far as the compiler knows)
  double fl[NF], fr[NF];
  const unsigned char mt = a.meta[il]; // with the other loads, not behind the Riemann solver
  if (a.P.viscous)
  {
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      fl[k] = a.fn[il + k * a.plane_f];
      fr[k] = a.fn[ir + k * a.plane_f];
    }
  }
  riemann_flux_t<ND, RS, true>(a.P, ul, ur, n, fn);
  if (a.P.viscous)
  {
    const double beta = (mt & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      // (1/2+b) F_L.n + (1/2-b) F_R.n - tau (u_R - u_L), n the left normal = -(right normal)
      double fv = (0.5 + beta) * fl[k] - (0.5 - beta) * fr[k];
      fv -= a.P.ldg_tau * (ur[k] - ul[k]);
      a.tconf[il + k * a.plane_f] = fn[k] * tl + fv * tl;
      a.tconf[ir + k * a.plane_f] = -fn[k] * tr + -fv * tr;
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

// div_tdisf + opp_3 (norm_tconf - norm_tdisf) -> RK update -> disu_fpts of the new state: a streaming kernel
// (after the sum-factorised flux kernel: (div_tdisf - opp_3 norm_tdisf) + opp_3 norm_tconf, norm_tdisf is not read)
#ifndef HFX_UPD_WAVES
#define HFX_UPD_WAVES 3
#endif
template <int ND, int N, bool BUF>
__global__ __launch_bounds__((SGeo<ND, N>::TB), HFX_UPD_WAVES) void split_update_kernel(const Split2Args a)
{
  using G = Geo<ND, N>;
  constexpr int NF = G::NF, NU = G::NU, NFP = G::NFP;
  constexpr int N3 = 2 * ND;
  __shared__ double su[NF][NU];
  __shared__ double sc[NF][NFP];
  const int t = threadIdx.x;
  const int tu = t < NU ? t : NU - 1, tf = t < NFP ? t : NFP - 1;
  const bool is_u = t < NU, is_f = t < NFP;
  const long ne = a.n_eles, plane_u = (long)NU * ne, plane_f = (long)NFP * ne;
  // this thread's rows of opp_3 (solution-point role) and opp_0 (flux-point role): the exact non-zeros in
  // ascending column order, values and columns in registers (no dictionary, no table look-ups)
  double c3[N3], c0[N];
  int i3[N3], i0[N];
#pragma unroll
  for (int q = 0; q < N3; q++)
  {
    c3[q] = q < a.o3w ? a.o3v[tu + (long)NU * q] : 0.0;
    i3[q] = q < a.o3w ? a.o3i[tu + (long)NU * q] : 0;
  }
#pragma unroll
  for (int q = 0; q < N; q++)
  {
    c0[q] = q < a.o0w ? a.o0v[tf + (long)NFP * q] : 0.0;
    i0[q] = q < a.o0w ? a.o0i[tf + (long)NFP * q] : 0;
  }

  const unsigned lu = tu, lf = tf;
  const long tot_u = plane_u * NF, tot_f = plane_f * NF;
  const GArr<BUF> g_u0(a.u0, tot_u), g_u1(a.u1, tot_u), g_div(a.div, tot_u), g_src(a.src, tot_u), g_dj(a.detjac_upts, plane_u);
  const GArr<BUF> g_tc(a.tconf, tot_f), g_nt(a.ntd_fpts, tot_f), g_dn(a.disu_next, tot_f);
  const EleOrder order(ne, a.xcd_order != 0, a.ele_list, a.n_list);
  // the RK formula of this launch (uniform): 0: u -= dt/div (dd - s), 1: u = ca u + cb u1 + dt/div rhs, 2: low storage
  int rk_form = 0;
  double rk_div = 1.0, rk_ca = 0.0, rk_cb = 0.0;
  bool rk_keep_u = false; // the stage's initial solution goes to disu_upts(1) first
  if (a.adv_type == 1)
  {
    rk_keep_u = a.in_step == 0;
    if (a.in_step < 3)
      rk_div = 3.0;
    else
    {
      rk_form = 1; rk_ca = 3.0 / 4.0; rk_cb = 1.0 / 4.0; rk_div = 4.0;
    }
  }
  else if (a.adv_type == 2)
  {
    rk_keep_u = a.in_step == 0;
    if (a.in_step < 2 || a.in_step == 3)
      rk_div = 2.0;
    else
    {
      rk_form = 1; rk_ca = 1.0 / 3.0; rk_cb = 2.0 / 3.0; rk_div = 6.0;
    }
  }
  else if (a.adv_type != 0)
    rk_form = 2;
  for (long kk = 0, e = order.at(0); e >= 0; kk++, e = order.at(kk))
  {
    const long eu = (long)NU * e, ef = (long)NFP * e;
    double u[NF], dvin[NF], u1v[NF], tcv[NF], sv[NF];
    // EVERY load of the element is requested before the first one is used (clamped lane offsets: no predicate).  With the
    // LDS writes between them each field's load was waited for on its own -- five memory latencies at the top of an element
#pragma unroll
    for (int k = 0; k < NF; k++) tcv[k] = g_tc.ld(ef + k * plane_f, lf);
    const double dt = a.dt_local_on ? a.dt_local[e] : a.dt;
    const double dj = g_dj.ld(eu, lu);
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      const long ok = eu + k * plane_u;
      u[k] = g_u0.ld(ok, lu);
      dvin[k] = g_div.ld(ok, lu);
    }
    if (a.need_u1)
    {
#pragma unroll
      for (int k = 0; k < NF; k++) u1v[k] = g_u1.ld(eu + k * plane_u, lu);
    }
    else
    {
#pragma unroll
      for (int k = 0; k < NF; k++) u1v[k] = 0.0;
    }
    if (a.src)
    {
#pragma unroll
      for (int k = 0; k < NF; k++) sv[k] = g_src.ld(eu + k * plane_u, lu);
    }
    else
    {
#pragma unroll
      for (int k = 0; k < NF; k++) sv[k] = 0.0;
    }
    if (!a.folded)
    {
      // the dictionary-row flux kernel leaves norm_tdisf in HBM: norm_tconf - norm_tdisf (the daxpy of src/eles.cpp:1746)
#pragma unroll
      for (int k = 0; k < NF; k++) tcv[k] += -1.0 * g_nt.ld(ef + k * plane_f, lf);
    }
    if (is_f)
    {
#pragma unroll
      for (int k = 0; k < NF; k++) sc[k][tf] = tcv[k];
    }
    lds_barrier();
    if (is_u)
    {
      // div_tconf = div_tdisf + opp_3 (norm_tconf - norm_tdisf): column outermost, NF accumulators per batch of reads
      double dva[NF];
#pragma unroll
      for (int k = 0; k < NF; k++) dva[k] = dvin[k];
#pragma unroll
      for (int q = 0; q < N3; q++)
      {
        double x[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) x[k] = ldsv(&sc[k][0] + i3[q]);
#pragma unroll
        for (int k = 0; k < NF; k++) dva[k] += c3[q] * x[k];
      }
      // AdvanceSolution (src/eles.cpp:1080-1180).  Which formula applies is the same for every point of the launch: the
      // choice is made ONCE per element, outside the loop over the fields, and the step's divisor is applied to dt once
      // (dt / 3.0 etc. as the reference writes it).  With the choice inside the loop the compiler evaluated every
      // formula's divisions for every field and waited for the previous field's stores before the next one's arithmetic.
      const double cdt = (rk_div == 1.0) ? dt : dt / rk_div;
      double unew[NF], r1v[NF];
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        const double dv = dva[k];
        if (dv != dv) atomicMin(a.nan_flag, (unsigned long long)(eu + k * plane_u + tu));
        const double dd = dv / dj;
        double un = u[k];
        if (rk_form == 0)
          un -= cdt * (dd - sv[k]);
        else if (rk_form == 1)
        {
          const double rhs = -dd + sv[k];
          un = rk_ca * un + rk_cb * u1v[k] + cdt * rhs;
        }
        else
        {
          const double rhs = -dd + sv[k];
          const double r1 = a.rk_a * u1v[k] + dt * rhs;
          r1v[k] = r1;
          un += a.rk_b * r1;
        }
        unew[k] = un;
      }
      // the stores of the element together, nothing between them
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        const long ok = eu + k * plane_u;
        if (a.write_div) g_div.st(ok, lu, dva[k]);
        if (rk_keep_u) g_u1.st(ok, lu, u[k]);
        if (rk_form == 2) g_u1.st(ok, lu, r1v[k]);
        g_u0.st(ok, lu, unew[k]);
        su[k][tu] = unew[k];
      }
    }
    lds_barrier();
    if (is_f)
    {
      double un[NF];
#pragma unroll
      for (int k = 0; k < NF; k++) un[k] = 0.0;
#pragma unroll
      for (int q = 0; q < N; q++)
      {
        double x[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) x[k] = ldsv(&su[k][0] + i0[q]);
#pragma unroll
        for (int k = 0; k < NF; k++) un[k] += c0[q] * x[k];
      }
#pragma unroll
      for (int k = 0; k < NF; k++) g_dn.st(ef + k * plane_f, lf, un[k]);
    }
    lds_barrier();
  }
}

} // namespace hfx
