// general.hip -- the fused RK stage for GENERAL (non-tensor-product) element classes: tetrahedra, triangular prisms.
//
// Reference side: the same 17 calls of CalcResidual + AdvanceSolution as everywhere (/root/reference/src/solver.cpp:50-223,
// src/HiFiLES.cpp:201-217); on these classes the seven operators (src/eles.cpp:3074-3596, built by src/eles_tets.cpp and
// src/eles_pris.cpp) are DENSE -- P3: 40 x 20 / 20 x 20 / 20 x 40 doubles on tetrahedra, 68 x 40 / 40 x 40 / 40 x 68 on
// prisms -- so the per-method path is a stream of 15+ dense GEMM and point kernels per stage (820 B per DOF-update).
// Here the stage is cut at the two places where data must cross elements, exactly like the split stage of the
// tensor-product classes (fused_hex.hip, fused = 3), and is FOUR launches per element block and stage:
//
//   gface_delta_kernel   thread per interior flux-point pair: LDG common solution -> delta_disu_fpts (both sides, which
//                        may belong to different element blocks: prism | tetrahedron faces)
//   general_flux_kernel  workgroup per batch of 16 elements: corrected gradient at the solution points
//                        (opp_4, opp_5), its extrapolation to the flux points (opp_6), both point physics blocks, the
//                        discontinuous divergence (opp_2) and normal flux (opp_1); leaves div_tdisf, norm_tdisf and
//                        each side's viscous flux projected on its own normal (Fn, 5 instead of 15 doubles per point)
//   gface_flux_multi_kernel  thread per pair, all face blocks in one launch: Riemann + LDG common flux from u and Fn of both sides -> norm_tconf
//   general_update_kernel  batch of 16 elements: opp_3 (norm_tconf - norm_tdisf), RK update, opp_0 of the new state
//
// Contractions run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).  A batch is 16 elements because 16 is the tile
// edge: the MFMA's "row" index is the ELEMENT of the batch, its "column" index an operator row, so one instruction
// applies 16 operator rows to 16 elements for 4 operator columns.  An accumulator register then holds 16 consecutive
// operator rows of one element: results go to HBM as 128-byte runs straight from the registers, and the flux-point
// physics is evaluated on the accumulators without a trip through LDS.  Operators are zero-padded on the host to
// (16 m') x (4 k') so that no tile needs a bounds check on its loads; they are the same ~100-300 kB for every workgroup
// and stay cache resident.  Element data sits in LDS as [operator column k][16 elements] planes (one per field /
// field x dimension) with the element index XOR-swizzled by the row: operand reads, accumulator writes and the
// point-wise passes are all bank-conflict free.
#include <vector>
#include "general.hpp"

#include <type_traits>

#include <algorithm>
#include <cstring>

#include "physics.hpp"

namespace hfx
{

typedef double g_f64x4 __attribute__((ext_vector_type(4)));

constexpr int GB = 16; // elements per batch = MFMA tile edge

struct GeneralData
{
  int KU = 0, KF = 0, MU = 0, MF = 0; // n_upts / n_fpts padded to 4 (contraction length) and to 16 (output rows)
  // zero-padded operators, column-major with leading dimension M*: o4/o5/o2 (MU rows), o6/o1/o0 (MF rows), o3 (MU rows)
  double *o0 = nullptr, *o1[3] = {}, *o2[3] = {}, *o3 = nullptr, *o4[3] = {}, *o5[3] = {}, *o6 = nullptr;
  unsigned char *meta = nullptr; // (n_fpts, n_eles): bit1 beta sign flipped (LEFT point of a pair), bit2 boundary point
  double *disu_alt = nullptr;    // second disu_fpts buffer (the update kernel writes the new state's flux-point solution)
  double *fn_fpts = nullptr;     // (n_fpts, n_eles, n_fields) viscous flux projected on the point's own normal
  int *nbr = nullptr;            // (n_fpts, n_eles) partner word of every flux point whose pair lies inside this block (GenArgs::nbr)
  double *o2f[3] = {};           // opp_2[d] - opp_3 opp_1[d], padded like o2: the folded correction (GenArgs::fold)
  hfx_eles *blocks[4] = {};      // the element blocks the partner words refer to (bits 3:2 of a word), captured at build
  int n_blocks = 0;
  bool any_bdy = false;
  bool built = false;
  long long *stamps = nullptr;
  double *les_len2 = nullptr; // (n_upts, n_eles) squared length scale of an LES closure (les_len2_upload)
};

void general_invalidate(hfx_eles *e)
{
  if (e && e->general) ((GeneralData *)e->general)->built = false;
}

void general_destroy(hfx_eles *e)
{
  if (!e || !e->general) return;
  GeneralData *g = (GeneralData *)e->general;
  void *p[] = {g->o0, g->o1[0], g->o1[1], g->o1[2], g->o2[0], g->o2[1], g->o2[2], g->o3, g->o4[0], g->o4[1], g->o4[2],
               g->o5[0], g->o5[1], g->o5[2], g->o6, g->meta, g->disu_alt, g->fn_fpts, g->stamps, g->nbr, g->o2f[0], g->o2f[1], g->o2f[2], g->les_len2};
  for (void *q : p)
    if (q) (void)hipFree(q);
  delete g;
  e->general = nullptr;
}

// LDS plane [row][16 elements], the element index swizzled by the row
__device__ __forceinline__ int sw(int row, int col) { return row * GB + (col ^ (row & 15)); }

// g_phys(d) = sum_l (inv_detjac * g_ref(l)) * JGinv(l,d)   (BLAS=NO branch of src/eles.cpp:1975-1979)
__device__ __forceinline__ void g_to_physical(const double inv_detjac, const double (&JG)[9], const double (&tg)[3], double (&cg)[3])
{
#pragma unroll
  for (int d = 0; d < 3; d++) cg[d] = 0.0;
#pragma unroll
  for (int l = 0; l < 3; l++)
  {
    const double temp = inv_detjac * tg[l];
#pragma unroll
    for (int d = 0; d < 3; d++) cg[d] += temp * JG[l + 3 * d];
  }
}

struct GenArgs
{
  int n_eles, nu, nfp, KU, KF, MU, MF;
  unsigned inv_nu, inv_nfp; // floor(2^32 / n) + 1: q / n == __umulhi(q, inv) for the q < 2^16 of the staging loops
  const double *o0, *o1[3], *o2[3], *o3, *o4[3], *o5[3], *o6;
  const double *u0, *delta, *disu;
  // fold != 0: o2[d] holds opp_2[d] - opp_3 opp_1[d], so that P4 leaves div_tdisf - opp_3 norm_tdisf in `div` and norm_tdisf is
  // neither formed nor stored; the update kernel then adds opp_3 norm_tconf alone (as split3's folded correction)
  int fold;
  // the LDG correction of a flux point whose partner lies in the SAME element block is formed in the flux kernel from the
  // partner's flux-point solution: (partner offset << 4) | partner's block << 2 | beta sign flipped << 1 | this point is the right
  // side; -1: a boundary point (its correction is in `delta`).  NULL: `delta` holds all of them.
  const int *nbr;
  const double *disu_b[4]; // flux-point solution of the blocks a partner word may name (its bits 3:2), and their plane strides
  long plane_b[4];
  const double *detjac_upts, *JGinv_upts, *detjac_fpts, *JGinv_fpts, *norm_fpts;
  const unsigned char *meta;
  double *div, *ntd, *fn, *grad_fpts; // grad_fpts: boundary points only (NULL: no boundary faces / inviscid)
  Phys P;
  // update kernel
  double *u0w, *u1;
  const double *tconf, *div_in, *src, *dt_local;
  double *disu_next;
  unsigned long long *nan_flag;
  int adv_type, in_step, dt_local_on, write_div, need_u1;
  double dt, rk_a, rk_b;
  long long *stamps; // diagnostics (option flux_stamps): cycle counter of every wave of ONE workgroup at the phase boundaries
  // LES closure evaluated in the flux kernel (LESG form): parameters (with the Leonard terms of the similarity models), the wall
  // distance of the Smagorinsky damping, and tdA at the flux points (F_sgs . n = (F~_sgs . n~) / tdA)
  LesParams les;
  const double *les_len2, *tdA_fpts;
  // over-integration: the de-aliased transformed inviscid flux (eles::evaluate_invFlux_over_int, formed by the dense contractions
  // before this launch), taken in P3 instead of the collocated one; NULL: none
  const double *tdisf_in;
};

// NA output tiles at once, sharing the operator fragments: acc[j] += op[rt*16 + (0..15)][0 .. 4 ksteps) . plane_j[k][16 elements],
// plane_j = plane + j * pstride.  op: zero-padded operator, leading dimension M; lane: li = lane & 15, lk = lane >> 4.
// The operator fragments of UK k-steps are requested together (they come from L2 / L1: one latency per UK * NA MFMAs).
template <int NA>
__device__ __forceinline__ void tile_mac(g_f64x4 (&acc)[NA], const double *__restrict__ op, int M, int rt, const double *plane, int pstride,
                                         int ksteps, int li, int lk)
{
  constexpr int UK = (NA > 5) ? 2 : 8; // k-steps whose operator fragments are requested together (15 tiles: their LDS operands fill the registers)
  // sw(4 s + lk, li) = 64 s + 16 lk + ((li ^ lk) ^ 4 (s & 3)): four per-lane offsets, the rest is an immediate
  const int lx = li ^ lk;
  const int ob[4] = {16 * lk + lx, 16 * lk + (lx ^ 4), 16 * lk + (lx ^ 8), 16 * lk + (lx ^ 12)};
  const double *ap = op + rt * 16 + li + (long)M * lk;
  const long ast = (long)M * 4;
  double av[UK], an[UK];
#pragma unroll
  for (int u = 0; u < UK; u++) av[u] = (u < ksteps) ? ap[ast * u] : 0.0;
  for (int s0 = 0; s0 < ksteps; s0 += UK)
  {
    // the next chunk's fragments are in flight while this chunk's MFMAs issue
    const double *apn = ap + ast * (s0 + UK);
#pragma unroll
    for (int u = 0; u < UK; u++) an[u] = (s0 + UK + u < ksteps) ? apn[ast * u] : 0.0;
    const double *pb = plane + 64 * s0;
#pragma unroll
    for (int u = 0; u < UK; u++)
      if (s0 + u < ksteps)
      {
#pragma unroll
        for (int j = 0; j < NA; j++)
          acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[j * pstride + 64 * u + ob[(UK % 4 == 0) ? (u & 3) : ((s0 + u) & 3)]], av[u], acc[j], 0, 0, 0);
      }
#pragma unroll
    for (int u = 0; u < UK; u++) av[u] = an[u];
  }
}

// n items of NF = 5 field tiles each over W waves: whole rounds item by item; the r = n % W items of the last round are
// cut into field groups so that W / r waves share one item (item(it, integral_constant<int, fields>, first field))
template <int W, class F>
__device__ __forceinline__ void split_rounds(int n, int wave, F &&item)
{
  const int whole = (n / W) * W, r = n - whole;
  for (int it = wave; it < whole; it += W) item(it, std::integral_constant<int, 5>{}, 0);
  if (r == 0) return;
  const int share = W / r; // waves per remaining item (>= 1)
  const int it = whole + wave / share, part = wave % share;
  if (it >= n) return; // (W not a multiple of r: the last few waves have nothing)
  if (share == 1)
    item(it, std::integral_constant<int, 5>{}, 0);
  else if (share == 2)
  {
    if (part == 0) item(it, std::integral_constant<int, 3>{}, 0);
    else item(it, std::integral_constant<int, 2>{}, 3);
  }
  else if (share == 3)
  {
    if (part == 0) item(it, std::integral_constant<int, 2>{}, 0);
    else if (part == 1) item(it, std::integral_constant<int, 2>{}, 2);
    else item(it, std::integral_constant<int, 1>{}, 4);
  }
  else if (share == 4)
  {
    if (part == 0) item(it, std::integral_constant<int, 2>{}, 0);
    else item(it, std::integral_constant<int, 1>{}, part + 1);
  }
  else if (part < 5)
    item(it, std::integral_constant<int, 1>{}, part);
}

// ---------------------------------------------------------------------------------------------------------------
// flux kernel: u, delta -> div_tdisf, norm_tdisf, Fn   (steps 3-4, 8 (both halves), 10-12 of CalcResidual's sequence)
// ---------------------------------------------------------------------------------------------------------------
// NUc, NFPc: the element class's point counts as compile-time constants (0: read from the arguments); the common sizes
// are instantiated so that trip counts, LDS offsets and the index divisions fold
// LESG: the eddy-viscosity / similarity closure of an LES run (eles::calc_sgsf_upts, src/eles.cpp:2395-2650) evaluated in P3 on the
// physical gradient the point physics holds.  The reference adds F_sgs to the total flux at the solution points
// (src/eles.cpp:2360-2392) and, at the flux points, extrapolates the TRANSFORMED SGS flux (opp_0), takes it back to physical space
// and adds it to each side's viscous flux in the common-flux sweep (src/eles.cpp:2817-2893, src/int_inters.cpp:299-313).  What the
// face kernel needs of that is F_sgs . n = (F~_sgs . n~) / tdA = (sum_d opp_1[d] F~_sgs,d) / tdA -- the contraction P4 runs for
// norm_tdisf in the form without the folded correction.  So P3 first leaves the transformed SGS flux ALONE in G (the rest of the
// point's transformed flux waits in registers), one more contraction adds its normal projection to the Fn this workgroup stored
// in P2, and then the rest joins G for P4.
template <int W, int NUc, int NFPc, bool LESG = false>
__global__ __launch_bounds__(64 * W, 2) void general_flux_kernel(const GenArgs a)
{
  static_assert(!LESG || NUc != 0, "the LES form is instantiated for the element classes with compile-time sizes");
  constexpr int NF = 5, ND = 3, T = 64 * W;
  extern __shared__ double lds[];
  const int nu = NUc ? NUc : a.nu, nfp = NFPc ? NFPc : a.nfp;
  const int KU = (nu + 3) & ~3, KF = (nfp + 3) & ~3, MU = (nu + 15) & ~15, MF = (nfp + 15) & ~15;
  const unsigned inv_nu = NUc ? (unsigned)(4294967296ull / (unsigned)(NUc ? NUc : 1)) + 1u : a.inv_nu;
  const unsigned inv_nfp = NFPc ? (unsigned)(4294967296ull / (unsigned)(NFPc ? NFPc : 1)) + 1u : a.inv_nfp;
  double *U = lds;                 // [NF][KU][16]      the state; later: nothing
  double *D = U + NF * KU * GB;    // [NF][KF][16]      delta_disu_fpts
  double *G = D + NF * KF * GB;    // [NF*ND][KU][16]   reference-space gradient, then the transformed total flux
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const long e0 = (long)blockIdx.x * GB;
  const int nval = (int)min((long)GB, a.n_eles - e0);
  const long plane_u = (long)nu * a.n_eles, plane_f = (long)nfp * a.n_eles;
  const bool visc = a.P.viscous != 0;
  auto stamp = [&](int slot) {
    if (a.stamps != nullptr && blockIdx.x == gridDim.x / 2 && lane == 0) a.stamps[wave * 16 + slot] = clock64();
  };
  stamp(0);

  // ---- P0: stage the state (and delta) of the batch: contiguous nu x 16 doubles per field in HBM -> [k][element].
  // ALL loads of a thread are requested before the first one is used (one memory latency for the batch instead of one per
  // trip of the staging loop: the stamps showed 21 000 - 48 000 of a batch's 100 000 cycles here)
  {
    constexpr int MAXQ_U = NUc ? (((NUc + 3) & ~3) * GB + T - 1) / T : 0, MAXQ_D = NFPc ? (((NFPc + 3) & ~3) * GB + T - 1) / T : 0;
    if constexpr (NUc != 0 && NFPc != 0 && (MAXQ_U + MAXQ_D) * NF <= 30) // (more would not fit the registers)
    {
      double ru[NF][MAXQ_U], rd[NF][MAXQ_D];
#pragma unroll
      for (int f = 0; f < NF; f++)
      {
        const double *src = a.u0 + e0 * nu + f * plane_u;
#pragma unroll
        for (int i = 0; i < MAXQ_U; i++)
        {
          const int q = tid + T * i;
          const int el = (int)__umulhi((unsigned)q, inv_nu);
          ru[f][i] = (q < nu * GB && el < nval) ? src[q] : 0.0;
        }
        if (visc && a.nbr == nullptr)
        {
          const double *sd = a.delta + e0 * nfp + f * plane_f;
#pragma unroll
          for (int i = 0; i < MAXQ_D; i++)
          {
            const int q = tid + T * i;
            const int el = (int)__umulhi((unsigned)q, inv_nfp);
            rd[f][i] = (q < nfp * GB && el < nval) ? sd[q] : 0.0;
          }
        }
      }
      if (visc && a.nbr != nullptr)
      {
        // LDG corrections formed here (the pairwise kernel is not launched for pairs inside this block): the point's partner
        // word, then its own and its partner's flux-point solution -- or the correction itself where the word is -1 --
        // delta = u_common - u_own, u_common = (u_L + u_R)/2 - beta (u_L - u_R) on the pair's orientation (src/inters.cpp:637)
        int w[MAXQ_D];
        bool okq[MAXQ_D];
#pragma unroll
        for (int i = 0; i < MAXQ_D; i++)
        {
          const int q = tid + T * i;
          const int el = (int)__umulhi((unsigned)q, inv_nfp);
          okq[i] = q < nfp * GB && el < nval;
          w[i] = okq[i] ? a.nbr[e0 * nfp + q] : -1;
        }
        double ro[NF][MAXQ_D];
#pragma unroll
        for (int i = 0; i < MAXQ_D; i++)
        {
          const long own = e0 * nfp + (okq[i] ? tid + T * i : 0);
          const int blk = (w[i] >> 2) & 3;
          const double *po = a.disu + own, *pp = (w[i] < 0) ? a.delta + own : a.disu_b[blk] + (w[i] >> 4);
          const long pstr = (w[i] < 0) ? plane_f : a.plane_b[blk];
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            ro[f][i] = po[f * plane_f];
            rd[f][i] = pp[f * pstr];
          }
        }
#pragma unroll
        for (int i = 0; i < MAXQ_D; i++)
        {
          const double beta = (w[i] & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            const double ul = (w[i] & 1) ? rd[f][i] : ro[f][i], ur = (w[i] & 1) ? ro[f][i] : rd[f][i];
            const double uc = 0.5 * (ul + ur) - beta * (ul - ur);
            rd[f][i] = !okq[i] ? 0.0 : ((w[i] < 0) ? rd[f][i] : uc - ro[f][i]);
          }
        }
      }
#pragma unroll
      for (int f = 0; f < NF; f++)
      {
#pragma unroll
        for (int i = 0; i < MAXQ_U; i++)
        {
          const int q = tid + T * i;
          if (q < KU * GB)
          {
            const int el = (int)__umulhi((unsigned)q, inv_nu), k = q - el * nu;
            const int p = q - nu * GB; // >= 0: one of the (KU - nu) * 16 padding entries
            U[f * KU * GB + ((q < nu * GB) ? sw(k, el) : sw(nu + p / GB, p % GB))] = ru[f][i];
          }
        }
        if (visc)
        {
#pragma unroll
          for (int i = 0; i < MAXQ_D; i++)
          {
            const int q = tid + T * i;
            if (q < KF * GB)
            {
              const int el = (int)__umulhi((unsigned)q, inv_nfp), k = q - el * nfp;
              const int p = q - nfp * GB;
              D[f * KF * GB + ((q < nfp * GB) ? sw(k, el) : sw(nfp + p / GB, p % GB))] = rd[f][i];
            }
          }
        }
      }
    }
    else
    {
      for (int f = 0; f < NF; f++)
      {
        const double *src = a.u0 + e0 * nu + f * plane_u;
        for (int q = tid; q < KU * GB; q += T)
        {
          const int el = (int)__umulhi((unsigned)q, inv_nu), k = q - el * nu; // q < nu*16: a real entry; the rest of the plane: K padding
          if (q < nu * GB)
            U[f * KU * GB + sw(k, el)] = (el < nval) ? src[q] : 0.0;
          else
          {
            const int p = q - nu * GB; // (KU - nu) * 16 padding entries
            U[f * KU * GB + sw(nu + p / GB, p % GB)] = 0.0;
          }
        }
        if (visc)
        {
          const double *sd = a.delta + e0 * nfp + f * plane_f;
          for (int q = tid; q < KF * GB; q += T)
          {
            const int el = (int)__umulhi((unsigned)q, inv_nfp), k = q - el * nfp;
            if (q < nfp * GB)
              D[f * KF * GB + sw(k, el)] = (el < nval) ? sd[q] : 0.0;
            else
            {
              const int p = q - nfp * GB;
              D[f * KF * GB + sw(nfp + p / GB, p % GB)] = 0.0;
            }
          }
        }
      }
    }
  }
  // K padding rows of G (read as zero by the opp_6 / opp_1 / opp_2 tiles)
  const int pad_rows = (KU - nu) * GB, pad_div = pad_rows > 0 ? pad_rows : 1; // (no padding rows where n_upts is a multiple of 4)
  for (int q = tid; q < NF * ND * pad_rows; q += T)
  {
    const int c = q / pad_div, p = q - c * pad_rows;
    G[c * KU * GB + sw(nu + p / GB, p % GB)] = 0.0;
  }
  stamp(1);
  __syncthreads();
  stamp(2);

  // ---- P3: point physics at the solution points (evaluate_invFlux, the transform of correct_gradient, evaluate_viscFlux;
  //          src/eles.cpp:1415,1973,2285): G(f,d) <- transformed total flux.  Points first, first + stride, ...
  auto p3_points = [&](int first, int stride) {
  // the point's metrics are requested one trip ahead (a trip is one memory latency otherwise: 7 000 cycles measured)
  double JGn[9], djn = 1.0;
  auto request_metrics = [&](int q) {
    const int el = (int)__umulhi((unsigned)q, inv_nu), pt = q - el * nu;
    const bool ok = q < nu * GB && el < nval;
    const long o = ok ? pt + (long)nu * (e0 + el) : 0;
#pragma unroll
    for (int c = 0; c < 9; c++) JGn[c] = ok ? a.JGinv_upts[o * 9 + c] : 0.0;
    djn = (ok && visc) ? a.detjac_upts[o] : 1.0;
  };
  request_metrics(first);
  for (int q = first; q < nu * GB; q += stride)
  {
    const int el = (int)__umulhi((unsigned)q, inv_nu), pt = q - el * nu;
    double JG[9];
#pragma unroll
    for (int c = 0; c < 9; c++) JG[c] = JGn[c];
    const double dj = djn;
    request_metrics(q + stride);
    if (el < nval)
    {
      const int so = sw(pt, el);
      double u[NF], F[NF * ND];
#pragma unroll
      for (int k = 0; k < NF; k++) u[k] = U[k * KU * GB + so];
      const bool oi = a.tdisf_in != nullptr;
      double td[NF * ND];
      if (oi)
      {
        const long o = pt + (long)nu * (e0 + el);
#pragma unroll
        for (int c = 0; c < NF * ND; c++)
        {
          td[c] = a.tdisf_in[o + c * plane_u];
          F[c] = 0.0;
        }
      }
      else
      {
#pragma unroll
        for (int c = 0; c < NF * ND; c++) td[c] = 0.0;
        calc_invf<ND, true>(a.P.gamma, u, F);
      }
      if (visc)
      {
        const double idj = 1.0 / dj;
        double g[NF * ND], fv[NF * ND];
#pragma unroll
        for (int k = 0; k < NF; k++)
        {
          double tg[ND], cg[ND];
#pragma unroll
          for (int d = 0; d < ND; d++) tg[d] = G[(k + NF * d) * KU * GB + so];
          g_to_physical(idj, JG, tg, cg);
#pragma unroll
          for (int d = 0; d < ND; d++) g[k + NF * d] = cg[d];
        }
        calc_visf<ND, true>(a.P, u, g, fv);
#pragma unroll
        for (int c = 0; c < NF * ND; c++) F[c] += fv[c];
      }
#pragma unroll
      for (int k = 0; k < NF; k++)
#pragma unroll
        for (int l = 0; l < ND; l++)
        {
          double t = td[k + NF * l];
#pragma unroll
          for (int m = 0; m < ND; m++) t += JG[l + ND * m] * F[k + NF * m];
          G[(k + NF * l) * KU * GB + so] = t;
        }
    }
    else
    {
      const int pt2 = q - el * nu, so = sw(pt2, el);
#pragma unroll
      for (int c = 0; c < NF * ND; c++) G[c * KU * GB + so] = 0.0;
    }
  }
  };

  if (visc)
  {
    // ---- P1: reference-space corrected gradient at the solution points (calculate_gradient + first half of
    //          correct_gradient, src/eles.cpp:1823,1900): G(f,d) = opp_4[d] U(f) + opp_5[d] D(f)
    const int n_rt = MU / 16;
    // one item = (row tile, dimension): NF output tiles that share the operator fragments.  Whole rounds of W items; the
    // items of the last, partial round are split by FIELD among the waves (W / r waves per item), so that no wave sits
    // idle while another does a whole second item
    auto p1_item = [&](int it, auto nf_c, int f0) {
      constexpr int NFS = decltype(nf_c)::value;
      const int d = it % ND, rt = it / ND;
      g_f64x4 acc[NFS];
#pragma unroll
      for (int f = 0; f < NFS; f++) acc[f] = g_f64x4{0.0, 0.0, 0.0, 0.0};
      tile_mac<NFS>(acc, a.o4[d], MU, rt, U + f0 * KU * GB, KU * GB, KU / 4, li, lk);
      tile_mac<NFS>(acc, a.o5[d], MU, rt, D + f0 * KF * GB, KF * GB, KF / 4, li, lk);
      const int row = rt * 16 + li;
      if (row < nu)
#pragma unroll
        for (int f = 0; f < NFS; f++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++) G[(f0 + f + NF * d) * KU * GB + sw(row, lk + 4 * rg)] = acc[f][rg];
    };
    split_rounds<W>(n_rt * ND, wave, p1_item);
    stamp(3);
    __syncthreads();
    stamp(4);

    // ---- P2: gradient at the flux points (second half of correct_gradient: opp_6, then the transform with the flux
    //          points' own metrics, src/eles.cpp:1930,1998) and the viscous flux there, projected on the point's normal
    const int n_ft = MF / 16;
    for (int rt = wave; rt < n_ft; rt += W)
    {
      g_f64x4 acc[NF * ND];
#pragma unroll
      for (int c = 0; c < NF * ND; c++) acc[c] = g_f64x4{0.0, 0.0, 0.0, 0.0};
      // the point inputs of (row, element lk + 4 rg) are requested one point ahead: rg = 0 before the MFMA loop
      const int row = rt * 16 + li;
      double pu[2][NF], pJ[2][9], pd[2], pn[2][ND];
      unsigned char pm[2] = {0, 0}; // the point's flag byte (bit2: boundary point), with the other inputs
      auto request = [&](int slot, int rg) {
        const int el = lk + 4 * rg;
        const bool ok = row < nfp && el < nval;
        // a lane without a point reads point 0 of the block (its values are not used): UNCONDITIONAL loads -- behind a
        // predicate every pass waited for the loads it had just issued for the next one (s_waitcnt vmcnt(0) at the join)
        const long o = ok ? row + (long)nfp * (e0 + el) : 0;
#pragma unroll
        for (int k = 0; k < NF; k++) pu[slot][k] = a.disu[o + k * plane_f];
#pragma unroll
        for (int q = 0; q < 9; q++) pJ[slot][q] = a.JGinv_fpts[o * 9 + q];
        pd[slot] = a.detjac_fpts[o];
#pragma unroll
        for (int m = 0; m < ND; m++) pn[slot][m] = a.norm_fpts[o + m * plane_f];
        if (a.grad_fpts != nullptr) pm[slot] = a.meta[o];
      };
      request(0, 0);
      tile_mac<NF * ND>(acc, a.o6, MF, rt, G, KU * GB, KU / 4, li, lk);
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
      {
        const int el = lk + 4 * rg, sl = rg & 1;
        if (rg < 3) request(sl ^ 1, rg + 1);
        if (row < nfp && el < nval)
        {
          const long o = row + (long)nfp * (e0 + el);
          double g[NF * ND], fv[NF * ND];
          const double idj = 1.0 / pd[sl];
#pragma unroll
          for (int k = 0; k < NF; k++)
          {
            double tg[ND], cg[ND];
#pragma unroll
            for (int d = 0; d < ND; d++) tg[d] = acc[k + NF * d][rg];
            g_to_physical(idj, pJ[sl], tg, cg);
#pragma unroll
            for (int d = 0; d < ND; d++) g[k + NF * d] = cg[d];
          }
          if (a.grad_fpts != nullptr && (pm[sl] & 4)) // boundary point: the boundary kernel reads the gradient
#pragma unroll
            for (int c = 0; c < NF * ND; c++) a.grad_fpts[o + c * plane_f] = g[c];
          calc_visf<ND, true>(a.P, pu[sl], g, fv);
#pragma unroll
          for (int k = 0; k < NF; k++)
          {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < ND; m++) s += fv[k + NF * m] * pn[sl][m];
            a.fn[o + k * plane_f] = s;
          }
        }
      }
    }
    stamp(5);
    __syncthreads();
    stamp(6);
  }

  if constexpr (LESG)
  {
    constexpr int TRIPS = (NUc * GB + T - 1) / T;
    double Tiv[TRIPS][NF * ND]; // transformed inviscid + viscous flux of this thread's points
#pragma unroll
    for (int tr = 0; tr < TRIPS; tr++)
    {
      const int q = tid + T * tr;
      const int el = (int)__umulhi((unsigned)q, inv_nu), pt = q - el * nu;
      const bool in = q < nu * GB, ok = in && el < nval;
      const long o = ok ? pt + (long)nu * (e0 + el) : 0;
      double JG[9];
#pragma unroll
      for (int c = 0; c < 9; c++) JG[c] = a.JGinv_upts[o * 9 + c];
      const double dj = a.detjac_upts[o], len2 = a.les_len2[o];
      const int so = sw(in ? pt : 0, in ? el : 0);
      double u[NF], F[NF * ND], g[NF * ND], fv[NF * ND], sg[NF * ND];
#pragma unroll
      for (int k = 0; k < NF; k++) u[k] = ok ? U[k * KU * GB + so] : 1.0;
      calc_invf<ND, true>(a.P.gamma, u, F);
      const double idj = 1.0 / dj;
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        double tg[ND], cg[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) tg[d] = G[(k + NF * d) * KU * GB + so];
        g_to_physical(idj, JG, tg, cg);
#pragma unroll
        for (int d = 0; d < ND; d++) g[k + NF * d] = ok ? cg[d] : 0.0;
      }
      calc_visf<ND, true>(a.P, u, g, fv);
      calc_sgsf_fast<ND>(a.P, a.les, u, g, len2, o, plane_u, sg);
#pragma unroll
      for (int k = 0; k < NF; k++)
#pragma unroll
        for (int l = 0; l < ND; l++)
        {
          double t = 0.0, ts = 0.0;
#pragma unroll
          for (int m = 0; m < ND; m++)
          {
            t += JG[l + ND * m] * (F[k + NF * m] + fv[k + NF * m]);
            ts += JG[l + ND * m] * sg[k + NF * m];
          }
          Tiv[tr][k + NF * l] = ok ? t : 0.0;
          if (in) G[(k + NF * l) * KU * GB + so] = ok ? ts : 0.0;
        }
    }
    __syncthreads();
    // F~_sgs . n~ at the flux points, joined to the Fn of P2: rows of opp_1[d] against G(., d), as P4's norm_tdisf tiles
    {
      const int n_ft = MF / 16;
      auto sgs_item = [&](int rt, auto nf_c, int f0) {
        constexpr int NFS = decltype(nf_c)::value;
        g_f64x4 acc[NFS];
#pragma unroll
        for (int f = 0; f < NFS; f++) acc[f] = g_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int d = 0; d < ND; d++) tile_mac<NFS>(acc, a.o1[d], MF, rt, G + (f0 + NF * d) * KU * GB, KU * GB, KU / 4, li, lk);
        const int row = rt * 16 + li;
        if (row < nfp)
#pragma unroll
          for (int rg = 0; rg < 4; rg++)
          {
            const int el = lk + 4 * rg;
            if (el < nval)
            {
              const long o = row + (long)nfp * (e0 + el);
              const double itd = 1.0 / a.tdA_fpts[o];
#pragma unroll
              for (int f = 0; f < NFS; f++)
              {
                // (the value another wave of this workgroup stored in P2: read past the first-level cache)
                double *p = a.fn + o + (f0 + f) * plane_f;
                const double old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *p = old + acc[f][rg] * itd;
              }
            }
          }
      };
      split_rounds<W>(n_ft, wave, sgs_item);
    }
    __syncthreads();
#pragma unroll
    for (int tr = 0; tr < TRIPS; tr++)
    {
      const int q = tid + T * tr;
      const int el = (int)__umulhi((unsigned)q, inv_nu), pt = q - el * nu;
      if (q < nu * GB)
      {
        const int so = sw(pt, el);
#pragma unroll
        for (int c = 0; c < NF * ND; c++) G[c * KU * GB + so] += Tiv[tr][c];
      }
    }
  }
  else
    p3_points(tid, T);
  stamp(7);
  __syncthreads();
  stamp(8);

  // ---- P4: discontinuous divergence (opp_2) and normal flux at the flux points (opp_1), summed over the dimensions
  //          (calculate_divergence, extrapolate_totalFlux; src/eles.cpp:1651,1549) -> HBM, 128-byte runs
  {
    const int n_ut = MU / 16, n_ft = MF / 16;
    auto p4_item = [&](int t, auto nf_c, int f0) {
      constexpr int NFS = decltype(nf_c)::value;
      const bool is_div = t < n_ut;
      const int rt = is_div ? t : t - n_ut;
      g_f64x4 acc[NFS];
#pragma unroll
      for (int f = 0; f < NFS; f++) acc[f] = g_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int d = 0; d < ND; d++)
        tile_mac<NFS>(acc, is_div ? a.o2[d] : a.o1[d], is_div ? MU : MF, rt, G + (f0 + NF * d) * KU * GB, KU * GB, KU / 4, li, lk);
      const int row = rt * 16 + li, n = is_div ? nu : nfp;
      double *out = is_div ? a.div : a.ntd;
      const long plane = is_div ? plane_u : plane_f;
      if (row < n)
#pragma unroll
        for (int f = 0; f < NFS; f++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++)
          {
            const int el = lk + 4 * rg;
            if (el < nval) out[row + (long)n * (e0 + el) + (f0 + f) * plane] = acc[f][rg];
          }
    };
    split_rounds<W>(a.fold ? n_ut : n_ut + n_ft, wave, p4_item); // (folded: the divergence tiles only)
  }
  stamp(9);
}

// ---------------------------------------------------------------------------------------------------------------
// update kernel: div_tdisf + opp_3 (norm_tconf - norm_tdisf) -> RK update -> disu_fpts of the new state
// (calculate_corrected_divergence, AdvanceSolution, extrapolate_solution; src/eles.cpp:1738,1080,1360)
// ---------------------------------------------------------------------------------------------------------------
template <int W, int NUc, int NFPc>
__global__ __launch_bounds__(64 * W) void general_update_kernel(const GenArgs a)
{
  constexpr int NF = 5, T = 64 * W;
  extern __shared__ double lds[];
  const int nu = NUc ? NUc : a.nu, nfp = NFPc ? NFPc : a.nfp;
  const int KU = (nu + 3) & ~3, KF = (nfp + 3) & ~3, MU = (nu + 15) & ~15, MF = (nfp + 15) & ~15;
  const unsigned inv_nu = NUc ? (unsigned)(4294967296ull / (unsigned)(NUc ? NUc : 1)) + 1u : a.inv_nu;
  const unsigned inv_nfp = NFPc ? (unsigned)(4294967296ull / (unsigned)(NFPc ? NFPc : 1)) + 1u : a.inv_nfp;
  double *X = lds;               // [NF][KF][16]  norm_tconf - norm_tdisf
  double *S = X + NF * KF * GB;  // [NF][KU][16]  the correction, then the new state
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const long e0 = (long)blockIdx.x * GB;
  const int nval = (int)min((long)GB, a.n_eles - e0);
  const long plane_u = (long)nu * a.n_eles, plane_f = (long)nfp * a.n_eles;

  // every load of a thread is requested before the first one is used (one memory latency per staging step, as in the
  // flux kernel's P0)
  constexpr int MAXQ_D = NFPc ? (((NFPc + 3) & ~3) * GB + T - 1) / T : 0, MAXQ_U = NUc ? (NUc * GB + T - 1) / T : 0;
  constexpr bool BATCH = NUc != 0 && NFPc != 0 && MAXQ_D * NF <= 20 && MAXQ_U * NF <= 15;
  if constexpr (BATCH)
  {
    double rt_[NF][MAXQ_D], rn_[NF][MAXQ_D];
#pragma unroll
    for (int f = 0; f < NF; f++)
    {
      const double *tc = a.tconf + e0 * nfp + f * plane_f, *nt = a.ntd + e0 * nfp + f * plane_f;
#pragma unroll
      for (int i = 0; i < MAXQ_D; i++)
      {
        const int q = tid + T * i;
        const int el = (int)__umulhi((unsigned)q, inv_nfp);
        const bool ok = q < nfp * GB && el < nval;
        rt_[f][i] = ok ? tc[q] : 0.0;
        rn_[f][i] = (ok && !a.fold) ? nt[q] : 0.0;
      }
    }
#pragma unroll
    for (int f = 0; f < NF; f++)
    {
#pragma unroll
      for (int i = 0; i < MAXQ_D; i++)
      {
        const int q = tid + T * i;
        if (q < KF * GB)
        {
          const int el = (int)__umulhi((unsigned)q, inv_nfp), k = q - el * nfp;
          const int p = q - nfp * GB;
          X[f * KF * GB + ((q < nfp * GB) ? sw(k, el) : sw(nfp + p / GB, p % GB))] = rt_[f][i] + -1.0 * rn_[f][i]; // the daxpy of src/eles.cpp:1746
        }
      }
      for (int q = tid; q < (KU - nu) * GB; q += T) S[f * KU * GB + sw(nu + q / GB, q % GB)] = 0.0;
    }
  }
  else
  {
  for (int f = 0; f < NF; f++)
  {
    const double *tc = a.tconf + e0 * nfp + f * plane_f, *nt = a.ntd + e0 * nfp + f * plane_f;
    for (int q = tid; q < KF * GB; q += T)
    {
      const int el = (int)__umulhi((unsigned)q, inv_nfp), k = q - el * nfp;
      if (q < nfp * GB)
        X[f * KF * GB + sw(k, el)] = (el < nval) ? (a.fold ? tc[q] : tc[q] + -1.0 * nt[q]) : 0.0; // the daxpy of src/eles.cpp:1746
      else
      {
        const int p = q - nfp * GB;
        X[f * KF * GB + sw(nfp + p / GB, p % GB)] = 0.0;
      }
    }
    for (int q = tid; q < (KU - nu) * GB; q += T) S[f * KU * GB + sw(nu + q / GB, q % GB)] = 0.0;
  }
  }
  __syncthreads();
  {
    const int n_rt = MU / 16;
    for (int rt = wave; rt < n_rt; rt += W)
    {
      g_f64x4 acc[NF];
#pragma unroll
      for (int f = 0; f < NF; f++) acc[f] = g_f64x4{0.0, 0.0, 0.0, 0.0};
      tile_mac<NF>(acc, a.o3, MU, rt, X, KF * GB, KF / 4, li, lk);
      const int row = rt * 16 + li;
      if (row < nu)
#pragma unroll
        for (int f = 0; f < NF; f++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++) S[f * KU * GB + sw(row, lk + 4 * rg)] = acc[f][rg];
    }
  }
  // the update's inputs, requested before the barrier that ends the opp_3 tiles (BATCH: all of them at once)
  double pdv[BATCH ? NF : 1][BATCH ? MAXQ_U : 1], pu0[BATCH ? NF : 1][BATCH ? MAXQ_U : 1], pu1[BATCH ? NF : 1][BATCH ? MAXQ_U : 1],
      pdj[BATCH ? MAXQ_U : 1];
  if constexpr (BATCH)
  {
#pragma unroll
    for (int i = 0; i < MAXQ_U; i++)
    {
      const int q = tid + T * i;
      const int el = (int)__umulhi((unsigned)q, inv_nu);
      const bool okq = q < nu * GB && el < nval;
      const long p = e0 * nu + q;
      pdj[i] = okq ? a.detjac_upts[p] : 1.0;
#pragma unroll
      for (int f = 0; f < NF; f++)
      {
        const long ok = p + f * plane_u;
        pdv[f][i] = okq ? a.div_in[ok] : 0.0;
        pu0[f][i] = okq ? a.u0w[ok] : 0.0;
        pu1[f][i] = (okq && a.need_u1) ? a.u1[ok] : 0.0;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int f = 0; f < NF; f++)
  {
#pragma unroll
    for (int i = 0; i < (BATCH ? MAXQ_U : 1); i++)
    for (int q = BATCH ? tid + T * i : tid; q < nu * GB; q += BATCH ? (1 << 30) : T)
    {
      const int el = (int)__umulhi((unsigned)q, inv_nu), pt = q - el * nu;
      const int so = f * KU * GB + sw(pt, el);
      if (el >= nval)
      {
        S[so] = 0.0;
        continue;
      }
      const long p = e0 * nu + q, ok = p + f * plane_u;
      const double dv = (BATCH ? pdv[f][i] : a.div_in[ok]) + S[so];
      if (dv != dv) atomicMin(a.nan_flag, (unsigned long long)ok);
      if (a.write_div) a.div[ok] = dv;
      const double s = a.src ? a.src[ok] : 0.0;
      const double dt = a.dt_local_on ? a.dt_local[e0 + el] : a.dt;
      const double dd = dv / (BATCH ? pdj[i] : a.detjac_upts[p]);
      double un = BATCH ? pu0[f][i] : a.u0w[ok];
      const double u1v = BATCH ? pu1[f][i] : (a.need_u1 ? a.u1[ok] : 0.0);
      if (a.adv_type == 0)
        un -= dt * (dd - s);
      else if (a.adv_type == 1)
      {
        if (a.in_step == 0) a.u1[ok] = un;
        if (a.in_step < 3)
          un -= dt / 3.0 * (dd - s);
        else
        {
          const double rhs = -dd + s;
          un = 3.0 / 4.0 * un + 1.0 / 4.0 * u1v + dt / 4.0 * rhs;
        }
      }
      else if (a.adv_type == 2)
      {
        if (a.in_step == 0) a.u1[ok] = un;
        if (a.in_step < 2 || a.in_step == 3)
          un -= dt / 2.0 * (dd - s);
        else if (a.in_step == 2)
        {
          const double rhs = -dd + s;
          un = 1.0 / 3.0 * un + 2.0 / 3.0 * u1v + dt / 6.0 * rhs;
        }
      }
      else
      {
        const double rhs = -dd + s;
        const double r1 = a.rk_a * u1v + dt * rhs;
        a.u1[ok] = r1;
        un += a.rk_b * r1;
      }
      a.u0w[ok] = un;
      S[so] = un;
    }
  }
  __syncthreads();
  {
    const int n_ft = MF / 16;
    for (int rt = wave; rt < n_ft; rt += W)
    {
      g_f64x4 acc[NF];
#pragma unroll
      for (int f = 0; f < NF; f++) acc[f] = g_f64x4{0.0, 0.0, 0.0, 0.0};
      tile_mac<NF>(acc, a.o0, MF, rt, S, KU * GB, KU / 4, li, lk);
      const int row = rt * 16 + li;
      if (row < nfp)
#pragma unroll
        for (int f = 0; f < NF; f++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++)
          {
            const int el = lk + 4 * rg;
            if (el < nval) a.disu_next[row + (long)nfp * (e0 + el) + f * plane_f] = acc[f][rg];
          }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pairwise face kernels; the two sides of a block may belong to different element blocks
// ---------------------------------------------------------------------------------------------------------------
struct GFaceArgs
{
  long npairs;
  const int *L, *R;
  const unsigned char *meta_l; // bit1 of the LEFT point: beta sign flipped (the switch of src/inters.cpp:568-581,620-633)
  long plane_l, plane_r;
  const double *disu_l, *disu_r, *fn_l, *fn_r, *norm_l, *tdA_l, *tdA_r;
  double *delta_l, *delta_r, *tconf_l, *tconf_r;
  Phys P;
};

__global__ __launch_bounds__(256) void gface_delta_kernel(const GFaceArgs a)
{
  constexpr int NF = 5;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.npairs) return;
  const long il = a.L[q], ir = a.R[q];
  const double beta = (a.meta_l[il] & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
  // every load before the first store (a load behind a store to memory that may overlap it waits for the store)
  double ul[NF], ur[NF];
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    ul[k] = a.disu_l[il + k * a.plane_l];
    ur[k] = a.disu_r[ir + k * a.plane_r];
  }
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    const double uc = 0.5 * (ul[k] + ur[k]) - beta * (ul[k] - ur[k]); // src/inters.cpp:637
    a.delta_l[il + k * a.plane_l] = uc - ul[k];
    a.delta_r[ir + k * a.plane_r] = uc - ur[k];
  }
}

// Riemann + LDG common flux of the interior pairs from u and Fn of both sides -> norm_tconf of both sides.
// ALL interior-face blocks of a stage in ONE launch.  A mixed mesh has one block per (left class, right
// class, face type) -- the channel four --, each a few tens of microseconds of work: launched one after the other every one
// pays its own ramp-up and tail (4 x 45 us for 208 MB, i.e. 1.2 TB/s).  Workgroups are dealt to the blocks in whole numbers
// (wg_start), so the block of a workgroup is uniform and its arguments come through scalar loads.
constexpr int GFACE_MAX_BLOCKS = 8;
struct GFaceMulti
{
  int nb;
  unsigned wg_start[GFACE_MAX_BLOCKS + 1];
  GFaceArgs blk[GFACE_MAX_BLOCKS];
};

template <int RS>
__global__ __launch_bounds__(256) void gface_flux_multi_kernel(const GFaceMulti m)
{
  constexpr int NF = 5, ND = 3;
  int b = 0;
  while (b + 1 < m.nb && blockIdx.x >= m.wg_start[b + 1]) b++;
  const GFaceArgs &a = m.blk[b];
  const long q = (long)(blockIdx.x - m.wg_start[b]) * 256 + threadIdx.x;
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
  for (int mm = 0; mm < ND; mm++) n[mm] = a.norm_l[il + mm * a.plane_l];
  const double tl = a.tdA_l[il], tr = a.tdA_r[ir];
  double fl[NF], fr[NF];
  const unsigned char mt = a.meta_l[il];
  const bool viscous = a.P.viscous;
  if (viscous)
  {
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      fl[k] = a.fn_l[il + k * a.plane_l];
      fr[k] = a.fn_r[ir + k * a.plane_r];
    }
  }
  riemann_flux_t<ND, RS, true>(a.P, ul, ur, n, fn);
  const double beta = (mt & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    double fv = 0.0;
    if (viscous)
    {
      fv = (0.5 + beta) * fl[k] - (0.5 - beta) * fr[k];
      fv -= a.P.ldg_tau * (ur[k] - ul[k]);
    }
    a.tconf_l[il + k * a.plane_l] = viscous ? fn[k] * tl + fv * tl : fn[k] * tl;
    a.tconf_r[ir + k * a.plane_r] = viscous ? -fn[k] * tr + -fv * tr : -fn[k] * tr;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// the consistent switch of src/inters.cpp:568-581 on the LEFT normal (exact zero tests), as physics.hpp's ldg_switch
static double ldg_switch_host(double beta, const double (&n)[3])
{
  if (n[0] < 0.0)
    beta = -beta;
  else if (n[0] == 0.0)
  {
    if ((n[0] + n[1]) < 0.0)
      beta = -beta;
    else if ((n[0] + n[1]) == 0.0)
    {
      if ((n[0] + n[2]) < 0.0) beta = -beta;
    }
  }
  return beta;
}

static int padded_operator(double **dst, const Operator &op, int M, int K)
{
  HFX_CHECK(op.present(), "general fused stage: an operator is missing");
  std::vector<double> h((size_t)op.m * op.k), p((size_t)M * K, 0.0);
  HFX_HIP(hipMemcpy(h.data(), op.dense, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  for (int c = 0; c < op.k; c++)
    for (int r = 0; r < op.m; r++) p[r + (size_t)M * c] = h[r + (size_t)op.m * c];
  if (*dst) (void)hipFree(*dst);
  HFX_HIP(hipMalloc((void **)dst, sizeof(double) * p.size()));
  HFX_HIP(hipMemcpy(*dst, p.data(), sizeof(double) * p.size(), hipMemcpyHostToDevice));
  return 0;
}

static size_t flux_lds_bytes(const GeneralData *g) { return sizeof(double) * GB * (5 * g->KU + 5 * g->KF + 15 * g->KU); }
static size_t update_lds_bytes(const GeneralData *g) { return sizeof(double) * GB * (5 * g->KF + 5 * g->KU); }

static int general_build(hfx_eles *e, hfx_inters *const *faces, int nfb, hfx_eles *const *eles, int neb)
{
  HFX_CHECK(e->n_dims == 3 && e->n_fields == 5, "general fused stage: three-dimensional Navier-Stokes / Euler blocks only");
  // (shock capturing follows the stage as its own step: general_shock_capture; an LES closure is evaluated in the flux kernel;
  // over-integration: the de-aliased flux is formed by the dense contractions before the flux kernel, which takes it in P3)
  HFX_CHECK(!(e->les_ready && e->over_int_ready), "general fused stage: an LES closure together with over-integration runs per method");
  if (e->les_ready)
  {
    static const int sizes[][2] = {{4, 12}, {10, 24}, {20, 40}, {6, 18}, {18, 39}, {40, 68}};
    bool known = false;
    for (const auto &sz : sizes) known = known || (e->n_upts == sz[0] && e->n_fpts == sz[1]);
    HFX_CHECK(known && e->ctx->params.viscous, "general fused stage: the LES closure is built for tetrahedra / prisms of orders 1..3 on a viscous run");
    // (partitioned blocks: the projected flux a partition face sends already contains F_sgs . n -- no third message; the
    // SVV closure filters the state after its flux-point values have left for the neighbours)
    for (int b = 0; b < nfb; b++)
      HFX_CHECK(!faces[b]->is_mpi || e->les.sgs_model != 3, "general fused stage: the SVV closure on partitioned blocks runs per method");
  }
  if (!e->general) e->general = new GeneralData();
  if (e->les_ready && e->les.sgs_model != 3 && les_len2_upload(e, &((GeneralData *)e->general)->les_len2)) return 1;
  const bool visc = e->ctx->params.viscous != 0;
  HFX_CHECK(!visc || e->viscous_ops, "general fused stage: viscous run but the block has no opp_4/5/6");
  if (!e->general) e->general = new GeneralData();
  GeneralData *g = (GeneralData *)e->general;
  const int nu = e->n_upts, nfp = e->n_fpts;
  g->KU = (nu + 3) & ~3; g->KF = (nfp + 3) & ~3; g->MU = (nu + 15) & ~15; g->MF = (nfp + 15) & ~15;
  HFX_CHECK(flux_lds_bytes(g) <= 160 * 1024, "general fused stage: a batch of this element class (%d solution, %d flux points) does not fit LDS", nu, nfp);
  if (padded_operator(&g->o0, e->opp_0, g->MF, g->KU) || padded_operator(&g->o3, e->opp_3, g->MU, g->KF)) return 1;
  for (int d = 0; d < 3; d++)
    if (padded_operator(&g->o1[d], e->opp_1[d], g->MF, g->KU) || padded_operator(&g->o2[d], e->opp_2[d], g->MU, g->KU)) return 1;
  {
    // the folded operators: O2f[d] = opp_2[d] - opp_3 . opp_1[d] (sum over the flux points in ascending order)
    std::vector<double> o3((size_t)nu * nfp), o1((size_t)nfp * nu), o2((size_t)nu * nu), pad((size_t)g->MU * g->KU);
    HFX_HIP(hipMemcpy(o3.data(), e->opp_3.dense, sizeof(double) * o3.size(), hipMemcpyDeviceToHost));
    for (int d = 0; d < 3; d++)
    {
      HFX_HIP(hipMemcpy(o1.data(), e->opp_1[d].dense, sizeof(double) * o1.size(), hipMemcpyDeviceToHost));
      HFX_HIP(hipMemcpy(o2.data(), e->opp_2[d].dense, sizeof(double) * o2.size(), hipMemcpyDeviceToHost));
      std::fill(pad.begin(), pad.end(), 0.0);
      for (int c = 0; c < nu; c++)
        for (int r = 0; r < nu; r++)
        {
          double t = 0.0;
          for (int j = 0; j < nfp; j++) t += o3[r + (size_t)nu * j] * o1[j + (size_t)nfp * c];
          pad[r + (size_t)g->MU * c] = o2[r + (size_t)nu * c] - t;
        }
      if (g->o2f[d]) (void)hipFree(g->o2f[d]);
      g->o2f[d] = nullptr;
      HFX_HIP(hipMalloc((void **)&g->o2f[d], sizeof(double) * pad.size()));
      HFX_HIP(hipMemcpy(g->o2f[d], pad.data(), sizeof(double) * pad.size(), hipMemcpyHostToDevice));
    }
  }
  if (visc)
  {
    if (padded_operator(&g->o6, e->opp_6, g->MF, g->KU)) return 1;
    for (int d = 0; d < 3; d++)
      if (padded_operator(&g->o4[d], e->opp_4[d], g->MU, g->KU) || padded_operator(&g->o5[d], e->opp_5[d], g->MU, g->KF)) return 1;
  }
  // per flux point: the LDG switch of the pair it is the LEFT point of (exact tests on the left normal, decided once),
  // and whether a boundary face owns it; every flux point must belong to exactly one registered face
  const long plane_f = (long)nfp * e->n_eles;
  std::vector<unsigned char> meta(plane_f, 0);
  std::vector<char> owned(plane_f, 0);
  std::vector<double> norm((size_t)plane_f * 3);
  HFX_HIP(hipMemcpy(norm.data(), e->norm_fpts, sizeof(double) * norm.size(), hipMemcpyDeviceToHost));
  g->any_bdy = false;
  for (int b = 0; b < nfb; b++)
  {
    hfx_inters *f = faces[b];
    if (f->is_mpi)
    {
      // partition faces (hfx_run_steps_partitioned_blocks): the point's correction and common flux come from the one-sided
      // kernels of kernels_mpi.hpp, like a boundary point's; what leaves for the neighbour is the point's Fn
      if (f->left == e)
        for (int o : f->hL) owned[o] = 1;
      continue;
    }
    if (f->is_bdy)
    {
      if (f->left != e) continue;
      g->any_bdy = true;
      for (int o : f->hL) { meta[o] |= 4; owned[o] = 1; }
      continue;
    }
    const long np = (long)f->n_inters * f->n_fpts_per_inter;
    if (f->left == e)
      for (long q = 0; q < np; q++)
      {
        const int il = f->hL[q];
        owned[il] = 1;
        const double n[3] = {norm[il], norm[il + plane_f], norm[il + 2 * plane_f]};
        if (ldg_switch_host(1.0, n) < 0) meta[il] |= 2;
      }
    if (f->right == e)
      for (long q = 0; q < np; q++) owned[f->hR[q]] = 1;
  }
  for (long o = 0; o < plane_f; o++) HFX_CHECK(owned[o], "general fused stage: flux point %ld belongs to no registered face", o);
  {
    // partner words: pairs inside this block and pairs with another block of this call (mixed meshes: prism | tetrahedron faces);
    // the beta sign is the LEFT point's (decided above for this block's left points; for a pair whose left side lies in another
    // block it is decided here the same way, from that block's normals)
    std::vector<int> nbr(plane_f, -1);
    bool fits = neb <= 4, any = false;
    for (int i = 0; i < neb; i++) fits = fits && (long)eles[i]->n_fpts * eles[i]->n_eles < (1L << 27);
    auto block_of = [&](const hfx_eles *x) { for (int i = 0; i < neb; i++) if (eles[i] == x) return i; return -1; };
    for (int b = 0; b < nfb && fits; b++)
    {
      hfx_inters *f = faces[b];
      if (f->is_bdy || f->is_mpi || (f->left != e && f->right != e)) continue;
      const int bl = block_of(f->left), br = block_of(f->right);
      if (bl < 0 || br < 0) { fits = false; break; }
      const long np = (long)f->n_inters * f->n_fpts_per_inter;
      std::vector<double> nl;
      const long plane_l = (long)f->left->n_fpts * f->left->n_eles;
      if (f->left != e)
      {
        nl.resize((size_t)plane_l * 3);
        HFX_HIP(hipMemcpy(nl.data(), f->left->norm_fpts, sizeof(double) * nl.size(), hipMemcpyDeviceToHost));
      }
      for (long q = 0; q < np; q++)
      {
        const int il = f->hL[q], ir = f->hR[q];
        int flip;
        if (f->left == e)
          flip = meta[il] & 2;
        else
        {
          const double n[3] = {nl[il], nl[il + plane_l], nl[il + 2 * plane_l]};
          flip = ldg_switch_host(1.0, n) < 0 ? 2 : 0;
        }
        if (f->left == e) nbr[il] = (ir << 4) | (br << 2) | flip;
        if (f->right == e) nbr[ir] = (il << 4) | (bl << 2) | flip | 1;
        any = true;
      }
    }
    if (g->nbr) { (void)hipFree(g->nbr); g->nbr = nullptr; }
    g->n_blocks = 0;
    if (fits && any)
    {
      HFX_HIP(hipMalloc((void **)&g->nbr, sizeof(int) * (size_t)plane_f));
      HFX_HIP(hipMemcpy(g->nbr, nbr.data(), sizeof(int) * (size_t)plane_f, hipMemcpyHostToDevice));
      g->n_blocks = neb;
      for (int i = 0; i < neb; i++) g->blocks[i] = eles[i];
    }
  }
  if (g->meta) (void)hipFree(g->meta);
  HFX_HIP(hipMalloc((void **)&g->meta, (size_t)plane_f));
  HFX_HIP(hipMemcpy(g->meta, meta.data(), (size_t)plane_f, hipMemcpyHostToDevice));
  if (!g->disu_alt) HFX_HIP(hipMalloc((void **)&g->disu_alt, sizeof(double) * plane_f * e->n_fields));
  if (!g->fn_fpts) HFX_HIP(hipMalloc((void **)&g->fn_fpts, sizeof(double) * plane_f * e->n_fields));
  g->built = true;
  return 0;
}

// does the flux kernel of this block stage its inputs with all loads requested up front (the form that can also form the
// LDG corrections itself)?  -- an instantiated element size whose staging registers fit (general_flux_kernel, P0)
static bool general_batched(const hfx_eles *e)
{
  const GeneralData *g = (const GeneralData *)e->general;
  if (!g) return false;
  static const int sizes[][2] = {{4, 12}, {10, 24}, {20, 40}, {6, 18}, {18, 39}, {40, 68}};
  bool known = false;
  for (const auto &sz : sizes) known = known || (e->n_upts == sz[0] && e->n_fpts == sz[1]);
  if (!known) return false;
  int w = e->ctx->opt.general_waves;
  if (w == 0) w = flux_lds_bytes(g) <= 80 * 1024 ? 4 : 8;
  const int T = 64 * w, mq_u = (g->KU * GB + T - 1) / T, mq_d = (g->KF * GB + T - 1) / T;
  return (mq_u + mq_d) * 5 <= 30;
}

static GenArgs gen_args(hfx_eles *e, int in_step, bool last_stage)
{
  GeneralData *g = (GeneralData *)e->general;
  const hfx_params &p = e->ctx->params;
  GenArgs a{};
  a.inv_nu = (unsigned)(4294967296ull / (unsigned)e->n_upts) + 1u; a.inv_nfp = (unsigned)(4294967296ull / (unsigned)e->n_fpts) + 1u;
  a.n_eles = e->n_eles; a.nu = e->n_upts; a.nfp = e->n_fpts; a.KU = g->KU; a.KF = g->KF; a.MU = g->MU; a.MF = g->MF;
  a.o0 = g->o0; a.o3 = g->o3; a.o6 = g->o6;
  a.fold = e->ctx->opt.fold_general ? 1 : 0;
  for (int d = 0; d < 3; d++) { a.o1[d] = g->o1[d]; a.o2[d] = a.fold ? g->o2f[d] : g->o2[d]; a.o4[d] = g->o4[d]; a.o5[d] = g->o5[d]; }
  a.u0 = e->arr[HFX_DISU_UPTS0]; a.delta = e->arr[HFX_DELTA_DISU_FPTS]; a.disu = e->arr[HFX_DISU_FPTS];
  a.detjac_upts = e->detjac_upts; a.JGinv_upts = e->JGinv_upts; a.detjac_fpts = e->detjac_fpts; a.JGinv_fpts = e->JGinv_fpts;
  a.norm_fpts = e->norm_fpts; a.meta = g->meta;
  a.nbr = (e->ctx->opt.gather_delta && p.viscous && general_batched(e)) ? g->nbr : nullptr;
  for (int i = 0; i < 4; i++)
  {
    a.disu_b[i] = (i < g->n_blocks) ? g->blocks[i]->arr[HFX_DISU_FPTS] : nullptr;
    a.plane_b[i] = (i < g->n_blocks) ? (long)g->blocks[i]->n_fpts * g->blocks[i]->n_eles : 0;
  }
  a.div = e->arr[HFX_DIV_TCONF_UPTS]; a.ntd = e->arr[HFX_NORM_TDISF_FPTS]; a.fn = g->fn_fpts;
  a.grad_fpts = (g->any_bdy && p.viscous) ? e->arr[HFX_GRAD_DISU_FPTS] : nullptr;
  a.P = e->ctx->phys();
  a.u0w = e->arr[HFX_DISU_UPTS0]; a.u1 = e->arr[HFX_DISU_UPTS1];
  a.tconf = e->arr[HFX_NORM_TCONF_FPTS]; a.div_in = e->arr[HFX_DIV_TCONF_UPTS];
  a.src = e->src_nonzero ? e->arr[HFX_SRC_UPTS] : nullptr;
  a.dt_local = e->arr[HFX_DT_LOCAL];
  a.disu_next = g->disu_alt;
  a.nan_flag = e->nan_flag;
  a.adv_type = p.adv_type; a.in_step = in_step; a.dt_local_on = p.dt_type == 2; a.dt = p.dt;
  a.rk_a = (p.adv_type >= 3) ? p.RK_a[in_step] : 0.0;
  a.rk_b = (p.adv_type >= 3) ? p.RK_b[in_step] : 0.0;
  a.need_u1 = (p.adv_type >= 3) || (p.adv_type == 1 && in_step == 3) || (p.adv_type == 2 && in_step == 2);
  if (e->ctx->opt.flux_stamps && !g->stamps)
  {
    if (hipMalloc((void **)&g->stamps, sizeof(long long) * 16 * 16) == hipSuccess) (void)hipMemset(g->stamps, 0, sizeof(long long) * 16 * 16);
  }
  a.stamps = g->stamps;
  a.les = e->les; a.les_len2 = g->les_len2; a.tdA_fpts = e->tdA_fpts;
  a.tdisf_in = e->over_int_ready ? e->arr[HFX_TDISF_UPTS] : nullptr;
  a.write_div = 1; // the flux kernel left the discontinuous part there: always complete it (the monitors read it)
  (void)last_stage;
  return a;
}

static GFaceArgs gface_args(hfx_inters *f)
{
  hfx_eles *l = f->left, *r = f->right;
  GeneralData *gl = (GeneralData *)l->general, *gr = (GeneralData *)r->general;
  GFaceArgs a{};
  a.npairs = (long)f->n_inters * f->n_fpts_per_inter;
  a.L = f->L; a.R = f->R; a.meta_l = gl->meta;
  a.plane_l = (long)l->n_fpts * l->n_eles; a.plane_r = (long)r->n_fpts * r->n_eles;
  a.disu_l = l->arr[HFX_DISU_FPTS]; a.disu_r = r->arr[HFX_DISU_FPTS];
  a.fn_l = gl->fn_fpts; a.fn_r = gr->fn_fpts;
  a.norm_l = l->norm_fpts; a.tdA_l = l->tdA_fpts; a.tdA_r = r->tdA_fpts;
  a.delta_l = l->arr[HFX_DELTA_DISU_FPTS]; a.delta_r = r->arr[HFX_DELTA_DISU_FPTS];
  a.tconf_l = l->arr[HFX_NORM_TCONF_FPTS]; a.tconf_r = r->arr[HFX_NORM_TCONF_FPTS];
  a.P = l->ctx->phys();
  return a;
}

template <int W, int NUc, int NFPc>
static int launch_element_kernels_t(hfx_eles *e, const GenArgs &a, bool flux)
{
  GeneralData *g = (GeneralData *)e->general;
  const unsigned grid = (unsigned)((e->n_eles + GB - 1) / GB);
  hipStream_t st = e->ctx->stream;
  if (flux)
  {
    const size_t lds = flux_lds_bytes(g);
    // (function attributes are per device: set whenever the image exceeds the default, as launch_dense does -- a cached
    // "configured" size would be wrong for a second context on another GPU of the same process)
    // a closure with an SGS flux (every model but the spectral vanishing viscosity, which only filters the state)
    const bool les = e->les_ready && e->les.sgs_model != 3 && e->ctx->params.viscous;
    if constexpr (NUc != 0)
    {
      if (les)
      {
        if (lds > 48 * 1024)
          HFX_HIP(hipFuncSetAttribute((const void *)general_flux_kernel<W, NUc, NFPc, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((general_flux_kernel<W, NUc, NFPc, true>), dim3(grid), dim3(64 * W), lds, st, a);
        HFX_HIP(hipGetLastError());
        return 0;
      }
    }
    HFX_CHECK(!les, "general fused stage: the LES closure is built for the element classes with compile-time sizes");
    if (lds > 48 * 1024)
      HFX_HIP(hipFuncSetAttribute((const void *)general_flux_kernel<W, NUc, NFPc>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((general_flux_kernel<W, NUc, NFPc>), dim3(grid), dim3(64 * W), lds, st, a);
  }
  else
  {
    const size_t lds = update_lds_bytes(g);
    if (lds > 48 * 1024)
      HFX_HIP(hipFuncSetAttribute((const void *)general_update_kernel<W, NUc, NFPc>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((general_update_kernel<W, NUc, NFPc>), dim3(grid), dim3(64 * W), lds, st, a);
  }
  HFX_HIP(hipGetLastError());
  return 0;
}

// the element classes of the reference's orders 1..3 (tetrahedra 4/12, 10/24, 20/40; prisms 6/18, 18/39, 40/68 solution /
// flux points) get instantiations with compile-time sizes; anything else runs the size-generic form
template <int W>
static int launch_element_kernels(hfx_eles *e, const GenArgs &a, bool flux)
{
#define HFX_GEN_CASE(NU_, NFP_) \
  if (a.nu == NU_ && a.nfp == NFP_) return launch_element_kernels_t<W, NU_, NFP_>(e, a, flux);
  HFX_GEN_CASE(4, 12) HFX_GEN_CASE(10, 24) HFX_GEN_CASE(20, 40) HFX_GEN_CASE(6, 18) HFX_GEN_CASE(18, 39) HFX_GEN_CASE(40, 68)
#undef HFX_GEN_CASE
  return launch_element_kernels_t<W, 0, 0>(e, a, flux);
}

// which: 0 the whole stage, 1 .. 4 one of its four parts (for the per-kernel timing)
static int general_stage(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int in_step, bool last_stage, int which)
{
  hfx_ctx *ctx = eles[0]->ctx;
  const Phys P = ctx->phys();
  hipStream_t st = ctx->stream;
  if (P.viscous && (which == 0 || which == 1))
    for (int b = 0; b < nfb; b++)
    {
      if (faces[b]->is_bdy)
      {
        if (hfx_bdy_launch_internal(faces[b], 0, 1)) return 1; // ghost state -> inviscid common flux, LDG common solution
        continue;
      }
      if (faces[b]->is_mpi) continue;
      {
        // both sides' flux kernels form these corrections themselves?
        auto own = [&](hfx_eles *x) {
          return x->general && ((GeneralData *)x->general)->nbr && ((GeneralData *)x->general)->n_blocks && general_batched(x);
        };
        if (ctx->opt.gather_delta && own(faces[b]->left) && own(faces[b]->right)) continue;
      }
      const GFaceArgs a = gface_args(faces[b]);
      if (a.npairs == 0) continue;
      hipLaunchKernelGGL(gface_delta_kernel, dim3((unsigned)((a.npairs + 255) / 256)), dim3(256), 0, st, a);
    }
  if (which == 0 || which == 2)
    for (int i = 0; i < neb; i++)
    {
      // polynomial de-aliasing (src/solver.cpp:82-91): tdisf_upts = over_int_filter . F(opp_over_int_cubpts . u)
      if (eles[i]->over_int_ready && hfx_eles_evaluate_invFlux_over_int(eles[i])) return 1;
      const GenArgs a = gen_args(eles[i], in_step, last_stage);
      // two workgroups per CU where the batch's LDS image allows it (4 waves each), otherwise one of 8 waves
      int w = ctx->opt.general_waves;
      if (w == 0) w = flux_lds_bytes((GeneralData *)eles[i]->general) <= 80 * 1024 ? 4 : 8;
      if (w == 3 ? launch_element_kernels<3>(eles[i], a, true) : w == 4 ? launch_element_kernels<4>(eles[i], a, true)
                                                                         : launch_element_kernels<8>(eles[i], a, true))
        return 1;
    }
  if (which == 0 || which == 3)
  {
    // boundary faces beside the interior ones: both need the flux kernels' results and write norm_tconf at disjoint points
    bool any_bdy = false;
    for (int b = 0; b < nfb; b++) any_bdy = any_bdy || (faces[b]->is_bdy && faces[b]->n_inters > 0);
    const bool beside = any_bdy && ctx->opt.bdy_beside;
    if (beside && side_stream_fork(ctx)) return 1;
    for (int b = 0; b < nfb; b++)
      if (faces[b]->is_bdy && hfx_bdy_launch_internal(faces[b], P.viscous ? 1 : 0, 1)) return 1;
    if (beside && side_stream_join(ctx)) return 1;
    // all interior-face blocks in one launch (groups of GFACE_MAX_BLOCKS)
    GFaceMulti m{};
    auto flush = [&]() -> int {
      if (m.nb == 0) return 0;
      const dim3 grid(m.wg_start[m.nb]);
      if (P.riemann == 0)
        hipLaunchKernelGGL(gface_flux_multi_kernel<0>, grid, dim3(256), 0, st, m);
      else if (P.riemann == 2)
        hipLaunchKernelGGL(gface_flux_multi_kernel<2>, grid, dim3(256), 0, st, m);
      else
        hipLaunchKernelGGL(gface_flux_multi_kernel<3>, grid, dim3(256), 0, st, m);
      m.nb = 0;
      return 0;
    };
    for (int b = 0; b < nfb; b++)
    {
      if (faces[b]->is_bdy || faces[b]->is_mpi) continue;
      const GFaceArgs a = gface_args(faces[b]);
      if (a.npairs == 0) continue;
      if (m.nb == 0) m.wg_start[0] = 0;
      m.blk[m.nb] = a;
      m.wg_start[m.nb + 1] = m.wg_start[m.nb] + (unsigned)((a.npairs + 255) / 256);
      if (++m.nb == GFACE_MAX_BLOCKS && flush()) return 1;
    }
    if (flush()) return 1;
    if (beside && side_stream_wait(ctx)) return 1;
  }
  if (which == 0 || which == 4)
  {
    for (int i = 0; i < neb; i++)
    {
      const GenArgs a = gen_args(eles[i], in_step, last_stage);
      // four waves where all of a thread's staging loads fit its registers at once (general_update_kernel, BATCH), else eight:
      // P3 prisms stage 5 + 3 doubles per field and thread on four waves -- past the limit, every trip of the staging loops then
      // waited for its own loads (0.233 ms for 451 MB)
      const int w_opt = ctx->opt.general_update_waves;
      const int mq_d = (a.KF * GB + 255) / 256, mq_u = (a.nu * GB + 255) / 256;
      const bool eight = w_opt ? w_opt == 8 : (mq_d * 5 > 20 || mq_u * 5 > 15);
      if (eight ? launch_element_kernels<8>(eles[i], a, false) : launch_element_kernels<4>(eles[i], a, false)) return 1;
    }
    for (int i = 0; i < neb; i++) std::swap(eles[i]->arr[HFX_DISU_FPTS], ((GeneralData *)eles[i]->general)->disu_alt);
  }
  HFX_HIP(hipGetLastError());
  return 0;
}

static int general_prepare(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb)
{
  hfx_ctx *ctx = eles[0]->ctx;
  for (int i = 0; i < neb; i++)
  {
    HFX_CHECK(eles[i] && eles[i]->ctx == ctx, "element blocks of different contexts");
    HFX_CHECK(eles[i]->n_eles > 0, "general fused stage: empty element block");
  }
  for (int b = 0; b < nfb; b++)
  {
    bool l = false, r = faces[b]->right == nullptr;
    for (int i = 0; i < neb; i++)
    {
      l = l || faces[b]->left == eles[i];
      r = r || faces[b]->right == eles[i];
    }
    HFX_CHECK(l && r, "face block %d refers to an element block that is not part of this call", b);
  }
  for (int i = 0; i < neb; i++)
  {
    GeneralData *g = (GeneralData *)eles[i]->general;
    bool same = g && g->built;
    if (same && g->n_blocks)
    {
      same = g->n_blocks == neb;
      for (int j = 0; j < neb && same; j++) same = g->blocks[j] == eles[j];
    }
    if (!same)
      if (general_build(eles[i], faces, nfb, eles, neb)) return 1;
  }
  return 0;
}

// the pieces of a stage for the partitioned driver (comm.hip): which = 1 .. 4 as in general_time_kernels; `faces` may hold
// partition-face blocks (they take part in the build -- every flux point needs its face -- and are skipped by the pairwise loops)
int general_stage_part(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int in_step, bool write_div, int which)
{
  return general_stage(eles, neb, faces, nfb, in_step, write_div, which);
}
const double *general_fn_fpts(const hfx_eles *e) { return e->general ? ((const GeneralData *)e->general)->fn_fpts : nullptr; }

int general_deferred_prepare(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb)
{
  return general_prepare(eles, neb, faces, nfb);
}

int general_deferred_stage(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int in_step, bool write_div)
{
  if (general_prepare(eles, neb, faces, nfb)) return 1; // (no-op unless the registration changed)
  return general_stage(eles, neb, faces, nfb, in_step, write_div, 0);
}

// eles::shock_capture behind a general fused stage (src/HiFiLES.cpp:214-216): the modal filter changes the state of the elements
// the sensor marks, so the flux-point values the update kernel left for the next stage are extrapolated again
int general_shock_capture(hfx_eles *const *eles, int neb)
{
  for (int i = 0; i < neb; i++)
    if (eles[i]->shock_ready)
    {
      if (hfx_eles_shock_capture(eles[i])) return 1;
      if (hfx_eles_extrapolate_solution(eles[i])) return 1;
    }
  return 0;
}

int general_run_steps(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int n_steps)
{
  if (general_prepare(eles, neb, faces, nfb)) return 1;
  if (n_steps <= 0) return 0;
  hfx_ctx *ctx = eles[0]->ctx;
  const int adv = ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  // disu_fpts of the current state (the caller may have changed disu_upts since the last call)
  for (int i = 0; i < neb; i++)
    if (hfx_eles_extrapolate_solution(eles[i])) return 1;
  for (int s = 0; s < n_steps; s++)
  {
    if (ctx->params.dt_type != 0)
    {
      double dt_min = 1e12;
      for (int i = 0; i < neb; i++)
      {
        if (calc_time_step(eles[i], nullptr)) return 1; /* src/HiFiLES.cpp:198 */
        dt_min = std::min(dt_min, ctx->params.dt);
      }
      ctx->params.dt = dt_min;
    }
    for (int rk = 0; rk < nst; rk++)
    {
      // closures that filter the solution do so at the first stage of a step (src/solver.cpp:55-62); the SVV closure replaces the
      // state, whose flux-point values are then recomputed
      if (rk == 0)
        for (int i = 0; i < neb; i++)
          if (eles[i]->les_ready && eles[i]->les.sgs_model >= 2)
          {
            if (hfx_eles_calc_sgs_terms(eles[i])) return 1;
            if (eles[i]->les.sgs_model == 3 && hfx_eles_extrapolate_solution(eles[i])) return 1;
          }
      if (general_stage(eles, neb, faces, nfb, rk, rk == nst - 1, 0)) return 1;
      if (general_shock_capture(eles, neb)) return 1;
    }
    advance_ramp_counters(faces, nfb); /* src/HiFiLES.cpp:224-225 */
  }
  return 0;
}

int general_time_kernels(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int reps, double *ms)
{
  if (general_prepare(eles, neb, faces, nfb)) return 1;
  hfx_ctx *ctx = eles[0]->ctx;
  const int adv = ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  hipStream_t st = ctx->stream;
  // one set of events per repetition and ONE synchronisation at the end: a host synchronisation per stage let the queue
  // run dry, and the first kernel after it (the flux kernel) then measured 10 % slower than in the running pipeline
  std::vector<hipEvent_t> ev((size_t)reps * 5);
  for (auto &x : ev) HFX_HIP(hipEventCreate(&x));
  for (int i = 0; i < neb; i++)
    if (hfx_eles_extrapolate_solution(eles[i])) return 1;
  double acc[4] = {0, 0, 0, 0};
  for (int r = 0; r < reps; r++)
  {
    const int rk = r % nst;
    for (int w = 1; w <= 4; w++)
    {
      HFX_HIP(hipEventRecord(ev[5 * r + w - 1], st));
      if (general_stage(eles, neb, faces, nfb, rk, rk == nst - 1, w)) return 1;
    }
    HFX_HIP(hipEventRecord(ev[5 * r + 4], st));
  }
  HFX_HIP(hipStreamSynchronize(st));
  for (int r = 0; r < reps; r++)
    for (int w = 0; w < 4; w++)
    {
      float t = 0;
      HFX_HIP(hipEventElapsedTime(&t, ev[5 * r + w], ev[5 * r + w + 1]));
      acc[w] += t;
    }
  for (auto &x : ev) (void)hipEventDestroy(x);
  for (int i = 0; i < 8; i++) ms[i] = (i < 4) ? acc[i] / reps : 0.0;
  for (int i = 0; i < neb; i++)
  {
    GeneralData *g = (GeneralData *)eles[i]->general;
    if (!g || !g->stamps) continue;
    long long h[16 * 16];
    HFX_HIP(hipMemcpy(h, g->stamps, sizeof h, hipMemcpyDeviceToHost));
    for (int w = 0; w < 8 && h[w * 16] != 0; w++)
    {
      fprintf(stderr, "general flux kernel (block %d) wave %d cycles: ", i, w);
      for (int q = 1; q <= 9; q++) fprintf(stderr, "%s%lld", q > 1 ? " " : "", h[w * 16 + q] - h[w * 16 + q - 1]);
      fprintf(stderr, "   (P0 | bar | P1 | bar | P2 | bar | P3 | bar | P4)  total %lld\n", h[w * 16 + 9] - h[w * 16]);
    }
  }
  return 0;
}

// ALGORITHMIC HBM bytes per launch of the four kernels, summed over the element blocks (doubles listed per element)
void general_kernel_bytes(hfx_eles *const *eles, int neb, double *bytes)
{
  for (int i = 0; i < 8; i++) bytes[i] = 0.0;
  for (int i = 0; i < neb; i++)
  {
    const hfx_eles *e = eles[i];
    const double nu = e->n_upts, nfp = e->n_fpts, nf = e->n_fields, nd = e->n_dims, ne = e->n_eles;
    bytes[0] += ne * (8.0 * (2 * nfp * nf) + 4.0 * nfp + nfp * 0.5);                                        // disu r, delta w, index + meta
    bytes[1] += ne * 8.0 * (nu * nf + nfp * nf + nu * (nd * nd + 1) + nfp * (nd * nd + 1) + nfp * nd + nfp * nf // u, delta, metrics, normals, disu r
                            + nu * nf + 2 * nfp * nf);                                                       // div, norm_tdisf, Fn w
    bytes[2] += ne * (8.0 * (nfp * nf + nfp * nf + 0.5 * nfp * nd + nfp + nfp * nf) + 4.0 * nfp);            // disu, Fn, normal(left), tdA r; tconf w
    bytes[3] += ne * 8.0 * (3 * nu * nf + nu + 2 * nfp * nf + 3 * nu * nf + nfp * nf);                       // u0,u1,div,detjac,tconf,ntd r; u0,u1,div,disu w
    if (e->ctx->opt.fold_general)
    {
      // folded correction: norm_tdisf is neither written (flux kernel) nor read (update kernel)
      bytes[1] -= ne * 8.0 * nfp * nf;
      bytes[3] -= ne * 8.0 * nfp * nf;
    }
    const GeneralData *g = (const GeneralData *)e->general;
    if (g && g->nbr && e->ctx->opt.gather_delta && e->ctx->params.viscous && general_batched(e))
    {
      // LDG corrections formed in the flux kernel: the pairwise kernel is not launched for the pairs inside the block (the
      // partner values it reads are as many doubles as the corrections it no longer reads) + a partner word per point
      bytes[0] -= ne * (8.0 * (2 * nfp * nf) + 4.0 * nfp + nfp * 0.5);
      bytes[1] += ne * 4.0 * nfp;
    }
  }
}

} // namespace hfx
