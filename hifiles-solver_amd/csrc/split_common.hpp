// split_common.hpp -- tables, element geometry and device helpers shared by the split fused kernels
// (device code of the split fused stage; included by fused_hex.hip only -- one translation unit, so that every kernel is
// instantiated once)
#pragma once
#include <vector>

#include "fused_hex.hpp"
#include "physics.hpp"

namespace hfx
{

constexpr int MAX_TAB = 256;

// minimum waves per SIMD the residual kernel of fused = 2 is compiled for (second __launch_bounds__ argument)
#ifndef HFX_SPLIT_WAVES_RES
#define HFX_SPLIT_WAVES_RES 4
#endif

struct FusedData
{
  unsigned char *meta = nullptr; // bit0: this point is the RIGHT side, bit1: beta sign flipped, bit2: boundary point
  double *disu_alt = nullptr;    // second disu_fpts buffer
  double *fn_fpts = nullptr;     // split variant 3: projected viscous flux per flux point (n_fpts,n_eles,n_fields)
  // tensor-product tables of the sum-factorised flux kernel (valid when tensor_ok)
  long long *stamps = nullptr; // diagnostics buffer (HFX_FLUX_STAMPS=1)
  bool tensor_ok = false;
  double *t_coef = nullptr; // Dm[N][N] | c5[ND][2][N] | Lf[ND][2][N] | L1[ND][2][N] | c3[ND][2][N]
  std::vector<double> h_coef; // host copy
  int *t_idx = nullptr;     // pf[ND][L][2] | fdq[NFP] | fbase[NFP]
  unsigned *pk_g = nullptr, *pk_r = nullptr; // packed operator rows of the gradient / residual kernel
  double *tab_g = nullptr, *tab_r = nullptr; // value tables (MAX_TAB doubles)
  int *o1m_dim = nullptr;                    // (n_fpts) dimension slab of the merged opp_1 row
  int *nbr = nullptr;                        // (n_fpts, n_eles) partner of every interior flux point (split3_kernels.hpp, Split2Args::nbr)
  // partitioned blocks: the elements that own a flux point without a registered face (= a partition-face point), and the rest
  int *upd_list_b = nullptr, *upd_list_i = nullptr;
  long n_list_b = 0, n_list_i = 0;
  long n_list_i1 = 0; // the flux kernel takes the others in two parts: upd_list_i[0 .. n_list_i1) and the rest
  double *les_len2 = nullptr;                // (n_upts, n_eles) squared length scale of the LES closure evaluated in the flux kernel
  bool gather_on = false;                    // the last stage formed the interior LDG corrections in the flux kernel (no face_delta launch)
  bool built = false;
};

constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }
constexpr int words_of(int w) { return (w + 1) / 2; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int ND, int N>
struct Geo
{
  static constexpr int NF = ND + 2;
  static constexpr int NU = ipow(N, ND);
  static constexpr int NFP = 2 * ND * ipow(N, ND - 1);
  static constexpr int WU = (NU + 63) / 64;  // solution-point waves
  static constexpr int WF = (NFP + 63) / 64; // flux-point waves
  static constexpr int TU = 64 * WU;
  static constexpr int TB = 64 * (WU + WF);       // gradient kernel: roles U, F
  static constexpr int TBR = 64 * (WU + 2 * WF);  // residual kernel: roles U, A, B
  static constexpr int UNP = (NF * NU + TU - 1) / TU; // doubles of the next element's state per upt thread
  static constexpr int WN = words_of(N);
  // packed-row layout, gradient kernel: opp_4[d] | opp_5[d] (rows = upts) | opp_0 | opp_6 (rows = fpts)
  static constexpr int G_O4 = 0;
  static constexpr int G_O5 = G_O4 + ND * WN * NU;
  static constexpr int G_O0 = G_O5 + ND * words_of(2) * NU;
  static constexpr int G_O6 = G_O0 + WN * NFP;
  static constexpr int G_END = G_O6 + WN * NFP;
  static constexpr int G_WU = ND * WN + ND * words_of(2); // words per upt thread
  static constexpr int G_WF = 2 * WN;                     // words per fpt thread
  // residual kernel: opp_2[d] | opp_3 (upts) | opp_0 | merged opp_1 (fpts)
  static constexpr int R_O2 = 0;
  static constexpr int R_O3 = R_O2 + ND * WN * NU;
  static constexpr int R_O0 = R_O3 + words_of(2 * ND) * NU;
  static constexpr int R_O1 = R_O0 + WN * NFP;
  static constexpr int R_END = R_O1 + WN * NFP;
  static constexpr int R_WU = ND * WN + words_of(2 * ND);
  static constexpr int R_WF = 2 * WN;
};

// acc += sum_q tab[vid_q] * data[col_q], ascending q (= ascending column).  `w` is a
// register array subscripted with compile-time constants only.
template <int W, int OFF, int PW>
__device__ __forceinline__ double row_dot(const unsigned (&w)[PW], const double *tab, const double *data, double acc)
{
#pragma unroll
  for (int i = 0; i < words_of(W); i++)
  {
    // The unpacked (value id, column) pairs are loop invariant; left alone the compiler hoists
    // all of them out of the persistent loop and the ~12 packed registers turn back into ~100
    // address registers.  The empty asm makes the word opaque so that it is unpacked at the use.
    unsigned word = w[OFF + i];
    asm volatile("" : "+v"(word));
    {
      const unsigned ent = word & 0xffffu;
      acc += tab[ent >> 8] * data[ent & 0xffu];
    }
    if (2 * i + 1 < W)
    {
      const unsigned ent = word >> 16;
      acc += tab[ent >> 8] * data[ent & 0xffu];
    }
  }
  return acc;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// GLOBAL load and store of the wave (s_waitcnt vmcnt(0)); the roles exchange data through LDS
// only, so waiting for the LDS counter is sufficient and result stores / prefetches stay in flight.
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// g_phys(d) = sum_l (inv_detjac * g_ref(l)) * JGinv(l,d)   (BLAS=NO branch of src/eles.cpp:1975-1979)
template <int ND>
__device__ __forceinline__ void to_physical(const double inv_detjac, const double (&JG)[ND * ND], const double (&tg)[ND],
                                            double (&cg)[ND])
{
#pragma unroll
  for (int d = 0; d < ND; d++) cg[d] = 0.0;
#pragma unroll
  for (int l = 0; l < ND; l++)
  {
    const double temp = inv_detjac * tg[l];
#pragma unroll
    for (int d = 0; d < ND; d++) cg[d] += temp * JG[l + ND * d];
  }
}

} // namespace hfx
