// fused_hex.hip -- the split fused stage for tensor-product elements (hexes, quads).
//
// One RK stage = the 17 calls of CalcResidual + AdvanceSolution
// (/root/reference/src/solver.cpp:50-223, src/HiFiLES.cpp:201-217).  Executed call by
// call they stream ~64 000 doubles per P4 hex through HBM.  Here the stage is cut at the two
// places where data must cross elements (the LDG common solution, the common fluxes) and everything element-local
// between two cuts is one kernel: four launches per stage.  Two variants:
//   fused = 2: keeps the reference's arrays (grad_disu_upts / grad_disu_fpts in HBM); carries the LES closure
//   fused = 3: the kernel that has the corrected gradient in registers goes straight on to the fluxes; the default
// plus the same stage cut into five phases for a partitioned block (hfx_stage_partitioned).
//
// Operator rows of the element kernels are either sum-factorised (1-D matrices through scalar registers, when the
// registered operators are bit-exactly tensor products) or DICTIONARY-COMPRESSED sparse rows: a tensor-product
// operator has only a handful of distinct values, so a row entry is 16 bits (value id, column), two per 32-bit
// register, the values in a 2 kB LDS table.  The arithmetic is unchanged: the same non-zeros, multiplied in the same
// ascending-column order as the reference dgemm (src/funcs.cpp:110-117).
#include "fused_hex.hpp"
#include "tensor_ops.hpp"

#include <algorithm>
#include <cstring>

#include "kernels_mpi.hpp"
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
  double *t_coef = nullptr; // Dm[N][N] | c5[ND][2][N] | Lf[ND][2][N] | L1[ND][2][N]
  int *t_idx = nullptr;     // pf[ND][L][2] | fdq[NFP] | fbase[NFP]
  unsigned *pk_g = nullptr, *pk_r = nullptr; // packed operator rows of the gradient / residual kernel
  double *tab_g = nullptr, *tab_r = nullptr; // value tables (MAX_TAB doubles)
  int *o1m_dim = nullptr;                    // (n_fpts) dimension slab of the merged opp_1 row
  bool built = false;
};

void fused_invalidate(hfx_eles *e)
{
  if (e && e->fused) e->fused->built = false;
}

void fused_destroy(hfx_eles *e)
{
  if (!e || !e->fused) return;
  FusedData *f = e->fused;
  void *p[] = {f->meta, f->disu_alt, f->fn_fpts, f->t_coef, f->t_idx, f->pk_g, f->pk_r, f->tab_g, f->tab_r, f->o1m_dim};
  for (void *q : p)
    if (q) (void)hipFree(q);
  delete f;
  e->fused = nullptr;
}

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

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static int tensor_n(const hfx_eles *e)
{
  // N with N^ND = n_upts and 2 ND N^(ND-1) = n_fpts, or 0
  for (int n = 2; n <= 8; n++)
    if (ipow(n, e->n_dims) == e->n_upts && 2 * e->n_dims * ipow(n, e->n_dims - 1) == e->n_fpts) return n;
  return 0;
}

struct Dict
{
  std::vector<double> vals;
  Dict() { vals.push_back(0.0); } // id 0 = +0.0: padding entries multiply by it
  int id(double v)
  {
    for (size_t i = 0; i < vals.size(); i++)
      if (std::memcmp(&vals[i], &v, sizeof v) == 0) return (int)i;
    vals.push_back(v);
    return (int)vals.size() - 1;
  }
};

// pack `w` entries per row of an operator given in host ELL form (width hw) at word offset `off`
static void pack_rows(std::vector<unsigned> &pk, size_t off, int m, int w, const double *hv, const int *hi, int hw, Dict &dict)
{
  for (int r = 0; r < m; r++)
    for (int q = 0; q < w; q++)
    {
      const double v = (q < hw) ? hv[r + (size_t)m * q] : 0.0;
      const int c = (q < hw) ? hi[r + (size_t)m * q] : hi[r];
      const unsigned ent = ((unsigned)dict.id(v) << 8) | (unsigned)c;
      pk[off + (size_t)(q >> 1) * m + r] |= ent << (16 * (q & 1));
    }
}

static int upload(void **dst, const void *src, size_t bytes)
{
  if (!*dst) HFX_HIP(hipMalloc(dst, std::max<size_t>(bytes, 8)));
  HFX_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}

template <int ND, int N>
static int build_packed(hfx_eles *e, FusedData *F, const std::vector<double> &o1v, const std::vector<int> &o1i)
{
  using G = Geo<ND, N>;
  constexpr int NU = G::NU, NFP = G::NFP, WN = G::WN;
  auto hw = [](const Operator &op) { return std::max(op.nnz_max, 1); };
  {
    Dict dict;
    std::vector<unsigned> pk(G::R_END, 0u);
    for (int d = 0; d < ND; d++)
      pack_rows(pk, G::R_O2 + (size_t)d * WN * NU, NU, N, e->opp_2[d].h_val.data(), e->opp_2[d].h_idx.data(), hw(e->opp_2[d]), dict);
    pack_rows(pk, G::R_O3, NU, 2 * ND, e->opp_3.h_val.data(), e->opp_3.h_idx.data(), hw(e->opp_3), dict);
    pack_rows(pk, G::R_O0, NFP, N, e->opp_0.h_val.data(), e->opp_0.h_idx.data(), hw(e->opp_0), dict);
    pack_rows(pk, G::R_O1, NFP, N, o1v.data(), o1i.data(), N, dict);
    HFX_CHECK(dict.vals.size() <= MAX_TAB, "fused path: operators hold %zu distinct values (> %d)", dict.vals.size(), MAX_TAB);
    dict.vals.resize(MAX_TAB, 0.0);
    if (upload((void **)&F->pk_r, pk.data(), sizeof(unsigned) * pk.size())) return 1;
    if (upload((void **)&F->tab_r, dict.vals.data(), sizeof(double) * MAX_TAB)) return 1;
  }
  if (e->viscous_ops)
  {
    Dict dict;
    std::vector<unsigned> pk(G::G_END, 0u);
    for (int d = 0; d < ND; d++)
    {
      pack_rows(pk, G::G_O4 + (size_t)d * WN * NU, NU, N, e->opp_4[d].h_val.data(), e->opp_4[d].h_idx.data(), hw(e->opp_4[d]), dict);
      pack_rows(pk, G::G_O5 + (size_t)d * NU, NU, 2, e->opp_5[d].h_val.data(), e->opp_5[d].h_idx.data(), hw(e->opp_5[d]), dict);
    }
    pack_rows(pk, G::G_O0, NFP, N, e->opp_0.h_val.data(), e->opp_0.h_idx.data(), hw(e->opp_0), dict);
    pack_rows(pk, G::G_O6, NFP, N, e->opp_6.h_val.data(), e->opp_6.h_idx.data(), hw(e->opp_6), dict);
    HFX_CHECK(dict.vals.size() <= MAX_TAB, "fused path: operators hold %zu distinct values (> %d)", dict.vals.size(), MAX_TAB);
    dict.vals.resize(MAX_TAB, 0.0);
    if (upload((void **)&F->pk_g, pk.data(), sizeof(unsigned) * pk.size())) return 1;
    if (upload((void **)&F->tab_g, dict.vals.data(), sizeof(double) * MAX_TAB)) return 1;
  }
  return 0;
}

static int dispatch_build_packed(hfx_eles *e, FusedData *F, int N, const std::vector<double> &o1v, const std::vector<int> &o1i)
{
  if (e->n_dims == 3)
  {
    switch (N)
    {
    case 2: return build_packed<3, 2>(e, F, o1v, o1i);
    case 3: return build_packed<3, 3>(e, F, o1v, o1i);
    case 4: return build_packed<3, 4>(e, F, o1v, o1i);
    case 5: return build_packed<3, 5>(e, F, o1v, o1i);
    case 6: return build_packed<3, 6>(e, F, o1v, o1i);
    }
  }
  else
  {
    switch (N)
    {
    case 2: return build_packed<2, 2>(e, F, o1v, o1i);
    case 3: return build_packed<2, 3>(e, F, o1v, o1i);
    case 4: return build_packed<2, 4>(e, F, o1v, o1i);
    case 5: return build_packed<2, 5>(e, F, o1v, o1i);
    case 6: return build_packed<2, 6>(e, F, o1v, o1i);
    }
  }
  set_error("fused path: no kernel for N = %d, n_dims = %d", N, e->n_dims);
  return 1;
}


// ---------------------------------------------------------------------------------------
// Tensor-product structure of the registered operators (checked bit for bit; anything else keeps
// the dictionary kernels).  With collocated solution / flux bases on a tensor-product element
//   opp_4[d](p, .) = opp_2[d](p, .) : N entries D[i_d(p)][m] on the pencil through p along d
//   opp_5[d](p, .)                  : the pencil's two flux points, c5[d][q][i_d(p)]
//   opp_0(f, .) = opp_6(f, .)       : N entries Lf[d][q][m] on the pencil that ends in f
//   opp_1 merged (f, .)             : the same pencil, L1[d][q][m]
// (definitions: /root/reference/src/eles.cpp:3074-3596).
// ---------------------------------------------------------------------------------------
static int line_of(int nd, int N, int d, int p, int &i_d)
{
  // pencil of point p along d: its index among the N^(nd-1) lines, and the position i_d on it
  const int S = ipow(N, d);
  i_d = (p / S) % N;
  const int base = p - i_d * S;
  if (nd == 2) return d == 0 ? base / N : base;
  if (d == 0) return base / N;                    // (j,k): base = N (j + N k)
  if (d == 1) return (base % N) + N * (base / (N * N)); // (i,k): base = i + N^2 k
  return base;                                    // (i,j)
}
static int base_of_line(int nd, int N, int d, int line)
{
  if (nd == 2) return d == 0 ? N * line : line;
  if (d == 0) return N * line;
  if (d == 1) return (line % N) + N * N * (line / N);
  return line;
}

static bool bits_equal(double a, double b) { return std::memcmp(&a, &b, sizeof a) == 0; }

static int tensor_build(hfx_eles *e, FusedData *F, int N, const std::vector<double> &o1v, const std::vector<int> &o1i,
                        const std::vector<int> &o1d)
{
  F->tensor_ok = false;
  const int nd = e->n_dims, nu = e->n_upts, nfp = e->n_fpts, L = ipow(N, nd - 1);
  auto V = [](const Operator &op, int r, int q) { return op.h_val[r + (size_t)op.m * q]; };
  auto I = [](const Operator &op, int r, int q) { return op.h_idx[r + (size_t)op.m * q]; };
  for (int d = 0; d < nd; d++)
    if (e->opp_2[d].nnz_max != N) return 0;
  if (e->opp_0.nnz_max != N) return 0;
  const bool visc = e->viscous_ops;
  std::vector<double> Dm((size_t)N * N), c5((size_t)nd * 2 * N, 0.0), Lf((size_t)nd * 2 * N), L1((size_t)nd * 2 * N);
  std::vector<int> pf((size_t)nd * L * 2, -1), fdq(nfp, -1), fbase(nfp, -1);
  for (int i = 0; i < N; i++)
    for (int m = 0; m < N; m++) Dm[i * N + m] = V(e->opp_2[0], i, m);
  if (visc)
    for (int d = 0; d < nd; d++)
    {
      if (e->opp_5[d].nnz_max != 2) return 0;
      for (int p = 0; p < nu; p++)
      {
        int i_d;
        if (line_of(nd, N, d, p, i_d) != 0) continue;
        for (int q = 0; q < 2; q++) c5[((size_t)d * 2 + q) * N + i_d] = V(e->opp_5[d], p, q);
      }
    }
  for (int d = 0; d < nd; d++)
  {
    const int S = ipow(N, d);
    for (int p = 0; p < nu; p++)
    {
      int i_d;
      const int line = line_of(nd, N, d, p, i_d), base = p - i_d * S;
      for (int m = 0; m < N; m++)
      {
        if (I(e->opp_2[d], p, m) != base + m * S || !bits_equal(V(e->opp_2[d], p, m), Dm[i_d * N + m])) return 0;
        if (visc && (e->opp_4[d].nnz_max != N || I(e->opp_4[d], p, m) != base + m * S ||
                     !bits_equal(V(e->opp_4[d], p, m), Dm[i_d * N + m])))
          return 0;
      }
      if (visc)
      {
        if (e->opp_5[d].nnz_max != 2) return 0;
        for (int q = 0; q < 2; q++)
        {
          int &slot = pf[((size_t)d * L + line) * 2 + q];
          const int f = I(e->opp_5[d], p, q);
          if (slot < 0) slot = f;
          if (slot != f) return 0;
          if (!bits_equal(V(e->opp_5[d], p, q), c5[((size_t)d * 2 + q) * N + i_d])) return 0;
        }
      }
    }
  }
  if (!visc)
  {
    // inviscid blocks have no opp_5: find the two flux points of a pencil from opp_0's columns
    for (int f = 0; f < nfp; f++)
    {
      const int c0 = I(e->opp_0, f, 0), S = I(e->opp_0, f, 1) - c0;
      int d = -1;
      for (int dd = 0; dd < nd; dd++)
        if (S == ipow(N, dd)) d = dd;
      if (d < 0) return 0;
      int i_d;
      const int line = line_of(nd, N, d, c0, i_d);
      if (i_d != 0) return 0;
      int *slot = &pf[((size_t)d * L + line) * 2];
      if (slot[0] < 0) slot[0] = f;
      else if (slot[1] < 0) slot[1] = f;
      else return 0;
    }
  }
  // every flux point is the end of exactly one pencil
  for (int d = 0; d < nd; d++)
    for (int line = 0; line < L; line++)
      for (int q = 0; q < 2; q++)
      {
        const int f = pf[((size_t)d * L + line) * 2 + q];
        if (f < 0 || f >= nfp || fdq[f] >= 0) return 0;
        fdq[f] = d * 2 + q;
        fbase[f] = base_of_line(nd, N, d, line);
        const int S = ipow(N, d);
        if (o1d[f] != d) return 0;
        for (int m = 0; m < N; m++)
        {
          if (I(e->opp_0, f, m) != fbase[f] + m * S || o1i[f + (size_t)nfp * m] != fbase[f] + m * S) return 0;
          if (visc && (e->opp_6.nnz_max != N || I(e->opp_6, f, m) != fbase[f] + m * S ||
                       !bits_equal(V(e->opp_6, f, m), V(e->opp_0, f, m))))
            return 0;
          double &lf = Lf[((size_t)d * 2 + q) * N + m], &l1 = L1[((size_t)d * 2 + q) * N + m];
          if (line == 0)
          {
            lf = V(e->opp_0, f, m);
            l1 = o1v[f + (size_t)nfp * m];
          }
          if (!bits_equal(lf, V(e->opp_0, f, m)) || !bits_equal(l1, o1v[f + (size_t)nfp * m])) return 0;
        }
      }
  for (int f = 0; f < nfp; f++)
    if (fdq[f] < 0) return 0;
  // merged opp_1 row = tnorm * opp_0 row with tnorm = +-1 (exact): keep the sign only
  for (int dq = 0; dq < nd * 2; dq++)
  {
    double sgn = 0.0;
    for (int m = 0; m < N; m++)
    {
      const double lf = Lf[(size_t)dq * N + m], l1 = L1[(size_t)dq * N + m];
      const double s_m = bits_equal(l1, lf) ? 1.0 : (bits_equal(l1, -lf) ? -1.0 : 0.0);
      if (s_m == 0.0) return 0;
      if (lf != 0.0)
      {
        if (sgn != 0.0 && s_m != sgn) return 0;
        sgn = s_m;
      }
    }
    if (sgn == 0.0) return 0;
    L1[(size_t)dq * N] = sgn;
  }
  // opp_3 (the divergence of the correction functions): row p has one entry per flux point at the ends of the ND pencils
  // through p, and the entry depends on (direction, end, position on the pencil) only.  The flux kernel then applies
  // -opp_3 . norm_tdisf pencil-wise, from the pencil values it holds in registers anyway (folded correction)
  std::vector<double> c3((size_t)nd * 2 * N, 0.0);
  {
    const int w3 = std::max(e->opp_3.nnz_max, 1);
    std::vector<char> have((size_t)nd * 2 * N, 0);
    for (int p = 0; p < nu; p++)
    {
      int seen = 0;
      for (int d = 0; d < nd; d++)
      {
        int i_d;
        const int line = line_of(nd, N, d, p, i_d);
        for (int q = 0; q < 2; q++)
        {
          const int f = pf[((size_t)d * L + line) * 2 + q];
          double v = 0.0;
          for (int c = 0; c < w3; c++)
            if (I(e->opp_3, p, c) == f && V(e->opp_3, p, c) != 0.0)
            {
              v = V(e->opp_3, p, c);
              seen++;
            }
          const size_t slot = ((size_t)d * 2 + q) * N + i_d;
          if (!have[slot])
          {
            c3[slot] = v;
            have[slot] = 1;
          }
          if (!bits_equal(c3[slot], v)) return 0;
        }
      }
      int nonzero = 0;
      for (int c = 0; c < w3; c++) nonzero += V(e->opp_3, p, c) != 0.0;
      if (nonzero != seen) return 0; // an entry outside the pencil ends
    }
  }
  std::vector<double> coef;
  coef.insert(coef.end(), Dm.begin(), Dm.end());
  coef.insert(coef.end(), c5.begin(), c5.end());
  coef.insert(coef.end(), Lf.begin(), Lf.end());
  coef.insert(coef.end(), L1.begin(), L1.end());
  coef.insert(coef.end(), c3.begin(), c3.end());
  std::vector<int> idx;
  idx.insert(idx.end(), pf.begin(), pf.end());
  idx.insert(idx.end(), fdq.begin(), fdq.end());
  idx.insert(idx.end(), fbase.begin(), fbase.end());
  if (F->t_coef) { (void)hipFree(F->t_coef); F->t_coef = nullptr; }
  if (F->t_idx) { (void)hipFree(F->t_idx); F->t_idx = nullptr; }
  if (upload((void **)&F->t_coef, coef.data(), sizeof(double) * coef.size())) return 1;
  if (upload((void **)&F->t_idx, idx.data(), sizeof(int) * idx.size())) return 1;
  F->tensor_ok = true;
  return 0;
}

static int fused_build(hfx_eles *e, hfx_inters *const *faces, int nfb, bool allow_unpaired = false)
{
  HFX_CHECK(e->ele_type == 4 || e->ele_type == 1, "fused path: tensor-product elements only (hexes, quads)");
  const int N = tensor_n(e);
  HFX_CHECK(N >= 2 && N <= 6, "fused path: built for orders 1..5 (n_upts %d, n_fpts %d)", e->n_upts, e->n_fpts);
  const int nd = e->n_dims, nfp = e->n_fpts;
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
  HFX_CHECK(e->n_upts <= 256 && e->n_fpts <= 256, "fused path: column indices must fit 8 bits");

  if (!e->fused) e->fused = new FusedData();
  FusedData *F = e->fused;

  // opp_1 merged over the dimension slabs (row k of opp_1[d] is l_j(fpt_k) * tnorm(d,k): one d per row)
  {
    std::vector<double> mv((size_t)nfp * N, 0.0);
    std::vector<int> mi((size_t)nfp * N, 0), md(nfp, 0);
    for (int r = 0; r < nfp; r++)
    {
      int dsel = -1;
      for (int d = 0; d < nd; d++)
      {
        const Operator &op = e->opp_1[d];
        const int w = std::max(op.nnz_max, 1);
        bool any = false;
        for (int q = 0; q < w; q++) any = any || (op.h_val[r + (size_t)nfp * q] != 0.0);
        if (any)
        {
          HFX_CHECK(dsel < 0, "fused path: row %d of opp_1 has entries in two dimension slabs", r);
          dsel = d;
        }
      }
      if (dsel < 0) dsel = 0;
      const Operator &op = e->opp_1[dsel];
      const int w = std::max(op.nnz_max, 1);
      md[r] = dsel;
      for (int q = 0; q < N; q++)
      {
        mv[r + (size_t)nfp * q] = (q < w) ? op.h_val[r + (size_t)nfp * q] : 0.0;
        mi[r + (size_t)nfp * q] = (q < w) ? op.h_idx[r + (size_t)nfp * q] : op.h_idx[r];
      }
    }
    if (upload((void **)&F->o1m_dim, md.data(), sizeof(int) * md.size())) return 1;
    if (dispatch_build_packed(e, F, N, mv, mi)) return 1;
    if (tensor_build(e, F, N, mv, mi, md)) return 1;
  }

  const long plane_f = (long)e->n_fpts * e->n_eles;
  std::vector<char> paired(plane_f, 0);
  std::vector<unsigned char> meta(plane_f, 0);
  std::vector<double> norm((size_t)plane_f * nd);
  HFX_HIP(hipMemcpy(norm.data(), e->norm_fpts, sizeof(double) * norm.size(), hipMemcpyDeviceToHost));
  for (int b = 0; b < nfb; b++)
  {
    hfx_inters *f = faces[b];
    if (f->is_bdy)
    {
      // boundary points: one-sided kernels; bit2 asks the flux kernel to store the gradient there
      HFX_CHECK(f->left == e, "fused path: boundary block of another element block");
      for (size_t q = 0; q < f->hL.size(); q++)
      {
        paired[f->hL[q]] = 1;
        meta[f->hL[q]] |= 4;
      }
      continue;
    }
    HFX_CHECK(f->left == e && f->right == e, "fused path: face blocks must connect the element block to itself");
    const long np = (long)f->n_inters * f->n_fpts_per_inter;
    for (long q = 0; q < np; q++)
    {
      const int il = f->hL[q], ir = f->hR[q];
      paired[il] = paired[ir] = 1;
      // the consistent switch of src/inters.cpp:568-581 on the LEFT normal (exact zero tests);
      // only the sign decision is stored
      const double n[3] = {norm[il], norm[il + plane_f], nd == 3 ? norm[il + 2 * plane_f] : 0.0};
      double bt = 1.0;
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
    }
  }
  if (!allow_unpaired)
    for (long o = 0; o < plane_f; o++)
      HFX_CHECK(paired[o], "fused path: flux point %ld belongs to no registered face (partition faces need "
                           "hfx_stage_partitioned / hfx_run_steps_partitioned)", o);
  if (upload((void **)&F->meta, meta.data(), plane_f)) return 1;
  if (!F->disu_alt) HFX_HIP(hipMalloc((void **)&F->disu_alt, sizeof(double) * plane_f * e->n_fields));
  F->built = true;
  return 0;
}

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
#pragma unroll
  for (int k = 0; k < NF; k++)
  {
    const double ul = a.disu[il + k * a.plane_f], ur = a.disu[ir + k * a.plane_f];
    const double uc = 0.5 * (ul + ur) - beta * (ul - ur); // src/inters.cpp:637
    a.delta[il + k * a.plane_f] = uc - ul;
    a.delta[ir + k * a.plane_f] = uc - ur;
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
  const double *tdisf_in;        // over-integration: the transformed inviscid flux, already evaluated (NULL: computed here)
  const unsigned char *meta;     // with grad_fpts: only flux points whose bit2 is set are written (NULL: all)
  int simd_roles;    // 1: the waves' parts are dealt by SIMD (split_flux_tensor_kernel)
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
#ifndef HFX_NO_PAIR
#define HFX_NO_PAIR 0 // 1: the unpaired point physics also in the loader-wave kernel (A/B builds)
#endif
#ifndef HFX_FLUX_FMETRICS
#define HFX_FLUX_FMETRICS 0
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
  __device__ __forceinline__ EleOrder(long n_eles, bool want) : ne(n_eles)
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
  typedef __attribute__((address_space(3))) void *lds_vp;
  __attribute__((address_space(3))) char *base = (__attribute__((address_space(3))) char *)lds_dst;
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
template <int ND, int N, int WV, bool BUF, bool OI, bool LW>
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
  // flux-point metrics of the current element (HFX_FLUX_FMETRICS: where they are requested -- 0: at their use in
  // phase B, behind the solution-point block; 1: at the top of phase B; 2: at the top of phase A)
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
  const EleOrder order(ne, a.xcd_order != 0);
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
        if (viscous)
        {
#pragma unroll
          for (int k = 0; k < NF; k++)
            dma16_region(g_delta, base + (U_DW / 2 + k * NFP), L_D, lane, (unsigned)((long)NFP * e + k * plane_f) * 8u);
        }
      };
      // metrics: 16 bytes per lane, partial last wave-instruction masked (an inactive lane writes nothing)
      constexpr int L_JGU = (NQ * NU + 1) / 2, L_DJU = (NU + 1) / 2, L_JGF = (NQ * NFP + 1) / 2, L_DJF = (NFP + 1) / 2; // lanes
      constexpr int MJ_MAX = (cmax(L_JGU, L_JGF) + 63) / 64;
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
        lstamp(kk, 2);
        if (e_next >= 0)
        {
          issue(e_next, (int)((kk + 1) & 1));
          if (viscous) HFX_VMCNT(N_UDV); else HFX_VMCNT(N_UDI); // the metrics of this element have landed
        }
        else
          HFX_VMCNT(0);
        lstamp(kk, 3);
        lds_barrier(); // 2: the compute waves may read the metric slot
        lstamp(kk, 4);
        if (viscous && !HFX_NO_PAIR)
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
#if HFX_FLUX_FMETRICS == 2
    fetch_fmetrics();
#endif
    // over-integration: the de-aliased inviscid flux of this point is requested here, a phase ahead of its use
    double td[OI ? NG : 1];
    if (OI && is_u)
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
#pragma unroll
        for (int m = 0; m < N; m++) xa[r][m] = ldsv(su_p + m * sr[r]);
        da[r] = ldsv(sd + it_fa[r]);
        db[r] = ldsv(sd + it_fb[r]);
      }
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
    double u[NF], uf[NF];
    if (is_u || LW) // (LW: every lane, on clamped point numbers -- the paired physics of phase B)
    {
#pragma unroll
      for (int k = 0; k < NF; k++) u[k] = ldsv(&su[k * NUS + tu]);
    }
    if (viscous && (is_f || LW))
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
#if HFX_FLUX_FMETRICS == 1
    fetch_fmetrics();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- B: gradient and projected viscous flux at the flux points; fluxes at the solution points
    if (LW && viscous && !HFX_NO_PAIR)
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
      if (is_f)
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
#pragma unroll
        for (int k = 0; k < NF; k++)
#pragma unroll
          for (int l = 0; l < ND; l++)
          {
            double s = OI ? td[k + NF * l] : 0.0;
#pragma unroll
            for (int m = 0; m < ND; m++) s += jg2[0][l + ND * m] * ft[k + NF * m];
            st[(k + NF * l) * NU + tu] = s;
          }
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
        // over-integration: the de-aliased inviscid flux was evaluated at the cubature points and projected back
#pragma unroll
        for (int q = 0; q < NG; q++) st[q * NU + tu] = td[q];
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
#if HFX_FLUX_FMETRICS == 0
      // flux-point metrics are fetched here, after the solution-point block has released its registers
      fetch_fmetrics();
#endif
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
        g_div.st(eu + k * plane_u, lu, s);
      }
    }
    stamp(9);
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
  riemann_flux_t<ND, RS, true>(a.P, ul, ur, n, fn);
  if (a.P.viscous)
  {
    const double beta = (a.meta[il] & 2) ? -a.P.ldg_beta : a.P.ldg_beta;
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      // (1/2+b) F_L.n + (1/2-b) F_R.n - tau (u_R - u_L), n the left normal = -(right normal)
      double fv = (0.5 + beta) * a.fn[il + k * a.plane_f] - (0.5 - beta) * a.fn[ir + k * a.plane_f];
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
  constexpr int NF = G::NF, NU = G::NU, NFP = G::NFP, TB = SGeo<ND, N>::TB;
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
  const EleOrder order(ne, a.xcd_order != 0);
  for (long kk = 0, e = order.at(0); e >= 0; kk++, e = order.at(kk))
  {
    const long eu = (long)NU * e, ef = (long)NFP * e;
    double u[NF], dvin[NF], u1v[NF];
    if (is_f)
    {
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        // folded: the flux kernel has subtracted opp_3 . norm_tdisf already (a.folded, the sum-factorised form)
        double v = g_tc.ld(ef + k * plane_f, lf);
        if (!a.folded) v += -1.0 * g_nt.ld(ef + k * plane_f, lf);
        sc[k][tf] = v;
      }
    }
    const double dt = a.dt_local_on ? a.dt_local[e] : a.dt;
    const double dj = g_dj.ld(eu, lu);
#pragma unroll
    for (int k = 0; k < NF; k++)
    {
      const long ok = eu + k * plane_u;
      u[k] = g_u0.ld(ok, lu);
      dvin[k] = g_div.ld(ok, lu);
      u1v[k] = a.need_u1 ? g_u1.ld(ok, lu) : 0.0;
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
#pragma unroll
      for (int k = 0; k < NF; k++)
      {
        const double dv = dva[k];
        const long ok = eu + k * plane_u;
        if (dv != dv) atomicMin(a.nan_flag, (unsigned long long)(ok + tu));
        if (a.write_div) g_div.st(ok, lu, dv);
        const double s = a.src ? g_src.ld(ok, lu) : 0.0;
        const double dd = dv / dj;
        double un = u[k];
        if (a.adv_type == 0)
          un -= dt * (dd - s);
        else if (a.adv_type == 1)
        {
          if (a.in_step == 0) g_u1.st(ok, lu, un);
          if (a.in_step < 3)
            un -= dt / 3.0 * (dd - s);
          else
          {
            const double rhs = -dd + s;
            un = 3.0 / 4.0 * un + 1.0 / 4.0 * u1v[k] + dt / 4.0 * rhs;
          }
        }
        else if (a.adv_type == 2)
        {
          if (a.in_step == 0) g_u1.st(ok, lu, un);
          if (a.in_step < 2 || a.in_step == 3)
            un -= dt / 2.0 * (dd - s);
          else if (a.in_step == 2)
          {
            const double rhs = -dd + s;
            un = 1.0 / 3.0 * un + 2.0 / 3.0 * u1v[k] + dt / 6.0 * rhs;
          }
        }
        else
        {
          const double rhs = -dd + s;
          const double r1 = a.rk_a * u1v[k] + dt * rhs;
          g_u1.st(ok, lu, r1);
          un += a.rk_b * r1;
        }
        g_u0.st(ok, lu, un);
        su[k][tu] = un;
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

// launch of the loader-wave form, instantiated only for element sizes it fits (loader_wave_fits)
template <int ND, int N, bool OI, bool FITS>
struct LoaderWaveLaunch
{
  static void go(int, hipStream_t, const Split2Args &, const double *, const int *) {}
};
template <int ND, int N, bool OI>
struct LoaderWaveLaunch<ND, N, OI, true>
{
  static void go(int grid, hipStream_t st, const Split2Args &e2, const double *coef, const int *idx)
  {
    hipLaunchKernelGGL((split_flux_tensor_kernel<ND, N, 2, true, OI, true>), dim3(grid), dim3(SGeo<ND, N>::TB + 64), 0, st, e2, coef, idx);
  }
};

template <int ND, int N>
static int launch_split_stage(hfx_eles *e, hfx_inters *const *faces, int nfb, SplitEleArgs &ea, int which, int variant)
{
  FusedData *F = e->fused;
  hipStream_t st = e->ctx->stream;
  const Phys P = e->ctx->phys();
  const long plane_f = (long)e->n_fpts * e->n_eles;
  const hfx_ctx::Options &opt = e->ctx->opt;
  const int grid_per_cu = std::max(1, opt.split_grid_per_cu);
  const int grid = (int)std::min<long>(e->n_eles, (long)e->ctx->n_cu * grid_per_cu);
  auto face_args = [&](hfx_inters *f) {
    SplitFaceArgs a{};
    a.npairs = (long)f->n_inters * f->n_fpts_per_inter;
    a.L = f->L; a.R = f->R; a.meta = F->meta; a.plane_f = plane_f;
    a.disu = e->arr[HFX_DISU_FPTS]; a.grad = e->arr[HFX_GRAD_DISU_FPTS]; a.fnorm = e->norm_fpts; a.tdA = e->tdA_fpts;
    a.delta = e->arr[HFX_DELTA_DISU_FPTS]; a.tconf = e->arr[HFX_NORM_TCONF_FPTS];
    a.sgsf = (e->les_ready && variant == 2) ? e->arr[HFX_SGSF_FPTS] : nullptr;
    a.jac_fpts = e->Jacobian_fpts; a.detjac_fpts = e->detjac_fpts;
    a.P = P;
    return a;
  };
  Split2Args e2{};
  if (variant == 3)
  {
    if (opt.flux_stamps && !F->stamps)
    {
      HFX_HIP(hipMalloc((void **)&F->stamps, sizeof(long long) * 64));
      HFX_HIP(hipMemset(F->stamps, 0, sizeof(long long) * 64));
    }
    if (!F->fn_fpts) HFX_HIP(hipMalloc((void **)&F->fn_fpts, sizeof(double) * (size_t)plane_f * e->n_fields));
    e2.n_eles = ea.n_eles;
    e2.xcd_order = opt.xcd_order ? 1 : 0;
    e2.pk_g = F->pk_g; e2.pk_r = F->pk_r; e2.tab_g = F->tab_g; e2.tab_r = F->tab_r; e2.o1m_dim = F->o1m_dim;
    e2.detjac_upts = ea.detjac_upts; e2.JGinv_upts = ea.JGinv_upts; e2.detjac_fpts = ea.detjac_fpts;
    e2.JGinv_fpts = ea.JGinv_fpts; e2.norm_fpts = e->norm_fpts;
    e2.u0 = ea.u0; e2.u1 = ea.u1; e2.delta = ea.delta; e2.tconf = ea.tconf;
    e2.fn_fpts = F->fn_fpts; e2.ntd_fpts = e->arr[HFX_NORM_TDISF_FPTS]; e2.div = ea.div_out;
    e2.folded = (F->tensor_ok && !opt.dictionary_rows) ? 1 : 0;
    e2.disu_next = ea.disu_next;
    bool any_bdy = false;
    for (int b = 0; b < nfb; b++) any_bdy = any_bdy || faces[b]->is_bdy;
    e2.grad_upts = nullptr;
    e2.grad_fpts = (any_bdy && P.viscous) ? e->arr[HFX_GRAD_DISU_FPTS] : nullptr; // boundary points only
    e2.meta = F->meta;
    e2.stamps = F->stamps;
    e2.stamp_it = std::max(2, opt.flux_stamps);
    e2.simd_roles = opt.simd_roles ? 1 : 0;
    e2.o3v = e->opp_3.ell_val; e2.o3i = e->opp_3.ell_idx; e2.o3w = std::max(e->opp_3.nnz_max, 1);
    e2.o0v = e->opp_0.ell_val; e2.o0i = e->opp_0.ell_idx; e2.o0w = std::max(e->opp_0.nnz_max, 1);
    e2.src = ea.src; e2.dt_local = ea.dt_local; e2.nan_flag = ea.nan_flag; e2.P = ea.P;
    e2.adv_type = ea.adv_type; e2.in_step = ea.in_step; e2.dt_local_on = ea.dt_local_on; e2.write_div = ea.write_div;
    e2.need_u1 = ea.need_u1; e2.dt = ea.dt; e2.rk_a = ea.rk_a; e2.rk_b = ea.rk_b;
  }
  if (P.viscous && (which == 0 || which == 1))
  {
    for (int b = 0; b < nfb; b++)
    {
      if (faces[b]->is_bdy)
      {
        // ghost state -> inviscid common flux and LDG common solution of the boundary points
        if (hfx_bdy_launch_internal(faces[b], 0, 1)) return 1;
        continue;
      }
      const SplitFaceArgs a = face_args(faces[b]);
      if (a.npairs == 0) continue;
      hipLaunchKernelGGL((face_delta_kernel<ND>), dim3((unsigned)((a.npairs + 255) / 256)), dim3(256), 0, st, a);
    }
  }
  if (variant == 3)
  {
    if (which == 0 || which == 2)
    {
      e2.tdisf_in = nullptr;
      if (e->over_int_ready)
      {
        // polynomial de-aliasing (src/solver.cpp:82-91): tdisf_upts = over_int_filter . F(opp_over_int_cubpts . u)
        if (hfx_eles_evaluate_invFlux_over_int(e)) return 1;
        e2.tdisf_in = e->arr[HFX_TDISF_UPTS];
      }
      const bool dict_only = opt.dictionary_rows != 0;
      const int waves = opt.flux_waves;
      // buffer-descriptor addressing needs 32-bit byte offsets into the largest array the kernel touches
      const bool nobuf = !opt.buffer_addressing;
      // (the largest array the launch really touches: the metric tensors at the flux points, and the n_fields * n_dims
      // component arrays only when they are in use -- gradients at boundary points, the de-aliased flux)
      // -- over BOTH point sets: quads with N >= 5 have more solution points than flux points
      const double plane_most = (double)std::max<long>(plane_f, (long)e->n_upts * e->n_eles);
      double most = plane_most * std::max(e->n_dims * e->n_dims, e->n_fields);
      if (e2.grad_fpts || e2.grad_upts) most = std::max(most, plane_most * e->n_fields * e->n_dims);
      if (e->over_int_ready) most = std::max(most, (double)e->n_upts * e->n_eles * e->n_fields * e->n_dims);
      const bool buf = !nobuf && most * 8.0 < 4294967296.0;
      const bool oi = e2.tdisf_in != nullptr;
#define HFX_FLUX_LAUNCH(WV_, BUF_, OI_, LW_)                                                                                  \
  hipLaunchKernelGGL((split_flux_tensor_kernel<ND, N, WV_, BUF_, OI_, LW_>), dim3(grid), dim3(SGeo<ND, N>::TB), 0, st, e2,   \
                     F->t_coef, F->t_idx)
      const bool no_lw = !opt.loader_wave;
      constexpr bool lw_fits = loader_wave_fits<ND, N>();
      const bool lw = lw_fits && buf && !no_lw && waves == 2;
      bool launched = false;
      if (F->tensor_ok && !dict_only && lw)
      {
        if (oi)
          LoaderWaveLaunch<ND, N, true, lw_fits>::go(grid, st, e2, F->t_coef, F->t_idx);
        else
          LoaderWaveLaunch<ND, N, false, lw_fits>::go(grid, st, e2, F->t_coef, F->t_idx);
        launched = true;
      }
      if (launched)
        ;
      else if (F->tensor_ok && !dict_only && oi && buf)
        HFX_FLUX_LAUNCH(2, true, true, false);
      else if (F->tensor_ok && !dict_only && oi)
        HFX_FLUX_LAUNCH(2, false, true, false);
      else if (F->tensor_ok && !dict_only && waves == 2 && buf)
        HFX_FLUX_LAUNCH(2, true, false, false);
      else if (F->tensor_ok && !dict_only && buf)
        HFX_FLUX_LAUNCH(3, true, false, false);
      else if (F->tensor_ok && !dict_only)
        HFX_FLUX_LAUNCH(2, false, false, false);
#undef HFX_FLUX_LAUNCH
      else
        hipLaunchKernelGGL((split_flux_kernel<ND, N>), dim3(grid), dim3(SGeo<ND, N>::TB), 0, st, e2);
    }
  }
  else if (P.viscous && (which == 0 || which == 2))
  {
    ea.pk = F->pk_g;
    ea.tab = F->tab_g;
    hipLaunchKernelGGL((split_gradient_kernel<ND, N>), dim3(grid), dim3(SGeo<ND, N>::TB), 0, st, ea);
    if (e->les_ready)
    {
      // LES (eddy-viscosity closures): SGS flux at the solution points from the corrected gradient, its extrapolation to
      // the flux points (src/solver.cpp:162-167); the face kernel adds it to each side, the residual kernel to the total
      if (hfx_les_sgsf_upts_internal(e)) return 1;
      if (hfx_les_extrapolate_reference_internal(e)) return 1; // the back-transform happens in the face kernel
    }
  }
  if (which == 0 || which == 3)
  {
    for (int b = 0; b < nfb; b++)
    {
      if (faces[b]->is_bdy)
      {
        if (hfx_bdy_launch_internal(faces[b], P.viscous ? 1 : 0, 1)) return 1;
        continue;
      }
      const SplitFaceArgs a = face_args(faces[b]);
      if (a.npairs == 0) continue;
      const unsigned nb = (unsigned)((a.npairs + 255) / 256);
      if (variant == 3)
      {
        Split2FaceArgs a2{};
        a2.npairs = a.npairs; a2.L = a.L; a2.R = a.R; a2.meta = a.meta; a2.plane_f = plane_f;
        a2.disu = a.disu; a2.fn = F->fn_fpts; a2.fnorm = a.fnorm; a2.tdA = a.tdA; a2.tconf = a.tconf; a2.P = P;
        if (P.riemann == 0)
          hipLaunchKernelGGL((face_flux2_kernel<ND, 0>), dim3(nb), dim3(256), 0, st, a2);
        else if (P.riemann == 2)
          hipLaunchKernelGGL((face_flux2_kernel<ND, 2>), dim3(nb), dim3(256), 0, st, a2);
        else
          hipLaunchKernelGGL((face_flux2_kernel<ND, 3>), dim3(nb), dim3(256), 0, st, a2);
      }
      else if (P.riemann == 0)
        hipLaunchKernelGGL((face_flux_kernel<ND, 0>), dim3(nb), dim3(256), 0, st, a);
      else if (P.riemann == 2)
        hipLaunchKernelGGL((face_flux_kernel<ND, 2>), dim3(nb), dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((face_flux_kernel<ND, 3>), dim3(nb), dim3(256), 0, st, a);
    }
  }
  if (which == 0 || which == 4)
  {
    if (variant == 3)
    {
      // buffer-descriptor addressing needs 32-bit byte offsets
      const bool nobuf = !opt.buffer_addressing;
      const bool small = (double)std::max<long>(plane_f, (long)e->n_upts * e->n_eles) * e->n_fields * 8.0 < 4294967296.0;
      if (small && !nobuf)
        hipLaunchKernelGGL((split_update_kernel<ND, N, true>), dim3(grid), dim3(SGeo<ND, N>::TB), 0, st, e2);
      else
        hipLaunchKernelGGL((split_update_kernel<ND, N, false>), dim3(grid), dim3(SGeo<ND, N>::TB), 0, st, e2);
    }
    else
    {
      ea.pk = F->pk_r;
      ea.tab = F->tab_r;
      hipLaunchKernelGGL((split_residual_kernel<ND, N>), dim3(grid), dim3(SGeo<ND, N>::TB), 0, st, ea);
    }
  }
  HFX_HIP(hipGetLastError());
  return 0;
}

static int split_stage(hfx_eles *e, hfx_inters *const *faces, int nfb, int in_step, bool last_stage, int which = 0,
                       int variant = 2)
{
  FusedData *F = e->fused;
  const hfx_params &p = e->ctx->params;
  SplitEleArgs a{};
  a.n_eles = e->n_eles;
  a.o1m_dim = F->o1m_dim;
  a.detjac_upts = e->detjac_upts; a.JGinv_upts = e->JGinv_upts;
  a.detjac_fpts = e->detjac_fpts; a.JGinv_fpts = e->JGinv_fpts;
  a.u0 = e->arr[HFX_DISU_UPTS0]; a.u1 = e->arr[HFX_DISU_UPTS1];
  a.delta = e->arr[HFX_DELTA_DISU_FPTS]; a.tconf = e->arr[HFX_NORM_TCONF_FPTS];
  a.disu_next = F->disu_alt;
  a.grad_upts = e->arr[HFX_GRAD_DISU_UPTS]; a.grad_fpts = e->arr[HFX_GRAD_DISU_FPTS];
  a.div_out = e->arr[HFX_DIV_TCONF_UPTS];
  a.sgsf_upts = (e->les_ready && variant == 2) ? e->arr[HFX_SGSF_UPTS] : nullptr;
  a.src = e->src_nonzero ? e->arr[HFX_SRC_UPTS] : nullptr;
  a.dt_local = e->arr[HFX_DT_LOCAL];
  a.nan_flag = e->nan_flag;
  a.P = e->ctx->phys();
  a.adv_type = p.adv_type; a.in_step = in_step; a.dt_local_on = p.dt_type == 2; a.dt = p.dt;
  a.rk_a = (p.adv_type >= 3) ? p.RK_a[in_step] : 0.0;
  a.rk_b = (p.adv_type >= 3) ? p.RK_b[in_step] : 0.0;
  a.need_u1 = (p.adv_type >= 3) || (p.adv_type == 1 && in_step == 3) || (p.adv_type == 2 && in_step == 2);
  a.write_div = last_stage ? 1 : 0;
  const int N = tensor_n(e);
  int rc = 1;
#define HFX_SPLIT_CASE(ND_, N_) \
  if (e->n_dims == ND_ && N == N_) rc = launch_split_stage<ND_, N_>(e, faces, nfb, a, which, variant);
  HFX_SPLIT_CASE(3, 2) HFX_SPLIT_CASE(3, 3) HFX_SPLIT_CASE(3, 4) HFX_SPLIT_CASE(3, 5) HFX_SPLIT_CASE(3, 6)
  HFX_SPLIT_CASE(2, 2) HFX_SPLIT_CASE(2, 3) HFX_SPLIT_CASE(2, 4) HFX_SPLIT_CASE(2, 5) HFX_SPLIT_CASE(2, 6)
#undef HFX_SPLIT_CASE
  if (rc) return 1;
  if (which == 0 || which == 4) std::swap(e->arr[HFX_DISU_FPTS], e->fused->disu_alt);
  return 0;
}

// shock capturing inside a split-path stage: disu_fpts must follow the filtered state.  The sum-factorised kernel
// rewrites the flux points of the elements it filters; the dense form is followed by a full extrapolate_solution.
static int shock_capture_keep_fpts(hfx_eles *e)
{
  if (tensor_shock_available(e) && e->ctx->contract_mode != HFX_CONTRACT_DENSE) return tensor_shock_launch(e, true);
  if (hfx_eles_shock_capture(e)) return 1;
  return hfx_eles_extrapolate_solution(e);
}

int split_run_steps(hfx_eles *e, hfx_inters *const *faces, int nfb, int n_steps, int variant)
{
  HFX_CHECK(e->n_eles > 0, "fused path: empty element block");
  // an LES closure reads the corrected gradients, which variant 3 keeps in registers: such a block runs variant 2
  if (e->les_ready && variant == 3) variant = 2;
  HFX_CHECK(!e->over_int_ready || variant == 3, "the split variant that keeps the gradients (fused 2, which LES selects) has no over-integration");
  if (!e->fused || !e->fused->built)
    if (fused_build(e, faces, nfb)) return 1;
  if (n_steps <= 0) return 0;
  const int adv = e->ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  if (hfx_eles_extrapolate_solution(e)) return 1;
  for (int s = 0; s < n_steps; s++)
  {
    if (calc_time_step(e, nullptr)) return 1; /* src/HiFiLES.cpp:198 */
    for (int rk = 0; rk < nst; rk++)
    {
      if (rk == 0 && e->les_ready && e->les.sgs_model >= 2)
      {
        // first stage of a step: filtered solution / Leonard terms (src/solver.cpp:55-62); the SVV closure replaces the
        // state, whose flux-point values the previous stage's update kernel has already written: redo them
        if (hfx_eles_calc_sgs_terms(e)) return 1;
        if (e->les.sgs_model == 3 && hfx_eles_extrapolate_solution(e)) return 1;
      }
      if (split_stage(e, faces, nfb, rk, rk == nst - 1, 0, variant)) return 1;
      if (e->shock_ready)
      {
        // the filter changes disu_upts(0) after the stage: redo the flux-point solution of the new state
        if (shock_capture_keep_fpts(e)) return 1;
      }
    }
    advance_ramp_counters(faces, nfb); /* src/HiFiLES.cpp:224-225 */
  }
  return 0;
}

int split_time_kernels(hfx_eles *e, hfx_inters *const *faces, int nfb, int reps, double *ms, char *names, int names_len,
                       int variant)
{
  if (!e->fused || !e->fused->built)
    if (fused_build(e, faces, nfb)) return 1;
  const int adv = e->ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  hipStream_t st = e->ctx->stream;
  hipEvent_t ev[5];
  for (auto &x : ev) HFX_HIP(hipEventCreate(&x));
  if (hfx_eles_extrapolate_solution(e)) return 1;
  double acc[4] = {0, 0, 0, 0};
  for (int r = 0; r < reps; r++)
  {
    const int rk = r % nst;
    for (int w = 1; w <= 4; w++)
    {
      HFX_HIP(hipEventRecord(ev[w - 1], st));
      if (split_stage(e, faces, nfb, rk, rk == nst - 1, w, variant)) return 1;
    }
    HFX_HIP(hipEventRecord(ev[4], st));
    HFX_HIP(hipStreamSynchronize(st));
    for (int w = 0; w < 4; w++)
    {
      float t = 0;
      HFX_HIP(hipEventElapsedTime(&t, ev[w], ev[w + 1]));
      acc[w] += t;
    }
  }
  for (auto &x : ev) (void)hipEventDestroy(x);
  for (int i = 0; i < 8; i++) ms[i] = 0.0;
  for (int w = 0; w < 4; w++) ms[w] = acc[w] / reps;
  if (e->fused->stamps)
  {
    long long h[64];
    HFX_HIP(hipMemcpy(h, e->fused->stamps, sizeof h, hipMemcpyDeviceToHost));
    for (int w = 0; w < 3; w++)
    {
      fprintf(stderr, "flux kernel wave %d cycles: ", w);
      for (int q = 1; q <= 9; q++) fprintf(stderr, "%s%lld", q > 1 ? " " : "", h[w * 16 + q] - h[w * 16 + q - 1]);
      fprintf(stderr, "   (fill | bar1 | A | bar2 | B | bar3 | C | bar4 | D)  total %lld\n", h[w * 16 + 9] - h[w * 16]);
    }
    fprintf(stderr, "loader wave cycles: ");
    for (int q = 1; q <= 7; q++) fprintf(stderr, "%s%lld", q > 1 ? " " : "", h[3 * 16 + q] - h[3 * 16 + q - 1]);
    fprintf(stderr, "   (wait state | bar1 | issue state, wait metrics | bar2 | bar3 | issue metrics | bar4)  total %lld\n", h[3 * 16 + 7] - h[3 * 16]);
  }
  const bool tensor = e->fused->tensor_ok && !e->ctx->opt.dictionary_rows;
  snprintf(names, names_len, "%s",
           variant == 3 ? (tensor ? "face_delta_kernel,split_flux_tensor_kernel,face_flux2_kernel,split_update_kernel"
                                  : "face_delta_kernel,split_flux_kernel,face_flux2_kernel,split_update_kernel")
                        : "face_delta_kernel,split_gradient_kernel,face_flux_kernel,split_residual_kernel");
  return 0;
}

void split_kernel_bytes(const hfx_eles *e, double *bytes, int variant)
{
  // ALGORITHMIC HBM bytes per launch (doubles listed per element)
  const double nu = e->n_upts, nfp = e->n_fpts, nf = e->n_fields, nd = e->n_dims, ne = e->n_eles;
  for (int i = 0; i < 8; i++) bytes[i] = 0.0;
  bytes[0] = ne * (8.0 * (2 * nfp * nf) + 4.0 * nfp + nfp * 0.5);                                   // disu r, delta w, index + meta
  bytes[1] = ne * 8.0 * (nu * nf + nfp * nf + nu * (nd * nd + 1) + nfp * (nd * nd + 1) + nfp * nf * nd); // + grad_fpts w
  bytes[2] = ne * (8.0 * (nfp * nf + nfp * nf * nd + 0.5 * nfp * nd + nfp + nfp * nf) + 4.0 * nfp);   // disu, grad, normal(left), tdA r; tconf w
  bytes[3] = ne * 8.0 * (nu * nf + nu * (nd * nd + 1) + nfp * nf + 3 * nu * nf + nfp * nf);           // u, metrics, tconf, u1 r; u0,u1,disu w
  if (variant == 3)
  {
    // u, delta, volume + flux-point metrics, own normals r ; div, norm_tdisf, Fn w
    // (norm_tdisf: the dictionary-row form only; the sum-factorised kernel folds opp_3 . norm_tdisf into div)
    const double ntd = (e->fused && e->fused->tensor_ok && !e->ctx->opt.dictionary_rows) ? 0.0 : nfp * nf;
    bytes[1] = ne * 8.0 * (nu * nf + nfp * nf + nu * (nd * nd + 1) + nfp * (nd * nd + 1) + nfp * nd + nu * nf + nfp * nf + ntd);
    bytes[2] = ne * (8.0 * (nfp * nf + nfp * nf + 0.5 * nfp * nd + nfp + nfp * nf) + 4.0 * nfp); // disu, Fn, normal(left), tdA r; tconf w
    bytes[3] = ne * 8.0 * (3 * nu * nf + nu + nfp * nf + ntd + 2 * nu * nf + nfp * nf);          // u0,u1,div,detjac,tconf(,ntd) r; u0,u1,disu w
  }
}


// ---------------------------------------------------------------------------------------
// split path on a partitioned block: interior pairs by the pairwise kernels, partition faces by the
// one-sided kernels of kernels_mpi.hpp; the caller exchanges the buffers between the phases
// ---------------------------------------------------------------------------------------
template <int ND>
static int mpi_launch(hfx_eles *e, hfx_inters *f, int what)
{
  if (f->n_inters == 0) return 0;
  MpiArgs a{};
  a.npairs = (long)f->n_inters * f->n_fpts_per_inter;
  a.nfpi = f->n_fpts_per_inter;
  a.L = f->L; a.Rlut = f->R;
  a.plane = (long)e->n_fpts * e->n_eles;
  a.disu = e->arr[HFX_DISU_FPTS]; a.grad = e->arr[HFX_GRAD_DISU_FPTS];
  a.norm = e->norm_fpts; a.tdA = e->tdA_fpts;
  a.tconf = e->arr[HFX_NORM_TCONF_FPTS];
  a.delta = (what == 3) ? nullptr : e->arr[HFX_DELTA_DISU_FPTS];
  a.out_disu = f->out_disu; a.out_grad = f->out_grad; a.in_disu = f->in_disu; a.in_grad = f->in_grad;
  a.fn = e->fused ? e->fused->fn_fpts : nullptr;
  a.P = e->ctx->phys();
  if (e->les_ready)
  {
    // the split path (variant 2) keeps sgsf_fpts in reference space: the partition-face kernels take it to physical space
    if (hfx_mpi_sgsf_buffers_internal(f)) return 1;
    a.sgsf = e->arr[HFX_SGSF_FPTS]; a.jac_fpts = e->Jacobian_fpts; a.detjac_fpts = e->detjac_fpts;
    a.out_sgsf = f->out_sgsf; a.in_sgsf = f->in_sgsf; a.sgs_ref = 1;
  }
  const dim3 g((unsigned)((a.npairs + 255) / 256)), b(256);
  hipStream_t st = e->ctx->stream;
  switch (what)
  {
  case 0: hipLaunchKernelGGL(mpi_pack_disu_kernel<ND>, g, b, 0, st, a); break;
  case 1: hipLaunchKernelGGL(mpi_delta_kernel<ND>, g, b, 0, st, a); break;
  case 2: hipLaunchKernelGGL(mpi_pack_grad_kernel<ND>, g, b, 0, st, a); break;
  case 3: hipLaunchKernelGGL((mpi_common_invflux_kernel<ND, true>), g, b, 0, st, a); break;
  case 4: hipLaunchKernelGGL((mpi_common_viscflux_kernel<ND, true>), g, b, 0, st, a); break;
  case 5: hipLaunchKernelGGL(mpi_pack_fn_kernel<ND>, g, b, 0, st, a); break;
  case 6: hipLaunchKernelGGL(mpi_common_flux2_kernel<ND>, g, b, 0, st, a); break;
  case 7: hipLaunchKernelGGL(mpi_pack_sgsf_kernel<ND>, g, b, 0, st, a); break;
  }
  HFX_HIP(hipGetLastError());
  return 0;
}

int split_variant(const hfx_eles *e) { return (e->ctx->fused_mode == 2 || e->les_ready) ? 2 : 3; }

int split_stage_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                            int phase, int in_step, int first)
{
  HFX_CHECK(e->n_eles > 0, "fused path: empty element block");
  for (int b = 0; b < n_mpi; b++) HFX_CHECK(mpi_faces[b]->is_mpi && mpi_faces[b]->left == e, "bad partition-face block");
  if (!e->fused || !e->fused->built)
    if (fused_build(e, int_faces, n_int, true)) return 1;
  const hfx_params &p = e->ctx->params;
  const int nst = (p.adv_type == 0) ? 1 : (p.adv_type <= 2) ? 4 : (p.adv_type == 3) ? 5 : 14;
  HFX_CHECK(in_step >= 0 && in_step < nst, "hfx_stage_partitioned: stage %d out of range", in_step);
  auto mpi_all = [&](int what) -> int {
    for (int b = 0; b < n_mpi; b++)
      if ((e->n_dims == 2 ? mpi_launch<2>(e, mpi_faces[b], what) : mpi_launch<3>(e, mpi_faces[b], what))) return 1;
    return 0;
  };
  const bool last = in_step == nst - 1;
  const int variant = split_variant(e); // 3: fluxes in the gradient kernel, Fn on the wire; 2 with an LES closure
  HFX_CHECK(!e->over_int_ready || variant == 3, "the split variant that keeps the gradients (fused 2, which LES selects) has no over-integration");
  switch (phase)
  {
  case 0:
    if (first && hfx_eles_extrapolate_solution(e)) return 1;
    return first ? mpi_all(0) : 0;
  case 1:
    if (!p.viscous) return 0;
    if (in_step == 0 && e->les_ready && e->les.sgs_model >= 2)
    {
      HFX_CHECK(e->les.sgs_model != 3, "hfx_stage_partitioned: the SVV closure filters the state at the first stage, after its flux-point "
                                       "values have left for the neighbours: run it per method");
      if (hfx_eles_calc_sgs_terms(e)) return 1; // Leonard terms of this step (src/solver.cpp:55-62)
    }
    return split_stage(e, int_faces, n_int, in_step, false, 1, variant); // interior LDG common solution
  case 2:
    if (variant == 3)
    {
      if (p.viscous && mpi_all(1)) return 1;
      if (split_stage(e, int_faces, n_int, in_step, false, 2, 3)) return 1; // gradient + fluxes; allocates fn_fpts
      return p.viscous ? mpi_all(5) : 0;
    }
    if (!p.viscous) return 0;
    if (mpi_all(1)) return 1;
    if (split_stage(e, int_faces, n_int, in_step, false, 2, 2)) return 1; // corrected gradients (+ the SGS chain with LES)
    if (mpi_all(2)) return 1;
    return e->les_ready ? mpi_all(7) : 0; // third message: the physical SGS flux (src/solver.cpp:168-178)
  case 3:
    if (split_stage(e, int_faces, n_int, in_step, last, 3, variant)) return 1; // interior common fluxes
    return variant == 3 ? 0 : mpi_all(3);
  case 4:
    if (variant == 3)
    {
      if (mpi_all(6)) return 1;
    }
    else if (p.viscous && mpi_all(4))
      return 1;
    if (split_stage(e, int_faces, n_int, in_step, last, 4, variant)) return 1; // residual, RK, new disu_fpts (swaps)
    if (e->shock_ready)
    {
      // src/HiFiLES.cpp:214-216: the filter changes disu_upts(0) after the stage -- redo the flux-point solution
      if (shock_capture_keep_fpts(e)) return 1;
    }
    return mpi_all(0);
  default:
    HFX_CHECK(false, "hfx_stage_partitioned: phase %d out of range", phase);
  }
  return 0;
}

} // namespace hfx
