// fused_hex.hip -- the split fused stage for tensor-product elements (hexes, quads): table setup, launchers, RK loops.
// Kernels: split_common.hpp (shared helpers), split2_kernels.hpp (variant 2), split3_kernels.hpp (variant 3, the default),
// split_partitioned.hpp (the five phases of a partitioned block).
//
// One RK stage = the 17 calls of CalcResidual + AdvanceSolution
// (/root/reference/src/solver.cpp:50-223, src/HiFiLES.cpp:201-217).  Executed call by
// call they stream ~64 000 doubles per P4 hex through HBM.  Here the stage is cut at the two
// places where data must cross elements (the LDG common solution, the common fluxes) and everything element-local
// between two cuts is one kernel: four launches per stage, three when the flux kernel forms the LDG corrections itself
// (DESIGN.md 3.2).  Two variants:
//   fused = 2: keeps the reference's arrays (grad_disu_upts / grad_disu_fpts in HBM); carries the LES closure
//   fused = 3: the kernel that has the corrected gradient in registers goes straight on to the fluxes; the default
// plus the same stage cut into five phases for a partitioned block (hfx_stage_partitioned).
//
// Operator rows of the element kernels are either sum-factorised (1-D matrices through scalar registers, when the
// registered operators are bit-exactly tensor products) or DICTIONARY-COMPRESSED sparse rows: a tensor-product
// operator has only a handful of distinct values, so a row entry is 16 bits (value id, column), two per 32-bit
// register, the values in a 2 kB LDS table.  The arithmetic is unchanged: the same non-zeros, multiplied in the same
// ascending-column order as the reference dgemm (src/funcs.cpp:110-117).
#include <vector>
#include "fused_hex.hpp"
#include "tensor_ops.hpp"

#include <algorithm>
#include <cstring>

#include "kernels_mpi.hpp"
#include "physics.hpp"
#include "split3_kernels.hpp" // -> split2_kernels.hpp -> split_common.hpp: the kernels of both variants

namespace hfx
{

void fused_invalidate(hfx_eles *e)
{
  if (!e || !e->fused) return;
  e->fused->built = false;
  if (e->fused->les_len2) { (void)hipFree(e->fused->les_len2); e->fused->les_len2 = nullptr; } // (follows the registered closure)
}

void fused_destroy(hfx_eles *e)
{
  if (!e || !e->fused) return;
  FusedData *f = e->fused;
  void *p[] = {f->meta, f->disu_alt, f->fn_fpts, f->t_coef, f->t_idx, f->pk_g, f->pk_r, f->tab_g, f->tab_r, f->o1m_dim, f->nbr, f->les_len2,
               f->upd_list_b, f->upd_list_i};
  for (void *q : p)
    if (q) (void)hipFree(q);
  delete f;
  e->fused = nullptr;
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
    case 7: return build_packed<2, 7>(e, F, o1v, o1i);
    case 8: return build_packed<2, 8>(e, F, o1v, o1i);
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
  F->h_coef = coef;
  if (upload((void **)&F->t_idx, idx.data(), sizeof(int) * idx.size())) return 1;
  F->tensor_ok = true;
  return 0;
}

static int fused_build(hfx_eles *e, hfx_inters *const *faces, int nfb, bool allow_unpaired = false)
{
  HFX_CHECK(e->ele_type == 4 || e->ele_type == 1, "fused path: tensor-product elements only (hexes, quads)");
  const int N = tensor_n(e);
  HFX_CHECK(N >= 2 && N <= (e->n_dims == 2 ? 8 : 6), "fused path: built for orders 1..5 (quads: 1..7) (n_upts %d, n_fpts %d)", e->n_upts, e->n_fpts);
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
  for (int **q : {&F->upd_list_b, &F->upd_list_i})
    if (*q) { (void)hipFree(*q); *q = nullptr; }
  F->n_list_b = F->n_list_i = 0;
  if (allow_unpaired)
  {
    // the elements with a partition-face point, and the others (split update launch of hfx_run_steps_partitioned)
    std::vector<int> lb, li;
    for (int el = 0; el < e->n_eles; el++)
    {
      bool any = false;
      for (int j = 0; j < nfp && !any; j++) any = !paired[(long)nfp * el + j];
      (any ? lb : li).push_back(el);
    }
    F->n_list_b = (long)lb.size();
    F->n_list_i = (long)li.size();
    F->n_list_i1 = F->n_list_i / 2;
    if (!lb.empty() && upload((void **)&F->upd_list_b, lb.data(), sizeof(int) * lb.size())) return 1;
    if (!li.empty() && upload((void **)&F->upd_list_i, li.data(), sizeof(int) * li.size())) return 1;
  }
  if (upload((void **)&F->meta, meta.data(), plane_f)) return 1;
  {
    // partner of every interior flux point for the flux kernel that forms the LDG corrections itself:
    // (partner offset << 2) | beta-sign flipped << 1 | this point is the right side;  -1: boundary or partition-face point
    std::vector<int> nbr(plane_f, -1);
    bool fits = plane_f < (1L << 29);
    for (int b = 0; b < nfb && fits; b++)
    {
      hfx_inters *f = faces[b];
      if (f->is_bdy) continue;
      const long np = (long)f->n_inters * f->n_fpts_per_inter;
      for (long q = 0; q < np; q++)
      {
        const int il = f->hL[q], ir = f->hR[q];
        nbr[il] = (ir << 2) | (meta[il] & 2);
        nbr[ir] = (il << 2) | (meta[il] & 2) | 1;
      }
    }
    if (F->nbr) { (void)hipFree(F->nbr); F->nbr = nullptr; }
    if (fits && upload((void **)&F->nbr, nbr.data(), sizeof(int) * nbr.size())) return 1;
  }
  if (!F->disu_alt) HFX_HIP(hipMalloc((void **)&F->disu_alt, sizeof(double) * plane_f * e->n_fields));
  F->built = true;
  return 0;
}

// workgroups of a persistent (grid-stride over elements) kernel that are resident on one CU at once -- registers and LDS
// decide.  A grid of exactly that many per CU has no workgroup waiting for a slot: with more, the late ones start when
// the others are part way through their elements and finish as a tail (LES residual kernel: five resident, 0.70 ms at
// five per CU, 1.00 ms at six, 0.81 ms at sixteen).
template <auto KERNEL>
static int resident_per_cu(int threads)
{
  static int n = 0;
  if (n == 0)
  {
    int q = 0;
    n = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, KERNEL, threads, 0) == hipSuccess && q > 0) ? q : 1;
  }
  return n;
}

// grid of a persistent element kernel: the resident workgroups, or `per_cu` of them per CU when the option asks.  `cap`:
// the streaming update kernel is fastest at three workgroups per CU (0.27 ms; 0.35 ms at the four that are resident, 0.29 ms
// at sixteen)
template <auto KERNEL>
static int element_grid(const hfx_eles *e, int threads, int per_cu, int cap = 1 << 30)
{
  const int pc = per_cu > 0 ? per_cu : std::min(cap, resident_per_cu<KERNEL>(threads));
  return (int)std::min<long>(e->n_eles, (long)e->ctx->n_cu * pc);
}

// launch of the loader-wave form, instantiated only for element sizes it fits (loader_wave_fits)
template <int ND, int N, bool OI, bool GA, bool LES, bool FITS>
struct LoaderWaveLaunch
{
  static void go(const hfx_eles *, int, hipStream_t, const Split2Args &, const double *, const int *) {}
};
template <int ND, int N, bool OI, bool GA, bool LES>
struct LoaderWaveLaunch<ND, N, OI, GA, LES, true>
{
  static void go(const hfx_eles *e, int per_cu, hipStream_t st, const Split2Args &e2, const double *coef, const int *idx)
  {
    constexpr int TB = SGeo<ND, N>::TB + 64;
    // (the over-integration form measured 3 % faster at sixteen workgroups per CU than at the two that are resident)
    if (OI && per_cu == 0) per_cu = 16;
    int grid = element_grid<split_flux_tensor_kernel<ND, N, 2, true, OI, true, GA, LES>>(e, TB, per_cu);
    if (e2.ele_list != nullptr) grid = (int)std::max<long>(1, std::min<long>(grid, e2.n_list));
    hipLaunchKernelGGL((split_flux_tensor_kernel<ND, N, 2, true, OI, true, GA, LES>), dim3(grid), dim3(TB), 0, st, e2, coef, idx);
  }
};

// The squared length scale of the eddy-viscosity closures at every solution point (les_len2_upload, hfx.hip): the flux kernel
// reads it instead of evaluating a cube root per point and stage.
static int les_len2_build(hfx_eles *e)
{
  FusedData *F = e->fused;
  if (F->les_len2) return 0;
  return les_len2_upload(e, &F->les_len2);
}

// runtime form of loader_wave_fits
static bool loader_wave_fits_rt(int nd, int N)
{
#define HFX_LWF(ND_, N_) \
  if (nd == ND_ && N == N_) return loader_wave_fits<ND_, N_>();
  HFX_LWF(3, 2) HFX_LWF(3, 3) HFX_LWF(3, 4) HFX_LWF(3, 5) HFX_LWF(3, 6) HFX_LWF(2, 2) HFX_LWF(2, 3) HFX_LWF(2, 4) HFX_LWF(2, 5) HFX_LWF(2, 6) HFX_LWF(2, 7) HFX_LWF(2, 8)
#undef HFX_LWF
  return false;
}

// Can the LES closure of this block be evaluated inside the flux kernel of variant 3 (split_flux_tensor_kernel<..., LES>)?  It
// is part of the loader-wave form of the sum-factorised kernel only; otherwise a block with a closure runs variant 2, which keeps
// the corrected gradient in HBM for the pointwise closure kernel.  Needs the block's fused tables (fused_build).
bool les_in_flux_kernel(const hfx_eles *e)
{
  const hfx_ctx::Options &opt = e->ctx->opt;
  if (!e->les_ready || !e->ctx->params.viscous || e->over_int_ready || !opt.les_flux_kernel) return false;
  if (!e->fused || !e->fused->built || !e->fused->tensor_ok) return false;
  if (opt.dictionary_rows || !opt.loader_wave || !opt.buffer_addressing || opt.flux_waves != 2) return false;
  if (!loader_wave_fits_rt(e->n_dims, tensor_n(e))) return false;
  const double plane_most = (double)std::max<long>((long)e->n_fpts * e->n_eles, (long)e->n_upts * e->n_eles);
  // every array the launch touches below 4 GiB (boundary blocks make it store the flux-point gradient array too)
  return plane_most * e->n_fields * e->n_dims * 8.0 < 4294967296.0;
}

template <int ND, int N>
static int launch_split_stage(hfx_eles *e, hfx_inters *const *faces, int nfb, SplitEleArgs &ea, int which, int variant)
{
  FusedData *F = e->fused;
  hipStream_t st = e->ctx->stream;
  const Phys P = e->ctx->phys();
  const long plane_f = (long)e->n_fpts * e->n_eles;
  const hfx_ctx::Options &opt = e->ctx->opt;
  // persistent grids: split_grid_per_cu workgroups per CU, 0 = as many as are resident (element_grid)
  const int per_cu = opt.split_grid_per_cu;
  const int flux_per_cu = opt.flux_grid_per_cu > 0 ? opt.flux_grid_per_cu : per_cu;
  constexpr int TB = SGeo<ND, N>::TB;
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
    e2.light_short = opt.light_wave_short ? 1 : 0;
    e2.o3v = e->opp_3.ell_val; e2.o3i = e->opp_3.ell_idx; e2.o3w = std::max(e->opp_3.nnz_max, 1);
    e2.o0v = e->opp_0.ell_val; e2.o0i = e->opp_0.ell_idx; e2.o0w = std::max(e->opp_0.nnz_max, 1);
    e2.src = ea.src; e2.dt_local = ea.dt_local; e2.nan_flag = ea.nan_flag; e2.P = ea.P;
    e2.adv_type = ea.adv_type; e2.in_step = ea.in_step; e2.dt_local_on = ea.dt_local_on; e2.write_div = ea.write_div;
    e2.need_u1 = ea.need_u1; e2.dt = ea.dt; e2.rk_a = ea.rk_a; e2.rk_b = ea.rk_b;
    e2.les = e->les; e2.tdA_fpts = e->tdA_fpts;
    if (e->les_ready && e->les.sgs_model != 3 && les_len2_build(e)) return 1;
    e2.les_len2 = F->les_len2;
  }
  // Will the flux kernel form the LDG corrections of the interior points itself?  (the loader-wave form of the sum-factorised
  // kernel only: same conditions as its selection below)
  // (with over-integration the loader-wave form is taken when the over-integration kernel can hand over its folded result)
  const bool oi_fold_ok = !e->over_int_ready ||
                          (opt.over_int_fold && tensor_over_int_available(e) && e->ctx->contract_mode != HFX_CONTRACT_DENSE);
  bool gather = false;
  if (variant == 3 && P.viscous && opt.gather_delta && F->nbr && F->tensor_ok && !opt.dictionary_rows && opt.loader_wave &&
      opt.buffer_addressing && opt.flux_waves == 2 && loader_wave_fits<ND, N>() && oi_fold_ok)
  {
    bool any_bdy = false;
    for (int b = 0; b < nfb; b++) any_bdy = any_bdy || faces[b]->is_bdy;
    const double plane_most = (double)std::max<long>(plane_f, (long)e->n_upts * e->n_eles);
    double most = plane_most * std::max(e->n_dims * e->n_dims, e->n_fields);
    if (any_bdy) most = std::max(most, plane_most * e->n_fields * e->n_dims);
    if (e->over_int_ready) most = std::max(most, (double)e->n_upts * e->n_eles * e->n_fields * e->n_dims);
    gather = most * 8.0 < 4294967296.0;
  }
  F->gather_on = gather;
  if (variant == 3)
  {
    e2.nbr = gather ? F->nbr : nullptr;
    e2.disu = e->arr[HFX_DISU_FPTS];
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
      if (gather) continue; // (the flux kernel reads the partner's flux-point solution itself)
      const SplitFaceArgs a = face_args(faces[b]);
      if (a.npairs == 0) continue;
      hipLaunchKernelGGL((face_delta_kernel<ND>), dim3((unsigned)((a.npairs + 255) / 256)), dim3(256), 0, st, a);
    }
  }
  // which (timing only): 5 = the over-integration kernel of part 2 alone, 7 = part 2 without it; 6 = the SGS kernels of part 2
  // (variant 2 with an LES closure) alone, 8 = part 2 without them
  if (variant == 3)
  {
    // Over-integration (src/solver.cpp:82-91).  With the loader-wave flux kernel the sum-factorised kernel hands over
    // sum_l Dc[l] tdisf_l -- the de-aliased flux's whole contribution to (div_tdisf - opp_3 norm_tdisf), n_fields values per
    // solution point (tensor_ops.hip) -- which that kernel adds to its divergence; otherwise tdisf_upts itself.
    const bool lw_form = loader_wave_fits<ND, N>() && opt.loader_wave && opt.flux_waves == 2 && opt.buffer_addressing &&
                         F->tensor_ok && !opt.dictionary_rows;
    const bool oi_fold = e->over_int_ready && lw_form && oi_fold_ok;
    auto run_over_int = [&]() -> int {
      if (!oi_fold) return hfx_eles_evaluate_invFlux_over_int(e);
      if (!tensor_over_int_folded(e))
      {
        // Dc[d] = D - c3[d][0] (L1 Lf)[d][0]^T - c3[d][1] (L1 Lf)[d][1]^T, as the flux kernel's prologue forms it (split3_kernels.hpp)
        using T = TGeo<ND, N>;
        const std::vector<double> &c = F->h_coef;
        std::vector<double> Dc((size_t)ND * N * N);
        for (int d = 0; d < ND; d++)
          for (int mp = 0; mp < N; mp++)
            for (int m = 0; m < N; m++)
            {
              const double ta = c[T::C_3 + (d * 2 + 0) * N + mp] * (c[T::C_L1 + (d * 2 + 0) * N] * c[T::C_LF + (d * 2 + 0) * N + m]);
              const double tb = c[T::C_3 + (d * 2 + 1) * N + mp] * (c[T::C_L1 + (d * 2 + 1) * N] * c[T::C_LF + (d * 2 + 1) * N + m]);
              Dc[((size_t)d * N + mp) * N + m] = c[T::C_D + mp * N + m] - ta - tb;
            }
        if (tensor_over_int_set_fold(e, Dc.data())) return 1;
      }
      return tensor_over_int_launch(e, true);
    };
    if (which == 5 && e->over_int_ready && run_over_int()) return 1;
    // 21 / 22 / 23: the flux kernel on a part of the elements (partitioned blocks, hfx_run_steps_partitioned): the first half of
    // the elements without partition-face points, those with, the second half -- the solution exchange runs beside the
    // first launch, the exchange of the projected fluxes beside the third
    const bool flux_part = which >= 21 && which <= 23;
    e2.ele_list = nullptr;
    e2.n_list = 0;
    if (which == 21) { e2.ele_list = F->upd_list_i; e2.n_list = F->n_list_i1; }
    if (which == 22) { e2.ele_list = F->upd_list_b; e2.n_list = F->n_list_b; }
    if (which == 23) { e2.ele_list = F->upd_list_i ? F->upd_list_i + F->n_list_i1 : nullptr; e2.n_list = F->n_list_i - F->n_list_i1; }
    if (flux_part && e2.n_list == 0) return 0;
    HFX_CHECK(!flux_part || (e2.ele_list != nullptr && !e->over_int_ready), "split flux kernel on element lists: no lists, or over-integration (which runs on all elements first)");
    if (which == 0 || which == 2 || which == 7 || flux_part)
    {
      e2.tdisf_in = nullptr;
      if (e->over_int_ready)
      {
        // polynomial de-aliasing (src/solver.cpp:82-91): tdisf_upts = over_int_filter . F(opp_over_int_cubpts . u)
        if (which != 7 && run_over_int()) return 1;
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
  hipLaunchKernelGGL((split_flux_tensor_kernel<ND, N, WV_, BUF_, OI_, LW_>),                                                  \
                     dim3(element_grid<split_flux_tensor_kernel<ND, N, WV_, BUF_, OI_, LW_>>(e, TB, per_cu)), dim3(TB), 0, st, \
                     e2, F->t_coef, F->t_idx)
      const bool no_lw = !opt.loader_wave;
      constexpr bool lw_fits = loader_wave_fits<ND, N>();
      // (a block whose de-aliased flux arrives whole -- dense over-integration -- takes the form without the loader wave)
      const bool lw = lw_fits && buf && !no_lw && waves == 2 && (!oi || oi_fold);
      HFX_CHECK(!oi_fold || (lw && F->tensor_ok && !dict_only), "over-integration: the folded form needs the loader-wave flux kernel");
      bool launched = false;
      // a closure with an SGS flux (every model but the spectral vanishing viscosity, which only filters the state)
      const bool les = e->les_ready && e->les.sgs_model != 3;
      HFX_CHECK(!les || (lw && F->tensor_ok && !dict_only && !oi && P.viscous),
                "split variant 3 with an LES closure needs the loader-wave flux kernel (les_in_flux_kernel): run variant 2");
      if (F->tensor_ok && !dict_only && lw)
      {
        const bool ga = e2.nbr != nullptr && P.viscous; // (the corrections formed in the kernel)
        if (les && ga)
          LoaderWaveLaunch<ND, N, false, true, true, lw_fits>::go(e, flux_per_cu, st, e2, F->t_coef, F->t_idx);
        else if (les)
          LoaderWaveLaunch<ND, N, false, false, true, lw_fits>::go(e, flux_per_cu, st, e2, F->t_coef, F->t_idx);
        else if (oi && ga)
          LoaderWaveLaunch<ND, N, true, true, false, lw_fits>::go(e, flux_per_cu, st, e2, F->t_coef, F->t_idx);
        else if (oi)
          LoaderWaveLaunch<ND, N, true, false, false, lw_fits>::go(e, flux_per_cu, st, e2, F->t_coef, F->t_idx);
        else if (ga)
          LoaderWaveLaunch<ND, N, false, true, false, lw_fits>::go(e, flux_per_cu, st, e2, F->t_coef, F->t_idx);
        else
          LoaderWaveLaunch<ND, N, false, false, false, lw_fits>::go(e, flux_per_cu, st, e2, F->t_coef, F->t_idx);
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
        hipLaunchKernelGGL((split_flux_kernel<ND, N>), dim3(element_grid<split_flux_kernel<ND, N>>(e, TB, per_cu)), dim3(TB), 0, st, e2);
    }
  }
  else if (P.viscous && (which == 0 || which == 2 || which == 6 || which == 8))
  {
    ea.pk = F->pk_g;
    ea.tab = F->tab_g;
    if (which != 6)
      hipLaunchKernelGGL((split_gradient_kernel<ND, N>), dim3(element_grid<split_gradient_kernel<ND, N>>(e, TB, per_cu)), dim3(TB), 0, st, ea);
    if (e->les_ready && which != 8)
    {
      // LES (eddy-viscosity closures): SGS flux at the solution points from the corrected gradient, its extrapolation to
      // the flux points (src/solver.cpp:162-167); the face kernel adds it to each side, the residual kernel to the total
      if (hfx_les_sgsf_upts_internal(e)) return 1;
      if (hfx_les_extrapolate_reference_internal(e)) return 1; // the back-transform happens in the face kernel
    }
  }
  if (which == 0 || which == 3)
  {
    // boundary faces on the side stream, beside the pairwise interior-face kernel: both need the flux kernel's results and
    // write norm_tconf at disjoint points
    bool any_bdy_faces = false;
    for (int b = 0; b < nfb; b++) any_bdy_faces = any_bdy_faces || (faces[b]->is_bdy && faces[b]->n_inters > 0);
    const bool beside = any_bdy_faces && opt.bdy_beside && e->ctx->mpi_stream == nullptr;
    if (beside && side_stream_fork(e->ctx)) return 1;
    for (int b = 0; b < nfb; b++)
      if (faces[b]->is_bdy && hfx_bdy_launch_internal(faces[b], P.viscous ? 1 : 0, 1)) return 1;
    if (beside && side_stream_join(e->ctx)) return 1;
    for (int b = 0; b < nfb; b++)
    {
      if (faces[b]->is_bdy) continue;
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
    if (beside && side_stream_wait(e->ctx)) return 1;
  }
  if (which == 0 || which == 4 || which == 41 || which == 42)
  {
    if (variant == 3)
    {
      // 41 / 42: the update on the elements with partition-face points / on the others (two launches: the first one's
      // flux-point solution leaves for the neighbours while the second runs); 42 comes behind the buffer swap of 41
      e2.ele_list = nullptr;
      e2.n_list = 0;
      if (which == 41) { e2.ele_list = F->upd_list_b; e2.n_list = F->n_list_b; }
      if (which == 42) { e2.ele_list = F->upd_list_i; e2.n_list = F->n_list_i; e2.disu_next = e->arr[HFX_DISU_FPTS]; }
      HFX_CHECK(which < 41 || e2.ele_list != nullptr || e2.n_list == 0, "split update: no element lists (the block was not built as a partitioned one)");
      // buffer-descriptor addressing needs 32-bit byte offsets
      const bool nobuf = !opt.buffer_addressing;
      const bool small = (double)std::max<long>(plane_f, (long)e->n_upts * e->n_eles) * e->n_fields * 8.0 < 4294967296.0;
      const long n_work = which >= 41 ? e2.n_list : (long)e->n_eles;
      if (n_work > 0)
      {
        if (small && !nobuf)
        {
          const int g = (int)std::min<long>(n_work, element_grid<split_update_kernel<ND, N, true>>(e, TB, per_cu, 3));
          hipLaunchKernelGGL((split_update_kernel<ND, N, true>), dim3(g), dim3(TB), 0, st, e2);
        }
        else
        {
          const int g = (int)std::min<long>(n_work, element_grid<split_update_kernel<ND, N, false>>(e, TB, per_cu, 3));
          hipLaunchKernelGGL((split_update_kernel<ND, N, false>), dim3(g), dim3(TB), 0, st, e2);
        }
      }
    }
    else
    {
      ea.pk = F->pk_r;
      ea.tab = F->tab_r;
      hipLaunchKernelGGL((split_residual_kernel<ND, N>), dim3(element_grid<split_residual_kernel<ND, N>>(e, TB, per_cu)), dim3(TB), 0, st, ea);
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
  HFX_SPLIT_CASE(2, 2) HFX_SPLIT_CASE(2, 3) HFX_SPLIT_CASE(2, 4) HFX_SPLIT_CASE(2, 5) HFX_SPLIT_CASE(2, 6) HFX_SPLIT_CASE(2, 7) HFX_SPLIT_CASE(2, 8)
#undef HFX_SPLIT_CASE
  if (rc) return 1;
  // (the buffer swap: behind the whole update, or behind its first part -- the second part then writes the new buffer by name)
  if (which == 0 || which == 4 || which == 41) std::swap(e->arr[HFX_DISU_FPTS], e->fused->disu_alt);
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

int split_deferred_prepare(hfx_eles *e, hfx_inters *const *faces, int nfb, bool partitioned)
{
  HFX_CHECK(e->n_eles > 0, "fused path: empty element block");
  if (e->fused && e->fused->built) return 0;
  return fused_build(e, faces, nfb, partitioned);
}

int split_deferred_stage(hfx_eles *e, hfx_inters *const *faces, int nfb, int in_step, bool write_div, bool shock)
{
  if (split_deferred_prepare(e, faces, nfb, false)) return 1;
  const int variant = split_variant(e);
  HFX_CHECK(!e->over_int_ready || variant == 3, "the split variant that keeps the gradients (fused 2, which LES without the in-kernel closure selects) has no over-integration");
  if (split_stage(e, faces, nfb, in_step, write_div, 0, variant)) return 1;
  // the filter changes disu_upts(0) after the stage: redo the flux-point solution of the new state
  return shock ? shock_capture_keep_fpts(e) : 0;
}

int split_run_steps(hfx_eles *e, hfx_inters *const *faces, int nfb, int n_steps, int variant)
{
  HFX_CHECK(e->n_eles > 0, "fused path: empty element block");
  if (!e->fused || !e->fused->built)
    if (fused_build(e, faces, nfb)) return 1;
  // an LES closure reads the corrected gradient: variant 3 evaluates it in the flux kernel where that kernel's loader-wave form
  // runs (les_in_flux_kernel); otherwise such a block runs variant 2, which keeps the gradient in HBM for a pointwise kernel
  if (e->les_ready && variant == 3 && !les_in_flux_kernel(e)) variant = 2;
  HFX_CHECK(!e->over_int_ready || variant == 3, "the split variant that keeps the gradients (fused 2, which LES without the in-kernel closure selects) has no over-integration");
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
  if (e->les_ready && variant == 3 && !les_in_flux_kernel(e)) variant = 2;
  const int adv = e->ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  hipStream_t st = e->ctx->stream;
  // one set of events per repetition and ONE synchronisation at the end: a host synchronisation per stage let the queue
  // run dry, and the first kernel after it (the flux kernel) then measured 10 % slower than in the running pipeline
  // the parts of a stage in launch order; part 2 in two pieces when the block de-aliases (variant 3: the over-integration kernel,
  // then the flux kernel) or carries an LES closure (variant 2: the gradient kernel, then the SGS kernels) -- slot 4 of `ms`
  const bool oi = variant == 3 && e->over_int_ready, sgs = variant == 2 && e->les_ready && e->ctx->params.viscous;
  std::vector<int> parts = {1};
  if (oi) { parts.push_back(5); parts.push_back(7); }
  else if (sgs) { parts.push_back(8); parts.push_back(6); }
  else parts.push_back(2);
  parts.push_back(3);
  parts.push_back(4);
  const int np = (int)parts.size();
  std::vector<hipEvent_t> ev((size_t)reps * (np + 1));
  for (auto &x : ev) HFX_HIP(hipEventCreate(&x));
  if (hfx_eles_extrapolate_solution(e)) return 1;
  double acc[9] = {};
  for (int r = 0; r < reps; r++)
  {
    const int rk = r % nst;
    for (int q = 0; q < np; q++)
    {
      HFX_HIP(hipEventRecord(ev[(np + 1) * r + q], st));
      if (split_stage(e, faces, nfb, rk, rk == nst - 1, parts[q], variant)) return 1;
    }
    HFX_HIP(hipEventRecord(ev[(np + 1) * r + np], st));
  }
  HFX_HIP(hipStreamSynchronize(st));
  for (int r = 0; r < reps; r++)
    for (int q = 0; q < np; q++)
    {
      float t = 0;
      HFX_HIP(hipEventElapsedTime(&t, ev[(np + 1) * r + q], ev[(np + 1) * r + q + 1]));
      acc[parts[q]] += t;
    }
  for (auto &x : ev) (void)hipEventDestroy(x);
  for (int i = 0; i < 8; i++) ms[i] = 0.0;
  ms[0] = acc[1] / reps;
  ms[1] = (acc[2] + acc[7] + acc[8]) / reps; // the element kernel of part 2 alone
  ms[2] = acc[3] / reps;
  ms[3] = acc[4] / reps;
  ms[4] = (acc[5] + acc[6]) / reps;          // over-integration kernel | SGS kernels
  if (e->fused->stamps)
  {
    long long h[64];
    HFX_HIP(hipMemcpy(h, e->fused->stamps, sizeof h, hipMemcpyDeviceToHost));
    for (int w = 0; w < 3; w++)
    {
      fprintf(stderr, "flux kernel wave %d cycles: ", w);
      for (int q = 1; q <= 9; q++) fprintf(stderr, "%s%lld", q > 1 ? " " : "", h[w * 16 + q] - h[w * 16 + q - 1]);
      fprintf(stderr, "   (fill | bar1 | A | bar2 | B | bar3 | C | bar4 | D)  total %lld;  C: reads landed after %lld\n", h[w * 16 + 9] - h[w * 16], h[w * 16 + 10] - h[w * 16 + 6]);
      if (h[w * 16 + 11])
        fprintf(stderr, "   A0 cycles: corrections written after %lld, next requests issued %lld, barrier 1b %lld\n", h[w * 16 + 11] - h[w * 16 + 2],
                h[w * 16 + 12] - h[w * 16 + 11], h[w * 16 + 13] - h[w * 16 + 12]);
    }
    fprintf(stderr, "loader wave cycles: ");
    for (int q = 1; q <= 7; q++) fprintf(stderr, "%s%lld", q > 1 ? " " : "", h[3 * 16 + q] - h[3 * 16 + q - 1]);
    fprintf(stderr, "   (wait state | bar1 | issue state, wait metrics | bar2 | bar3 | issue metrics | bar4)  total %lld\n", h[3 * 16 + 7] - h[3 * 16]);
  }
  const bool tensor = e->fused->tensor_ok && !e->ctx->opt.dictionary_rows;
  snprintf(names, names_len, "%s%s",
           variant == 3 ? (tensor ? "face_delta_kernel,split_flux_tensor_kernel,face_flux2_kernel,split_update_kernel"
                                  : "face_delta_kernel,split_flux_kernel,face_flux2_kernel,split_update_kernel")
                        : "face_delta_kernel,split_gradient_kernel,face_flux_kernel,split_residual_kernel",
           oi ? (tensor_over_int_available(e) && e->ctx->contract_mode != HFX_CONTRACT_DENSE ? ",overint_tensor_kernel" : ",evaluate_invFlux_over_int (dense)")
              : sgs ? ",sgsf_upts_kernel + ell_apply_kernel (SGS flux)" : "");
  return 0;
}

void split_kernel_bytes(const hfx_eles *e, double *bytes, int variant)
{
  if (e->les_ready && variant == 3 && !les_in_flux_kernel(e)) variant = 2;
  // ALGORITHMIC HBM bytes per launch (doubles listed per element)
  const double nu = e->n_upts, nfp = e->n_fpts, nf = e->n_fields, nd = e->n_dims, ne = e->n_eles;
  for (int i = 0; i < 8; i++) bytes[i] = 0.0;
  bytes[0] = ne * (8.0 * (2 * nfp * nf) + 4.0 * nfp + nfp * 0.5);                                   // disu r, delta w, index + meta
  bytes[1] = ne * 8.0 * (nu * nf + nfp * nf + nu * (nd * nd + 1) + nfp * (nd * nd + 1) + nfp * nf * nd); // + grad_fpts w
  bytes[2] = ne * (8.0 * (nfp * nf + nfp * nf * nd + 0.5 * nfp * nd + nfp + nfp * nf) + 4.0 * nfp);   // disu, grad, normal(left), tdA r; tconf w
  bytes[3] = ne * 8.0 * (nu * nf + nu * (nd * nd + 1) + nfp * nf + 3 * nu * nf + nfp * nf);           // u, metrics, tconf, u1 r; u0,u1,disu w
  if (variant == 2 && e->les_ready)
    // SGS flux at the solution points (u, corrected gradient, metrics r; sgsf_upts w) and its extrapolation (sgsf_upts r, sgsf_fpts w)
    bytes[4] = ne * 8.0 * (nu * nf + nu * nf * nd + nu * (nd * nd + 1) + 2 * nu * nf * nd + nfp * nf * nd);
  if (variant == 3)
  {
    // u, delta, volume + flux-point metrics, own normals r ; div, norm_tdisf, Fn w
    // (norm_tdisf: the dictionary-row form only; the sum-factorised kernel folds opp_3 . norm_tdisf into div)
    const double ntd = (e->fused && e->fused->tensor_ok && !e->ctx->opt.dictionary_rows) ? 0.0 : nfp * nf;
    bytes[1] = ne * 8.0 * (nu * nf + nfp * nf + nu * (nd * nd + 1) + nfp * (nd * nd + 1) + nfp * nd + nu * nf + nfp * nf + ntd);
    bytes[2] = ne * (8.0 * (nfp * nf + nfp * nf + 0.5 * nfp * nd + nfp + nfp * nf) + 4.0 * nfp); // disu, Fn, normal(left), tdA r; tconf w
    bytes[3] = ne * 8.0 * (3 * nu * nf + nu + nfp * nf + ntd + 2 * nu * nf + nfp * nf);          // u0,u1,div,detjac,tconf(,ntd) r; u0,u1,disu w
    if (e->les_ready)
    {
      // the closure in the flux kernel: tdA at the flux points, the squared length scale, the Leonard terms of the similarity models
      const int m = e->les.sgs_model;
      bytes[1] += ne * 8.0 * (nfp + nu + ((m == 2 || m == 4) ? nu * (nd == 3 ? 9.0 : 5.0) : 0.0)); // tdA, length scale, Leonard terms
    }
    if (e->over_int_ready)
    {
      // the flux kernel reads the de-aliased flux too; the over-integration kernel: u, the metric tensors at the cubature points r, tdisf w
      bytes[1] += ne * 8.0 * nu * nf * nd;
      bytes[4] = ne * 8.0 * (nu * nf + nd * nd * e->n_cubpts + nu * nf * nd);
    }
    if (e->fused && e->fused->gather_on)
    {
      // the flux kernel reads the partners' flux-point solution (as many doubles as the corrections it no longer reads) and
      // a partner word per point; the pairwise LDG kernel is not launched
      bytes[0] = 0.0;
      bytes[1] += ne * 4.0 * nfp;
    }
  }
}


#include "split_partitioned.hpp"

} // namespace hfx
