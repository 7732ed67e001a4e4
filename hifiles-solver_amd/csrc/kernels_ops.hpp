// kernels_ops.hpp -- operator contractions  out(m x ncols) (=|+=) sum_t op_t(m x k) * in_t(k x ncols)
//
// These replace the seven dgemm families of the reference
// (/root/reference/src/eles.cpp:1370,1559,1661,1758,1835,1900,1930; dgemm semantics
// /root/reference/src/funcs.cpp:49-123).  A "column" is one (element, field[, dim]) pair.
//
// Sparse form (this file): the operators of tensor-product elements at
// collocated Gauss / Lobatto points hold 1..N exact non-zeros per row
// (SURVEY.md 7.6); the reference's own `sparse_* 1` option targets the same
// structure (src/eles.cpp:3114-3119).  Skipping exact zeros does not change
// any sum (x + 0*y = x), and the non-zeros of a row are accumulated in the
// same ascending-column order as the reference dgemm, so the result differs
// from the dense reference only by FMA contraction.
//
// Data movement: each workgroup owns a tile of CT consecutive columns; the
// k x CT input tile(s) are contiguous in HBM (hf_array layout) and are staged
// through LDS with fully coalesced loads; every thread keeps the non-zeros of
// ONE operator row in registers and walks the tile's columns, so the output
// rows of a column are written by consecutive lanes (coalesced m-double runs).
#pragma once
#include "hfx_internal.hpp"

namespace hfx
{

template <int NT, int NO>
struct EllArgs
{
  int m, k;           // operator is m x k
  long ncols;         // columns of in / out
  int ct;             // columns per workgroup tile
  int rg;             // row groups per workgroup (threads = rg*m rounded up to 64)
  int beta;           // 0: out = ..., 1: out += ...
  const double *val[NT * NO];
  const int *idx[NT * NO];
  const double *in[NT];  // NT input slabs (k x ncols each)
  double *out[NO];       // NO output slabs (m x ncols each)
  // optional fused prologue of calculate_corrected_divergence (src/eles.cpp:1746):
  // in_0 := in_0 - sub (written back), used as the input
  const double *sub;
  double *in_writeback;
  unsigned long long *nan_flag; // optional NaN report on out_0 (src/eles.cpp:1781-1795)
};

template <int NNZ, int NT, int NO>
__global__ void ell_apply_kernel(const EllArgs<NT, NO> a)
{
  extern __shared__ double tile[]; // NT tiles of k*ct doubles
  const int m = a.m, k = a.k, ct = a.ct;
  const long c0 = (long)blockIdx.x * ct;
  const int ncl = (int)min((long)ct, a.ncols - c0); // columns in this tile
  const int tid = threadIdx.x;

  // stage the input tile(s): contiguous k*ncl doubles each
  const int tile_len = k * ncl;
#pragma unroll
  for (int t = 0; t < NT; t++)
  {
    const double *src = a.in[t] + c0 * k;
    double *dst = tile + t * (k * ct);
    if (t == 0 && a.sub != nullptr)
    {
      const double *sb = a.sub + c0 * k;
      double *wb = a.in_writeback + c0 * k;
      for (int q = tid; q < tile_len; q += blockDim.x)
      {
        const double v = src[q] + -1.0 * sb[q];
        dst[q] = v;
        wb[q] = v;
      }
    }
    else
    {
      for (int q = tid; q < tile_len; q += blockDim.x) dst[q] = src[q];
    }
  }

  const int g = tid / m;  // row group
  const int r = tid - g * m;
  const bool active = g < a.rg;

  double val[NO * NT][NNZ];
  int idx[NO * NT][NNZ];
  if (active)
  {
#pragma unroll
    for (int ot = 0; ot < NO * NT; ot++)
#pragma unroll
      for (int q = 0; q < NNZ; q++)
      {
        val[ot][q] = a.val[ot][r + m * q];
        idx[ot][q] = a.idx[ot][r + m * q];
      }
  }
  __syncthreads();
  if (!active) return;

  for (int c = g; c < ncl; c += a.rg)
  {
#pragma unroll
    for (int o = 0; o < NO; o++)
    {
      double *op = a.out[o] + (c0 + c) * m + r;
      double acc = a.beta ? *op : 0.0;
#pragma unroll
      for (int t = 0; t < NT; t++)
      {
        const double *tl = tile + t * (k * ct) + c * k;
#pragma unroll
        for (int q = 0; q < NNZ; q++) acc += val[o * NT + t][q] * tl[idx[o * NT + t][q]];
      }
      *op = acc;
      if (o == 0 && a.nan_flag != nullptr && acc != acc)
        atomicMin(a.nan_flag, (unsigned long long)((c0 + c) * m + r));
    }
  }
}

// ---------------------------------------------------------------------------
// Dense form: FP64 MFMA GEMM  C(m x ncols) = beta*C + A(m x k) * B(k x ncols), A small.
// v_mfma_f64_16x16x4_f64: lane l holds A[i = l&15][kk = l>>4], B[kk = l>>4][j = l&15],
// D[row = (l>>4) + 4*reg][col = l&15]  (cdna_hip_programming.md section 3).
//
// Workgroup = 4 or 8 waves owns a 64-column tile of B/C: the k x 64 B
// tile is contiguous in HBM and is staged once through LDS (coalesced); A is
// read from global/L2 (it is the same 100-150 kB for every workgroup and stays
// cache resident).  The (16-row tile, group of 16-column sub-tiles) units are dealt
// to the waves; an A fragment loaded from L2 feeds as many MFMAs as the group has sub-tiles.
// C is produced transposed inside the MFMA (operands swapped) so that stores
// are 128-byte contiguous runs along the operator-row index.
// The k-loop runs l ascending in steps of 4, the reference's summation order
// within rounding of the MFMA's internal 4-term accumulation.
// ---------------------------------------------------------------------------
struct DenseArgs
{
  int m, k;
  long ncols;
  int beta;
  const double *A;  // (m,k) col-major
  const double *Ap; // the same zero-padded to (mpad, kpad), mpad = 16 ceil(m/16), kpad = 4 ceil(k/4)
  int mpad;
  const double *B; // (k,ncols)
  double *C;       // (m,ncols)
  const double *sub;
  double *in_writeback;
  unsigned long long *nan_flag;
};

typedef double f64x4 __attribute__((ext_vector_type(4)));

// The k loop of one unit of work (a 16-row tile of the operator x NS 16-column sub-tiles of the B tile in LDS), branch-free:
// operator fragments from the zero-padded copy (no bounds checks), whole groups of UK k-steps with every load of the group
// requested before its first MFMA, then the remaining k-steps one by one.  (Written with per-load predicates and a per-step
// "past the end?" test the loop compiled to a branch per load and `ds_read; s_waitcnt; v_mfma` per step.)
template <int NS, int LDB>
__device__ __forceinline__ void dense_unit_kloop(const double *__restrict__ Ap, int mpad, int kpad, const double *bt, int row, int lk,
                                                 int li, int s0, f64x4 (&acc)[NS])
{
  constexpr int UK = 8;
  const double *ap = Ap + row + (long)mpad * lk;
  const double *bp = bt + lk * LDB + s0 * 16 + li;
  int kb = 0;
  for (; kb + 4 * UK <= kpad; kb += 4 * UK)
  {
    double av[UK], bv[UK][NS];
#pragma unroll
    for (int u = 0; u < UK; u++) av[u] = ap[(long)mpad * (kb + 4 * u)];
#pragma unroll
    for (int u = 0; u < UK; u++)
#pragma unroll
      for (int s = 0; s < NS; s++) bv[u][s] = bp[(kb + 4 * u) * LDB + s * 16];
#pragma unroll
    for (int u = 0; u < UK; u++)
#pragma unroll
      for (int s = 0; s < NS; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[u][s], av[u], acc[s], 0, 0, 0);
  }
  for (; kb < kpad; kb += 4)
  {
    const double av = ap[(long)mpad * kb];
#pragma unroll
    for (int s = 0; s < NS; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(bp[kb * LDB + s * 16], av, acc[s], 0, 0, 0);
  }
}

constexpr int DENSE_CT = 64; // columns per workgroup (32 / 16 for operators with many columns: the B tile holds all of k)

// NW waves per workgroup; a unit of work is one 16-row tile of the operator times CT/SPL columns, dealt round-robin to
// the waves (SPLQ asked, at most one unit per 16-column sub-tile)
template <int CT, int NW, int SPLQ>
__global__ __launch_bounds__(64 * NW) void dense_mfma_kernel(const DenseArgs a)
{
  constexpr int DENSE_CT = CT, DENSE_THREADS = 64 * NW;
  extern __shared__ double btile[]; // [kpad][DENSE_CT+pad] transposed: bt[kk*LDB + j]
  const int m = a.m, k = a.k;
  const int kpad = (k + 3) & ~3;
  constexpr int LDB = DENSE_CT + 1; // +1 double pad: conflict-free transposed store
  const long c0 = (long)blockIdx.x * DENSE_CT;
  const int ncl = (int)min((long)DENSE_CT, a.ncols - c0);
  const int tid = threadIdx.x;

  // stage B tile: global is [col][k] contiguous; LDS is [k][col]
  {
    const double *src = a.B + c0 * k;
    const int tile_len = k * ncl;
    // SB entries per thread are requested before the first one is used: with `LDS = load` per trip every trip waits for its
    // own load (a tile of k = 125 rows is 31 trips)
    constexpr int SB = 8;
    if (a.sub != nullptr)
    {
      const double *sb = a.sub + c0 * k;
      double *wb = a.in_writeback + c0 * k;
      for (int q0 = tid; q0 < tile_len; q0 += DENSE_THREADS * SB)
      {
        double v[SB], w[SB];
#pragma unroll
        for (int i = 0; i < SB; i++)
        {
          const int q = q0 + DENSE_THREADS * i, qc = q < tile_len ? q : tile_len - 1;
          v[i] = src[qc];
          w[i] = sb[qc];
        }
#pragma unroll
        for (int i = 0; i < SB; i++)
        {
          const int q = q0 + DENSE_THREADS * i;
          if (q < tile_len)
          {
            const double x = v[i] + -1.0 * w[i];
            wb[q] = x;
            const int j = q / k, kk = q - j * k;
            btile[kk * LDB + j] = x;
          }
        }
      }
    }
    else
    {
      for (int q0 = tid; q0 < tile_len; q0 += DENSE_THREADS * SB)
      {
        double v[SB];
#pragma unroll
        for (int i = 0; i < SB; i++)
        {
          const int q = q0 + DENSE_THREADS * i;
          v[i] = src[q < tile_len ? q : tile_len - 1];
        }
#pragma unroll
        for (int i = 0; i < SB; i++)
        {
          const int q = q0 + DENSE_THREADS * i;
          if (q < tile_len)
          {
            const int j = q / k, kk = q - j * k;
            btile[kk * LDB + j] = v[i];
          }
        }
      }
    }
    // zero the k padding rows and the missing columns so that MFMA adds exact zeros
    for (int q = tid; q < (kpad - k) * DENSE_CT; q += DENSE_THREADS)
    {
      const int kk = k + q / DENSE_CT, j = q % DENSE_CT;
      btile[kk * LDB + j] = 0.0;
    }
    if (ncl < DENSE_CT)
      for (int q = tid; q < kpad * (DENSE_CT - ncl); q += DENSE_THREADS)
      {
        const int kk = q / (DENSE_CT - ncl), j = ncl + q % (DENSE_CT - ncl);
        btile[kk * LDB + j] = 0.0;
      }
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int n_rt = (m + 15) >> 4;

  constexpr int SPL = CT / 16 >= SPLQ ? SPLQ : CT / 16;
  constexpr int NS = CT / 16 / SPL; // 16-column sub-tiles per unit of work
  for (int un = wave; un < n_rt * SPL; un += NW)
  {
    const int rt = un / SPL, s0 = (un % SPL) * NS;
    // MFMA operands are swapped (C^T = B^T A^T) so that the 16 lanes li of an
    // accumulator register hold 16 CONSECUTIVE operator rows of one C column:
    // D[i = lk + 4*reg -> column of the tile][j = li -> operator row]
    const int row = rt * 16 + li;
    const bool row_ok = row < m;
    f64x4 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; s++)
    {
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
      {
        const int ocol = (s0 + s) * 16 + lk + 4 * rg;
        acc[s][rg] = (a.beta && row_ok && ocol < ncl) ? a.C[(c0 + ocol) * m + row] : 0.0;
      }
    }
    dense_unit_kloop<NS, LDB>(a.Ap, a.mpad, kpad, btile, row, lk, li, s0, acc);
#pragma unroll
    for (int s = 0; s < NS; s++)
    {
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
      {
        const int ocol = (s0 + s) * 16 + lk + 4 * rg;
        if (row_ok && ocol < ncl)
        {
          const double v = acc[s][rg];
          a.C[(c0 + ocol) * m + row] = v;
          if (a.nan_flag != nullptr && v != v) atomicMin(a.nan_flag, (unsigned long long)((c0 + ocol) * m + row));
        }
      }
    }
  }
}

} // namespace hfx
