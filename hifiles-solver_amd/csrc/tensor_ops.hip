// tensor_ops.hip -- sum-factorised over-integration and shock capturing for tensor-product elements.
//
// The reference applies over-integration (eles::evaluate_invFlux_over_int, src/eles.cpp:1480-1545) and the Persson
// sensor / exponential filter (eles::shock_capture :2918-2959, eles_hexas::shock_det_persson src/eles_hexas.cpp:1007)
// as dense per-element matrix products: n_cub x n_upts and n_upts x n_upts matrices (343 x 125, 125 x 125 at P4).
// On hexes and quads each of those matrices is a Kronecker power of ONE small 1-D matrix
// (csrc/host/eles_modal.cpp says why), so the product is three passes of a (No x Ni) matrix along the three index
// directions -- 12x fewer multiply-adds at P4 / 7 cubature points per direction -- and, because a whole element fits
// in LDS, interpolation -> flux at the cubature points -> projection run as ONE kernel that reads the state and the
// cubature-point metrics and writes the de-aliased flux: 5.6k doubles of HBM traffic per element instead of the 21k
// of the three-launch dense form.
//
// The C ABI is unchanged: the caller registers the reference's dense matrices (hfx_eles_set_over_int,
// hfx_eles_set_shock_capture); the 1-D factors are recovered from them here and VERIFIED (the Kronecker power must
// reproduce every entry to 1e-12 of the matrix scale); anything that does not factor stays on the dense MFMA path.
#include "tensor_ops.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "physics.hpp"

namespace hfx
{

static int ipow_i(int b, int e)
{
  int r = 1;
  for (int i = 0; i < e; i++) r *= b;
  return r;
}

// M (nr1^nd x nc1^nd, column-major, first direction fastest in rows and in columns) = A (x) ... (x) A ?
// A (nr1 x nc1, column-major).  For nd = 3 the real cube root fixes the sign; for nd = 2 either sign gives the same M.
bool kron_factor(const double *M, int nd, int nr1, int nc1, std::vector<double> &A, double tol)
{
  const int nr = ipow_i(nr1, nd), nc = ipow_i(nc1, nd);
  auto rdiag = [&](int p) { int r = 0; for (int d = nd - 1; d >= 0; d--) r = r * nr1 + p; return r; };
  auto cdiag = [&](int q) { int c = 0; for (int d = nd - 1; d >= 0; d--) c = c * nc1 + q; return c; };
  // pivot: the largest "diagonal" entry M[(p,p,p),(q,q,q)] = A[p,q]^nd
  int bp = 0, bq = 0;
  double best = 0.0;
  for (int p = 0; p < nr1; p++)
    for (int q = 0; q < nc1; q++)
    {
      const double v = std::fabs(M[rdiag(p) + (size_t)nr * cdiag(q)]);
      if (v > best) { best = v; bp = p; bq = q; }
    }
  if (best == 0.0) return false;
  const double m0 = M[rdiag(bp) + (size_t)nr * cdiag(bq)];
  double apq;
  if (nd == 3)
    apq = std::cbrt(m0);
  else
  {
    if (m0 < 0.0) return false;
    apq = std::sqrt(m0);
  }
  const double den = (nd == 3) ? apq * apq : apq;
  A.assign((size_t)nr1 * nc1, 0.0);
  // A[i,j] = M[(i,p,..,p),(j,q,..,q)] / A[p,q]^(nd-1)
  for (int i = 0; i < nr1; i++)
    for (int j = 0; j < nc1; j++)
    {
      int r = 0, c = 0;
      for (int d = nd - 1; d >= 1; d--) { r = r * nr1 + bp; c = c * nc1 + bq; }
      r = r * nr1 + i;
      c = c * nc1 + j;
      A[i + (size_t)nr1 * j] = M[r + (size_t)nr * c] / den;
    }
  double scale = 0.0, err = 0.0;
  for (size_t q = 0; q < (size_t)nr * nc; q++) scale = std::max(scale, std::fabs(M[q]));
  for (int c = 0; c < nc; c++)
    for (int r = 0; r < nr; r++)
    {
      double v = 1.0;
      int rr = r, cc = c;
      for (int d = 0; d < nd; d++)
      {
        v *= A[(rr % nr1) + (size_t)nr1 * (cc % nc1)];
        rr /= nr1;
        cc /= nc1;
      }
      err = std::max(err, std::fabs(v - M[r + (size_t)nr * c]));
    }
  return err <= tol * scale;
}

// hierarchical mode order of the reference's tensor Legendre basis -> tensor index (first direction fastest)
// (eval_legendre_basis_3D_hierarchical src/eles_hexas.cpp:1364, 2-D src/eles_quads.cpp:1116)
static void hierarchical_to_tensor(int nd, int N, std::vector<int> &t_of_h)
{
  const int p = N - 1;
  t_of_h.clear();
  if (nd == 3)
  {
    for (int l = 0; l < 3 * p + 1; l++)
      for (int k = 0; k < l + 1; k++)
        for (int j = 0; j < l - k + 1; j++)
        {
          const int i = l - k - j;
          if (i <= p && j <= p && k <= p) t_of_h.push_back(i + N * (j + N * k));
        }
  }
  else
  {
    for (int k = 0; k < 2 * p + 1; k++)
      for (int j = 0; j < k + 1; j++)
      {
        const int i = k - j;
        if (i <= p && j <= p) t_of_h.push_back(i + N * j);
      }
  }
}

struct TensorOps
{
  // over-integration
  bool over_int = false;
  int N = 0, Nc = 0;
  double *I1 = nullptr, *F1 = nullptr; // device: (Nc x N) interpolation, (N x Nc) projection
  std::vector<double> hF1;             // host copy of F1
  double *F1f = nullptr;               // device: n_dims matrices (N x Nc): the projection followed by a 1-D operator of the caller
  bool folded = false;
  // shock capturing
  bool shock = false;
  double *W1 = nullptr, *E1 = nullptr;         // device: (N x N) modal transform, (N x N) filter
  double *wnum = nullptr, *wden = nullptr;     // device: sensor weights per TENSOR mode
};

static void free_dev(double *&p)
{
  if (p) (void)hipFree(p);
  p = nullptr;
}

void tensor_ops_destroy(hfx_eles *e)
{
  TensorOps *T = (TensorOps *)e->tensor_ops;
  if (!T) return;
  for (double **p : {&T->I1, &T->F1, &T->F1f, &T->W1, &T->E1, &T->wnum, &T->wden}) free_dev(*p);
  delete T;
  e->tensor_ops = nullptr;
}

static TensorOps *ops_of(hfx_eles *e)
{
  if (!e->tensor_ops) e->tensor_ops = new TensorOps();
  return (TensorOps *)e->tensor_ops;
}

static int tensor_n1(const hfx_eles *e)
{
  if (e->ele_type != 4 && e->ele_type != 1) return 0;
  const int N = e->order + 1;
  return ipow_i(N, e->n_dims) == e->n_upts ? N : 0;
}

static int upload_d(double **dst, const std::vector<double> &v)
{
  free_dev(*dst);
  HFX_HIP(hipMalloc((void **)dst, sizeof(double) * std::max<size_t>(v.size(), 1)));
  HFX_HIP(hipMemcpy(*dst, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice));
  return 0;
}

static const double KRON_TOL = 1e-12;

int tensor_over_int_setup(hfx_eles *e, int n_cubpts, const double *opp_cub, const double *filter)
{
  TensorOps *T = ops_of(e);
  T->over_int = false;
  const int N = tensor_n1(e);
  if (!N) return 0;
  int Nc = 0;
  for (int c = 1; c <= 64; c++)
    if (ipow_i(c, e->n_dims) == n_cubpts) Nc = c;
  if (!Nc || Nc < N || Nc > 10 || N > 6) return 0; // instantiated: N = 2..6 (P1..P5), N <= Nc <= 10
  // LDS image: three regions of n_fields * Nc^nd doubles
  const size_t lds = sizeof(double) * 3 * (size_t)e->n_fields * n_cubpts;
  if (lds > 160 * 1024) return 0;
  std::vector<double> I1, F1;
  if (!kron_factor(opp_cub, e->n_dims, Nc, N, I1, KRON_TOL)) return 0;
  if (!kron_factor(filter, e->n_dims, N, Nc, F1, KRON_TOL)) return 0;
  if (upload_d(&T->I1, I1) || upload_d(&T->F1, F1)) return 1;
  T->hF1 = F1;
  T->folded = false;
  T->N = N;
  T->Nc = Nc;
  T->over_int = true;
  return 0;
}

// The fused stage consumes the de-aliased flux through ONE linear map per direction, res = sum_l A_l tdisf_l with
// A_l = opp_2[l] - opp_3 opp_1[l] (divergence minus the folded correction, fused_hex.hip), which on a tensor-product element acts
// along direction l only: a 1-D matrix Dc[l] (N x N, row-major, the caller's).  Composed with the projection,
// res = sum_l (F1 (x) .. Dc[l] F1 .. (x) F1) t_l: the SAME passes with another matrix in direction l, and the kernel writes
// n_fields values per solution point instead of n_fields * n_dims.
int tensor_over_int_set_fold(hfx_eles *e, const double *Dc)
{
  TensorOps *T = (TensorOps *)e->tensor_ops;
  HFX_CHECK(T && T->over_int, "over-integration: no tensor factors");
  const int N = T->N, Nc = T->Nc, nd = e->n_dims;
  std::vector<double> M((size_t)nd * N * Nc, 0.0);
  for (int d = 0; d < nd; d++)
    for (int c = 0; c < Nc; c++)
      for (int mp = 0; mp < N; mp++)
      {
        double s = 0.0;
        for (int m = 0; m < N; m++) s += Dc[((size_t)d * N + mp) * N + m] * T->hF1[m + (size_t)N * c];
        M[(size_t)d * N * Nc + mp + (size_t)N * c] = s;
      }
  if (upload_d(&T->F1f, M)) return 1;
  T->folded = true;
  return 0;
}
bool tensor_over_int_folded(const hfx_eles *e) { return tensor_over_int_available(e) && ((TensorOps *)e->tensor_ops)->folded; }

int tensor_shock_setup(hfx_eles *e, const double *inv_vandermonde, const double *exp_filter, const double *norm_basis,
                       const int *high_modes)
{
  TensorOps *T = ops_of(e);
  T->shock = false;
  const int N = tensor_n1(e);
  if (!N || N > 6) return 0; // instantiated: P1..P5
  const int nu = e->n_upts, nd = e->n_dims;
  std::vector<int> t_of_h;
  hierarchical_to_tensor(nd, N, t_of_h);
  if ((int)t_of_h.size() != nu) return 0;
  // rows of inv_vandermonde re-ordered from the hierarchical to the tensor mode order
  std::vector<double> Wt((size_t)nu * nu), wn(nu), wd(nu), W1, E1;
  for (int h = 0; h < nu; h++)
  {
    const int t = t_of_h[h];
    for (int pt = 0; pt < nu; pt++) Wt[t + (size_t)nu * pt] = inv_vandermonde[h + (size_t)nu * pt];
    wd[t] = norm_basis[h];
    wn[t] = high_modes[h] ? norm_basis[h] : 0.0;
  }
  if (!kron_factor(Wt.data(), nd, N, N, W1, KRON_TOL)) return 0;
  if (!kron_factor(exp_filter, nd, N, N, E1, KRON_TOL)) return 0;
  if (upload_d(&T->W1, W1) || upload_d(&T->E1, E1) || upload_d(&T->wnum, wn) || upload_d(&T->wden, wd)) return 1;
  T->N = N;
  T->shock = true;
  return 0;
}

bool tensor_over_int_available(const hfx_eles *e) { return e->ctx->opt.tensor_ops && e->tensor_ops && ((TensorOps *)e->tensor_ops)->over_int; }
bool tensor_shock_available(const hfx_eles *e) { return e->ctx->opt.tensor_ops && e->tensor_ops && ((TensorOps *)e->tensor_ops)->shock; }

// ---- device ---------------------------------------------------------------------------------------------------

// one pass of a tensor contraction, LDS -> LDS: the arrays are indexed (pre, i, post) with `pre` fastest (P values),
// the contracted index i (MI values) next, everything slower (higher directions, fields) in `post`;
//   out(pre, a, post) = sum_i M(a, i) in(pre, i, post),   M column-major (MO x MI)
// PENCIL-wise: a work item is one (pre, post); it reads its MI inputs from LDS once and produces the MO outputs with
// MO independent accumulators.  The extents are compile-time, so the pass is straight-line code; every lane walks
// the same (a, i) sequence, so M is read through the constant address space into scalar registers and costs no LDS
// bandwidth (the same arrangement as the flux kernel's sum-factorised contractions).
typedef const double __attribute__((address_space(4))) *tcptr;

// workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for every outstanding GLOBAL load and
// store of the wave (vmcnt(0)), i.e. drain the prefetches and expose the latency of each result store
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MO, int MI, typename MT>
__device__ __forceinline__ void tpass(const double *in, double *out, const MT &M, int P, int n_pencils)
{
  for (int w = threadIdx.x; w < n_pencils; w += blockDim.x)
  {
    const int pre = w % P, post = w / P;
    const double *ip = in + pre + P * MI * post;
    double *op = out + pre + P * MO * post;
    double x[MI];
#pragma unroll
    for (int i = 0; i < MI; i++) x[i] = ip[P * i];
#pragma unroll
    for (int a = 0; a < MO; a++)
    {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < MI; i++) acc += M[a + MO * i] * x[i];
      op[P * a] = acc;
    }
  }
}

// all ND passes of (NIN per direction) -> (NOUT per direction) on nf fields; ping-pongs between a and b and returns
// the buffer that holds the result.  Every pass ends in a barrier.  M is either a pointer into the constant address
// space (re-read by every pass: the barrier's memory clobber forbids keeping it) or a register array the caller loaded
// once -- wave-uniform values, i.e. scalar registers.
template <int ND, int NOUT, int NIN, typename MT>
__device__ __forceinline__ double *tapply(double *a, double *b, const MT &M, int nf)
{
  double *src = a, *dst = b;
  int P = 1;
#pragma unroll
  for (int d = 0; d < ND; d++)
  {
    int rest = nf;
#pragma unroll
    for (int q = d + 1; q < ND; q++) rest *= NIN;
    tpass<NOUT, NIN>(src, dst, M, P, P * rest);
    lds_sync();
    P *= NOUT;
    double *t = src;
    src = dst;
    dst = t;
  }
  return src;
}

constexpr int cpow(int b, int e) { return e == 0 ? 1 : b * cpow(b, e - 1); }

struct OverIntArgs
{
  long n_eles;
  const double *u, *JGc, *I1, *F1;
  double *tdisf;
  double gamma;
  const double *F1f; // FOLD: n_dims matrices (N x NC), the projection followed by the caller's 1-D operator of that direction
};

// the projection passes of flux direction `l` with the folded matrix in direction l
template <int ND, int N, int NC, typename MT>
__device__ __forceinline__ double *tproject_fold(double *a, double *b, const MT &F, const MT &Ff, int nf, int l)
{
  double *src = a, *dst = b;
  int P = 1;
#pragma unroll
  for (int d = 0; d < ND; d++)
  {
    int rest = nf;
#pragma unroll
    for (int q = d + 1; q < ND; q++) rest *= NC;
    if (d == l)
      tpass<N, NC>(src, dst, Ff, P, P * rest);
    else
      tpass<N, NC>(src, dst, F, P, P * rest);
    lds_sync();
    P *= N;
    double *t = src;
    src = dst;
    dst = t;
  }
  return src;
}

// threads of the over-integration workgroup (7^3 = 343 points on 6 waves, one point per thread, took 98 registers instead
// of 150 and 20.7 k cycles per element instead of 23.5 k, but with three 6-wave workgroups per CU the stage was not faster)
// (round 3, with the folded form: 384 threads 0.56 ms against 0.466; 256 threads compiled for four waves per SIMD -- 128 registers,
// 40 of them spilled -- 0.61 ms)
constexpr int oi_threads(int) { return 256; }

template <int ND, int N, int NC, bool FOLD = false>
__global__ __launch_bounds__(oi_threads(cpow(NC, ND))) void overint_tensor_kernel(const OverIntArgs a)
{
  constexpr int NF = ND + 2, NQ = ND * ND, nu = cpow(N, ND), nc = cpow(NC, ND);
  constexpr int THR = oi_threads(nc);
  constexpr int PPT = (nc + THR - 1) / THR; // cubature points per thread
  extern __shared__ double lds[];
  double *R0 = lds, *R1 = R0 + NF * nc, *R2 = R1 + NF * nc;
  const tcptr cI = (tcptr)(uintptr_t)a.I1, cF = (tcptr)(uintptr_t)a.F1;
  const long plane_u = (long)nu * a.n_eles;
  // software pipeline: the next element's state is requested while this one is worked on
  constexpr int UPT = (NF * nu + THR - 1) / THR;
  double pu[UPT];
  auto fetch_u = [&](long e) {
#pragma unroll
    for (int i = 0; i < UPT; i++)
    {
      const int q = threadIdx.x + THR * i;
      if (q < NF * nu)
      {
        const int k = q / nu, pt = q - k * nu;
        pu[i] = a.u[pt + (long)nu * e + k * plane_u];
      }
    }
  };
  // the metric tensors of this thread's cubature points travel the same way: those of the next element are requested as
  // soon as this element's transformed flux is formed, so that they are in flight during the projection passes
  double jg[PPT][NQ];
  auto fetch_jg = [&](long e) {
#pragma unroll
    for (int r = 0; r < PPT; r++)
    {
      const int q = threadIdx.x + THR * r;
      const double *src = a.JGc + NQ * ((long)(q < nc ? q : nc - 1) + (long)nc * e);
#pragma unroll
      for (int m = 0; m < NQ; m++) jg[r][m] = src[m];
    }
  };
  if ((long)blockIdx.x < a.n_eles)
  {
    fetch_u(blockIdx.x);
    fetch_jg(blockIdx.x);
  }
  for (long e = blockIdx.x; e < a.n_eles; e += gridDim.x)
  {
#pragma unroll
    for (int i = 0; i < UPT; i++)
      if (threadIdx.x + THR * i < NF * nu) R0[threadIdx.x + THR * i] = pu[i];
    lds_sync();
    if (e + gridDim.x < a.n_eles) fetch_u(e + gridDim.x);
    // solution at the cubature points (opp_over_int_cubpts . u)
    double *ucub = tapply<ND, NC, N>(R0, R1, cI, NF);
    double *fa = (ucub == R0) ? R1 : R0; // two free regions
    double *fb = R2;
    // the transformed Euler flux at this thread's cubature points, once: t(k, l) = sum_m JGinv(l, m) F(k, m)
    double tf[PPT][NF * ND];
#pragma unroll
    for (int r = 0; r < PPT; r++)
    {
      const int q = threadIdx.x + THR * r;
      if (q < nc)
      {
        double u[NF], fx[NF * ND];
#pragma unroll
        for (int k = 0; k < NF; k++) u[k] = ucub[q + nc * k];
        calc_invf<ND>(a.gamma, u, fx);
#pragma unroll
        for (int l = 0; l < ND; l++)
#pragma unroll
          for (int k = 0; k < NF; k++)
          {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < ND; m++) s += jg[r][l + ND * m] * fx[k + NF * m];
            tf[r][k + NF * l] = s;
          }
      }
    }
    if (e + gridDim.x < a.n_eles) fetch_jg(e + gridDim.x);
    double racc[FOLD ? UPT : 1];
    if constexpr (FOLD)
    {
#pragma unroll
      for (int i = 0; i < UPT; i++) racc[i] = 0.0;
    }
#pragma unroll
    for (int l = 0; l < ND; l++)
    {
#pragma unroll
      for (int r = 0; r < PPT; r++)
      {
        const int q = threadIdx.x + THR * r;
        if (q < nc)
        {
#pragma unroll
          for (int k = 0; k < NF; k++) fa[q + nc * k] = tf[r][k + NF * l];
        }
      }
      lds_sync();
      if constexpr (FOLD)
      {
        // projection followed by the caller's operator along l; the directions' results are summed in registers (l ascending)
        const tcptr cFf = (tcptr)(uintptr_t)(a.F1f + (size_t)l * N * NC);
        double *res = tproject_fold<ND, N, NC>(fa, fb, cF, cFf, NF, l);
#pragma unroll
        for (int i = 0; i < UPT; i++)
        {
          const int q = threadIdx.x + THR * i;
          if (q < NF * nu) racc[i] += res[q];
        }
        lds_sync();
        continue;
      }
      // projection back on the solution points (over_int_filter . t)
      double *res = tapply<ND, N, NC>(fa, fb, cF, NF);
      for (int q = threadIdx.x; q < NF * nu; q += blockDim.x)
      {
        const int k = q / nu, pt = q - k * nu;
        a.tdisf[pt + (long)nu * e + (k + NF * l) * plane_u] = res[q];
      }
      lds_sync();
    }
    if constexpr (FOLD)
    {
#pragma unroll
      for (int i = 0; i < UPT; i++)
      {
        const int q = threadIdx.x + THR * i;
        if (q < NF * nu)
        {
          const int k = q / nu, pt = q - k * nu;
          a.tdisf[pt + (long)nu * e + k * plane_u] = racc[i];
        }
      }
    }
  }
}

constexpr int TMAX = 10; // largest 1-D extent instantiated

template <int ND, int N, int NC, bool FOLD>
static int oi_launch_form(hfx_eles *e, const OverIntArgs &a, size_t lds, int grid)
{
  HFX_HIP(hipFuncSetAttribute((const void *)overint_tensor_kernel<ND, N, NC, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // persistent grid: exactly the workgroups that are resident at once (registers and LDS decide), so that no workgroup
  // waits for a slot while the others are half way through their elements
  int per_cu = 0;
  HFX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, overint_tensor_kernel<ND, N, NC, FOLD>, oi_threads(cpow(NC, ND)), lds));
  grid = (int)std::min<long>(e->n_eles, (long)e->ctx->n_cu * std::max(per_cu, 1));
  hipLaunchKernelGGL((overint_tensor_kernel<ND, N, NC, FOLD>), dim3(grid), dim3(oi_threads(cpow(NC, ND))), lds, e->ctx->stream, a);
  HFX_HIP(hipGetLastError());
  return 0;
}

template <int ND, int N, int NC>
static int oi_launch_one(hfx_eles *e, const OverIntArgs &a, size_t lds, int grid)
{
  return a.F1f ? oi_launch_form<ND, N, NC, true>(e, a, lds, grid) : oi_launch_form<ND, N, NC, false>(e, a, lds, grid);
}

template <int ND, int N, int NC>
static int oi_pick_nc(hfx_eles *e, const OverIntArgs &a, size_t lds, int grid, int Nc)
{
  if (Nc == NC) return oi_launch_one<ND, N, NC>(e, a, lds, grid);
  if constexpr (NC < TMAX)
    return oi_pick_nc<ND, N, NC + 1>(e, a, lds, grid, Nc);
  else
  {
    set_error("over-integration: no kernel for %d cubature points per direction", Nc);
    return 1;
  }
}

template <int ND, int N>
static int oi_pick_n(hfx_eles *e, const OverIntArgs &a, size_t lds, int grid, int n, int Nc)
{
  if (n == N) return oi_pick_nc<ND, N, N>(e, a, lds, grid, Nc);
  if constexpr (N < 6)
    return oi_pick_n<ND, N + 1>(e, a, lds, grid, n, Nc);
  else
  {
    set_error("over-integration: no kernel for order %d", n - 1);
    return 1;
  }
}

int tensor_over_int_launch(hfx_eles *e, bool folded)
{
  TensorOps *T = (TensorOps *)e->tensor_ops;
  HFX_CHECK(T && T->over_int, "over-integration: no tensor factors");
  HFX_CHECK(!folded || T->folded, "over-integration: the folded form was not set up");
  OverIntArgs a{};
  a.n_eles = e->n_eles;
  a.u = e->arr[HFX_DISU_UPTS0]; a.JGc = e->JGinv_over_int_cubpts; a.I1 = T->I1; a.F1 = T->F1;
  a.F1f = folded ? T->F1f : nullptr;
  // (folded: n_fields values per solution point, in the first n_fields planes of tdisf_upts)
  a.tdisf = e->arr[HFX_TDISF_UPTS];
  a.gamma = e->ctx->params.gamma;
  // two regions of n_fields * Nc^nd doubles and the projection's intermediate, n_fields * N * Nc^(nd-1)
  const size_t lds = sizeof(double) * (size_t)e->n_fields * (2 * (size_t)e->n_cubpts + (size_t)T->N * (e->n_cubpts / T->Nc));
  const int per_cu = std::max(1, (int)((160 * 1024) / lds));
  const int grid = (int)std::min<long>(e->n_eles, (long)e->ctx->n_cu * std::min(per_cu, 8));
  return e->n_dims == 2 ? oi_pick_n<2, 2>(e, a, lds, grid, T->N, T->Nc) : oi_pick_n<3, 2>(e, a, lds, grid, T->N, T->Nc);
}

struct ShockArgs
{
  long n_eles;
  int field;
  double *u;
  const double *W1, *E1, *wnum, *wden;
  double *sensor;
  double s0;
  // optional: the flux-point solution of the filtered elements (opp_0 in ELL form), for callers that keep disu_fpts
  // of the current state (the split fused paths) -- saves them a full extrapolate_solution after the filter
  double *disu_fpts;
  const double *o0v;
  const int *o0i;
  int o0w, nfp;
};

// Persson sensor of every element and, where it fires, the exponentially filtered state -- one workgroup per element
template <int ND, int N>
__global__ __launch_bounds__(256) void shock_tensor_kernel(const ShockArgs a)
{
  constexpr int NF = ND + 2, nu = cpow(N, ND);
  __shared__ double R0[NF * nu], R1[NF * nu], red[8];
  const tcptr sW = (tcptr)(uintptr_t)a.W1, sE = (tcptr)(uintptr_t)a.E1;
  const long plane_u = (long)nu * a.n_eles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a thread's sensor weights once, and the sensor field of the NEXT element requested while this one is worked on: an element's
  // chain was three memory latencies long (field, weights, nothing else to do in between) for a few hundred multiply-adds
  constexpr int QPT = (nu + 255) / 256;
  double wn[QPT], wd[QPT], pf[QPT];
  // the modal transform's 1-D matrix stays in (scalar) registers across the elements: every pass re-read it through the scalar
  // cache otherwise, N * N doubles behind a wait, three times per element
  double Wm[N * N];
#pragma unroll
  for (int i = 0; i < N * N; i++) Wm[i] = sW[i];
#pragma unroll
  for (int i = 0; i < QPT; i++)
  {
    const int q = threadIdx.x + 256 * i, qc = q < nu ? q : nu - 1;
    wn[i] = a.wnum[qc];
    wd[i] = a.wden[qc];
    pf[i] = ((long)blockIdx.x < a.n_eles) ? a.u[qc + (long)nu * blockIdx.x + a.field * plane_u] : 0.0;
  }
  for (long e = blockIdx.x; e < a.n_eles; e += gridDim.x)
  {
    lds_sync();
#pragma unroll
    for (int i = 0; i < QPT; i++)
      if (threadIdx.x + 256 * i < nu) R0[threadIdx.x + 256 * i] = pf[i];
    lds_sync();
    {
      const long en = e + gridDim.x < a.n_eles ? e + gridDim.x : e;
#pragma unroll
      for (int i = 0; i < QPT; i++)
      {
        const int q = threadIdx.x + 256 * i, qc = q < nu ? q : nu - 1;
        pf[i] = a.u[qc + (long)nu * en + a.field * plane_u];
      }
    }
    // modal coefficients of the sensor field (tensor mode order)
    const double *modal = tapply<ND, N, N>(R0, R1, Wm, 1);
    double sn = 0.0, sd = 0.0;
#pragma unroll
    for (int i = 0; i < QPT; i++)
    {
      const int q = threadIdx.x + 256 * i;
      if (q < nu)
      {
        const double m2 = modal[q] * modal[q];
        sn += m2 * wn[i];
        sd += wd[i] * m2;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
      sn += __shfl_down(sn, off, 64);
      sd += __shfl_down(sd, off, 64);
    }
    if (lane == 0)
    {
      red[2 * wave] = sn;
      red[2 * wave + 1] = sd;
    }
    lds_sync();
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++)
    {
      num += red[2 * w];
      den += red[2 * w + 1];
    }
    const double s = num / den;
    if (threadIdx.x == 0) a.sensor[e] = s;
    if (s >= a.s0) // uniform over the workgroup (src/eles.cpp:2936)
    {
      lds_sync();
      for (int q = threadIdx.x; q < NF * nu; q += blockDim.x)
      {
        const int k = q / nu, pt = q - k * nu;
        R0[q] = a.u[pt + (long)nu * e + k * plane_u];
      }
      lds_sync();
      const double *res = tapply<ND, N, N>(R0, R1, sE, NF);
      for (int q = threadIdx.x; q < NF * nu; q += blockDim.x)
      {
        const int k = q / nu, pt = q - k * nu;
        a.u[pt + (long)nu * e + k * plane_u] = res[q];
      }
      if (a.disu_fpts)
      {
        // disu_fpts = opp_0 . u of the filtered state, rows in the ascending-column order of the contraction kernels
        const long plane_f = (long)a.nfp * a.n_eles;
        for (int q = threadIdx.x; q < NF * a.nfp; q += blockDim.x)
        {
          const int k = q / a.nfp, r = q - k * a.nfp;
          double acc = 0.0;
          for (int w = 0; w < a.o0w; w++) acc += a.o0v[r + (long)a.nfp * w] * res[k * nu + a.o0i[r + (long)a.nfp * w]];
          a.disu_fpts[r + (long)a.nfp * e + k * plane_f] = acc;
        }
      }
    }
  }
}

template <int ND, int N>
static int shock_pick_n(hfx_eles *e, const ShockArgs &a, int grid, int n)
{
  if (n == N)
  {
    // persistent grid: the workgroups resident at once, or all of a small mesh
    int per_cu = 0;
    HFX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shock_tensor_kernel<ND, N>, 256, 0));
    grid = (int)std::min<long>(grid, (long)e->ctx->n_cu * std::max(per_cu, 1));
    hipLaunchKernelGGL((shock_tensor_kernel<ND, N>), dim3(grid), dim3(256), 0, e->ctx->stream, a);
    HFX_HIP(hipGetLastError());
    return 0;
  }
  if constexpr (N < 6)
    return shock_pick_n<ND, N + 1>(e, a, grid, n);
  else
  {
    set_error("shock capturing: no kernel for order %d", n - 1);
    return 1;
  }
}

int tensor_shock_launch(hfx_eles *e, bool refresh_disu_fpts)
{
  TensorOps *T = (TensorOps *)e->tensor_ops;
  HFX_CHECK(T && T->shock, "shock capturing: no tensor factors");
  ShockArgs a{};
  a.n_eles = e->n_eles;
  a.field = e->shock_det_field == 0 ? 0 : e->n_dims + 1;
  a.u = e->arr[HFX_DISU_UPTS0];
  a.W1 = T->W1; a.E1 = T->E1; a.wnum = T->wnum; a.wden = T->wden;
  a.sensor = e->arr[HFX_SENSOR];
  a.s0 = e->s0;
  if (refresh_disu_fpts)
  {
    a.disu_fpts = e->arr[HFX_DISU_FPTS];
    a.o0v = e->opp_0.ell_val; a.o0i = e->opp_0.ell_idx; a.o0w = std::max(e->opp_0.nnz_max, 1); a.nfp = e->n_fpts;
  }
  const int grid = (int)std::min<long>(e->n_eles, (long)e->ctx->n_cu * 8);
  return e->n_dims == 2 ? shock_pick_n<2, 2>(e, a, grid, T->N) : shock_pick_n<3, 2>(e, a, grid, T->N);
}

} // namespace hfx
