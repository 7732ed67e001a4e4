// hfx.hip -- C ABI of libhfx (include/hfx.h): handles, registration, per-method
// launchers mirroring eles::* / int_inters::* of the reference, and the
// CalcResidual / RK-loop drivers.  gfx950 only.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

#include "fused_hex.hpp"
#include "general.hpp"
#include "tensor_ops.hpp"
#include "hfx_internal.hpp"
#include "kernels_ops.hpp"
#include "kernels_bdy.hpp"
#include "kernels_mpi.hpp"
#include "kernels_point.hpp"

namespace hfx
{
static thread_local std::string g_err;
void set_error(const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

static int dev_alloc_copy(double **dst, const double *src, long n)
{
  HFX_HIP(hipMalloc((void **)dst, sizeof(double) * (size_t)std::max<long>(n, 1)));
  if (src) HFX_HIP(hipMemcpy(*dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  return 0;
}

// ---- operator registration -------------------------------------------------
static int make_operator(Operator &op, const double *host, int m, int k)
{
  op.m = m;
  op.k = k;
  if (dev_alloc_copy(&op.dense, host, (long)m * k)) return 1;
  {
    op.mpad = (m + 15) & ~15;
    op.kpad = (k + 3) & ~3;
    std::vector<double> pad((size_t)op.mpad * op.kpad, 0.0);
    for (int c = 0; c < k; c++)
      for (int r = 0; r < m; r++) pad[r + (size_t)op.mpad * c] = host[r + (long)m * c];
    if (dev_alloc_copy(&op.dense_pad, pad.data(), (long)pad.size())) return 1;
  }
  // ELL: exact non-zeros, ascending column
  int nnz_max = 0;
  long total = 0;
  std::vector<int> cnt(m, 0);
  for (int r = 0; r < m; r++)
  {
    for (int c = 0; c < k; c++)
      if (host[r + (long)m * c] != 0.0) cnt[r]++;
    nnz_max = std::max(nnz_max, cnt[r]);
    total += cnt[r];
  }
  op.nnz_max = nnz_max;
  op.nnz_total = total;
  const int w = std::max(nnz_max, 1);
  std::vector<double> val((size_t)m * w, 0.0);
  std::vector<int> idx((size_t)m * w, 0);
  for (int r = 0; r < m; r++)
  {
    int q = 0, first = 0;
    for (int c = 0; c < k; c++)
      if (host[r + (long)m * c] != 0.0)
      {
        if (q == 0) first = c;
        val[r + (size_t)m * q] = host[r + (long)m * c];
        idx[r + (size_t)m * q] = c;
        q++;
      }
    for (; q < w; q++)
    {
      val[r + (size_t)m * q] = 0.0;
      idx[r + (size_t)m * q] = first;
    }
  }
  HFX_HIP(hipMalloc((void **)&op.ell_val, sizeof(double) * val.size()));
  HFX_HIP(hipMalloc((void **)&op.ell_idx, sizeof(int) * idx.size()));
  HFX_HIP(hipMemcpy(op.ell_val, val.data(), sizeof(double) * val.size(), hipMemcpyHostToDevice));
  HFX_HIP(hipMemcpy(op.ell_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
  op.h_val = val;
  op.h_idx = idx;
  return 0;
}

static void free_operator(Operator &op)
{
  if (op.dense) (void)hipFree(op.dense);
  if (op.dense_pad) (void)hipFree(op.dense_pad);
  if (op.ell_val) (void)hipFree(op.ell_val);
  if (op.ell_idx) (void)hipFree(op.ell_idx);
  op = Operator();
}

constexpr int ELL_MAX_NNZ = 8;

// sparse when every term has <= ELL_MAX_NNZ non-zeros per row (tensor-product
// collocation gives <= order+2); anything else is a genuinely dense operator
static bool use_sparse(const hfx_ctx *ctx, const Operator *const *ops, int n)
{
  if (ctx->contract_mode == HFX_CONTRACT_DENSE) return false;
  for (int i = 0; i < n; i++)
    if (ops[i]->nnz_max > ELL_MAX_NNZ) return false;
  return true;
}

template <int NNZ, int NT, int NO>
static int launch_ell_t(hfx_ctx *ctx, EllArgs<NT, NO> &a)
{
  // threads: rg row groups of m rows, rounded up to whole waves
  const int m = a.m;
  int best_rg = 1, best_bd = ((m + 63) / 64) * 64;
  double best_eff = (double)m / best_bd;
  for (int rg = 2; rg <= 16; rg++)
  {
    const int bd = ((rg * m + 63) / 64) * 64;
    if (bd > 512) break;
    const double eff = (double)(rg * m) / bd;
    if (eff > best_eff + 1e-9)
    {
      best_eff = eff;
      best_rg = rg;
      best_bd = bd;
    }
  }
  HFX_CHECK(best_bd <= 1024, "operator with %d rows does not fit one workgroup", m);
  a.rg = best_rg;
  // columns per tile: ~40 kB of LDS for the input tile(s), a multiple of rg
  const long bytes_per_col = (long)NT * a.k * sizeof(double);
  int ct = (int)std::max<long>(1, (40 * 1024) / bytes_per_col);
  ct = std::max(best_rg, (ct / best_rg) * best_rg);
  ct = (int)std::min<long>(ct, std::max<long>(a.ncols, 1));
  a.ct = ct;
  const size_t lds = (size_t)NT * a.k * ct * sizeof(double);
  HFX_CHECK(lds <= 160 * 1024, "ELL tile does not fit LDS");
  const long nblk = (a.ncols + ct - 1) / ct;
  if (nblk == 0) return 0;
  if (lds > 48 * 1024)
    HFX_HIP(hipFuncSetAttribute((const void *)ell_apply_kernel<NNZ, NT, NO>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  hipLaunchKernelGGL((ell_apply_kernel<NNZ, NT, NO>), dim3((unsigned)nblk), dim3(best_bd), lds, ctx->stream, a);
  HFX_HIP(hipGetLastError());
  return 0;
}

template <int NT, int NO>
static int launch_ell(hfx_ctx *ctx, EllArgs<NT, NO> &a, int nnz)
{
  switch (std::max(nnz, 1))
  {
  case 1: return launch_ell_t<1, NT, NO>(ctx, a);
  case 2: return launch_ell_t<2, NT, NO>(ctx, a);
  case 3: return launch_ell_t<3, NT, NO>(ctx, a);
  case 4: return launch_ell_t<4, NT, NO>(ctx, a);
  case 5: return launch_ell_t<5, NT, NO>(ctx, a);
  case 6: return launch_ell_t<6, NT, NO>(ctx, a);
  case 7: return launch_ell_t<7, NT, NO>(ctx, a);
  case 8: return launch_ell_t<8, NT, NO>(ctx, a);
  }
  set_error("launch_ell: nnz %d out of range", nnz);
  return 1;
}

static int launch_dense(hfx_ctx *ctx, const Operator &op, const double *B, double *C, long ncols, int beta,
                        const double *sub = nullptr, double *wb = nullptr, unsigned long long *nan_flag = nullptr)
{
  DenseArgs a;
  a.m = op.m;
  a.k = op.k;
  a.ncols = ncols;
  a.beta = beta;
  a.A = op.dense;
  a.Ap = op.dense_pad;
  a.mpad = op.mpad;
  a.B = B;
  a.C = C;
  a.sub = sub;
  a.in_writeback = wb;
  a.nan_flag = nan_flag;
  const int kpad = (op.k + 3) & ~3;
  // the B tile holds all k rows of its columns: narrower tiles for operators with many columns
  int ct = DENSE_CT;
  while (ct > 16 && (size_t)kpad * (ct + 1) * sizeof(double) > 160 * 1024) ct >>= 1;
  const size_t lds = (size_t)kpad * (ct + 1) * sizeof(double);
  HFX_CHECK(lds <= 160 * 1024, "dense contraction: k = %d does not fit LDS", op.k);
  const long nblk = (ncols + ct - 1) / ct;
  if (nblk == 0) return 0;
  // shape of the workgroup: operators with many 16-row tiles keep eight waves busy on half-tiles; small ones (simplex
  // classes: 2-5 row tiles) deal quarter-tiles to four waves (measured)
  const int n_rt = (op.m + 15) / 16;
  int nw = ctx->opt.dense_waves ? ctx->opt.dense_waves : (n_rt >= 6 ? 8 : 4);
  int spl = ctx->opt.dense_split ? ctx->opt.dense_split : (n_rt >= 6 ? 2 : 4);
#define HFX_DENSE_L(CT_, NW_, SP_)                                                                                      \
  {                                                                                                                     \
    if (lds > 48 * 1024)                                                                                                \
      HFX_HIP(hipFuncSetAttribute((const void *)dense_mfma_kernel<CT_, NW_, SP_>,                                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
    hipLaunchKernelGGL((dense_mfma_kernel<CT_, NW_, SP_>), dim3((unsigned)nblk), dim3(64 * NW_), lds, ctx->stream, a);  \
  }
#define HFX_DENSE_S(CT_, NW_)                                                                                           \
  {                                                                                                                     \
    if (spl == 1) HFX_DENSE_L(CT_, NW_, 1) else if (spl == 2) HFX_DENSE_L(CT_, NW_, 2) else HFX_DENSE_L(CT_, NW_, 4)    \
  }
#define HFX_DENSE(CT_)                                                                                                  \
  {                                                                                                                     \
    if (nw == 8) HFX_DENSE_S(CT_, 8) else HFX_DENSE_S(CT_, 4)                                                           \
  }
  if (ct == 64) HFX_DENSE(64) else if (ct == 32) HFX_DENSE(32) else HFX_DENSE(16)
#undef HFX_DENSE_L
#undef HFX_DENSE_S
#undef HFX_DENSE
  HFX_HIP(hipGetLastError());
  return 0;
}

// out = [beta out] + sum_t op[t] * in[t]   (NT terms; in[t] are slabs of k x ncols)
static int contract_multi_in(hfx_ctx *ctx, const Operator *const *ops, int nt, const double *const *in, double *out,
                             long ncols, int beta, const double *sub = nullptr, double *wb = nullptr,
                             unsigned long long *nan_flag = nullptr)
{
  if (use_sparse(ctx, ops, nt))
  {
    int nnz = 0;
    for (int t = 0; t < nt; t++) nnz = std::max(nnz, ops[t]->nnz_max);
    // pad terms with fewer non-zeros: their ELL arrays are only nnz_max(t) wide, so
    // fall back to one launch per term when widths differ
    bool same = true;
    for (int t = 0; t < nt; t++) same = same && (std::max(ops[t]->nnz_max, 1) == std::max(nnz, 1));
    if (same && nt == 1)
    {
      EllArgs<1, 1> a{};
      a.m = ops[0]->m; a.k = ops[0]->k; a.ncols = ncols; a.beta = beta;
      a.val[0] = ops[0]->ell_val; a.idx[0] = ops[0]->ell_idx; a.in[0] = in[0]; a.out[0] = out;
      a.sub = sub; a.in_writeback = wb; a.nan_flag = nan_flag;
      return launch_ell<1, 1>(ctx, a, nnz);
    }
    if (same && nt == 2 && !sub)
    {
      EllArgs<2, 1> a{};
      a.m = ops[0]->m; a.k = ops[0]->k; a.ncols = ncols; a.beta = beta;
      for (int t = 0; t < 2; t++) { a.val[t] = ops[t]->ell_val; a.idx[t] = ops[t]->ell_idx; a.in[t] = in[t]; }
      a.out[0] = out; a.nan_flag = nan_flag;
      return launch_ell<2, 1>(ctx, a, nnz);
    }
    if (same && nt == 3 && !sub)
    {
      EllArgs<3, 1> a{};
      a.m = ops[0]->m; a.k = ops[0]->k; a.ncols = ncols; a.beta = beta;
      for (int t = 0; t < 3; t++) { a.val[t] = ops[t]->ell_val; a.idx[t] = ops[t]->ell_idx; a.in[t] = in[t]; }
      a.out[0] = out; a.nan_flag = nan_flag;
      return launch_ell<3, 1>(ctx, a, nnz);
    }
    for (int t = 0; t < nt; t++)
    {
      EllArgs<1, 1> a{};
      a.m = ops[t]->m; a.k = ops[t]->k; a.ncols = ncols; a.beta = (t == 0) ? beta : 1;
      a.val[0] = ops[t]->ell_val; a.idx[0] = ops[t]->ell_idx; a.in[0] = in[t]; a.out[0] = out;
      if (t == 0) { a.sub = sub; a.in_writeback = wb; }
      if (t == nt - 1) a.nan_flag = nan_flag;
      if (launch_ell<1, 1>(ctx, a, ops[t]->nnz_max)) return 1;
    }
    return 0;
  }
  for (int t = 0; t < nt; t++)
    if (launch_dense(ctx, *ops[t], in[t], out, ncols, (t == 0) ? beta : 1, (t == 0) ? sub : nullptr,
                     (t == 0) ? wb : nullptr, (t == nt - 1) ? nan_flag : nullptr))
      return 1;
  return 0;
}

// out[o] = [beta out[o]] + op[o] * in   (NO outputs share one input)
static int contract_multi_out(hfx_ctx *ctx, const Operator *const *ops, int no, const double *in, double *const *out,
                              long ncols, int beta)
{
  if (use_sparse(ctx, ops, no))
  {
    int nnz = 0;
    for (int o = 0; o < no; o++) nnz = std::max(nnz, ops[o]->nnz_max);
    bool same = true;
    for (int o = 0; o < no; o++) same = same && (std::max(ops[o]->nnz_max, 1) == std::max(nnz, 1));
    if (same && no == 2)
    {
      EllArgs<1, 2> a{};
      a.m = ops[0]->m; a.k = ops[0]->k; a.ncols = ncols; a.beta = beta; a.in[0] = in;
      for (int o = 0; o < 2; o++) { a.val[o] = ops[o]->ell_val; a.idx[o] = ops[o]->ell_idx; a.out[o] = out[o]; }
      return launch_ell<1, 2>(ctx, a, nnz);
    }
    if (same && no == 3)
    {
      EllArgs<1, 3> a{};
      a.m = ops[0]->m; a.k = ops[0]->k; a.ncols = ncols; a.beta = beta; a.in[0] = in;
      for (int o = 0; o < 3; o++) { a.val[o] = ops[o]->ell_val; a.idx[o] = ops[o]->ell_idx; a.out[o] = out[o]; }
      return launch_ell<1, 3>(ctx, a, nnz);
    }
    for (int o = 0; o < no; o++)
    {
      EllArgs<1, 1> a{};
      a.m = ops[o]->m; a.k = ops[o]->k; a.ncols = ncols; a.beta = beta;
      a.val[0] = ops[o]->ell_val; a.idx[0] = ops[o]->ell_idx; a.in[0] = in; a.out[0] = out[o];
      if (launch_ell<1, 1>(ctx, a, ops[o]->nnz_max)) return 1;
    }
    return 0;
  }
  for (int o = 0; o < no; o++)
    if (launch_dense(ctx, *ops[o], in, out[o], ncols, beta)) return 1;
  return 0;
}

static inline unsigned nblocks(long n, int b) { return (unsigned)((n + b - 1) / b); }

// ---- the side stream of the boundary-face kernels (hfx_ctx::side_stream) ----
int side_stream_fork(hfx_ctx *ctx)
{
  if (!ctx->side_stream)
  {
    HFX_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    HFX_HIP(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
    HFX_HIP(hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming));
  }
  HFX_HIP(hipEventRecord(ctx->side_fork, ctx->stream));
  HFX_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
  ctx->bdy_stream = ctx->side_stream;
  return 0;
}
int side_stream_join(hfx_ctx *ctx)
{
  ctx->bdy_stream = nullptr;
  HFX_HIP(hipEventRecord(ctx->side_done, ctx->side_stream));
  return 0;
}
int side_stream_wait(hfx_ctx *ctx)
{
  HFX_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
  return 0;
}

} // namespace hfx

namespace hfx
{
// The squared length scale of the eddy-viscosity closures at every solution point, min(y^2 Kappa^2, C_s^2 Delta^2) for the damped
// Smagorinsky model and C_s^2 Delta^2 for WALE (src/eles.cpp:2436-2520, Delta = filter_ratio vol^(1/n_dims) / (order + 1) with
// vol = detjac * the reference element's volume): it depends on the metrics and the wall distance only, so the fused stages' flux
// kernels read it (calc_sgsf_fast) instead of evaluating a cube root per point and stage.  Evaluated once on the host with the
// reference's own expression; *dst: a new device array (n_upts, n_eles).
int les_len2_upload(hfx_eles *e, double **dst)
{
  const long plane = (long)e->n_upts * e->n_eles;
  std::vector<double> dj(plane), len2(plane), wd;
  HFX_HIP(hipMemcpy(dj.data(), e->detjac_upts, sizeof(double) * plane, hipMemcpyDeviceToHost));
  const int nd = e->n_dims;
  if (e->les.sgs_model == 0)
  {
    HFX_CHECK(e->wall_distance, "the Smagorinsky closure needs wall_distance");
    wd.resize((size_t)plane * nd);
    HFX_HIP(hipMemcpy(wd.data(), e->wall_distance, sizeof(double) * wd.size(), hipMemcpyDeviceToHost));
  }
  for (long p = 0; p < plane; p++)
  {
    const double vol = dj[p] * e->les.vol_factor;
    const double delta = e->les.filter_ratio * std::pow(vol, 1. / nd) / (e->les.order + 1.);
    double l2 = e->les.C_s * e->les.C_s * delta * delta;
    if (e->les.sgs_model == 0)
    {
      double y = 0.0;
      for (int i = 0; i < nd; i++) y += wd[p + (size_t)i * plane] * wd[p + (size_t)i * plane];
      y = std::sqrt(y);
      l2 = std::fmin(y * y * e->les.Kappa * e->les.Kappa, l2);
    }
    len2[p] = l2;
  }
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  HFX_HIP(hipMalloc((void **)dst, sizeof(double) * std::max<long>(plane, 1)));
  HFX_HIP(hipMemcpy(*dst, len2.data(), sizeof(double) * plane, hipMemcpyHostToDevice));
  return 0;
}

} // namespace hfx

using namespace hfx;

// ===========================================================================
extern "C" {

const char *hfx_last_error(void) { return g_err.c_str(); }
int hfx_version(void) { return 1; }

int hfx_ctx_create(int device, hfx_ctx **out)
{
  HFX_CHECK(out != nullptr, "hfx_ctx_create: out is NULL");
  int n = 0;
  HFX_HIP(hipGetDeviceCount(&n));
  HFX_CHECK(device >= 0 && device < n, "hfx_ctx_create: device %d of %d", device, n);
  HFX_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  HFX_HIP(hipGetDeviceProperties(&prop, device));
  HFX_CHECK(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0,
            "libhfx is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
  hfx_ctx *c = new hfx_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  HFX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return 0;
}

int hfx_ctx_destroy(hfx_ctx *ctx)
{
  if (!ctx) return 0;
  if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
  if (ctx->side_done) (void)hipEventDestroy(ctx->side_done);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

int hfx_ctx_set_params(hfx_ctx *ctx, const hfx_params *p)
{
  HFX_CHECK(ctx && p, "hfx_ctx_set_params: NULL argument");
  HFX_CHECK(p->riemann_solve_type == 0 || p->riemann_solve_type == 2 || p->riemann_solve_type == 3,
            "Riemann solver not implemented"); /* src/int_inters.cpp:210 */
  HFX_CHECK(p->vis_riemann_solve_type == 0, "Viscous Riemann solver not implemented"); /* src/int_inters.cpp:233 */
  HFX_CHECK(p->adv_type >= 0 && p->adv_type <= 4, "ERROR: Time integration type not recognised ... "); /* src/eles.cpp:1262 */
  HFX_CHECK(p->n_rk >= 0 && p->n_rk <= 16, "hfx_ctx_set_params: n_rk out of range");
  HFX_IMMEDIATE(ctx, 0); // (what has been recorded runs with the parameters it was recorded under)
  ctx->params = *p;
  ctx->have_params = true;
  return 0;
}

int hfx_ctx_set_contract_mode(hfx_ctx *ctx, int mode)
{
  HFX_CHECK(ctx, "NULL ctx");
  HFX_CHECK(mode >= 0 && mode <= 2, "hfx_ctx_set_contract_mode: bad mode %d", mode);
  HFX_IMMEDIATE(ctx, 0);
  ctx->contract_mode = mode;
  return 0;
}

int hfx_ctx_set_fused_mode(hfx_ctx *ctx, int mode)
{
  HFX_CHECK(ctx, "NULL ctx");
  HFX_CHECK(mode == 2 || mode == 3, "hfx_ctx_set_fused_mode: mode must be 2 (split, the reference's arrays kept) or 3 (split, fluxes in the gradient kernel)");
  HFX_IMMEDIATE(ctx, 0);
  ctx->defer.plans.clear();
  ctx->fused_mode = mode;
  return 0;
}

int hfx_ctx_set_CFL(hfx_ctx *ctx, double CFL)
{
  HFX_CHECK(ctx, "NULL ctx");
  HFX_CHECK(CFL > 0.0, "hfx_ctx_set_CFL: CFL must be positive");
  HFX_IMMEDIATE(ctx, 0);
  ctx->CFL = CFL;
  ctx->have_CFL = true;
  return 0;
}

int hfx_ctx_set_option(hfx_ctx *ctx, const char *name, int value)
{
  HFX_CHECK(ctx && name, "hfx_ctx_set_option: NULL argument");
  HFX_IMMEDIATE(ctx, 0);
  ctx->defer.plans.clear();
  hfx_ctx::Options &o = ctx->opt;
  const std::string n(name);
  if (n == "deferred") ctx->defer.on = value != 0;
  else if (n == "split_grid_per_cu") { HFX_CHECK(value >= 0, "split_grid_per_cu must be >= 0"); o.split_grid_per_cu = value; }
  else if (n == "flux_grid_per_cu") { HFX_CHECK(value >= 0, "flux_grid_per_cu must be >= 0"); o.flux_grid_per_cu = value; }
  else if (n == "xcd_order") o.xcd_order = value != 0;
  else if (n == "over_int_fold") o.over_int_fold = value != 0;
  else if (n == "general_update_waves") { HFX_CHECK(value == 0 || value == 4 || value == 8, "general_update_waves must be 0, 4 or 8"); o.general_update_waves = value; }
  else if (n == "dictionary_rows") o.dictionary_rows = value != 0;
  else if (n == "flux_waves") { HFX_CHECK(value == 2 || value == 3, "flux_waves must be 2 or 3"); o.flux_waves = value; }
  else if (n == "buffer_addressing") o.buffer_addressing = value != 0;
  else if (n == "loader_wave") o.loader_wave = value != 0;
  else if (n == "gather_delta") o.gather_delta = value != 0;
  else if (n == "fold_general") o.fold_general = value != 0;
  else if (n == "comm_stream_faces") o.comm_stream_faces = value != 0;
  else if (n == "split_update") o.split_update = value != 0;
  else if (n == "split_flux") o.split_flux = value != 0;
  else if (n == "flux_stamps") o.flux_stamps = value;
  else if (n == "simd_roles") o.simd_roles = value != 0;
  else if (n == "les_flux_kernel") o.les_flux_kernel = value != 0;
  else if (n == "bdy_beside") o.bdy_beside = value != 0;
  else if (n == "light_wave_short") o.light_wave_short = value != 0;
  else if (n == "tensor_ops") o.tensor_ops = value != 0;
  else if (n == "dense_waves") { HFX_CHECK(value == 0 || value == 4 || value == 8, "dense_waves must be 0, 4 or 8"); o.dense_waves = value; }
  else if (n == "dense_split") { HFX_CHECK(value == 0 || value == 1 || value == 2 || value == 4, "dense_split must be 0, 1, 2 or 4"); o.dense_split = value; }
  else if (n == "general_waves") { HFX_CHECK(value == 0 || value == 3 || value == 4 || value == 8, "general_waves must be 0, 3, 4 or 8"); o.general_waves = value; }
  else HFX_CHECK(false, "hfx_ctx_set_option: unknown option %s", name);
  return 0;
}

int hfx_ctx_get_dt(hfx_ctx *ctx, double *dt)
{
  HFX_CHECK(ctx && dt, "hfx_ctx_get_dt: NULL argument");
  *dt = ctx->params.dt;
  return 0;
}

int hfx_ctx_synchronize(hfx_ctx *ctx)
{
  HFX_CHECK(ctx, "NULL ctx");
  HFX_IMMEDIATE(ctx, 0);
  HFX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int hfx_ctx_flush(hfx_ctx *ctx)
{
  HFX_CHECK(ctx, "NULL ctx");
  return defer_flush(ctx, 0);
}

int hfx_ctx_deferred_stats(hfx_ctx *ctx, long *n_fused, long *n_replayed, const char **why)
{
  HFX_CHECK(ctx, "NULL ctx");
  if (n_fused) *n_fused = ctx->defer.n_fused;
  if (n_replayed) *n_replayed = ctx->defer.n_replayed;
  if (why) *why = ctx->defer.last_why.c_str();
  return 0;
}

void *hfx_ctx_stream(hfx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---------------------------------------------------------------------------
int hfx_eles_create(hfx_ctx *ctx, const hfx_eles_desc *d, hfx_eles **out)
{
  HFX_CHECK(ctx && d && out, "hfx_eles_create: NULL argument");
  HFX_CHECK(d->n_dims == 2 || d->n_dims == 3, "Invalid number of dimensions!"); /* src/eles.cpp:1446 */
  HFX_CHECK(d->n_fields == d->n_dims + 2, "hfx_eles_create: n_fields must be n_dims+2 (Euler / Navier-Stokes)");
  HFX_CHECK(d->n_eles >= 0 && d->n_upts > 0 && d->n_fpts > 0, "hfx_eles_create: bad sizes");
  HFX_CHECK(d->opp_0 && d->opp_3 && d->detjac_upts && d->JGinv_upts && d->tdA_fpts && d->norm_fpts,
            "hfx_eles_create: missing operator or metric");
  HFX_IMMEDIATE(ctx, 0);
  ctx->defer.plans.clear();
  HFX_HIP(hipSetDevice(ctx->device));
  hfx_eles *e = new hfx_eles();
  e->ctx = ctx;
  e->n_eles = d->n_eles; e->n_upts = d->n_upts; e->n_fpts = d->n_fpts; e->n_fields = d->n_fields;
  e->n_dims = d->n_dims; e->ele_type = d->ele_type; e->order = d->order;
  const int nu = d->n_upts, nfp = d->n_fpts, nd = d->n_dims, nf = d->n_fields;
  const long ne = d->n_eles;
  e->viscous_ops = d->opp_4[0] != nullptr;
  if (make_operator(e->opp_0, d->opp_0, nfp, nu)) return 1;
  if (make_operator(e->opp_3, d->opp_3, nu, nfp)) return 1;
  for (int i = 0; i < nd; i++)
  {
    HFX_CHECK(d->opp_1[i] && d->opp_2[i], "hfx_eles_create: missing opp_1/opp_2");
    if (make_operator(e->opp_1[i], d->opp_1[i], nfp, nu)) return 1;
    if (make_operator(e->opp_2[i], d->opp_2[i], nu, nu)) return 1;
    if (e->viscous_ops)
    {
      HFX_CHECK(d->opp_4[i] && d->opp_5[i] && d->opp_6, "hfx_eles_create: missing opp_4/5/6");
      if (make_operator(e->opp_4[i], d->opp_4[i], nu, nu)) return 1;
      if (make_operator(e->opp_5[i], d->opp_5[i], nu, nfp)) return 1;
    }
  }
  if (e->viscous_ops)
  {
    HFX_CHECK(d->detjac_fpts && d->JGinv_fpts, "hfx_eles_create: missing flux-point metrics");
    if (make_operator(e->opp_6, d->opp_6, nfp, nu)) return 1;
  }
  if (dev_alloc_copy(&e->detjac_upts, d->detjac_upts, nu * ne)) return 1;
  if (dev_alloc_copy(&e->JGinv_upts, d->JGinv_upts, (long)nd * nd * nu * ne)) return 1;
  if (d->detjac_fpts && dev_alloc_copy(&e->detjac_fpts, d->detjac_fpts, nfp * ne)) return 1;
  if (d->JGinv_fpts && dev_alloc_copy(&e->JGinv_fpts, d->JGinv_fpts, (long)nd * nd * nfp * ne)) return 1;
  if (dev_alloc_copy(&e->tdA_fpts, d->tdA_fpts, nfp * ne)) return 1;
  if (dev_alloc_copy(&e->norm_fpts, d->norm_fpts, (long)nfp * ne * nd)) return 1;

  const long pu = nu * ne, pf = nfp * ne;
  long len[HFX_N_ARRAYS];
  len[HFX_DISU_UPTS0] = pu * nf; len[HFX_DISU_UPTS1] = pu * nf; len[HFX_DISU_FPTS] = pf * nf;
  len[HFX_TDISF_UPTS] = pu * nf * nd; len[HFX_NORM_TDISF_FPTS] = pf * nf; len[HFX_NORM_TCONF_FPTS] = pf * nf;
  len[HFX_DIV_TCONF_UPTS] = pu * nf; len[HFX_DELTA_DISU_FPTS] = pf * nf;
  len[HFX_GRAD_DISU_UPTS] = pu * nf * nd; len[HFX_GRAD_DISU_FPTS] = pf * nf * nd;
  len[HFX_SRC_UPTS] = pu * nf; len[HFX_DT_LOCAL] = ne; len[HFX_SENSOR] = ne;
  len[HFX_SGSF_UPTS] = pu * nf * nd; len[HFX_SGSF_FPTS] = pf * nf * nd;
  len[HFX_DISUF_UPTS] = pu * nf; len[HFX_LU] = pu * (nd == 2 ? 3 : 6); len[HFX_LE] = pu * nd;
  for (int i = 0; i < HFX_N_ARRAYS; i++)
  {
    e->arr_len[i] = len[i];
    if (i == HFX_SRC_UPTS || i == HFX_DT_LOCAL) continue;       // allocated on first upload
    if (i == HFX_SGSF_UPTS || i == HFX_SGSF_FPTS) continue;     // allocated by hfx_eles_set_les
    if (i == HFX_DISUF_UPTS || i == HFX_LU || i == HFX_LE) continue; // allocated by hfx_eles_set_les_filter
    HFX_HIP(hipMalloc((void **)&e->arr[i], sizeof(double) * (size_t)std::max<long>(len[i], 1)));
    // the reference zero-initialises its arrays (hf_array::setup + initialize_to_zero, src/eles.cpp:100-215)
    HFX_HIP(hipMemset(e->arr[i], 0, sizeof(double) * (size_t)std::max<long>(len[i], 1)));
  }
  HFX_HIP(hipMalloc((void **)&e->nan_flag, sizeof(unsigned long long)));
  HFX_HIP(hipMemset(e->nan_flag, 0xff, sizeof(unsigned long long)));
  e->red_blocks = 1024;
  HFX_HIP(hipMalloc((void **)&e->red_buf, sizeof(double) * e->red_blocks));
  *out = e;
  return 0;
}

int hfx_eles_destroy(hfx_eles *e)
{
  if (!e) return 0;
  {
    // (a record that still names this block cannot run any more: drop it with the plans that point here)
    hfx::Deferred &d = e->ctx->defer;
    d.log.clear();
    d.plans.clear();
  }
  free_operator(e->opp_0); free_operator(e->opp_3); free_operator(e->opp_6);
  free_operator(e->inv_vandermonde); free_operator(e->exp_filter);
  free_operator(e->opp_over_int_cubpts); free_operator(e->over_int_filter);
  free_operator(e->opp_volume_cubpts);
  free_operator(e->opp_p);
  free_operator(e->filter_upts);
  for (double *p : {e->sgs_uu, e->sgs_ue})
    if (p) (void)hipFree(p);
  if (e->disu_ppts) (void)hipFree(e->disu_ppts);
  for (double *p : {e->weight_volume_cubpts, e->vol_detjac_vol_cubpts, e->iq_u, e->iq_g})
    if (p) (void)hipFree(p);
  for (double *p : {e->JGinv_over_int_cubpts, e->u_cub, e->t_cub})
    if (p) (void)hipFree(p);
  if (e->persson_num) (void)hipFree(e->persson_num);
  if (e->persson_den) (void)hipFree(e->persson_den);
  tensor_ops_destroy(e);
  for (int i = 0; i < 3; i++)
  {
    free_operator(e->opp_1[i]); free_operator(e->opp_2[i]); free_operator(e->opp_4[i]); free_operator(e->opp_5[i]);
  }
  double *m[] = {e->detjac_upts, e->JGinv_upts, e->detjac_fpts, e->JGinv_fpts, e->tdA_fpts, e->norm_fpts};
  for (double *p : m)
    if (p) (void)hipFree(p);
  for (int i = 0; i < HFX_N_ARRAYS; i++)
    if (e->arr[i]) (void)hipFree(e->arr[i]);
  if (e->h_ref) (void)hipFree(e->h_ref);
  if (e->wall_distance) (void)hipFree(e->wall_distance);
  if (e->Jacobian_fpts) (void)hipFree(e->Jacobian_fpts);
  if (e->nan_flag) (void)hipFree(e->nan_flag);
  if (e->red_buf) (void)hipFree(e->red_buf);
  fused_destroy(e);
  general_destroy(e);
  delete e;
  return 0;
}

int hfx_eles_upload(hfx_eles *e, int id, const double *host)
{
  HFX_CHECK(e && host, "hfx_eles_upload: NULL argument");
  HFX_CHECK(id >= 0 && id < HFX_N_ARRAYS, "hfx_eles_upload: bad array id %d", id);
  HFX_IMMEDIATE(e->ctx, 0);
  if (id == HFX_DISU_UPTS0 || id == HFX_DISU_FPTS) invalidate_fpts(e);
  e->stale &= ~(1u << id);
  if (!e->arr[id]) HFX_HIP(hipMalloc((void **)&e->arr[id], sizeof(double) * (size_t)std::max<long>(e->arr_len[id], 1)));
  HFX_HIP(hipStreamSynchronize(e->ctx->stream));
  HFX_HIP(hipMemcpy(e->arr[id], host, sizeof(double) * (size_t)e->arr_len[id], hipMemcpyHostToDevice));
  if (id == HFX_SRC_UPTS) e->src_nonzero = true;
  return 0;
}

int hfx_eles_download(hfx_eles *e, int id, double *host)
{
  HFX_CHECK(e && host, "hfx_eles_download: NULL argument");
  HFX_CHECK(id >= 0 && id < HFX_N_ARRAYS, "hfx_eles_download: bad array id %d", id);
  HFX_CHECK(e->arr[id] != nullptr, "hfx_eles_download: array %d was never uploaded", id);
  HFX_IMMEDIATE(e->ctx, 1u << id);
  HFX_CHECK(!(e->ctx->defer.on && (e->stale & (1u << id))),
            "hfx_eles_download: array %d was not materialised by the fused stage that ran last (deferred execution): read it before the "
            "next stage begins, or switch the option \"deferred\" off", id);
  HFX_HIP(hipStreamSynchronize(e->ctx->stream));
  HFX_HIP(hipMemcpy(host, e->arr[id], sizeof(double) * (size_t)e->arr_len[id], hipMemcpyDeviceToHost));
  return 0;
}

int hfx_eles_is_current(hfx_eles *e, int id, int *current)
{
  HFX_CHECK(e && current, "hfx_eles_is_current: NULL argument");
  HFX_CHECK(id >= 0 && id < HFX_N_ARRAYS, "hfx_eles_is_current: bad array id %d", id);
  // (a pending record is not run for this question)
  const hfx::Deferred &d = e->ctx->defer;
  if (!(d.on && (e->stale & (1u << id))))
    *current = 1;
  else
  {
    // stale now -- but a pending stage that a request for this array would turn into a replay refreshes it
    bool pending_stage = false;
    for (const hfx::DeferCall &c : d.log) pending_stage = pending_stage || (c.method == hfx::DM_CORRECTED_DIVERGENCE && c.e == e);
    *current = pending_stage ? 2 : 0;
  }
  return 0;
}

int hfx_eles_device_ptr(hfx_eles *e, int id, double **dev)
{
  HFX_CHECK(e && dev, "hfx_eles_device_ptr: NULL argument");
  HFX_CHECK(id >= 0 && id < HFX_N_ARRAYS, "hfx_eles_device_ptr: bad array id %d", id);
  HFX_IMMEDIATE(e->ctx, 1u << id);
  if (id == HFX_DISU_UPTS0 || id == HFX_DISU_FPTS) invalidate_fpts(e); // (the caller may write through the pointer)
  *dev = e->arr[id];
  return 0;
}

// ---- eles::* ----------------------------------------------------------------
int hfx_eles_extrapolate_solution(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0; /* src/eles.cpp:1362 */
  HFX_DEFER(e->ctx, DM_EXTRAPOLATE_SOLUTION, e, nullptr, nullptr, 0, 0);
  e->fpts_valid = true;
  e->stale &= ~(1u << HFX_DISU_FPTS);
  const Operator *ops[1] = {&e->opp_0};
  const double *in[1] = {e->arr[HFX_DISU_UPTS0]};
  return contract_multi_in(e->ctx, ops, 1, in, e->arr[HFX_DISU_FPTS], (long)e->n_eles * e->n_fields, 0);
}

int hfx_eles_calculate_gradient(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_CHECK(e->viscous_ops, "calculate_gradient: block was registered without opp_4");
  HFX_DEFER(e->ctx, DM_CALCULATE_GRADIENT, e, nullptr, nullptr, 0, 0);
  e->stale &= ~(1u << HFX_GRAD_DISU_UPTS);
  const long slab = (long)e->n_upts * e->n_eles * e->n_fields;
  const Operator *ops[3] = {&e->opp_4[0], &e->opp_4[1], &e->opp_4[2]};
  double *out[3] = {e->arr[HFX_GRAD_DISU_UPTS], e->arr[HFX_GRAD_DISU_UPTS] + slab, e->arr[HFX_GRAD_DISU_UPTS] + 2 * slab};
  return contract_multi_out(e->ctx, ops, e->n_dims, e->arr[HFX_DISU_UPTS0], out, (long)e->n_eles * e->n_fields, 0);
}

int hfx_eles_evaluate_invFlux(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  HFX_DEFER(e->ctx, DM_EVALUATE_INVFLUX, e, nullptr, nullptr, 0, 0);
  e->stale &= ~(1u << HFX_TDISF_UPTS);
  const long plane = (long)e->n_upts * e->n_eles;
  if (e->n_dims == 2)
    hipLaunchKernelGGL(invflux_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane,
                       e->ctx->params.gamma, e->arr[HFX_DISU_UPTS0], e->JGinv_upts, e->arr[HFX_TDISF_UPTS]);
  else
    hipLaunchKernelGGL(invflux_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane,
                       e->ctx->params.gamma, e->arr[HFX_DISU_UPTS0], e->JGinv_upts, e->arr[HFX_TDISF_UPTS]);
  HFX_HIP(hipGetLastError());
  return 0;
}

int hfx_eles_correct_gradient(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_CHECK(e->viscous_ops, "correct_gradient: block was registered without opp_5/opp_6");
  hfx_ctx *ctx = e->ctx;
  HFX_DEFER(ctx, DM_CORRECT_GRADIENT, e, nullptr, nullptr, 0, 0);
  e->stale &= ~((1u << HFX_GRAD_DISU_UPTS) | (1u << HFX_GRAD_DISU_FPTS));
  const long slab_u = (long)e->n_upts * e->n_eles * e->n_fields;
  const long ncols = (long)e->n_eles * e->n_fields;
  // (i) grad_disu_upts(:,:,:,d) += opp_5[d] * delta_disu_fpts
  {
    const Operator *ops[3] = {&e->opp_5[0], &e->opp_5[1], &e->opp_5[2]};
    double *out[3] = {e->arr[HFX_GRAD_DISU_UPTS], e->arr[HFX_GRAD_DISU_UPTS] + slab_u, e->arr[HFX_GRAD_DISU_UPTS] + 2 * slab_u};
    if (contract_multi_out(ctx, ops, e->n_dims, e->arr[HFX_DELTA_DISU_FPTS], out, ncols, 1)) return 1;
  }
  // (ii) grad_disu_fpts(:,:,:,d) = opp_6 * grad_disu_upts(:,:,:,d): the dim slabs are contiguous -> one launch
  {
    const Operator *ops[1] = {&e->opp_6};
    const double *in[1] = {e->arr[HFX_GRAD_DISU_UPTS]};
    if (contract_multi_in(ctx, ops, 1, in, e->arr[HFX_GRAD_DISU_FPTS], ncols * e->n_dims, 0)) return 1;
  }
  // (iii) reference -> physical, in place, at solution and flux points
  const long pu = (long)e->n_upts * e->n_eles, pf = (long)e->n_fpts * e->n_eles;
  if (e->n_dims == 2)
  {
    hipLaunchKernelGGL(grad_transform_kernel<2>, dim3(nblocks(pu, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, pu,
                       e->detjac_upts, e->JGinv_upts, e->arr[HFX_GRAD_DISU_UPTS]);
    hipLaunchKernelGGL(grad_transform_kernel<2>, dim3(nblocks(pf, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, pf,
                       e->detjac_fpts, e->JGinv_fpts, e->arr[HFX_GRAD_DISU_FPTS]);
  }
  else
  {
    hipLaunchKernelGGL(grad_transform_kernel<3>, dim3(nblocks(pu, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, pu,
                       e->detjac_upts, e->JGinv_upts, e->arr[HFX_GRAD_DISU_UPTS]);
    hipLaunchKernelGGL(grad_transform_kernel<3>, dim3(nblocks(pf, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, pf,
                       e->detjac_fpts, e->JGinv_fpts, e->arr[HFX_GRAD_DISU_FPTS]);
  }
  HFX_HIP(hipGetLastError());
  return 0;
}

int hfx_eles_evaluate_viscFlux(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  HFX_DEFER(e->ctx, DM_EVALUATE_VISCFLUX, e, nullptr, nullptr, 0, 0);
  e->stale &= ~((1u << HFX_TDISF_UPTS) | (1u << HFX_SGSF_UPTS));
  const long plane = (long)e->n_upts * e->n_eles;
  const Phys P = e->ctx->phys();
  if (e->les_ready)
  {
    if (e->n_dims == 2)
      hipLaunchKernelGGL(viscflux_les_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane, P,
                         e->les, e->arr[HFX_DISU_UPTS0], e->arr[HFX_GRAD_DISU_UPTS], e->JGinv_upts, e->detjac_upts,
                         e->wall_distance, e->arr[HFX_TDISF_UPTS], e->arr[HFX_SGSF_UPTS]);
    else
      hipLaunchKernelGGL(viscflux_les_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane, P,
                         e->les, e->arr[HFX_DISU_UPTS0], e->arr[HFX_GRAD_DISU_UPTS], e->JGinv_upts, e->detjac_upts,
                         e->wall_distance, e->arr[HFX_TDISF_UPTS], e->arr[HFX_SGSF_UPTS]);
    HFX_HIP(hipGetLastError());
    return 0;
  }
  if (e->n_dims == 2)
    hipLaunchKernelGGL(viscflux_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane, P,
                       e->arr[HFX_DISU_UPTS0], e->arr[HFX_GRAD_DISU_UPTS], e->JGinv_upts, e->arr[HFX_TDISF_UPTS]);
  else
    hipLaunchKernelGGL(viscflux_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane, P,
                       e->arr[HFX_DISU_UPTS0], e->arr[HFX_GRAD_DISU_UPTS], e->JGinv_upts, e->arr[HFX_TDISF_UPTS]);
  HFX_HIP(hipGetLastError());
  return 0;
}

int hfx_eles_extrapolate_totalFlux(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_DEFER(e->ctx, DM_EXTRAPOLATE_TOTALFLUX, e, nullptr, nullptr, 0, 0);
  e->stale &= ~(1u << HFX_NORM_TDISF_FPTS);
  const long slab = (long)e->n_upts * e->n_eles * e->n_fields;
  const Operator *ops[3] = {&e->opp_1[0], &e->opp_1[1], &e->opp_1[2]};
  const double *in[3] = {e->arr[HFX_TDISF_UPTS], e->arr[HFX_TDISF_UPTS] + slab, e->arr[HFX_TDISF_UPTS] + 2 * slab};
  return contract_multi_in(e->ctx, ops, e->n_dims, in, e->arr[HFX_NORM_TDISF_FPTS], (long)e->n_eles * e->n_fields, 0);
}

int hfx_eles_calculate_divergence(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_DEFER(e->ctx, DM_CALCULATE_DIVERGENCE, e, nullptr, nullptr, 0, 0);
  e->stale &= ~(1u << HFX_DIV_TCONF_UPTS);
  const long slab = (long)e->n_upts * e->n_eles * e->n_fields;
  const Operator *ops[3] = {&e->opp_2[0], &e->opp_2[1], &e->opp_2[2]};
  const double *in[3] = {e->arr[HFX_TDISF_UPTS], e->arr[HFX_TDISF_UPTS] + slab, e->arr[HFX_TDISF_UPTS] + 2 * slab};
  return contract_multi_in(e->ctx, ops, e->n_dims, in, e->arr[HFX_DIV_TCONF_UPTS], (long)e->n_eles * e->n_fields, 0);
}

int hfx_eles_calculate_corrected_divergence(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_DEFER(e->ctx, DM_CORRECTED_DIVERGENCE, e, nullptr, nullptr, 0, 0);
  e->stale &= ~((1u << HFX_DIV_TCONF_UPTS) | (1u << HFX_NORM_TCONF_FPTS));
  // norm_tconf -= norm_tdisf (overwriting norm_tconf, src/eles.cpp:1746) is fused into the tile load;
  // div_tconf += opp_3 * norm_tconf; NaN scan -> device flag
  const Operator *ops[1] = {&e->opp_3};
  const double *in[1] = {e->arr[HFX_NORM_TCONF_FPTS]};
  return contract_multi_in(e->ctx, ops, 1, in, e->arr[HFX_DIV_TCONF_UPTS], (long)e->n_eles * e->n_fields, 1,
                           e->arr[HFX_NORM_TDISF_FPTS], e->arr[HFX_NORM_TCONF_FPTS], e->nan_flag);
}

int hfx_eles_AdvanceSolution(hfx_eles *e, int in_step, int adv_type)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  hfx_ctx *ctx = e->ctx;
  HFX_CHECK(ctx->have_params, "parameters not set");
  HFX_CHECK(adv_type >= 0 && adv_type <= 4, "ERROR: Time integration type not recognised ... ");
  const int nst = (adv_type == 0) ? 1 : (adv_type <= 2) ? 4 : (adv_type == 3) ? 5 : 14;
  HFX_CHECK(in_step >= 0 && in_step < nst, "AdvanceSolution: stage %d out of range for adv_type %d", in_step, adv_type);
  HFX_DEFER(ctx, DM_ADVANCE_SOLUTION, e, nullptr, nullptr, in_step, adv_type);
  invalidate_fpts(e);
  AdvArgs a;
  a.n = (long)e->n_upts * e->n_eles * e->n_fields;
  a.plane = (long)e->n_upts * e->n_eles;
  a.n_upts = e->n_upts;
  a.adv_type = adv_type;
  a.in_step = in_step;
  a.dt_local_on = (ctx->params.dt_type == 2);
  HFX_CHECK(!a.dt_local_on || e->arr[HFX_DT_LOCAL], "dt_type 2 needs HFX_DT_LOCAL uploaded");
  a.dt = ctx->params.dt;
  a.rk_a = (adv_type >= 3) ? ctx->params.RK_a[in_step] : 0.0;
  a.rk_b = (adv_type >= 3) ? ctx->params.RK_b[in_step] : 0.0;
  a.u0 = e->arr[HFX_DISU_UPTS0];
  a.u1 = e->arr[HFX_DISU_UPTS1];
  a.div = e->arr[HFX_DIV_TCONF_UPTS];
  a.detjac = e->detjac_upts;
  a.src = e->src_nonzero ? e->arr[HFX_SRC_UPTS] : nullptr;
  a.dt_local = e->arr[HFX_DT_LOCAL];
  hipLaunchKernelGGL(advance_kernel, dim3(nblocks(a.n, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, a);
  HFX_HIP(hipGetLastError());
  return 0;
}

int hfx_eles_check_nan(hfx_eles *e, long *first_nan)
{
  HFX_CHECK(e && first_nan, "NULL argument");
  HFX_IMMEDIATE(e->ctx, 0);
  unsigned long long v = 0;
  HFX_HIP(hipStreamSynchronize(e->ctx->stream));
  HFX_HIP(hipMemcpy(&v, e->nan_flag, sizeof v, hipMemcpyDeviceToHost));
  *first_nan = (v == ~0ull) ? -1 : (long)v;
  if (v != ~0ull) HFX_HIP(hipMemset(e->nan_flag, 0xff, sizeof(unsigned long long)));
  return 0;
}

int hfx_eles_compute_res_upts(hfx_eles *e, int norm_type, int field, double *out)
{
  HFX_CHECK(e && out, "NULL argument");
  HFX_CHECK(norm_type >= 0 && norm_type <= 2, "compute_res_upts: bad norm type");
  HFX_CHECK(field >= 0 && field < e->n_fields, "compute_res_upts: bad field");
  HFX_IMMEDIATE(e->ctx, 1u << HFX_DIV_TCONF_UPTS);
  HFX_CHECK(!(e->ctx->defer.on && (e->stale & (1u << HFX_DIV_TCONF_UPTS))),
            "compute_res_upts: div_tconf_upts was not stored by the fused stage that ran last (deferred execution stores it at the last "
            "stage of a step, or when asked before the next stage begins)");
  const long plane = (long)e->n_upts * e->n_eles;
  const int nb = (int)std::min<long>(e->red_blocks, std::max<long>(1, (plane + PT_BLOCK - 1) / PT_BLOCK));
  hipLaunchKernelGGL(res_partial_kernel, dim3(nb), dim3(PT_BLOCK), 0, e->ctx->stream, plane, norm_type,
                     e->arr[HFX_DIV_TCONF_UPTS] + field * plane, e->detjac_upts,
                     e->src_nonzero ? e->arr[HFX_SRC_UPTS] + field * plane : nullptr, e->red_buf);
  HFX_HIP(hipGetLastError());
  std::vector<double> part(nb);
  HFX_HIP(hipStreamSynchronize(e->ctx->stream));
  HFX_HIP(hipMemcpy(part.data(), e->red_buf, sizeof(double) * nb, hipMemcpyDeviceToHost));
  double s = 0.0;
  for (int i = 0; i < nb; i++) s = (norm_type == 0) ? std::max(s, part[i]) : s + part[i];
  *out = s;
  return 0;
}

// ---- int_inters -----------------------------------------------------------------
int hfx_int_inters_create(hfx_ctx *ctx, hfx_eles *left, hfx_eles *right, int n_inters, int nfpi, const int *L,
                          const int *R, hfx_inters **out)
{
  HFX_CHECK(ctx && left && right && out, "hfx_int_inters_create: NULL argument");
  HFX_IMMEDIATE(ctx, 0);
  ctx->defer.plans.clear();
  HFX_CHECK(n_inters >= 0 && nfpi > 0, "hfx_int_inters_create: bad sizes");
  HFX_CHECK(n_inters == 0 || (L && R), "hfx_int_inters_create: NULL table");
  HFX_CHECK(left->n_dims == right->n_dims && left->n_fields == right->n_fields, "left/right blocks differ in n_dims");
  const long np = (long)n_inters * nfpi;
  const long pl = (long)left->n_fpts * left->n_eles, pr = (long)right->n_fpts * right->n_eles;
  // validate the tables on the host before any kernel dereferences them, and check that
  // every flux point is owned by at most one face side (race-free scatter)
  {
    std::vector<unsigned char> seen_l(pl, 0), seen_r(left == right ? 0 : pr, 0);
    std::vector<unsigned char> &sr = (left == right) ? seen_l : seen_r;
    for (long q = 0; q < np; q++)
    {
      HFX_CHECK(L[q] >= 0 && L[q] < pl, "face table L[%ld] = %d out of range", q, L[q]);
      HFX_CHECK(R[q] >= 0 && R[q] < pr, "face table R[%ld] = %d out of range", q, R[q]);
      HFX_CHECK(!seen_l[L[q]], "flux point %d appears on two face sides", L[q]);
      seen_l[L[q]] = 1;
      HFX_CHECK(!sr[R[q]], "flux point %d appears on two face sides", R[q]);
      sr[R[q]] = 1;
    }
  }
  hfx_inters *f = new hfx_inters();
  f->ctx = ctx; f->left = left; f->right = right; f->n_inters = n_inters; f->n_fpts_per_inter = nfpi;
  f->hL.assign(L, L + np);
  f->hR.assign(R, R + np);
  HFX_HIP(hipMalloc((void **)&f->L, sizeof(int) * (size_t)std::max<long>(np, 1)));
  HFX_HIP(hipMalloc((void **)&f->R, sizeof(int) * (size_t)std::max<long>(np, 1)));
  HFX_HIP(hipMemcpy(f->L, L, sizeof(int) * (size_t)np, hipMemcpyHostToDevice));
  HFX_HIP(hipMemcpy(f->R, R, sizeof(int) * (size_t)np, hipMemcpyHostToDevice));
  left->faces_attached.push_back(f);
  if (right != left) right->faces_attached.push_back(f);
  fused_invalidate(left);
  fused_invalidate(right);
  general_invalidate(left);
  general_invalidate(right);
  *out = f;
  return 0;
}

int hfx_inters_destroy(hfx_inters *f)
{
  if (!f) return 0;
  // (a record that still names this block cannot run any more: drop it with the plans that point here)
  f->ctx->defer.log.clear();
  f->ctx->defer.plans.clear();
  for (hfx_eles *e : {f->left, f->right})
    if (e)
    {
      auto &v = e->faces_attached;
      v.erase(std::remove(v.begin(), v.end(), f), v.end());
      fused_invalidate(e);
      general_invalidate(e);
    }
  if (f->L) (void)hipFree(f->L);
  if (f->R) (void)hipFree(f->R);
  for (double *b : {f->out_disu, f->in_disu, f->out_grad, f->in_grad, f->out_sgsf, f->in_sgsf})
    if (b) (void)hipFree(b);
  if (f->boundary_id) (void)hipFree(f->boundary_id);
  if (f->bcs) (void)hipFree(f->bcs);
  delete f;
  return 0;
}

// ---- LES closure ----------------------------------------------------------------------------
int hfx_eles_set_les(hfx_eles *e, const hfx_les *les, const double *wall_distance, const double *Jacobian_fpts)
{
  HFX_CHECK(e && les && Jacobian_fpts, "hfx_eles_set_les: NULL argument");
  HFX_IMMEDIATE(e->ctx, 0);
  e->ctx->defer.plans.clear();
  HFX_CHECK(les->sgs_model >= 0 && les->sgs_model <= 4, "SGS model not implemented"); /* src/eles.cpp:2461 */
  HFX_CHECK(les->sgs_model != 0 || wall_distance, "hfx_eles_set_les: the Smagorinsky model needs wall_distance");
  HFX_CHECK(e->viscous_ops, "LES not supported with inviscid flow"); /* src/input.cpp:570 */
  e->les.sgs_model = les->sgs_model; e->les.order = e->order;
  {
    static const double ref_vol[5] = {2.0, 4.0, 8.0 / 6.0, 4.0, 8.0}; // tri, quad, tet, prism, hex
    HFX_CHECK(e->ele_type >= 0 && e->ele_type <= 4, "hfx_eles_set_les: unknown element class %d", e->ele_type);
    e->les.vol_factor = ref_vol[e->ele_type];
  }
  e->les.C_s = les->C_s; e->les.filter_ratio = les->filter_ratio; e->les.Kappa = les->Kappa; e->les.prandtl_t = les->prandtl_t;
  e->les.Lu = e->arr[HFX_LU]; e->les.Le = e->arr[HFX_LE]; // NULL until hfx_eles_set_les_filter
  for (double **p : {&e->wall_distance, &e->Jacobian_fpts})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (wall_distance && dev_alloc_copy(&e->wall_distance, wall_distance, (long)e->n_upts * e->n_eles * e->n_dims)) return 1;
  if (dev_alloc_copy(&e->Jacobian_fpts, Jacobian_fpts, (long)e->n_dims * e->n_dims * e->n_fpts * e->n_eles)) return 1;
  for (int id : {HFX_SGSF_UPTS, HFX_SGSF_FPTS})
    if (!e->arr[id])
    {
      HFX_HIP(hipMalloc((void **)&e->arr[id], sizeof(double) * (size_t)std::max<long>(e->arr_len[id], 1)));
      HFX_HIP(hipMemset(e->arr[id], 0, sizeof(double) * (size_t)std::max<long>(e->arr_len[id], 1)));
    }
  e->les_ready = true;
  fused_invalidate(e);
  general_invalidate(e); // (the general stage takes the closure / the de-aliased flux from the registration too)
  return 0;
}

int hfx_eles_set_les_filter(hfx_eles *e, const double *filter_upts)
{
  HFX_CHECK(e && filter_upts, "hfx_eles_set_les_filter: NULL argument");
  HFX_IMMEDIATE(e->ctx, 0);
  e->ctx->defer.plans.clear();
  free_operator(e->filter_upts);
  if (make_operator(e->filter_upts, filter_upts, e->n_upts, e->n_upts)) return 1;
  for (int id : {HFX_DISUF_UPTS, HFX_LU, HFX_LE})
    if (!e->arr[id])
    {
      HFX_HIP(hipMalloc((void **)&e->arr[id], sizeof(double) * (size_t)std::max<long>(e->arr_len[id], 1)));
      HFX_HIP(hipMemset(e->arr[id], 0, sizeof(double) * (size_t)std::max<long>(e->arr_len[id], 1)));
    }
  if (!e->sgs_uu) HFX_HIP(hipMalloc((void **)&e->sgs_uu, sizeof(double) * (size_t)std::max<long>(e->arr_len[HFX_LU], 1)));
  if (!e->sgs_ue) HFX_HIP(hipMalloc((void **)&e->sgs_ue, sizeof(double) * (size_t)std::max<long>(e->arr_len[HFX_LE], 1)));
  e->les.Lu = e->arr[HFX_LU];
  e->les.Le = e->arr[HFX_LE];
  return 0;
}

int hfx_eles_calc_sgs_terms(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_CHECK(e->les_ready, "calc_sgs_terms: hfx_eles_set_les was not called");
  const int model = e->les.sgs_model;
  if (model < 2) return 0; /* src/solver.cpp:57 */
  HFX_CHECK(e->filter_upts.present(), "calc_sgs_terms: SGS model %d filters the solution: register filter_upts (hfx_eles_set_les_filter)", model);
  hfx_ctx *ctx = e->ctx;
  HFX_DEFER(ctx, DM_CALC_SGS_TERMS, e, nullptr, nullptr, 0, 0);
  if (model == 3) invalidate_fpts(e); // (the filtered solution replaces the state)
  e->stale &= ~((1u << HFX_DISUF_UPTS) | (1u << HFX_LU) | (1u << HFX_LE));
  hipStream_t st = ctx->stream;
  const long plane = (long)e->n_upts * e->n_eles;
  const Operator *ops[1] = {&e->filter_upts};
  const double *in_u[1] = {e->arr[HFX_DISU_UPTS0]};
  if (contract_multi_in(ctx, ops, 1, in_u, e->arr[HFX_DISUF_UPTS], (long)e->n_eles * e->n_fields, 0)) return 1;
  const bool sim = model == 2 || model == 4;
  // products of the unfiltered solution (similarity models) and the NaN scan of the filtered one
  if (e->n_dims == 2)
    hipLaunchKernelGGL(sgs_products_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, st, plane, e->arr[HFX_DISU_UPTS0],
                       e->arr[HFX_DISUF_UPTS], sim ? e->sgs_uu : nullptr, sim ? e->sgs_ue : nullptr, e->nan_flag);
  else
    hipLaunchKernelGGL(sgs_products_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, st, plane, e->arr[HFX_DISU_UPTS0],
                       e->arr[HFX_DISUF_UPTS], sim ? e->sgs_uu : nullptr, sim ? e->sgs_ue : nullptr, e->nan_flag);
  HFX_HIP(hipGetLastError());
  if (model == 3)
  {
    // spectral vanishing viscosity: the filtered solution replaces the solution (src/eles.cpp:2088-2090)
    HFX_HIP(hipMemcpyAsync(e->arr[HFX_DISU_UPTS0], e->arr[HFX_DISUF_UPTS], sizeof(double) * (size_t)e->arr_len[HFX_DISU_UPTS0],
                           hipMemcpyDeviceToDevice, st));
    return 0;
  }
  const double *in_uu[1] = {e->sgs_uu}, *in_ue[1] = {e->sgs_ue};
  if (contract_multi_in(ctx, ops, 1, in_uu, e->arr[HFX_LU], (long)e->n_eles * (e->n_dims == 2 ? 3 : 6), 0)) return 1;
  if (contract_multi_in(ctx, ops, 1, in_ue, e->arr[HFX_LE], (long)e->n_eles * e->n_dims, 0)) return 1;
  if (e->n_dims == 2)
    hipLaunchKernelGGL(sgs_leonard_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, st, plane, e->arr[HFX_DISUF_UPTS],
                       e->arr[HFX_LU], e->arr[HFX_LE]);
  else
    hipLaunchKernelGGL(sgs_leonard_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, st, plane, e->arr[HFX_DISUF_UPTS],
                       e->arr[HFX_LU], e->arr[HFX_LE]);
  HFX_HIP(hipGetLastError());
  return 0;
}

// sgsf_upts from disu_upts(0) and the corrected gradient in grad_disu_upts (internal: the split fused path, variant 2)
extern "C" int hfx_les_sgsf_upts_internal(hfx_eles *e)
{
  HFX_CHECK(e && e->les_ready, "LES closure not set");
  hfx_ctx *ctx = e->ctx;
  const long plane = (long)e->n_upts * e->n_eles;
  if (e->n_dims == 2)
    hipLaunchKernelGGL(sgsf_upts_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, plane, ctx->phys(), e->les,
                       e->arr[HFX_DISU_UPTS0], e->arr[HFX_GRAD_DISU_UPTS], e->JGinv_upts, e->detjac_upts, e->wall_distance,
                       e->arr[HFX_SGSF_UPTS]);
  else
    hipLaunchKernelGGL(sgsf_upts_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, plane, ctx->phys(), e->les,
                       e->arr[HFX_DISU_UPTS0], e->arr[HFX_GRAD_DISU_UPTS], e->JGinv_upts, e->detjac_upts, e->wall_distance,
                       e->arr[HFX_SGSF_UPTS]);
  HFX_HIP(hipGetLastError());
  return 0;
}

// first half of extrapolate_sgsFlux only: sgsf_fpts = opp_0 * sgsf_upts, still in reference space (internal: the split
// fused path applies |J|^-1 J where the face kernel consumes the flux, instead of a separate sweep over sgsf_fpts)
extern "C" int hfx_les_extrapolate_reference_internal(hfx_eles *e)
{
  HFX_CHECK(e && e->les_ready, "LES closure not set");
  const Operator *ops[1] = {&e->opp_0};
  const double *in[1] = {e->arr[HFX_SGSF_UPTS]};
  return contract_multi_in(e->ctx, ops, 1, in, e->arr[HFX_SGSF_FPTS], (long)e->n_eles * e->n_fields * e->n_dims, 0);
}

int hfx_eles_extrapolate_sgsFlux(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0;
  HFX_CHECK(e->les_ready, "extrapolate_sgsFlux: hfx_eles_set_les was not called");
  HFX_DEFER(e->ctx, DM_EXTRAPOLATE_SGSFLUX, e, nullptr, nullptr, 0, 0);
  e->stale &= ~(1u << HFX_SGSF_FPTS);
  // sgsf_fpts(:,:,:,d) = opp_0 * sgsf_upts(:,:,:,d): the dim slabs are contiguous -> one launch
  {
    const Operator *ops[1] = {&e->opp_0};
    const double *in[1] = {e->arr[HFX_SGSF_UPTS]};
    if (contract_multi_in(e->ctx, ops, 1, in, e->arr[HFX_SGSF_FPTS], (long)e->n_eles * e->n_fields * e->n_dims, 0)) return 1;
  }
  const long plane = (long)e->n_fpts * e->n_eles;
  if (e->n_dims == 2)
    hipLaunchKernelGGL(sgsf_to_physical_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane,
                       e->detjac_fpts, e->Jacobian_fpts, e->arr[HFX_SGSF_FPTS]);
  else
    hipLaunchKernelGGL(sgsf_to_physical_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, e->ctx->stream, plane,
                       e->detjac_fpts, e->Jacobian_fpts, e->arr[HFX_SGSF_FPTS]);
  HFX_HIP(hipGetLastError());
  return 0;
}

// ---- integral diagnostics -------------------------------------------------------------------
int hfx_eles_set_volume_cubpts(hfx_eles *e, int n_cubpts, const double *opp_volume_cubpts, const double *weight_volume_cubpts,
                               const double *vol_detjac_vol_cubpts)
{
  HFX_CHECK(e && opp_volume_cubpts && weight_volume_cubpts && vol_detjac_vol_cubpts && n_cubpts > 0,
            "hfx_eles_set_volume_cubpts: bad argument");
  HFX_IMMEDIATE(e->ctx, 0);
  free_operator(e->opp_volume_cubpts);
  for (double **p : {&e->weight_volume_cubpts, &e->vol_detjac_vol_cubpts, &e->iq_u, &e->iq_g})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  e->n_vol_cubpts = n_cubpts;
  if (make_operator(e->opp_volume_cubpts, opp_volume_cubpts, n_cubpts, e->n_upts)) return 1;
  if (dev_alloc_copy(&e->weight_volume_cubpts, weight_volume_cubpts, n_cubpts)) return 1;
  if (dev_alloc_copy(&e->vol_detjac_vol_cubpts, vol_detjac_vol_cubpts, (long)n_cubpts * e->n_eles)) return 1;
  return 0;
}

int hfx_eles_CalcIntegralQuantities(hfx_eles *e, int n_q, const int *quantity_ids, double *integral_quantities)
{
  HFX_CHECK(e && quantity_ids && integral_quantities, "hfx_eles_CalcIntegralQuantities: NULL argument");
  HFX_IMMEDIATE(e->ctx, (1u << HFX_DISU_UPTS0) | (1u << HFX_GRAD_DISU_UPTS));
  HFX_CHECK(n_q >= 0 && n_q <= IQ_MAX, "hfx_eles_CalcIntegralQuantities: at most %d quantities per call", IQ_MAX);
  if (e->n_eles == 0 || n_q == 0) return 0;
  HFX_CHECK(e->opp_volume_cubpts.dense, "CalcIntegralQuantities: hfx_eles_set_volume_cubpts was not called");
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  for (int m = 0; m < n_q; m++)
    HFX_CHECK(quantity_ids[m] >= 0 && quantity_ids[m] <= 4, "integral diagnostic quantity not recognized"); /* src/eles.cpp:5618 */
  hfx_ctx *ctx = e->ctx;
  hipStream_t st = ctx->stream;
  const int nc = e->n_vol_cubpts;
  const long pc = (long)nc * e->n_eles;
  if (!e->iq_u) HFX_HIP(hipMalloc((void **)&e->iq_u, sizeof(double) * (size_t)pc * e->n_fields));
  if (!e->iq_g) HFX_HIP(hipMalloc((void **)&e->iq_g, sizeof(double) * (size_t)pc * e->n_fields * e->n_dims));
  // state and corrected gradient at the cubature points: dense contractions
  {
    const Operator *ops[1] = {&e->opp_volume_cubpts};
    const double *in[1] = {e->arr[HFX_DISU_UPTS0]};
    if (contract_multi_in(ctx, ops, 1, in, e->iq_u, (long)e->n_eles * e->n_fields, 0)) return 1;
    in[0] = e->arr[HFX_GRAD_DISU_UPTS];
    if (contract_multi_in(ctx, ops, 1, in, e->iq_g, (long)e->n_eles * e->n_fields * e->n_dims, 0)) return 1;
  }
  const int nblk = (int)std::min<long>(e->red_blocks / IQ_MAX, nblocks(pc, PT_BLOCK));
  int *d_ids = nullptr;
  HFX_HIP(hipMalloc((void **)&d_ids, sizeof(int) * IQ_MAX));
  HFX_HIP(hipMemcpyAsync(d_ids, quantity_ids, sizeof(int) * n_q, hipMemcpyHostToDevice, st));
  if (e->n_dims == 2)
    hipLaunchKernelGGL(integral_quantities_kernel<2>, dim3(nblk), dim3(PT_BLOCK), 0, st, nc, (long)e->n_eles, e->iq_u, e->iq_g,
                       e->weight_volume_cubpts, e->vol_detjac_vol_cubpts, ctx->params.gamma, n_q, d_ids, e->red_buf);
  else
    hipLaunchKernelGGL(integral_quantities_kernel<3>, dim3(nblk), dim3(PT_BLOCK), 0, st, nc, (long)e->n_eles, e->iq_u, e->iq_g,
                       e->weight_volume_cubpts, e->vol_detjac_vol_cubpts, ctx->params.gamma, n_q, d_ids, e->red_buf);
  HFX_HIP(hipGetLastError());
  std::vector<double> part((size_t)nblk * n_q);
  HFX_HIP(hipMemcpyAsync(part.data(), e->red_buf, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
  HFX_HIP(hipStreamSynchronize(st));
  (void)hipFree(d_ids);
  for (int m = 0; m < n_q; m++)
  {
    double s = 0.0;
    for (int b = 0; b < nblk; b++) s += part[b + (size_t)nblk * m];
    integral_quantities[m] += s; /* the reference accumulates over element classes (src/output.cpp:2023-2028) */
  }
  return 0;
}

// ---- CFL time stepping --------------------------------------------------------------------
int hfx_eles_set_h_ref(hfx_eles *e, const double *h_ref)
{
  HFX_CHECK(e && h_ref, "hfx_eles_set_h_ref: NULL argument");
  HFX_IMMEDIATE(e->ctx, 0);
  if (e->h_ref) (void)hipFree(e->h_ref);
  if (e->wall_distance) (void)hipFree(e->wall_distance);
  if (e->Jacobian_fpts) (void)hipFree(e->Jacobian_fpts);
  e->h_ref = nullptr;
  return dev_alloc_copy(&e->h_ref, h_ref, e->n_eles);
}

int hfx_eles_calc_dt_local(hfx_eles *e, double CFL, double *dt_min)
{
  HFX_CHECK(e && dt_min, "hfx_eles_calc_dt_local: NULL argument");
  HFX_IMMEDIATE(e->ctx, 0);
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  *dt_min = 1e12; /* src/solver.cpp:490 */
  if (e->n_eles == 0) return 0;
  HFX_CHECK(e->h_ref, "calc_dt_local: hfx_eles_set_h_ref was not called");
  hipStream_t st = e->ctx->stream;
  if (!e->arr[HFX_DT_LOCAL]) HFX_HIP(hipMalloc((void **)&e->arr[HFX_DT_LOCAL], sizeof(double) * (size_t)e->n_eles));
  const double big = 1e12;
  HFX_HIP(hipMemcpyAsync(e->red_buf, &big, sizeof(double), hipMemcpyHostToDevice, st));
  const Phys P = e->ctx->phys();
  if (e->n_dims == 2)
    hipLaunchKernelGGL(dt_local_kernel<2>, dim3((unsigned)e->n_eles), dim3(64), 0, st, e->n_upts, (long)e->n_eles,
                       e->arr[HFX_DISU_UPTS0], e->h_ref, P, CFL, e->order, e->arr[HFX_DT_LOCAL], (unsigned long long *)e->red_buf);
  else
    hipLaunchKernelGGL(dt_local_kernel<3>, dim3((unsigned)e->n_eles), dim3(64), 0, st, e->n_upts, (long)e->n_eles,
                       e->arr[HFX_DISU_UPTS0], e->h_ref, P, CFL, e->order, e->arr[HFX_DT_LOCAL], (unsigned long long *)e->red_buf);
  HFX_HIP(hipGetLastError());
  HFX_HIP(hipMemcpyAsync(dt_min, e->red_buf, sizeof(double), hipMemcpyDeviceToHost, st));
  HFX_HIP(hipStreamSynchronize(st));
  return 0;
}

// ---- plot-point interpolation --------------------------------------------------------------
int hfx_eles_set_opp_p(hfx_eles *e, int n_ppts, const double *opp_p)
{
  HFX_CHECK(e && opp_p && n_ppts > 0, "hfx_eles_set_opp_p: bad argument");
  HFX_IMMEDIATE(e->ctx, 0);
  free_operator(e->opp_p);
  if (e->disu_ppts) { (void)hipFree(e->disu_ppts); e->disu_ppts = nullptr; }
  if (make_operator(e->opp_p, opp_p, n_ppts, e->n_upts)) return 1;
  if (dev_alloc_copy(&e->disu_ppts, nullptr, (long)n_ppts * e->n_eles * e->n_fields)) return 1;
  e->n_ppts = n_ppts;
  return 0;
}

int hfx_eles_calc_disu_ppts(hfx_eles *e, double *host)
{
  HFX_CHECK(e && host, "hfx_eles_calc_disu_ppts: NULL argument");
  HFX_IMMEDIATE(e->ctx, 0);
  if (e->n_eles == 0) return 0; /* src/eles.cpp:3759 */
  HFX_CHECK(e->n_ppts > 0, "calc_disu_ppts: hfx_eles_set_opp_p was not called");
  hfx_ctx *ctx = e->ctx;
  const Operator *ops[1] = {&e->opp_p};
  const double *in[1] = {e->arr[HFX_DISU_UPTS0]};
  if (contract_multi_in(ctx, ops, 1, in, e->disu_ppts, (long)e->n_eles * e->n_fields, 0)) return 1;
  HFX_HIP(hipMemcpyAsync(host, e->disu_ppts, sizeof(double) * (size_t)e->n_ppts * e->n_eles * e->n_fields, hipMemcpyDeviceToHost,
                         ctx->stream));
  HFX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- over-integration (polynomial de-aliasing of the inviscid flux) ------------------------
int hfx_eles_set_over_int(hfx_eles *e, int n_cubpts, const double *opp_over_int_cubpts, const double *over_int_filter,
                          const double *JGinv_over_int_cubpts)
{
  HFX_CHECK(e && opp_over_int_cubpts && over_int_filter && JGinv_over_int_cubpts && n_cubpts > 0,
            "hfx_eles_set_over_int: bad argument");
  HFX_IMMEDIATE(e->ctx, 0);
  e->ctx->defer.plans.clear();
  free_operator(e->opp_over_int_cubpts);
  free_operator(e->over_int_filter);
  for (double **p : {&e->JGinv_over_int_cubpts, &e->u_cub, &e->t_cub})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  e->n_cubpts = n_cubpts;
  if (make_operator(e->opp_over_int_cubpts, opp_over_int_cubpts, n_cubpts, e->n_upts)) return 1;
  if (make_operator(e->over_int_filter, over_int_filter, e->n_upts, n_cubpts)) return 1;
  const long pc = (long)n_cubpts * e->n_eles;
  if (dev_alloc_copy(&e->JGinv_over_int_cubpts, JGinv_over_int_cubpts, pc * e->n_dims * e->n_dims)) return 1;
  if (dev_alloc_copy(&e->u_cub, nullptr, pc * e->n_fields)) return 1;
  if (dev_alloc_copy(&e->t_cub, nullptr, pc * e->n_fields * e->n_dims)) return 1;
  e->over_int_ready = true;
  fused_invalidate(e);
  general_invalidate(e); // (the general stage takes the closure / the de-aliased flux from the registration too)
  // tensor-product elements: 1-D factors of the two matrices for the sum-factorised kernel (tensor_ops.hip)
  return tensor_over_int_setup(e, n_cubpts, opp_over_int_cubpts, over_int_filter);
}

int hfx_eles_evaluate_invFlux_over_int(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0; /* src/eles.cpp:1482 */
  HFX_CHECK(e->over_int_ready, "evaluate_invFlux_over_int: hfx_eles_set_over_int was not called");
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  hfx_ctx *ctx = e->ctx;
  HFX_DEFER(ctx, DM_EVALUATE_INVFLUX, e, nullptr, nullptr, 1, 0);
  e->stale &= ~(1u << HFX_TDISF_UPTS);
  if (tensor_over_int_available(e) && ctx->contract_mode != HFX_CONTRACT_DENSE) return tensor_over_int_launch(e);
  // interpolate the solution to the over-integration cubature points
  {
    const Operator *ops[1] = {&e->opp_over_int_cubpts};
    const double *in[1] = {e->arr[HFX_DISU_UPTS0]};
    if (contract_multi_in(ctx, ops, 1, in, e->u_cub, (long)e->n_eles * e->n_fields, 0)) return 1;
  }
  // flux + transform there
  const long plane = (long)e->n_cubpts * e->n_eles;
  if (e->n_dims == 2)
    hipLaunchKernelGGL(invflux_kernel<2>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, plane,
                       ctx->params.gamma, e->u_cub, e->JGinv_over_int_cubpts, e->t_cub);
  else
    hipLaunchKernelGGL(invflux_kernel<3>, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, plane,
                       ctx->params.gamma, e->u_cub, e->JGinv_over_int_cubpts, e->t_cub);
  HFX_HIP(hipGetLastError());
  // project back to the solution points (the over-integration filter)
  {
    const Operator *ops[1] = {&e->over_int_filter};
    const double *in[1] = {e->t_cub};
    if (contract_multi_in(ctx, ops, 1, in, e->arr[HFX_TDISF_UPTS], (long)e->n_eles * e->n_fields * e->n_dims, 0)) return 1;
  }
  return 0;
}

// ---- shock capturing --------------------------------------------------------------------
int hfx_eles_set_shock_capture(hfx_eles *e, const double *inv_vandermonde, const double *exp_filter,
                               const double *norm_basis_persson, const int *high_modes, double s0, int shock_det_field)
{
  HFX_CHECK(e && inv_vandermonde && exp_filter && norm_basis_persson && high_modes, "hfx_eles_set_shock_capture: NULL argument");
  HFX_IMMEDIATE(e->ctx, 0);
  e->ctx->defer.plans.clear();
  HFX_CHECK(shock_det_field == 0 || shock_det_field == 1, "Unsupported shock capturing field."); /* src/eles_hexas.cpp:1034 */
  free_operator(e->inv_vandermonde);
  free_operator(e->exp_filter);
  if (make_operator(e->inv_vandermonde, inv_vandermonde, e->n_upts, e->n_upts)) return 1;
  if (make_operator(e->exp_filter, exp_filter, e->n_upts, e->n_upts)) return 1;
  std::vector<double> num(e->n_upts), den(e->n_upts);
  for (int j = 0; j < e->n_upts; j++)
  {
    den[j] = norm_basis_persson[j];
    num[j] = high_modes[j] ? norm_basis_persson[j] : 0.0;
  }
  if (e->persson_num) (void)hipFree(e->persson_num);
  if (e->persson_den) (void)hipFree(e->persson_den);
  e->persson_num = e->persson_den = nullptr;
  if (dev_alloc_copy(&e->persson_num, num.data(), e->n_upts)) return 1;
  if (dev_alloc_copy(&e->persson_den, den.data(), e->n_upts)) return 1;
  e->s0 = s0;
  e->shock_det_field = shock_det_field;
  e->shock_ready = true;
  return tensor_shock_setup(e, inv_vandermonde, exp_filter, norm_basis_persson, high_modes);
}

int hfx_eles_shock_capture(hfx_eles *e)
{
  HFX_CHECK(e, "NULL eles");
  if (e->n_eles == 0) return 0; /* src/eles.cpp:2920 */
  HFX_CHECK(e->shock_ready, "shock_capture: hfx_eles_set_shock_capture was not called");
  hfx_ctx *ctx = e->ctx;
  HFX_DEFER(ctx, DM_SHOCK_CAPTURE, e, nullptr, nullptr, 0, 0);
  invalidate_fpts(e);
  e->stale &= ~(1u << HFX_SENSOR);
  if (tensor_shock_available(e) && ctx->contract_mode != HFX_CONTRACT_DENSE) return tensor_shock_launch(e);
  const long plane = (long)e->n_upts * e->n_eles;
  double *scratch = e->arr[HFX_TDISF_UPTS]; // free between AdvanceSolution and the next stage's evaluate_invFlux
  double *u = e->arr[HFX_DISU_UPTS0];
  // 1. modal coefficients of the sensor field, all elements at once: a dense (n_upts x n_upts) contraction
  {
    const Operator *ops[1] = {&e->inv_vandermonde};
    const double *in[1] = {u + (e->shock_det_field == 0 ? 0 : (long)(e->n_dims + 1) * plane)};
    if (contract_multi_in(ctx, ops, 1, in, scratch, e->n_eles, 0)) return 1;
  }
  // 2. sensor
  hipLaunchKernelGGL(persson_sensor_kernel, dim3((unsigned)e->n_eles), dim3(64), 0, ctx->stream, e->n_upts, (long)e->n_eles,
                     scratch, e->persson_num, e->persson_den, e->arr[HFX_SENSOR]);
  // 3. filtered state of every element (dense contraction), 4. kept where the sensor fires
  {
    const Operator *ops[1] = {&e->exp_filter};
    const double *in[1] = {u};
    if (contract_multi_in(ctx, ops, 1, in, scratch, (long)e->n_eles * e->n_fields, 0)) return 1;
  }
  hipLaunchKernelGGL(shock_select_kernel, dim3(nblocks(plane, PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, e->n_upts,
                     (long)e->n_eles, e->n_fields, e->s0, e->arr[HFX_SENSOR], scratch, u);
  HFX_HIP(hipGetLastError());
  return 0;
}

// ---- bdy_inters ----------------------------------------------------------------------
int hfx_bdy_inters_create(hfx_ctx *ctx, hfx_eles *left, int n_inters, int nfpi, const int *L, const int *boundary_id,
                          const hfx_bc *bcs, int n_bcs, double R_ref, hfx_inters **out)
{
  HFX_CHECK(ctx && left && out, "hfx_bdy_inters_create: NULL argument");
  HFX_IMMEDIATE(ctx, 0);
  ctx->defer.plans.clear();
  HFX_CHECK(n_inters >= 0 && nfpi > 0 && n_bcs >= 0, "hfx_bdy_inters_create: bad sizes");
  HFX_CHECK(n_inters == 0 || (L && boundary_id && bcs && n_bcs > 0), "hfx_bdy_inters_create: NULL table");
  const long np = (long)n_inters * nfpi;
  const long pl = (long)left->n_fpts * left->n_eles;
  for (long q = 0; q < np; q++) HFX_CHECK(L[q] >= 0 && L[q] < pl, "boundary face table L[%ld] = %d out of range", q, L[q]);
  for (int i = 0; i < n_inters; i++)
  {
    HFX_CHECK(boundary_id[i] >= 0 && boundary_id[i] < n_bcs, "boundary_id[%d] = %d out of range", i, boundary_id[i]);
    const hfx_bc &b = bcs[boundary_id[i]];
    HFX_CHECK(b.flag >= HFX_BC_SUB_IN_SIMP && b.flag <= HFX_BC_SLIP_WALL_DUAL && b.flag != HFX_BC_CYCLIC,
              "Boundary condition not implemented yet (bc_flag %d)", b.flag); /* src/input.cpp:341 */
    HFX_CHECK(!((b.flag == HFX_BC_ISOTHERM_WALL || b.flag == HFX_BC_ADIABAT_WALL) && b.use_wm),
              "boundary group %d uses the wall model, which is not part of this path", boundary_id[i]);
    HFX_CHECK(b.flag != HFX_BC_ADIABAT_WALL || ctx->params.viscous || !ctx->have_params,
              "Adiabatic wall boundary only available to viscous simulation"); /* src/input.cpp:427 */
  }
  hfx_inters *f = new hfx_inters();
  f->ctx = ctx; f->left = left; f->right = nullptr; f->n_inters = n_inters; f->n_fpts_per_inter = nfpi;
  f->is_bdy = true;
  f->n_bcs = n_bcs;
  for (int b = 0; b < n_bcs; b++) f->any_ramp = f->any_ramp || (bcs[b].flag == HFX_BC_SUB_IN_CHAR && bcs[b].pressure_ramp);
  f->R_ref = R_ref;
  f->hL.assign(L, L + np);
  const size_t ni = (size_t)std::max<long>(np, 1);
  HFX_HIP(hipMalloc((void **)&f->L, sizeof(int) * ni));
  HFX_HIP(hipMemcpy(f->L, L, sizeof(int) * (size_t)np, hipMemcpyHostToDevice));
  HFX_HIP(hipMalloc((void **)&f->boundary_id, sizeof(int) * (size_t)std::max(n_inters, 1)));
  HFX_HIP(hipMemcpy(f->boundary_id, boundary_id, sizeof(int) * (size_t)n_inters, hipMemcpyHostToDevice));
  HFX_HIP(hipMalloc((void **)&f->bcs, sizeof(hfx_bc) * (size_t)std::max(n_bcs, 1)));
  HFX_HIP(hipMemcpy(f->bcs, bcs, sizeof(hfx_bc) * (size_t)n_bcs, hipMemcpyHostToDevice));
  left->faces_attached.push_back(f);
  fused_invalidate(left);
  general_invalidate(left);
  *out = f;
  return 0;
}

int hfx_bdy_inters_set_ramp_counter(hfx_inters *f, int ramp_counter)
{
  HFX_CHECK(f && f->is_bdy, "not a boundary-face block");
  {
    // between two time steps (src/HiFiLES.cpp:224-225): behind a recorded stage the new value waits for that stage to run
    hfx::Deferred &d = f->ctx->defer;
    if (d.on && !d.busy && !d.log.empty() && d.log.back().method >= DM_ADVANCE_SOLUTION)
      return defer_record(f->ctx, DM_SET_RAMP_COUNTER, nullptr, f, nullptr, ramp_counter, 0);
  }
  HFX_IMMEDIATE(f->ctx, 0);
  f->ramp_counter = ramp_counter;
  return 0;
}

static BdyArgs bdy_args(hfx_inters *f)
{
  BdyArgs a{};
  hfx_eles *l = f->left;
  a.npts = (long)f->n_inters * f->n_fpts_per_inter;
  a.nfpi = f->n_fpts_per_inter;
  a.L = f->L; a.boundary_id = f->boundary_id; a.bcs = f->bcs;
  a.plane = (long)l->n_fpts * l->n_eles;
  a.disu = l->arr[HFX_DISU_FPTS]; a.grad = l->arr[HFX_GRAD_DISU_FPTS];
  a.norm = l->norm_fpts; a.tdA = l->tdA_fpts;
  a.tconf = l->arr[HFX_NORM_TCONF_FPTS]; a.delta = l->arr[HFX_DELTA_DISU_FPTS];
  a.P = f->ctx->phys();
  a.R_ref = f->R_ref;
  a.ramp_counter = f->ramp_counter;
  return a;
}

// fast != 0: the fused paths' reciprocal-multiply physics
int hfx_bdy_launch_internal(hfx_inters *f, int visc, int fast)
{
  HFX_CHECK(f && f->is_bdy, "not a boundary-face block");
  HFX_CHECK(f->ctx->have_params, "parameters not set");
  if (f->n_inters == 0) return 0;
  const BdyArgs a = bdy_args(f);
  const dim3 g((unsigned)((a.npts + 255) / 256)), b(256);
  hipStream_t st = f->ctx->bdy_stream ? f->ctx->bdy_stream : f->ctx->stream;
  const int nd = f->left->n_dims;
#define HFX_BDY(K)                                                             \
  if (nd == 2 && fast) hipLaunchKernelGGL((K<2, true>), g, b, 0, st, a);       \
  else if (nd == 2) hipLaunchKernelGGL((K<2, false>), g, b, 0, st, a);         \
  else if (fast) hipLaunchKernelGGL((K<3, true>), g, b, 0, st, a);             \
  else hipLaunchKernelGGL((K<3, false>), g, b, 0, st, a);
  if (visc)
  {
    HFX_BDY(bdy_viscflux_kernel)
  }
  else
  {
    HFX_BDY(bdy_invflux_kernel)
  }
#undef HFX_BDY
  HFX_HIP(hipGetLastError());
  return 0;
}

int hfx_bdy_inters_evaluate_boundaryConditions_invFlux(hfx_inters *f, double /*time_bound*/)
{
  HFX_CHECK(f && f->is_bdy, "not a boundary-face block");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_BDY_INVFLUX, nullptr, f, nullptr, 0, 0);
  f->left->stale &= ~((1u << HFX_NORM_TCONF_FPTS) | (1u << HFX_DELTA_DISU_FPTS));
  return hfx_bdy_launch_internal(f, 0, 0);
}
int hfx_bdy_inters_evaluate_boundaryConditions_viscFlux(hfx_inters *f, double /*time_bound*/)
{
  HFX_CHECK(f && f->is_bdy, "not a boundary-face block");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_BDY_VISCFLUX, nullptr, f, nullptr, 0, 0);
  return hfx_bdy_launch_internal(f, 1, 0);
}

// ---- mpi_inters ----------------------------------------------------------------------
int hfx_mpi_inters_create(hfx_ctx *ctx, hfx_eles *left, int n_inters, int nfpi, const int *L, const int *Rlut,
                          hfx_inters **out)
{
  HFX_CHECK(ctx && left && out, "hfx_mpi_inters_create: NULL argument");
  HFX_IMMEDIATE(ctx, 0);
  ctx->defer.plans.clear();
  HFX_CHECK(n_inters >= 0 && nfpi > 0, "hfx_mpi_inters_create: bad sizes");
  HFX_CHECK(n_inters == 0 || (L && Rlut), "hfx_mpi_inters_create: NULL table");
  const long np = (long)n_inters * nfpi;
  const long pl = (long)left->n_fpts * left->n_eles;
  for (long q = 0; q < np; q++)
  {
    HFX_CHECK(L[q] >= 0 && L[q] < pl, "partition face table L[%ld] = %d out of range", q, L[q]);
    HFX_CHECK(Rlut[q] >= 0 && Rlut[q] < nfpi, "partition face table Rlut[%ld] = %d out of range", q, Rlut[q]);
  }
  hfx_inters *f = new hfx_inters();
  f->ctx = ctx; f->left = left; f->right = nullptr; f->n_inters = n_inters; f->n_fpts_per_inter = nfpi;
  f->is_mpi = true;
  f->hL.assign(L, L + np);
  f->hR.assign(Rlut, Rlut + np);
  const size_t ni = (size_t)std::max<long>(np, 1);
  HFX_HIP(hipMalloc((void **)&f->L, sizeof(int) * ni));
  HFX_HIP(hipMalloc((void **)&f->R, sizeof(int) * ni));
  HFX_HIP(hipMemcpy(f->L, L, sizeof(int) * (size_t)np, hipMemcpyHostToDevice));
  HFX_HIP(hipMemcpy(f->R, Rlut, sizeof(int) * (size_t)np, hipMemcpyHostToDevice));
  const size_t nd = ni * left->n_fields, ng = nd * left->n_dims;
  HFX_HIP(hipMalloc((void **)&f->out_disu, sizeof(double) * nd));
  HFX_HIP(hipMalloc((void **)&f->in_disu, sizeof(double) * nd));
  HFX_HIP(hipMalloc((void **)&f->out_grad, sizeof(double) * ng));
  HFX_HIP(hipMalloc((void **)&f->in_grad, sizeof(double) * ng));
  HFX_HIP(hipMemset(f->in_disu, 0, sizeof(double) * nd));
  HFX_HIP(hipMemset(f->in_grad, 0, sizeof(double) * ng));
  *out = f;
  return 0;
}

static MpiArgs mpi_args(hfx_inters *f)
{
  MpiArgs a{};
  hfx_eles *l = f->left;
  a.npairs = (long)f->n_inters * f->n_fpts_per_inter;
  a.nfpi = f->n_fpts_per_inter;
  a.L = f->L; a.Rlut = f->R;
  a.plane = (long)l->n_fpts * l->n_eles;
  a.disu = l->arr[HFX_DISU_FPTS]; a.grad = l->arr[HFX_GRAD_DISU_FPTS];
  a.norm = l->norm_fpts; a.tdA = l->tdA_fpts;
  a.tconf = l->arr[HFX_NORM_TCONF_FPTS]; a.delta = l->arr[HFX_DELTA_DISU_FPTS];
  a.out_disu = f->out_disu; a.out_grad = f->out_grad; a.in_disu = f->in_disu; a.in_grad = f->in_grad;
  a.P = f->ctx->phys();
  if (l->les_ready && f->out_sgsf) // per-method path: sgsf_fpts is physical (hfx_eles_extrapolate_sgsFlux took it back)
  {
    a.sgsf = l->arr[HFX_SGSF_FPTS]; a.jac_fpts = l->Jacobian_fpts; a.detjac_fpts = l->detjac_fpts;
    a.out_sgsf = f->out_sgsf; a.in_sgsf = f->in_sgsf; a.sgs_ref = 0;
  }
  return a;
}

// the SGS-flux buffers of a partition-face block whose left block carries an LES closure (allocated on first use)
extern "C" int hfx_mpi_sgsf_buffers_internal(hfx_inters *f)
{
  if (f->out_sgsf || !f->left->les_ready) return 0;
  const size_t ng = (size_t)std::max<long>((long)f->n_inters * f->n_fpts_per_inter, 1) * f->left->n_fields * f->left->n_dims;
  HFX_HIP(hipMalloc((void **)&f->out_sgsf, sizeof(double) * ng));
  HFX_HIP(hipMalloc((void **)&f->in_sgsf, sizeof(double) * ng));
  HFX_HIP(hipMemset(f->in_sgsf, 0, sizeof(double) * ng));
  return 0;
}

// METHOD: the DeferMethod of a recordable call, or -1 for the packing halves (not part of the reference's call sequence:
// what has been recorded runs first)
#define HFX_MPI_LAUNCH(METHOD, KERNEL2, KERNEL3)                                                            \
  do                                                                                                        \
  {                                                                                                         \
    HFX_CHECK(f && f->is_mpi, "not a partition-face block");                                                \
    if (f->n_inters == 0) return 0;                                                                         \
    HFX_CHECK(f->ctx->have_params, "parameters not set");                                                   \
    if ((METHOD) >= 0) { HFX_DEFER(f->ctx, METHOD, nullptr, f, nullptr, 0, 0); }                            \
    HFX_IMMEDIATE(f->ctx, 0);                                                                               \
    if (hfx_mpi_sgsf_buffers_internal(f)) return 1;                                                         \
    const MpiArgs a = mpi_args(f);                                                                          \
    if (f->left->n_dims == 2)                                                                               \
      hipLaunchKernelGGL(KERNEL2, dim3(nblocks(a.npairs, 256)), dim3(256), 0, f->ctx->stream, a);           \
    else                                                                                                    \
      hipLaunchKernelGGL(KERNEL3, dim3(nblocks(a.npairs, 256)), dim3(256), 0, f->ctx->stream, a);           \
    HFX_HIP(hipGetLastError());                                                                             \
    return 0;                                                                                               \
  } while (0)

int hfx_mpi_inters_pack_solution(hfx_inters *f) { HFX_MPI_LAUNCH(-1, mpi_pack_disu_kernel<2>, mpi_pack_disu_kernel<3>); }
int hfx_mpi_inters_pack_corrected_gradient(hfx_inters *f) { HFX_MPI_LAUNCH(-1, mpi_pack_grad_kernel<2>, mpi_pack_grad_kernel<3>); }
int hfx_mpi_inters_pack_sgsf(hfx_inters *f)
{
  HFX_CHECK(f && f->is_mpi && f->left->les_ready, "hfx_mpi_inters_pack_sgsf: the left block has no LES closure (hfx_eles_set_les)");
  HFX_MPI_LAUNCH(-1, mpi_pack_sgsf_kernel<2>, mpi_pack_sgsf_kernel<3>);
}
int hfx_mpi_inters_calculate_common_invFlux(hfx_inters *f)
{
  HFX_MPI_LAUNCH(DM_MPI_COMMON_INVFLUX, (mpi_common_invflux_kernel<2, false>), (mpi_common_invflux_kernel<3, false>));
}
int hfx_mpi_inters_calculate_common_viscFlux(hfx_inters *f)
{
  HFX_MPI_LAUNCH(DM_MPI_COMMON_VISCFLUX, (mpi_common_viscflux_kernel<2, false>), (mpi_common_viscflux_kernel<3, false>));
}

int hfx_mpi_inters_buffer(hfx_inters *f, int which, double **dev, long *n)
{
  HFX_CHECK(f && f->is_mpi && dev && n, "hfx_mpi_inters_buffer: bad argument");
  HFX_IMMEDIATE(f->ctx, 0);
  HFX_CHECK(which >= 0 && which <= 7, "hfx_mpi_inters_buffer: which must be 0..7");
  if (which >= 6)
  {
    HFX_CHECK(f->left->les_ready, "hfx_mpi_inters_buffer: the SGS-flux buffers exist only with an LES closure on the left block");
    if (hfx_mpi_sgsf_buffers_internal(f)) return 1;
  }
  const long nd = (long)f->n_inters * f->n_fpts_per_inter * f->left->n_fields;
  double *b[8] = {f->out_disu, f->in_disu, f->out_grad, f->in_grad, f->out_grad, f->in_grad, f->out_sgsf, f->in_sgsf};
  *dev = b[which];
  *n = (which == 2 || which == 3 || which >= 6) ? nd * f->left->n_dims : nd;
  return 0;
}

int hfx_stage_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                          int phase, int in_step, int first)
{
  HFX_CHECK(e, "NULL eles");
  HFX_IMMEDIATE(e->ctx, 0);
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  HFX_CHECK(phase >= 0 && phase <= 4, "hfx_stage_partitioned: phase must be 0..4");
  return split_stage_partitioned(e, int_faces, n_int, mpi_faces, n_mpi, phase, in_step, first);
}

static FaceArgs face_args(hfx_inters *f)
{
  FaceArgs a;
  hfx_eles *l = f->left, *r = f->right;
  a.npairs = (long)f->n_inters * f->n_fpts_per_inter;
  a.L = f->L; a.R = f->R;
  a.plane_l = (long)l->n_fpts * l->n_eles; a.plane_r = (long)r->n_fpts * r->n_eles;
  a.disu_l = l->arr[HFX_DISU_FPTS]; a.disu_r = r->arr[HFX_DISU_FPTS];
  a.norm_l = l->norm_fpts;
  a.tdA_l = l->tdA_fpts; a.tdA_r = r->tdA_fpts;
  a.tconf_l = l->arr[HFX_NORM_TCONF_FPTS]; a.tconf_r = r->arr[HFX_NORM_TCONF_FPTS];
  a.delta_l = l->arr[HFX_DELTA_DISU_FPTS]; a.delta_r = r->arr[HFX_DELTA_DISU_FPTS];
  a.grad_l = l->arr[HFX_GRAD_DISU_FPTS]; a.grad_r = r->arr[HFX_GRAD_DISU_FPTS];
  const bool les = l->les_ready && r->les_ready;
  a.sgsf_l = les ? l->arr[HFX_SGSF_FPTS] : nullptr;
  a.sgsf_r = les ? r->arr[HFX_SGSF_FPTS] : nullptr;
  return a;
}

int hfx_int_inters_calculate_common_invFlux(hfx_inters *f)
{
  HFX_CHECK(f, "NULL inters");
  if (f->n_inters == 0) return 0;
  HFX_CHECK(f->ctx->have_params, "parameters not set");
  HFX_DEFER(f->ctx, DM_INT_COMMON_INVFLUX, nullptr, f, nullptr, 0, 0);
  for (hfx_eles *x : {f->left, f->right}) x->stale &= ~((1u << HFX_NORM_TCONF_FPTS) | (1u << HFX_DELTA_DISU_FPTS));
  const FaceArgs a = face_args(f);
  const Phys P = f->ctx->phys();
  if (f->left->n_dims == 2)
    hipLaunchKernelGGL(common_invflux_kernel<2>, dim3(nblocks(a.npairs, PT_BLOCK)), dim3(PT_BLOCK), 0, f->ctx->stream, a, P);
  else
    hipLaunchKernelGGL(common_invflux_kernel<3>, dim3(nblocks(a.npairs, PT_BLOCK)), dim3(PT_BLOCK), 0, f->ctx->stream, a, P);
  HFX_HIP(hipGetLastError());
  return 0;
}

int hfx_int_inters_calculate_common_viscFlux(hfx_inters *f)
{
  HFX_CHECK(f, "NULL inters");
  if (f->n_inters == 0) return 0;
  HFX_CHECK(f->ctx->have_params, "parameters not set");
  HFX_DEFER(f->ctx, DM_INT_COMMON_VISCFLUX, nullptr, f, nullptr, 0, 0);
  const FaceArgs a = face_args(f);
  const Phys P = f->ctx->phys();
  if (f->left->n_dims == 2)
    hipLaunchKernelGGL(common_viscflux_kernel<2>, dim3(nblocks(a.npairs, PT_BLOCK)), dim3(PT_BLOCK), 0, f->ctx->stream, a, P);
  else
    hipLaunchKernelGGL(common_viscflux_kernel<3>, dim3(nblocks(a.npairs, PT_BLOCK)), dim3(PT_BLOCK), 0, f->ctx->stream, a, P);
  HFX_HIP(hipGetLastError());
  return 0;
}

// ---- the caller contract -----------------------------------------------------------
int hfx_CalcResidual_blocks(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb)
{
  HFX_CHECK(eles && neb > 0, "hfx_CalcResidual_blocks: no element blocks");
  HFX_IMMEDIATE(eles[0]->ctx, 0);
  hfx_ctx *ctx = eles[0]->ctx;
  for (int i = 0; i < neb; i++) HFX_CHECK(eles[i] && eles[i]->ctx == ctx, "element blocks of different contexts");
  HFX_CHECK(ctx->have_params, "parameters not set");
  const int viscous = ctx->params.viscous;
  for (int b = 0; b < nfb; b++)
  {
    // every face block must connect element blocks of this call
    bool l = false, r = faces[b]->right == nullptr;
    for (int i = 0; i < neb; i++)
    {
      l = l || faces[b]->left == eles[i];
      r = r || faces[b]->right == eles[i];
    }
    HFX_CHECK(l && r, "face block %d refers to an element block that is not part of this call", b);
  }
#define HFX_EACH(call)             \
  for (int i = 0; i < neb; i++)    \
    if (call(eles[i])) return 1
  /* the call order of src/solver.cpp:65-216, every method over all element classes */
  HFX_EACH(hfx_eles_extrapolate_solution);
  if (viscous) HFX_EACH(hfx_eles_calculate_gradient);
  for (int i = 0; i < neb; i++) /* src/solver.cpp:82-91 */
    if (eles[i]->over_int_ready ? hfx_eles_evaluate_invFlux_over_int(eles[i]) : hfx_eles_evaluate_invFlux(eles[i])) return 1;
  for (int b = 0; b < nfb; b++)
    if (!faces[b]->is_bdy && hfx_int_inters_calculate_common_invFlux(faces[b])) return 1;
  for (int b = 0; b < nfb; b++)
    if (faces[b]->is_bdy && hfx_bdy_inters_evaluate_boundaryConditions_invFlux(faces[b], 0.0)) return 1;
  if (viscous)
  {
    HFX_EACH(hfx_eles_correct_gradient);
    HFX_EACH(hfx_eles_evaluate_viscFlux);
    for (int i = 0; i < neb; i++)
      if (eles[i]->les_ready && hfx_eles_extrapolate_sgsFlux(eles[i])) return 1; /* src/solver.cpp:162-167 */
  }
  HFX_EACH(hfx_eles_extrapolate_totalFlux);
  HFX_EACH(hfx_eles_calculate_divergence);
  if (viscous)
  {
    for (int b = 0; b < nfb; b++)
      if (!faces[b]->is_bdy && hfx_int_inters_calculate_common_viscFlux(faces[b])) return 1;
    for (int b = 0; b < nfb; b++)
      if (faces[b]->is_bdy && hfx_bdy_inters_evaluate_boundaryConditions_viscFlux(faces[b], 0.0)) return 1;
  }
  HFX_EACH(hfx_eles_calculate_corrected_divergence);
#undef HFX_EACH
  return 0;
}

int hfx_CalcResidual(hfx_eles *e, hfx_inters *const *faces, int nfb)
{
  HFX_CHECK(e, "NULL eles");
  return hfx_CalcResidual_blocks(&e, 1, faces, nfb);
}

// calc_time_step over several blocks: the minimum over the blocks (src/solver.cpp:498-505)
static int calc_time_step_blocks(hfx_eles *const *eles, int neb)
{
  hfx_ctx *ctx = eles[0]->ctx;
  if (ctx->params.dt_type != 1 && ctx->params.dt_type != 2) return 0;
  double dt_min = 1e12;
  for (int i = 0; i < neb; i++)
  {
    if (calc_time_step(eles[i], nullptr)) return 1;
    dt_min = std::min(dt_min, ctx->params.dt);
  }
  ctx->params.dt = dt_min;
  return 0;
}

int hfx_run_steps_blocks(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int n_steps, int fused)
{
  HFX_CHECK(eles && neb > 0 && eles[0], "hfx_run_steps_blocks: no element blocks");
  HFX_IMMEDIATE(eles[0]->ctx, 0);
  hfx_ctx *ctx = eles[0]->ctx;
  HFX_CHECK(ctx->have_params, "parameters not set");
  const int adv = ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14; /* src/HiFiLES.cpp:143-150 */
  if (fused == 4) return general_run_steps(eles, neb, faces, nfb, n_steps);
  HFX_CHECK(fused == 0 || neb == 1, "hfx_run_steps_blocks: the split fused stage (fused %d) takes one tensor-product element block; "
                                    "several blocks run per method (0) or through the general fused stage (4)", fused);
  HFX_CHECK(fused == 0 || fused == 2 || fused == 3, "hfx_run_steps: fused must be 0 (per-method), 2 or 3 (split fused stage), 4 (general fused "
                                                    "stage); the gather-style variant 1 of earlier versions has been retired");
  if (fused) return split_run_steps(eles[0], faces, nfb, n_steps, fused);
  for (int s = 0; s < n_steps; s++)
  {
    if (calc_time_step_blocks(eles, neb)) return 1; /* src/HiFiLES.cpp:198 */
    for (int rk = 0; rk < nst; rk++)
    {
      if (rk == 0) /* src/solver.cpp:55-62 */
        for (int i = 0; i < neb; i++)
          if (eles[i]->les_ready && hfx_eles_calc_sgs_terms(eles[i])) return 1;
      if (hfx_CalcResidual_blocks(eles, neb, faces, nfb)) return 1;
      for (int i = 0; i < neb; i++)
        if (hfx_eles_AdvanceSolution(eles[i], rk, adv)) return 1;
      for (int i = 0; i < neb; i++)
        if (eles[i]->shock_ready && hfx_eles_shock_capture(eles[i])) return 1; /* src/HiFiLES.cpp:214-216 */
    }
    advance_ramp_counters(faces, nfb); /* src/HiFiLES.cpp:224-225 */
  }
  return 0;
}

int hfx_run_steps(hfx_eles *e, hfx_inters *const *faces, int nfb, int n_steps, int fused)
{
  HFX_CHECK(e, "NULL eles");
  return hfx_run_steps_blocks(&e, 1, faces, nfb, n_steps, fused);
}

int hfx_time_methods(hfx_eles *e, hfx_inters *const *faces, int nfb, int reps, double ms[HFX_N_TIMED_METHODS])
{
  HFX_CHECK(e && ms && reps > 0, "hfx_time_methods: bad argument");
  HFX_IMMEDIATE(e->ctx, 0);
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  hipStream_t st = e->ctx->stream;
  const int adv = e->ctx->params.adv_type;
  const int nst = (adv == 0) ? 1 : (adv <= 2) ? 4 : (adv == 3) ? 5 : 14;
  const int viscous = e->ctx->params.viscous;
  hipEvent_t ev[HFX_N_TIMED_METHODS + 1];
  for (auto &x : ev) HFX_HIP(hipEventCreate(&x));
  for (int i = 0; i < HFX_N_TIMED_METHODS; i++) ms[i] = 0.0;
  for (int r = 0; r < reps; r++)
  {
    int k = 0;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (hfx_eles_extrapolate_solution(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (viscous && hfx_eles_calculate_gradient(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (hfx_eles_evaluate_invFlux(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    for (int b = 0; b < nfb; b++)
      if (hfx_int_inters_calculate_common_invFlux(faces[b])) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (viscous && hfx_eles_correct_gradient(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (viscous && hfx_eles_evaluate_viscFlux(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (hfx_eles_extrapolate_totalFlux(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (hfx_eles_calculate_divergence(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (viscous)
      for (int b = 0; b < nfb; b++)
        if (hfx_int_inters_calculate_common_viscFlux(faces[b])) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (hfx_eles_calculate_corrected_divergence(e)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    if (hfx_eles_AdvanceSolution(e, r % nst, adv)) return 1;
    HFX_HIP(hipEventRecord(ev[k++], st));
    HFX_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < HFX_N_TIMED_METHODS; i++)
    {
      float t = 0;
      HFX_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      ms[i] += t;
    }
  }
  for (int i = 0; i < HFX_N_TIMED_METHODS; i++) ms[i] /= reps;
  for (auto &x : ev) (void)hipEventDestroy(x);
  return 0;
}

int hfx_time_fused_kernels(hfx_eles *e, hfx_inters *const *faces, int nfb, int reps, double ms[8], char names[256])
{
  HFX_CHECK(e && ms && names && reps > 0, "hfx_time_fused_kernels: bad argument");
  HFX_IMMEDIATE(e->ctx, 0);
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  return split_time_kernels(e, faces, nfb, reps, ms, names, 256, e->ctx->fused_mode);
}

int hfx_time_general_kernels(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int reps, double ms[8], char names[256])
{
  HFX_CHECK(eles && neb > 0 && ms && names && reps > 0, "hfx_time_general_kernels: bad argument");
  HFX_IMMEDIATE(eles[0]->ctx, 0);
  HFX_CHECK(eles[0]->ctx->have_params, "parameters not set");
  snprintf(names, 256, "gface_delta_kernel,general_flux_kernel,gface_flux_multi_kernel,general_update_kernel");
  return general_time_kernels(eles, neb, faces, nfb, reps, ms);
}

int hfx_general_kernel_bytes(hfx_eles *const *eles, int neb, double bytes[8])
{
  HFX_CHECK(eles && neb > 0 && bytes, "hfx_general_kernel_bytes: bad argument");
  general_kernel_bytes(eles, neb, bytes);
  return 0;
}

int hfx_fused_kernel_bytes(hfx_eles *e, double bytes[8])
{
  HFX_CHECK(e && bytes, "hfx_fused_kernel_bytes: bad argument");
  split_kernel_bytes(e, bytes, e->ctx->fused_mode);
  return 0;
}

} // extern "C"
