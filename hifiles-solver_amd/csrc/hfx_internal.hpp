// hfx_internal.hpp -- internal types of libhfx (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/hfx.h"
#include "physics.hpp"

namespace hfx
{

void set_error(const char *fmt, ...);
struct FusedData;

#define HFX_HIP(call)                                                                          \
  do                                                                                           \
  {                                                                                            \
    hipError_t _e = (call);                                                                    \
    if (_e != hipSuccess)                                                                      \
    {                                                                                          \
      hfx::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

#define HFX_CHECK(cond, ...)       \
  do                               \
  {                                \
    if (!(cond))                   \
    {                              \
      hfx::set_error(__VA_ARGS__); \
      return 1;                    \
    }                              \
  } while (0)

// One registered operator matrix (m x k, column-major) on the device, in the
// forms the contraction kernels consume.
struct Operator
{
  int m = 0, k = 0;
  double *dense = nullptr; // (m,k) column-major
  // the same zero-padded to whole MFMA tiles, (mpad, kpad) column-major with mpad = 16 ceil(m/16), kpad = 4 ceil(k/4): the
  // dense contraction kernels read operator fragments from it without bounds checks
  double *dense_pad = nullptr;
  int mpad = 0, kpad = 0;
  // ELL form: exact non-zeros of every row in ASCENDING column order (the
  // reference dgemm sums l ascending, src/funcs.cpp:110-117), padded with
  // (val 0, col = first column of the row).  Row-interleaved: entry q of row r
  // at [r + m*q].
  int nnz_max = 0;
  double *ell_val = nullptr;
  int *ell_idx = nullptr;
  long nnz_total = 0;
  std::vector<double> h_val; // host copy of the ELL arrays (width max(nnz_max,1), row-interleaved)
  std::vector<int> h_idx;
  bool present() const { return dense != nullptr; }
};

} // namespace hfx

struct hfx_inters;
struct hfx_comm;

namespace hfx
{
// ---- deferred execution (deferred.hip) -------------------------------------------------------------------------------
// With the option "deferred" the per-method entry points -- the calls CalcResidual (src/solver.cpp:59-221) and the RK loop
// (src/HiFiLES.cpp:201-217) make -- are RECORDED instead of launched.  When the record is complete (the next stage begins,
// or any other entry point needs the device state) it is compared with CalcResidual's canonical order: a whole stage runs as
// the split / general / partitioned fused stage, anything else is replayed call by call.
enum DeferMethod
{
  DM_CALC_SGS_TERMS = 0, DM_EXTRAPOLATE_SOLUTION, DM_MPI_SEND_SOLUTION, DM_CALCULATE_GRADIENT, DM_EVALUATE_INVFLUX,
  DM_INT_COMMON_INVFLUX, DM_BDY_INVFLUX, DM_MPI_RECEIVE_SOLUTION, DM_MPI_COMMON_INVFLUX, DM_CORRECT_GRADIENT,
  DM_MPI_SEND_GRADIENT, DM_EVALUATE_VISCFLUX, DM_EXTRAPOLATE_SGSFLUX, DM_MPI_SEND_SGSF, DM_EXTRAPOLATE_TOTALFLUX,
  DM_CALCULATE_DIVERGENCE, DM_INT_COMMON_VISCFLUX, DM_BDY_VISCFLUX, DM_MPI_RECEIVE_GRADIENT, DM_MPI_RECEIVE_SGSF,
  DM_MPI_COMMON_VISCFLUX, DM_CORRECTED_DIVERGENCE, DM_ADVANCE_SOLUTION, DM_SHOCK_CAPTURE, DM_N_METHODS,
  // not a method of the stage: `run_input.ramp_counter++` between two time steps (src/HiFiLES.cpp:224-225).  Recorded behind a
  // whole stage so that it does not force that stage to run before the caller has said what it wants of it; applied after it
  DM_SET_RAMP_COUNTER = DM_N_METHODS
};
struct DeferCall
{
  int method = 0;         // DeferMethod; the enumeration is in CalcResidual's order, so it doubles as the call's phase
  hfx_eles *e = nullptr;  // element-block methods
  hfx_inters *f = nullptr; // face-block methods
  hfx_comm *c = nullptr;  // send_* / receive_*
  int i0 = 0, i1 = 0;     // AdvanceSolution: in_step, adv_type; evaluate_invFlux: 1 = the over-integration form; ramp counter: value
};
struct DeferPlan
{
  std::vector<DeferCall> signature; // the record this plan was made for (methods and objects; stage number excluded)
  int kind = 0;                     // 0 replay, 1 split fused stage, 2 partitioned split stage, 3 general fused stage, 4 partitioned general stage
  std::string why;                  // kind 0: why the record is not run as a fused stage
  std::vector<hfx_eles *> eles;
  std::vector<hfx_inters *> faces, mpi_faces; // interior + boundary blocks | partition-face blocks
  hfx_comm *comm = nullptr;
  bool sgs_terms = false, shock = false;
};
struct Deferred
{
  bool on = false;   // option "deferred"
  bool busy = false; // a flush / replay / immediate entry point is running: calls execute at once
  std::vector<DeferCall> log;
  std::vector<DeferPlan> plans; // one per distinct record seen (a run has one or two)
  long n_fused = 0, n_replayed = 0; // stages run fused | records replayed call by call (hfx_ctx_deferred_stats)
  std::string last_why;
};
} // namespace hfx

struct hfx_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  // a second stream for the boundary-face kernels of the fused stages, which then run BESIDE the pairwise interior-face kernel
  // (side_stream_fork: the side stream waits for what the main stream holds so far and boundary launches go there;
  // side_stream_join: launches go to the main stream again; side_stream_wait: the main stream waits for the side stream)
  hipStream_t side_stream = nullptr, bdy_stream = nullptr;
  hipEvent_t side_fork = nullptr, side_done = nullptr;
  hipStream_t mpi_stream = nullptr; // when set: the one-sided partition-face kernels of the split path are launched here (hfx_run_steps_partitioned)
  hfx_params params{};
  bool have_params = false;
  int contract_mode = HFX_CONTRACT_AUTO;
  int fused_mode = 3; // which split variant hfx_time_fused_kernels / hfx_fused_kernel_bytes / hfx_stage_partitioned use (2 or 3)
  int n_cu = 256;
  // measurement knobs (hfx_ctx_set_option): kernel variants that give the same results; defaults are the product path
  struct Options
  {
    int split_grid_per_cu = 0;  // persistent workgroups per CU of the split element kernels; 0: as many as are resident
    int flux_grid_per_cu = 0;   // the same for the loader-wave flux kernel alone (0: split_grid_per_cu)
    int xcd_order = 1;          // workgroups of one XCD walk one contiguous eighth of the elements
    int dictionary_rows = 0;    // 1: the dictionary-row flux kernel even when the operators are tensor products
    int flux_waves = 2;         // waves per SIMD the sum-factorised flux kernel is launched for (2 or 3)
    int buffer_addressing = 1;  // buffer-descriptor addressing where every array is below 4 GiB
    int split_flux = 1;         // hfx_run_steps_partitioned: the flux kernel in three launches (half of the elements without partition-face points | those with | the other half)
    int split_update = 1;       // hfx_run_steps_partitioned: the update kernel in two launches, partition-face elements first (their exchange hidden behind the rest)
    int comm_stream_faces = 1;  // hfx_run_steps_partitioned: partition-face kernels on the communication stream, beside the interior ones
    int fold_general = 1;       // general fused stage: opp_3 norm_tdisf folded into the divergence operator (no norm_tdisf traffic)
    int gather_delta = 1;       // the loader-wave flux kernel forms the LDG corrections of interior points itself (no face_delta launch)
    int loader_wave = 1;        // the LDS-DMA loader wave of the flux kernel where the element size fits
    int simd_roles = 1;         // 1: the flux kernel deals its waves' parts by SIMD (one heavy wave per SIMD)
    int flux_stamps = 0;        // 1: phase time stamps of one workgroup of the flux kernel (printed by hfx_time_fused_kernels)
    int tensor_ops = 1;         // sum-factorised over-integration / shock capturing on tensor-product classes
    int dense_waves = 0;        // waves per workgroup of the dense MFMA contraction: 0 by the operator's rows, else 4 or 8
    int dense_split = 0;        // column groups per 16-row tile dealt to the waves: 0 by the operator's rows, else 1, 2 or 4
    int general_update_waves = 0; // waves per workgroup of the general stage's update kernel: 0 by the staging registers, else 4 or 8
    int over_int_fold = 1;      // 1: with the loader-wave flux kernel the over-integration kernel hands over its contribution to the divergence (n_fields values per point), not tdisf_upts
    int light_wave_short = 1;   // 1: a flux-kernel wave without solution points runs the flux-point physics alone (not the paired form)
    int bdy_beside = 0;         // 1: the fused stages' viscous boundary-face kernels run on a side stream beside the interior-face kernel (measured neutral: off)
    int les_flux_kernel = 1;    // 1: the LES closure is evaluated in the flux kernel of split variant 3 where its loader-wave form runs
    int general_waves = 0;      // waves per workgroup of the general flux kernel: 0 by the LDS image (4 or 8), else 3, 4 or 8
  } opt;
  hfx::Deferred defer;
  double CFL = 0.0; // run_input.CFL (hfx_ctx_set_CFL); dt_type 1 / 2 only
  bool have_CFL = false;
  hfx::Phys phys() const
  {
    hfx::Phys P;
    P.gamma = params.gamma;
    P.prandtl = params.prandtl;
    P.rt_inf = params.rt_inf;
    P.mu_inf = params.mu_inf;
    P.c_sth = params.c_sth;
    P.fix_vis = params.fix_vis;
    P.ldg_beta = params.ldg_beta;
    P.ldg_tau = params.ldg_tau;
    P.riemann = params.riemann_solve_type;
    P.viscous = params.viscous;
    return P;
  }
};

struct hfx_eles
{
  hfx_ctx *ctx = nullptr;
  int n_eles = 0, n_upts = 0, n_fpts = 0, n_fields = 0, n_dims = 0, ele_type = 0, order = 0;
  bool viscous_ops = false;
  double *h_ref = nullptr; // (n_eles) eles::h_ref for calc_dt_local
  // LES closure (hfx_eles_set_les)
  bool les_ready = false;
  hfx::LesParams les{};
  double *wall_distance = nullptr, *Jacobian_fpts = nullptr;
  // similarity-type closures (sgs_model 2, 3, 4): the filter matrix and the work arrays of calc_sgs_terms
  hfx::Operator filter_upts;
  double *sgs_uu = nullptr, *sgs_ue = nullptr;
  // integral diagnostics (hfx_eles_set_volume_cubpts)
  int n_vol_cubpts = 0;
  hfx::Operator opp_volume_cubpts;
  double *weight_volume_cubpts = nullptr, *vol_detjac_vol_cubpts = nullptr, *iq_u = nullptr, *iq_g = nullptr;
  // plot-point interpolation (hfx_eles_set_opp_p)
  int n_ppts = 0;
  hfx::Operator opp_p;
  double *disu_ppts = nullptr;
  // over-integration (hfx_eles_set_over_int)
  bool over_int_ready = false;
  int n_cubpts = 0;
  hfx::Operator opp_over_int_cubpts, over_int_filter;
  double *JGinv_over_int_cubpts = nullptr, *u_cub = nullptr, *t_cub = nullptr;
  // shock capturing (hfx_eles_set_shock_capture)
  bool shock_ready = false;
  hfx::Operator inv_vandermonde, exp_filter;
  double *persson_num = nullptr, *persson_den = nullptr; // (n_upts) weights of the sensor's two sums
  double s0 = 0.0;
  int shock_det_field = 0;
  hfx::Operator opp_0, opp_1[3], opp_2[3], opp_3, opp_4[3], opp_5[3], opp_6;
  // metrics
  double *detjac_upts = nullptr, *JGinv_upts = nullptr, *detjac_fpts = nullptr, *JGinv_fpts = nullptr,
         *tdA_fpts = nullptr, *norm_fpts = nullptr;
  // state / work arrays, indexed by hfx_array_id
  double *arr[HFX_N_ARRAYS] = {};
  long arr_len[HFX_N_ARRAYS] = {};
  bool src_nonzero = false;
  // deferred execution: disu_fpts holds opp_0 . disu_upts(0) of the CURRENT state (the fused stages leave it so; anything
  // else that changes the state clears it), and on a partitioned block that flux-point solution is already on its way to
  // the neighbours; stale: bit i = array i was not refreshed by the last fused stage (its contents are older)
  bool fpts_valid = false, fpts_sent = false;
  unsigned stale = 0;
  unsigned long long *nan_flag = nullptr; // device: smallest flat index of a NaN in div_tconf, or ~0
  double *red_buf = nullptr;              // device partial sums for reductions
  int red_blocks = 0;
  void *tensor_ops = nullptr; // hfx::TensorOps: 1-D factors of the over-integration / shock-capturing matrices
  // fused-path private data (built lazily)
  hfx::FusedData *fused = nullptr;
  void *general = nullptr; // hfx::GeneralData: the general (non-tensor-product) fused stage (general.hip)
  std::vector<hfx_inters *> faces_attached;
};

struct hfx_inters
{
  hfx_ctx *ctx = nullptr;
  hfx_eles *left = nullptr, *right = nullptr;
  int n_inters = 0, n_fpts_per_inter = 0;
  int *L = nullptr, *R = nullptr; // device (n_fpts_per_inter, n_inters)
  std::vector<int> hL, hR;        // host copies (for building per-element tables)
  // partition faces (is_mpi): R holds the received-record slot lut(j); buffers are owned here
  bool is_mpi = false;
  double *out_disu = nullptr, *in_disu = nullptr, *out_grad = nullptr, *in_grad = nullptr;
  double *out_sgsf = nullptr, *in_sgsf = nullptr; // LES: physical SGS flux records (allocated when the left block has a closure)
  // neighbour segments (hfx_mpi_inters_set_neighbours): faces [send[s], send[s]+count[s]) go to peer[s], its faces arrive at recv[s]
  std::vector<int> seg_peer, seg_send, seg_recv, seg_count;
  // boundary faces (is_bdy): left side only
  bool is_bdy = false;
  int *boundary_id = nullptr; // device (n_inters)
  hfx_bc *bcs = nullptr;      // device (n_bcs)
  int n_bcs = 0, ramp_counter = 0;
  bool any_ramp = false; // a group of this block ramps its total pressure: run_input.pressure_ramp (src/input.cpp:374-377)
  double R_ref = 0.0;
};

// RCCL transport of the partition-face buffers (comm.hip)
struct hfx_comm
{
  hfx_ctx *ctx = nullptr;
  void *nccl = nullptr;         // ncclComm_t
  hipStream_t stream = nullptr; // communication stream
  int nranks = 1, rank = 0;
  hipEvent_t packed[3] = {nullptr, nullptr, nullptr};   // compute -> comm: buffers of kind 0 / 1 / 2 are packed
  hipEvent_t received[3] = {nullptr, nullptr, nullptr}; // comm -> compute: exchange of kind 0 / 1 / 2 complete
  double *scratch = nullptr;                   // device scratch of the small all-reduces
};

namespace hfx
{
// calc_time_step (src/solver.cpp:484-549) for one block: dt_type 1 sets params.dt to the minimum CFL step (over the ranks
// of `comm` when given), dt_type 2 refreshes HFX_DT_LOCAL; dt_type 0: nothing (comm.hip)
int calc_time_step(hfx_eles *e, hfx_comm *comm);
// run_input.ramp_counter++ after a time step for the boundary blocks with a ramping group (src/HiFiLES.cpp:224-225)
void advance_ramp_counters(hfx_inters *const *faces, int nfb);
} // namespace hfx

namespace hfx
{
// ---- deferred execution (deferred.hip) ----
int defer_record(hfx_ctx *ctx, int method, hfx_eles *e, hfx_inters *f, hfx_comm *c, int i0, int i1);
// run what has been recorded.  need: bit i = array i (hfx_array_id) must hold the reference's values afterwards -- a record
// that a fused stage would leave without them is replayed call by call
int defer_flush(hfx_ctx *ctx, unsigned need = 0);
struct DeferBusy
{
  hfx_ctx *c;
  bool prev;
  explicit DeferBusy(hfx_ctx *ctx) : c(ctx), prev(ctx->defer.busy) { c->defer.busy = true; }
  ~DeferBusy() { c->defer.busy = prev; }
};
inline void invalidate_fpts(hfx_eles *e) { e->fpts_valid = e->fpts_sent = false; }
int side_stream_fork(hfx_ctx *ctx);
int side_stream_join(hfx_ctx *ctx);
int side_stream_wait(hfx_ctx *ctx);
// (hfx.hip) the squared length scale of the eddy-viscosity closures at every solution point, for the fused stages' flux kernels
int les_len2_upload(hfx_eles *e, double **dst);
} // namespace hfx
// a per-method entry point: recorded while the context defers (and is not replaying)
#define HFX_DEFER(ctx_, method_, e_, f_, c_, i0_, i1_)  \
  if ((ctx_)->defer.on && !(ctx_)->defer.busy) return hfx::defer_record(ctx_, method_, e_, f_, c_, i0_, i1_)
// any other entry point that reads or changes device state: what has been recorded runs first
#define HFX_IMMEDIATE(ctx_, need_)                \
  if (hfx::defer_flush(ctx_, need_)) return 1;    \
  hfx::DeferBusy _defer_busy(ctx_)

// boundary-face kernels (hfx.hip); visc 0: inviscid sweep (+ LDG common solution), 1: viscous sweep;
// fast: the fused paths' reciprocal-multiply physics
extern "C" int hfx_bdy_launch_internal(hfx_inters *f, int visc, int fast);
// LES: sgsf_upts = JGinv * F_sgs from disu_upts(0) and grad_disu_upts (hfx.hip)
extern "C" int hfx_mpi_sgsf_buffers_internal(hfx_inters *f); // allocates out / in_sgsf when the left block has a closure
extern "C" int hfx_les_sgsf_upts_internal(hfx_eles *e);
extern "C" int hfx_les_extrapolate_reference_internal(hfx_eles *e); // sgsf_fpts = opp_0 * sgsf_upts, not yet back-transformed
