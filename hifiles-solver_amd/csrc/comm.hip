// comm.hip -- the partition-face transport of libhfx: RCCL point-to-point over xGMI, owned by the library.
//
// Reference side: mpi_inters::send_solution / receive_solution / send_corrected_gradient / receive_corrected_gradient
// (/root/reference/src/mpi_inters.cpp:218-336: pack, MPI_Isend + MPI_Irecv per neighbour rank, MPI_Waitall) and the
// places CalcResidual calls them (src/solver.cpp:68-72,131-139,148-155,197-210).  Here a message is one
// ncclSend / ncclRecv pair per neighbour inside one group on a communication stream; "send" = record `packed` on the
// compute stream, make the communication stream wait for it, enqueue the group, record `received`; "receive" = make the
// compute stream wait for `received`.  The host never blocks inside a stage.
//
// librccl is resolved at run time so that a process that never creates a communicator does not need it, and so that
// a launcher that already mapped a librccl (PyTorch ships one with the same soname) shares that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

#include <vector>

#include "fused_hex.hpp"
#include "general.hpp"
#include "hfx_internal.hpp"

namespace hfx
{

struct Rccl
{
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;

static int rccl_load()
{
  if (g_rccl.handle) return 0;
  void *h = nullptr;
  // a copy that is already mapped (same soname) first, then the system one
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
  if (!h)
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  HFX_CHECK(h, "librccl.so.1 could not be loaded: %s", dlerror());
#define HFX_SYM(field, name)                                             \
  g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                 \
  HFX_CHECK(g_rccl.field, "librccl lacks the symbol %s", name)
  HFX_SYM(GetUniqueId, "ncclGetUniqueId");
  HFX_SYM(CommInitRank, "ncclCommInitRank");
  HFX_SYM(CommDestroy, "ncclCommDestroy");
  HFX_SYM(GroupStart, "ncclGroupStart");
  HFX_SYM(GroupEnd, "ncclGroupEnd");
  HFX_SYM(Send, "ncclSend");
  HFX_SYM(Recv, "ncclRecv");
  HFX_SYM(AllReduce, "ncclAllReduce");
  HFX_SYM(CommCount, "ncclCommCount");
  HFX_SYM(CommCuDevice, "ncclCommCuDevice");
  HFX_SYM(CommUserRank, "ncclCommUserRank");
  HFX_SYM(GetErrorString, "ncclGetErrorString");
#undef HFX_SYM
  g_rccl.handle = h;
  return 0;
}

#define HFX_NCCL(call)                                                                                       \
  do                                                                                                         \
  {                                                                                                          \
    ncclResult_t _r = (call);                                                                                \
    if (_r != ncclSuccess)                                                                                   \
    {                                                                                                        \
      hfx::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hfx::g_rccl.GetErrorString(_r));     \
      return 1;                                                                                              \
    }                                                                                                        \
  } while (0)

// which buffers an exchange moves and how long a face record is.  kind 0: the flux-point solution; kind 2: the SGS flux; kind 1: the
// corrected gradient (per-method path and fused mode 2, as the reference sends it) or, with `projected`, each side's
// viscous flux projected on its own normal (fused mode 3: n_fields instead of n_fields * n_dims doubles per flux point)
static void exchange_buffers(const hfx_inters *f, int kind, bool projected, double *&out, double *&in, long &rec)
{
  const hfx_eles *l = f->left;
  rec = (long)f->n_fpts_per_inter * l->n_fields;
  if (kind == 0)
  {
    out = f->out_disu;
    in = f->in_disu;
    return;
  }
  if (kind == 2) // LES: the physical SGS flux, laid out like the gradient (src/mpi_inters.cpp:65-66)
  {
    out = f->out_sgsf;
    in = f->in_sgsf;
    rec *= l->n_dims;
    return;
  }
  out = f->out_grad;
  in = f->in_grad;
  if (!projected) rec *= l->n_dims;
}

// the grouped exchange of `kind` for the given partition-face blocks, ordered after everything the compute stream has
// been given so far
// ordered: the communication stream is already behind the kernels that packed the buffers (they ran on it)
static int start_exchange(hfx_comm *c, hfx_inters *const *mpi_faces, int n_mpi, int kind, bool projected, bool ordered = false)
{
  hipStream_t cs = c->stream;
  if (!ordered)
  {
    HFX_HIP(hipEventRecord(c->packed[kind], c->ctx->stream));
    HFX_HIP(hipStreamWaitEvent(cs, c->packed[kind], 0));
  }
  bool any = false;
  for (int b = 0; b < n_mpi; b++) any = any || !mpi_faces[b]->seg_peer.empty();
  if (any)
  {
    ncclComm_t comm = (ncclComm_t)c->nccl;
    // (checked before the group opens: a failure inside it must not leave the thread's group depth at 1)
    for (int b = 0; b < n_mpi; b++)
      for (int peer : mpi_faces[b]->seg_peer)
        HFX_CHECK(peer >= 0 && peer < c->nranks, "partition-face block: neighbour rank %d is not one of the communicator's %d ranks", peer, c->nranks);
    HFX_NCCL(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int b = 0; b < n_mpi && r == ncclSuccess; b++)
    {
      const hfx_inters *f = mpi_faces[b];
      double *out, *in;
      long rec;
      exchange_buffers(f, kind, projected, out, in, rec);
      for (size_t s = 0; s < f->seg_peer.size() && r == ncclSuccess; s++)
      {
        const size_t n = (size_t)f->seg_count[s] * rec;
        r = g_rccl.Send(out + (size_t)f->seg_send[s] * rec, n, ncclDouble, f->seg_peer[s], comm, cs);
        if (r == ncclSuccess) r = g_rccl.Recv(in + (size_t)f->seg_recv[s] * rec, n, ncclDouble, f->seg_peer[s], comm, cs);
      }
    }
    if (r != ncclSuccess)
    {
      (void)g_rccl.GroupEnd(); // close the group whatever it says, so that later RCCL calls on this thread are not queued into it
      set_error("%s:%d: ncclSend / ncclRecv of exchange %d failed: %s", __FILE__, __LINE__, kind, g_rccl.GetErrorString(r));
      return 1;
    }
    HFX_NCCL(g_rccl.GroupEnd());
  }
  HFX_HIP(hipEventRecord(c->received[kind], cs));
  return 0;
}

static int wait_exchange(hfx_comm *c, int kind)
{
  HFX_HIP(hipStreamWaitEvent(c->ctx->stream, c->received[kind], 0));
  return 0;
}

static int n_rk_stages(const hfx_params &p) { return (p.adv_type == 0) ? 1 : (p.adv_type <= 2) ? 4 : (p.adv_type == 3) ? 5 : 14; }

// calc_time_step (src/solver.cpp:484-549): per-element CFL steps, minimum over the block and over the ranks
int calc_time_step(hfx_eles *e, hfx_comm *comm)
{
  hfx_ctx *ctx = e->ctx;
  if (ctx->params.dt_type != 1 && ctx->params.dt_type != 2) return 0;
  HFX_CHECK(ctx->have_CFL, "dt_type %d needs run_input.CFL: call hfx_ctx_set_CFL", ctx->params.dt_type);
  double dt_min = 0.0;
  if (hfx_eles_calc_dt_local(e, ctx->CFL, &dt_min)) return 1;
  if (comm && comm->nranks > 1 && hfx_comm_allreduce(comm, &dt_min, 1, 0)) return 1;
  ctx->params.dt = dt_min;
  return 0;
}

// run_input.ramp_counter++ after a time step when a boundary group ramps (src/HiFiLES.cpp:224-225)
void advance_ramp_counters(hfx_inters *const *faces, int nfb)
{
  for (int b = 0; b < nfb; b++)
    if (faces[b]->is_bdy && faces[b]->any_ramp) faces[b]->ramp_counter++;
}

struct StageTimers
{
  hipEvent_t ph[5] = {}, x0[2] = {}, x1[2] = {}, fk[2] = {};
  double acc[8] = {};
  int create()
  {
    for (auto &x : ph) HFX_HIP(hipEventCreate(&x));
    for (auto &x : x0) HFX_HIP(hipEventCreate(&x));
    for (auto &x : x1) HFX_HIP(hipEventCreate(&x));
    for (auto &x : fk) HFX_HIP(hipEventCreate(&x));
    return 0;
  }
  void destroy()
  {
    for (auto &x : ph) (void)hipEventDestroy(x);
    for (auto &x : x0) (void)hipEventDestroy(x);
    for (auto &x : x1) (void)hipEventDestroy(x);
    for (auto &x : fk) (void)hipEventDestroy(x);
  }
};

static int check_partition_blocks(hfx_eles *e, hfx_inters *const *mpi_faces, int n_mpi, hfx_comm *comm)
{
  HFX_CHECK(comm && comm->ctx == e->ctx, "the communicator belongs to another context");
  for (int b = 0; b < n_mpi; b++)
  {
    const hfx_inters *f = mpi_faces[b];
    HFX_CHECK(f->is_mpi && f->left == e, "bad partition-face block");
    int listed = 0;
    for (int c : f->seg_count) listed += c;
    HFX_CHECK(listed == f->n_inters, "partition-face block: %d of %d faces have a neighbour (hfx_mpi_inters_set_neighbours)", listed,
              f->n_inters);
    for (int peer : f->seg_peer)
      HFX_CHECK(peer >= 0 && peer < comm->nranks, "partition-face block: neighbour rank %d is not one of the communicator's %d ranks", peer, comm->nranks);
  }
  return 0;
}

// ONE RK stage of the split fused path on a partitioned block with the library's transport, in CalcResidual's order
// (src/solver.cpp:68-72,131-139,148-155,197-210).  start: the flux-point solution of the current state has not been packed
// and sent yet (the first stage after the caller changed the state; later stages find the exchange that the previous
// stage started after its update).  Leaves the exchange of the NEW state's flux-point solution in flight.
int partitioned_stage(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi, hfx_comm *comm,
                      int rk, bool start, StageTimers *T)
{
  hfx_ctx *ctx = e->ctx;
  const bool visc = ctx->params.viscous != 0;
  if (split_deferred_prepare(e, int_faces, n_int, true)) return 1; // (the variant below depends on the block's fused tables)
  const bool projected = split_variant(e) == 3; // variant 3 sends the projected viscous flux
  // third message: the SGS flux (src/solver.cpp:168-178,203-206) -- variant 2 only; in variant 3 it is part of the projected flux
  const bool les = e->les_ready && !projected;
  hipStream_t st = ctx->stream, cs = comm->stream;
  auto phase = [&](int ph, int stage, int first) {
    return split_stage_partitioned(e, int_faces, n_int, mpi_faces, n_mpi, ph, stage, first);
  };
  // (the timed form keeps every kernel of a phase on the compute stream, so that its events bracket the phase)
  const bool beside = ctx->opt.comm_stream_faces && projected && !les && visc && T == nullptr;
  if (start)
  {
    if (phase(0, rk, 1)) return 1;
    if (start_exchange(comm, mpi_faces, n_mpi, 0, false)) return 1; // later stages: started after phase 4 of the previous one
  }
  if (beside)
  {
    // ---- partition-face kernels and exchanges on the communication stream, in its order; the compute stream meets it
    // at two events per stage: before the flux kernel (LDG corrections at the partition faces are in place) and before
    // the update kernel (common fluxes at the partition faces are in place)
    if (phase(1, rk, 0)) return 1; // interior LDG pairs                                      | compute stream
    ctx->mpi_stream = cs;
    const int r5 = phase(5, rk, 0); // LDG corrections at the partition faces (behind the receive) | communication stream
    ctx->mpi_stream = nullptr;
    if (r5) return 1;
    HFX_HIP(hipEventRecord(comm->received[0], cs));
    // the flux kernel in three launches (option split_flux): half of the elements WITHOUT partition-face points first -- they
    // need nothing from the neighbours, so the solution exchange runs beside them --, then the elements with, whose projected
    // fluxes then leave beside the other half (and the interior common-flux kernel)
    const bool split_flux = ctx->opt.split_flux && !e->over_int_ready && n_mpi > 0;
    if (split_flux && phase(13, rk, 0)) return 1;
    HFX_HIP(hipStreamWaitEvent(st, comm->received[0], 0));
    if (split_flux ? phase(14, rk, 0) : phase(6, rk, 0)) return 1; // gradient + flux kernel
    HFX_HIP(hipEventRecord(comm->packed[1], st));
    HFX_HIP(hipStreamWaitEvent(cs, comm->packed[1], 0));
    ctx->mpi_stream = cs;
    int rc = phase(7, rk, 0); // pack the projected flux
    if (!rc) rc = start_exchange(comm, mpi_faces, n_mpi, 1, true, true);
    if (!rc) rc = phase(8, rk, 0); // common fluxes at the partition faces
    ctx->mpi_stream = nullptr;
    if (rc) return 1;
    HFX_HIP(hipEventRecord(comm->received[1], cs));
    if (split_flux && phase(15, rk, 0)) return 1;
    if (phase(3, rk, 0)) return 1; // interior common fluxes                                  | compute stream
    HFX_HIP(hipStreamWaitEvent(st, comm->received[1], 0));
    // the update: first the elements with partition-face points, whose new flux-point solution is packed and sent (communication
    // stream) while the others are updated -- the exchange the next stage's flux kernel waits for is hidden behind them
    const bool split_update = ctx->opt.split_update && !e->shock_ready && n_mpi > 0;
    if (split_update ? phase(11, rk, 0) : phase(9, rk, 0)) return 1; // update (+ shock capturing)
    HFX_HIP(hipEventRecord(comm->packed[0], st));
    HFX_HIP(hipStreamWaitEvent(cs, comm->packed[0], 0));
    ctx->mpi_stream = cs;
    rc = phase(10, rk, 0); // pack the new flux-point solution
    if (!rc) rc = start_exchange(comm, mpi_faces, n_mpi, 0, false, true);
    ctx->mpi_stream = nullptr;
    if (!rc && split_update) rc = phase(12, rk, 0);
    return rc;
  }
  if (T) HFX_HIP(hipEventRecord(T->ph[0], st));
  if (phase(1, rk, 0)) return 1;
  if (T) HFX_HIP(hipEventRecord(T->ph[1], st));
  if (wait_exchange(comm, 0)) return 1;
  const bool pieces = T && projected && visc; // phase 2 in its three pieces, the element kernel bracketed on its own
  if (pieces)
  {
    if (phase(5, rk, 0)) return 1;
    HFX_HIP(hipEventRecord(T->fk[0], st));
    if (phase(6, rk, 0)) return 1;
    HFX_HIP(hipEventRecord(T->fk[1], st));
    if (phase(7, rk, 0)) return 1;
  }
  else if (phase(2, rk, 0))
    return 1;
  if (visc)
  {
    if (T) HFX_HIP(hipEventRecord(T->x1[0], st));
    if (start_exchange(comm, mpi_faces, n_mpi, 1, projected)) return 1;
    if (les && start_exchange(comm, mpi_faces, n_mpi, 2, false)) return 1;
    if (T) HFX_HIP(hipEventRecord(T->x1[1], cs));
  }
  if (T) HFX_HIP(hipEventRecord(T->ph[2], st));
  if (phase(3, rk, 0)) return 1;
  if (T) HFX_HIP(hipEventRecord(T->ph[3], st));
  if (visc && wait_exchange(comm, 1)) return 1;
  if (visc && les && wait_exchange(comm, 2)) return 1;
  if (phase(4, rk, 0)) return 1;
  if (T) HFX_HIP(hipEventRecord(T->x0[0], st));
  if (start_exchange(comm, mpi_faces, n_mpi, 0, false)) return 1;
  if (T)
  {
    HFX_HIP(hipEventRecord(T->x0[1], cs));
    HFX_HIP(hipEventRecord(T->ph[4], st));
    HFX_HIP(hipStreamSynchronize(st));
    HFX_HIP(hipStreamSynchronize(cs));
    float t = 0;
    for (int w = 0; w < 4; w++)
    {
      HFX_HIP(hipEventElapsedTime(&t, T->ph[w], T->ph[w + 1]));
      T->acc[w] += t;
    }
    HFX_HIP(hipEventElapsedTime(&t, T->x0[0], T->x0[1]));
    T->acc[4] += t;
    if (visc)
    {
      HFX_HIP(hipEventElapsedTime(&t, T->x1[0], T->x1[1]));
      T->acc[5] += t;
    }
    HFX_HIP(hipEventElapsedTime(&t, T->ph[0], T->ph[4]));
    T->acc[6] += t;
    if (pieces)
    {
      HFX_HIP(hipEventElapsedTime(&t, T->fk[0], T->fk[1]));
      T->acc[7] += t;
    }
  }
  return 0;
}

// the deferred scheduler's stage (deferred.hip): as above, no timers
int partitioned_stage_deferred(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                               hfx_comm *comm, int rk, bool start)
{
  if (check_partition_blocks(e, mpi_faces, n_mpi, comm)) return 1;
  return partitioned_stage(e, int_faces, n_int, mpi_faces, n_mpi, comm, rk, start, nullptr);
}

// ONE RK stage of the GENERAL fused stage (general.hip: tetrahedra, prisms, several element blocks) on partitioned blocks: the
// stage's four parts with the one-sided partition-face kernels and the two exchanges between them, in CalcResidual's order
// (src/solver.cpp:68-72,131-139,148-155,197-210; mpi_inters::set_mpi takes any element class, src/mpi_inters.cpp:154):
//   [start: pack the flux-point solution, exchange]            interior LDG pairs / boundary ghost states
//   wait; LDG corrections at the partition faces               flux kernels of all blocks
//   pack each side's projected viscous flux Fn, exchange       interior common fluxes, boundary viscous fluxes
//   wait; common fluxes at the partition faces                 update kernels (new flux-point solution)
//   pack the new flux-point solution, exchange (for the next stage)
int general_partitioned_stage(hfx_eles *const *eles, int neb, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces,
                              int n_mpi, hfx_comm *comm, int rk, bool start, bool shock)
{
  HFX_CHECK(neb > 0 && comm, "general partitioned stage: bad argument");
  hfx_ctx *ctx = eles[0]->ctx;
  HFX_CHECK(comm->ctx == ctx, "the communicator belongs to another context");
  const bool visc = ctx->params.viscous != 0;
  const int nst = n_rk_stages(ctx->params);
  // the blocks' tables are built with ALL face blocks (every flux point needs its face)
  std::vector<hfx_inters *> all(int_faces, int_faces + n_int);
  for (int b = 0; b < n_mpi; b++)
  {
    const hfx_inters *f = mpi_faces[b];
    HFX_CHECK(f->is_mpi, "bad partition-face block");
    bool mine = false;
    for (int i = 0; i < neb; i++) mine = mine || f->left == eles[i];
    HFX_CHECK(mine, "a partition-face block belongs to an element block that is not part of this call");
    int listed = 0;
    for (int c : f->seg_count) listed += c;
    HFX_CHECK(listed == f->n_inters, "partition-face block: %d of %d faces have a neighbour (hfx_mpi_inters_set_neighbours)", listed, f->n_inters);
    all.push_back(mpi_faces[b]);
  }
  if (general_deferred_prepare(eles, neb, all.data(), (int)all.size())) return 1;
  auto mpi_all = [&](int what) -> int {
    for (int b = 0; b < n_mpi; b++)
      if (mpi_launch_general(mpi_faces[b]->left, mpi_faces[b], what, general_fn_fpts(mpi_faces[b]->left))) return 1;
    return 0;
  };
  auto part = [&](int which) { return general_stage_part(eles, neb, all.data(), (int)all.size(), rk, rk == nst - 1, which); };
  if (start)
  {
    if (mpi_all(0)) return 1;
    if (start_exchange(comm, mpi_faces, n_mpi, 0, false)) return 1;
  }
  if (part(1)) return 1;
  if (wait_exchange(comm, 0)) return 1;
  if (visc && mpi_all(1)) return 1;
  if (part(2)) return 1;
  if (visc)
  {
    if (mpi_all(5)) return 1;
    if (start_exchange(comm, mpi_faces, n_mpi, 1, true)) return 1;
  }
  if (part(3)) return 1;
  if (visc && wait_exchange(comm, 1)) return 1;
  if (mpi_all(6)) return 1;
  if (part(4)) return 1;
  // eles::shock_capture (src/HiFiLES.cpp:214-216) before the new flux-point solution is packed: the filter, then the flux-point
  // values of the filtered state
  if (shock && general_shock_capture(eles, neb)) return 1;
  if (mpi_all(0)) return 1;
  return start_exchange(comm, mpi_faces, n_mpi, 0, false);
}

// the loop of hfx_run_steps_partitioned / hfx_time_partitioned.  n_stages_total < 0: n_steps whole time steps
static int run_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                           hfx_comm *comm, int n_steps, int n_stages_total, StageTimers *T)
{
  hfx_ctx *ctx = e->ctx;
  if (check_partition_blocks(e, mpi_faces, n_mpi, comm)) return 1;
  const int nst = n_rk_stages(ctx->params);
  bool first = true;
  int done = 0;
  for (int s = 0; n_stages_total >= 0 ? done < n_stages_total : s < n_steps; s++)
  {
    if (calc_time_step(e, comm)) return 1; /* src/HiFiLES.cpp:198 */
    for (int rk = 0; rk < nst && (n_stages_total < 0 || done < n_stages_total); rk++, done++)
    {
      if (partitioned_stage(e, int_faces, n_int, mpi_faces, n_mpi, comm, rk, first, T)) return 1;
      first = false;
    }
    advance_ramp_counters(int_faces, n_int);
  }
  // the exchange started after the last stage belongs to a stage that is not run: the compute stream waits for it so
  // that nothing is in flight when the caller reads or changes the state (a following call starts over with `first`)
  return wait_exchange(comm, 0);
}

} // namespace hfx

using namespace hfx;

extern "C" {

int hfx_comm_get_unique_id(char id[HFX_COMM_ID_BYTES])
{
  HFX_CHECK(id, "hfx_comm_get_unique_id: NULL argument");
  static_assert(HFX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (rccl_load()) return 1;
  ncclUniqueId u;
  HFX_NCCL(g_rccl.GetUniqueId(&u));
  std::memcpy(id, u.internal, HFX_COMM_ID_BYTES);
  return 0;
}

int hfx_comm_create(hfx_ctx *ctx, const char id[HFX_COMM_ID_BYTES], int nranks, int rank, hfx_comm **out)
{
  HFX_CHECK(ctx && id && out, "hfx_comm_create: NULL argument");
  HFX_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "hfx_comm_create: rank %d of %d", rank, nranks);
  if (rccl_load()) return 1;
  HFX_HIP(hipSetDevice(ctx->device));
  hfx_comm *c = new hfx_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  ncclUniqueId u;
  std::memcpy(u.internal, id, HFX_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, u, rank);
  if (r != ncclSuccess)
  {
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
    delete c;
    return 1;
  }
  c->nccl = comm;
  // (a failure from here on must not leak the communicator: the peers hold its other members)
  auto rest = [&]() -> int {
    HFX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int k = 0; k < 3; k++)
    {
      HFX_HIP(hipEventCreateWithFlags(&c->packed[k], hipEventDisableTiming));
      HFX_HIP(hipEventCreateWithFlags(&c->received[k], hipEventDisableTiming));
    }
    HFX_HIP(hipMalloc((void **)&c->scratch, sizeof(double) * 64));
    return 0;
  };
  if (rest())
  {
    const std::string msg = hfx_last_error();
    (void)hfx_comm_destroy(c);
    set_error("%s", msg.c_str());
    return 1;
  }
  *out = c;
  return 0;
}

int hfx_comm_destroy(hfx_comm *c)
{
  if (!c) return 0;
  if (c->ctx)
  {
    // what has been recorded may name this communicator: run it, then forget the plans that do
    (void)defer_flush(c->ctx, 0);
    c->ctx->defer.plans.clear();
  }
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->nccl) (void)g_rccl.CommDestroy((ncclComm_t)c->nccl);
  for (int k = 0; k < 3; k++)
  {
    if (c->packed[k]) (void)hipEventDestroy(c->packed[k]);
    if (c->received[k]) (void)hipEventDestroy(c->received[k]);
  }
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int hfx_comm_info(hfx_comm *c, int *nranks, int *rank, int *device, char pci_bus_id[32])
{
  HFX_CHECK(c && c->nccl, "hfx_comm_info: no communicator");
  // what RCCL itself says about this communicator (not what it was asked for)
  int n = 0, r = 0, d = 0;
  HFX_NCCL(g_rccl.CommCount((ncclComm_t)c->nccl, &n));
  HFX_NCCL(g_rccl.CommUserRank((ncclComm_t)c->nccl, &r));
  HFX_NCCL(g_rccl.CommCuDevice((ncclComm_t)c->nccl, &d));
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (device) *device = d;
  if (pci_bus_id)
  {
    pci_bus_id[0] = 0;
    HFX_HIP(hipDeviceGetPCIBusId(pci_bus_id, 32, d));
  }
  return 0;
}

int hfx_comm_allreduce(hfx_comm *c, double *values, int n, int op)
{
  HFX_CHECK(c && values && n >= 1 && n <= 64, "hfx_comm_allreduce: bad argument");
  HFX_CHECK(op >= 0 && op <= 2, "hfx_comm_allreduce: op must be 0 (min), 1 (max) or 2 (sum)");
  const ncclRedOp_t ops[3] = {ncclMin, ncclMax, ncclSum};
  HFX_HIP(hipMemcpyAsync(c->scratch, values, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HFX_NCCL(g_rccl.AllReduce(c->scratch, c->scratch, (size_t)n, ncclDouble, ops[op], (ncclComm_t)c->nccl, c->stream));
  HFX_HIP(hipMemcpyAsync(values, c->scratch, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HFX_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int hfx_mpi_inters_set_neighbours(hfx_inters *f, int n_seg, const int *peer, const int *send_first, const int *recv_first,
                                  const int *count)
{
  HFX_CHECK(f && f->is_mpi, "not a partition-face block");
  HFX_CHECK(n_seg >= 0 && (n_seg == 0 || (peer && send_first && recv_first && count)), "hfx_mpi_inters_set_neighbours: bad argument");
  HFX_IMMEDIATE(f->ctx, 0);
  f->ctx->defer.plans.clear();
  std::vector<char> sent((size_t)f->n_inters, 0), got((size_t)f->n_inters, 0);
  for (int s = 0; s < n_seg; s++)
  {
    HFX_CHECK(peer[s] >= 0 && count[s] > 0, "segment %d: peer %d, %d faces", s, peer[s], count[s]);
    HFX_CHECK(send_first[s] >= 0 && send_first[s] + count[s] <= f->n_inters && recv_first[s] >= 0 && recv_first[s] + count[s] <= f->n_inters,
              "segment %d leaves the block's %d faces", s, f->n_inters);
    for (int i = 0; i < count[s]; i++)
    {
      HFX_CHECK(!sent[send_first[s] + i] && !got[recv_first[s] + i], "segment %d overlaps another one", s);
      sent[send_first[s] + i] = got[recv_first[s] + i] = 1;
    }
  }
  f->seg_peer.assign(peer, peer + n_seg);
  f->seg_send.assign(send_first, send_first + n_seg);
  f->seg_recv.assign(recv_first, recv_first + n_seg);
  f->seg_count.assign(count, count + n_seg);
  return 0;
}

int hfx_mpi_inters_send_solution(hfx_inters *f, hfx_comm *c)
{
  HFX_CHECK(f && f->is_mpi && c, "hfx_mpi_inters_send_solution: bad argument");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_MPI_SEND_SOLUTION, nullptr, f, c, 0, 0);
  if (hfx_mpi_inters_pack_solution(f)) return 1;
  return start_exchange(c, &f, 1, 0, false);
}

int hfx_mpi_inters_receive_solution(hfx_inters *f, hfx_comm *c)
{
  HFX_CHECK(f && f->is_mpi && c, "hfx_mpi_inters_receive_solution: bad argument");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_MPI_RECEIVE_SOLUTION, nullptr, f, c, 0, 0);
  return wait_exchange(c, 0);
}

int hfx_mpi_inters_send_corrected_gradient(hfx_inters *f, hfx_comm *c)
{
  HFX_CHECK(f && f->is_mpi && c, "hfx_mpi_inters_send_corrected_gradient: bad argument");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_MPI_SEND_GRADIENT, nullptr, f, c, 0, 0);
  if (hfx_mpi_inters_pack_corrected_gradient(f)) return 1;
  return start_exchange(c, &f, 1, 1, false);
}

int hfx_mpi_inters_receive_corrected_gradient(hfx_inters *f, hfx_comm *c)
{
  HFX_CHECK(f && f->is_mpi && c, "hfx_mpi_inters_receive_corrected_gradient: bad argument");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_MPI_RECEIVE_GRADIENT, nullptr, f, c, 0, 0);
  return wait_exchange(c, 1);
}

int hfx_mpi_inters_send_sgsf_fpts(hfx_inters *f, hfx_comm *c)
{
  HFX_CHECK(f && f->is_mpi && c, "hfx_mpi_inters_send_sgsf_fpts: bad argument");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_MPI_SEND_SGSF, nullptr, f, c, 0, 0);
  if (hfx_mpi_inters_pack_sgsf(f)) return 1;
  return start_exchange(c, &f, 1, 2, false);
}

int hfx_mpi_inters_receive_sgsf_fpts(hfx_inters *f, hfx_comm *c)
{
  HFX_CHECK(f && f->is_mpi && c, "hfx_mpi_inters_receive_sgsf_fpts: bad argument");
  if (f->n_inters == 0) return 0;
  HFX_DEFER(f->ctx, DM_MPI_RECEIVE_SGSF, nullptr, f, c, 0, 0);
  return wait_exchange(c, 2);
}

int hfx_run_steps_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                              hfx_comm *comm, int n_steps)
{
  HFX_CHECK(e && comm, "hfx_run_steps_partitioned: NULL argument");
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  HFX_IMMEDIATE(e->ctx, 0);
  return run_partitioned(e, int_faces, n_int, mpi_faces, n_mpi, comm, n_steps, -1, nullptr);
}

int hfx_run_steps_partitioned_blocks(hfx_eles *const *eles, int n_ele_blocks, hfx_inters *const *int_faces, int n_int,
                                     hfx_inters *const *mpi_faces, int n_mpi, hfx_comm *comm, int n_steps)
{
  HFX_CHECK(eles && n_ele_blocks > 0 && eles[0] && comm, "hfx_run_steps_partitioned_blocks: NULL argument");
  hfx_ctx *ctx = eles[0]->ctx;
  HFX_CHECK(ctx->have_params, "parameters not set");
  HFX_IMMEDIATE(ctx, 0);
  HFX_CHECK(ctx->params.dt_type == 0, "hfx_run_steps_partitioned_blocks: CFL time steps (dt_type 1 / 2) over several element blocks are the caller's "
                                      "(hfx_eles_calc_dt_local per block + hfx_comm_allreduce, then one step at a time)");
  const int nst = n_rk_stages(ctx->params);
  for (int i = 0; i < n_ele_blocks; i++)
    if (hfx_eles_extrapolate_solution(eles[i])) return 1;
  bool start = true;
  for (int s = 0; s < n_steps; s++)
  {
    for (int rk = 0; rk < nst; rk++)
    {
      // closures that filter the solution do so at the first stage of a step (src/solver.cpp:55-62)
      if (rk == 0)
        for (int i = 0; i < n_ele_blocks; i++)
          if (eles[i]->les_ready && eles[i]->les.sgs_model >= 2 && hfx_eles_calc_sgs_terms(eles[i])) return 1;
      if (general_partitioned_stage(eles, n_ele_blocks, int_faces, n_int, mpi_faces, n_mpi, comm, rk, start, true)) return 1;
      start = false;
    }
    advance_ramp_counters(int_faces, n_int);
  }
  // (nothing in flight when the caller reads or changes the state)
  return n_steps > 0 ? wait_exchange(comm, 0) : 0;
}

int hfx_time_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                         hfx_comm *comm, int reps, double ms[8])
{
  HFX_CHECK(e && comm && ms && reps > 0, "hfx_time_partitioned: bad argument");
  HFX_CHECK(e->ctx->have_params, "parameters not set");
  HFX_IMMEDIATE(e->ctx, 0);
  StageTimers T;
  if (T.create()) return 1;
  const int rc = run_partitioned(e, int_faces, n_int, mpi_faces, n_mpi, comm, 0, reps, &T);
  T.destroy();
  if (rc) return 1;
  for (int i = 0; i < 8; i++) ms[i] = T.acc[i] / reps;
  return 0;
}

} // extern "C"
