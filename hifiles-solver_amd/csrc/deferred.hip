// deferred.hip -- deferred execution: the reference's UNCHANGED call sequence drives the fused stages.
//
// Reference side: CalcResidual (/root/reference/src/solver.cpp:50-223) calls seventeen methods of eles / int_inters /
// bdy_inters / mpi_inters per RK stage, the RK loop (src/HiFiLES.cpp:201-217) follows with AdvanceSolution (and
// shock_capture).  Each of them has an entry point in the C ABI (include/hfx.h) that launches its own kernels -- the
// per-method path, ~17 GB of HBM traffic per stage at 32^3 P4 where the fused stages move 3.5.
//
// With hfx_ctx_set_option(ctx, "deferred", 1) those entry points only RECORD the call.  The record is looked at when
//   * the next stage begins (extrapolate_solution / calc_sgs_terms arrives behind an AdvanceSolution), or
//   * any other entry point needs the device state (download, monitors, calc_dt_local, synchronize, new parameters ...).
// A record that is exactly one stage in CalcResidual's order -- every method over every block it belongs to, the phases in
// the reference's order, AdvanceSolution for every element block with one stage number -- runs as ONE fused stage:
//   the split stage (fused_hex.hip)           one tensor-product block, interior + boundary faces
//   the partitioned split stage (comm.hip)    the same with partition faces whose send / receive calls name one hfx_comm
//   the general stage (general.hip)           tetrahedra / prisms / several blocks
// Everything else (a partial stage, an unusual order, blocks the fused stages refuse, a caller that asks for an array the
// fused stage keeps on chip) is REPLAYED call by call, which is exactly what the calls would have done at once.
//
// What makes the two agree: the fused stages produce disu_upts(0), disu_upts(1), disu_fpts of the new state (and
// div_tconf_upts at the last stage of a step or on demand); every other array is marked stale and a download of a stale
// array fails loudly instead of returning an older stage's values.
#include <algorithm>
#include <cstring>

#include "fused_hex.hpp"
#include "general.hpp"
#include "hfx_internal.hpp"

namespace hfx
{

static const char *const method_names[DM_N_METHODS + 1] = {
    "calc_sgs_terms", "extrapolate_solution", "send_solution", "calculate_gradient", "evaluate_invFlux", "int calculate_common_invFlux",
    "evaluate_boundaryConditions_invFlux", "receive_solution", "mpi calculate_common_invFlux", "correct_gradient",
    "send_corrected_gradient", "evaluate_viscFlux", "extrapolate_sgsFlux", "send_sgsf_fpts", "extrapolate_totalFlux",
    "calculate_divergence", "int calculate_common_viscFlux", "evaluate_boundaryConditions_viscFlux", "receive_corrected_gradient",
    "receive_sgsf_fpts", "mpi calculate_common_viscFlux", "calculate_corrected_divergence", "AdvanceSolution", "shock_capture",
    "set_ramp_counter"};

static int n_rk_stages(const hfx_params &p) { return (p.adv_type == 0) ? 1 : (p.adv_type <= 2) ? 4 : (p.adv_type == 3) ? 5 : 14; }

static bool same_call(const DeferCall &a, const DeferCall &b)
{
  // (the stage number of AdvanceSolution is not part of a record's identity)
  return a.method == b.method && a.e == b.e && a.f == b.f && a.c == b.c && (a.method == DM_ADVANCE_SOLUTION || a.i0 == b.i0) && a.i1 == b.i1;
}

template <class T>
static bool contains(const std::vector<T *> &v, const T *x)
{
  return std::find(v.begin(), v.end(), x) != v.end();
}

// Is the record one whole stage in CalcResidual's order?  Fills the plan's block lists; `why` says what is different.
static bool match_stage(hfx_ctx *ctx, const std::vector<DeferCall> &log, DeferPlan &P)
{
  const hfx_params &prm = ctx->params;
  const bool visc = prm.viscous != 0;
  char buf[256];
  auto no = [&](const char *fmt, const char *a = "", const char *b = "") {
    snprintf(buf, sizeof buf, fmt, a, b);
    P.why = buf;
    return false;
  };
  if (!ctx->have_params) return no("parameters not set");
  // the blocks: element blocks by their extrapolate_solution, face blocks by their inviscid call
  std::vector<hfx_inters *> ints, bdys;
  for (const DeferCall &c : log)
  {
    if (c.method == DM_EXTRAPOLATE_SOLUTION && !contains(P.eles, c.e)) P.eles.push_back(c.e);
    if (c.method == DM_INT_COMMON_INVFLUX && !contains(ints, c.f)) ints.push_back(c.f);
    if (c.method == DM_BDY_INVFLUX && !contains(bdys, c.f)) bdys.push_back(c.f);
    if (c.method == DM_MPI_COMMON_INVFLUX && !contains(P.mpi_faces, c.f)) P.mpi_faces.push_back(c.f);
  }
  if (P.eles.empty()) return no("the record has no extrapolate_solution");
  bool any_les = false, any_sgs_terms = false, any_shock = false;
  for (hfx_eles *e : P.eles) any_les = any_les || e->les_ready;
  for (const DeferCall &c : log)
  {
    any_sgs_terms = any_sgs_terms || c.method == DM_CALC_SGS_TERMS;
    any_shock = any_shock || c.method == DM_SHOCK_CAPTURE;
  }
  // every face block registered on these element blocks must take part (the fused stages need every flux point's face)
  for (hfx_eles *e : P.eles)
    for (hfx_inters *f : e->faces_attached)
    {
      if (f->n_inters == 0) continue;
      const bool in = f->is_mpi ? contains(P.mpi_faces, f) : f->is_bdy ? contains(bdys, f) : contains(ints, f);
      if (!in) return no("a face block of an element block of the record was not called");
    }
  for (hfx_inters *f : ints)
    if (!contains(P.eles, f->left) || !contains(P.eles, f->right)) return no("an interior-face block connects an element block that is not in the record");
  for (hfx_inters *f : bdys)
    if (!contains(P.eles, f->left)) return no("a boundary-face block belongs to an element block that is not in the record");
  for (hfx_inters *f : P.mpi_faces)
    if (!contains(P.eles, f->left)) return no("a partition-face block belongs to an element block that is not in the record");
  P.faces = ints;
  P.faces.insert(P.faces.end(), bdys.begin(), bdys.end());
  const bool mpi = !P.mpi_faces.empty();
  if (any_les && !visc) return no("LES closure on an inviscid run");

  // what each phase must hold
  auto expected = [&](int m, std::vector<hfx_eles *> &E, std::vector<hfx_inters *> &F) {
    E.clear();
    F.clear();
    switch (m)
    {
    case DM_CALC_SGS_TERMS:
      if (any_sgs_terms)
        for (hfx_eles *e : P.eles)
          if (e->les_ready && e->les.sgs_model >= 2) E.push_back(e);
      break;
    case DM_EXTRAPOLATE_SOLUTION: case DM_EVALUATE_INVFLUX: case DM_EXTRAPOLATE_TOTALFLUX: case DM_CALCULATE_DIVERGENCE:
    case DM_CORRECTED_DIVERGENCE: case DM_ADVANCE_SOLUTION:
      E = P.eles;
      break;
    case DM_CALCULATE_GRADIENT: case DM_CORRECT_GRADIENT: case DM_EVALUATE_VISCFLUX:
      if (visc) E = P.eles;
      break;
    case DM_EXTRAPOLATE_SGSFLUX:
      if (visc)
        for (hfx_eles *e : P.eles)
          if (e->les_ready) E.push_back(e);
      break;
    case DM_SHOCK_CAPTURE:
      if (any_shock)
        for (hfx_eles *e : P.eles)
          if (e->shock_ready) E.push_back(e);
      break;
    case DM_INT_COMMON_INVFLUX: F = ints; break;
    case DM_BDY_INVFLUX: F = bdys; break;
    case DM_INT_COMMON_VISCFLUX: if (visc) F = ints; break;
    case DM_BDY_VISCFLUX: if (visc) F = bdys; break;
    case DM_MPI_SEND_SOLUTION: case DM_MPI_RECEIVE_SOLUTION: case DM_MPI_COMMON_INVFLUX:
      if (mpi) F = P.mpi_faces;
      break;
    case DM_MPI_SEND_GRADIENT: case DM_MPI_RECEIVE_GRADIENT: case DM_MPI_COMMON_VISCFLUX:
      if (mpi && visc) F = P.mpi_faces;
      break;
    case DM_MPI_SEND_SGSF: case DM_MPI_RECEIVE_SGSF:
      if (mpi && visc)
        for (hfx_inters *f : P.mpi_faces)
          if (f->left->les_ready) F.push_back(f);
      break;
    }
  };
  size_t pos = 0;
  std::vector<hfx_eles *> E, seenE;
  std::vector<hfx_inters *> F, seenF;
  int in_step = -1;
  for (int m = 0; m < DM_N_METHODS; m++)
  {
    expected(m, E, F);
    seenE.clear();
    seenF.clear();
    while (pos < log.size() && log[pos].method == m)
    {
      const DeferCall &c = log[pos++];
      if (c.e)
      {
        if (!contains(E, c.e) || contains(seenE, c.e)) return no("%s is called for a block it is not expected for, or twice", method_names[m]);
        seenE.push_back(c.e);
      }
      else
      {
        if (!contains(F, c.f) || contains(seenF, c.f)) return no("%s is called for a block it is not expected for, or twice", method_names[m]);
        seenF.push_back(c.f);
      }
      if (m == DM_EVALUATE_INVFLUX && (c.i0 != 0) != c.e->over_int_ready)
        return no("evaluate_invFlux / evaluate_invFlux_over_int does not follow the block's registration (the fused stages de-alias where over_int is registered)");
      if (m == DM_ADVANCE_SOLUTION)
      {
        if (c.i1 != prm.adv_type) return no("AdvanceSolution with an adv_type other than the context's");
        if (in_step >= 0 && c.i0 != in_step) return no("AdvanceSolution with different stage numbers");
        in_step = c.i0;
      }
      if (m >= DM_MPI_SEND_SOLUTION && c.c)
      {
        if (P.comm && P.comm != c.c) return no("two communicators in one stage");
        P.comm = c.c;
      }
    }
    if (seenE.size() != E.size() || seenF.size() != F.size())
      return no(pos < log.size() ? "%s is missing for a block (next in the record: %s)" : "%s is missing for a block (the record ends)", method_names[m],
                pos < log.size() ? method_names[log[pos].method] : "");
  }
  if (pos != log.size()) return no("%s is out of CalcResidual's order", method_names[log[pos].method]);
  P.sgs_terms = any_sgs_terms;
  P.shock = any_shock;
  return true;
}

// which fused stage runs this record (or none)
static void make_plan(hfx_ctx *ctx, const std::vector<DeferCall> &log, DeferPlan &P)
{
  P.signature = log;
  P.kind = 0;
  if (!match_stage(ctx, log, P)) return;
  const bool mpi = !P.mpi_faces.empty();
  hfx_eles *e0 = P.eles[0];
  const bool tensor = P.eles.size() == 1 && (e0->ele_type == 4 || e0->ele_type == 1);
  if (mpi)
  {
    if (!P.comm) { P.why = "partition faces without the library's communicator"; return; }
    if (P.comm->ctx != ctx) { P.why = "the communicator belongs to another context"; return; }
    if (e0->les_ready && e0->les.sgs_model == 3) { P.why = "the SVV closure filters the state after its flux-point values have left for the neighbours"; return; }
  }
  if (tensor)
  {
    if (split_deferred_prepare(e0, P.faces.data(), (int)P.faces.size(), mpi))
    {
      P.why = hfx_last_error();
      return;
    }
    if (e0->over_int_ready && split_variant(e0) != 3) { P.why = "the split variant that keeps the gradients (LES, fused mode 2) has no over-integration"; return; }
    P.kind = mpi ? 2 : 1;
    return;
  }

  {
    std::vector<hfx_inters *> all = P.faces;
    all.insert(all.end(), P.mpi_faces.begin(), P.mpi_faces.end());
    if (general_deferred_prepare(P.eles.data(), (int)P.eles.size(), all.data(), (int)all.size()))
    {
      P.why = hfx_last_error();
      return;
    }
  }
  P.kind = mpi ? 4 : 3;
}

// arrays a fused stage of this kind leaves with the reference's values (given write_div)
static unsigned fresh_after(const DeferPlan &P, bool write_div)
{
  unsigned m = (1u << HFX_DISU_UPTS0) | (1u << HFX_DISU_UPTS1) | (1u << HFX_DISU_FPTS) | (1u << HFX_SRC_UPTS) | (1u << HFX_DT_LOCAL) |
               (1u << HFX_SENSOR) | (1u << HFX_DISUF_UPTS) | (1u << HFX_LU) | (1u << HFX_LE);
  if (write_div) m |= 1u << HFX_DIV_TCONF_UPTS;
  return m;
}

int replay_call(const DeferCall &c)
{
  switch (c.method)
  {
  case DM_CALC_SGS_TERMS: return hfx_eles_calc_sgs_terms(c.e);
  case DM_EXTRAPOLATE_SOLUTION: return hfx_eles_extrapolate_solution(c.e);
  case DM_MPI_SEND_SOLUTION: return hfx_mpi_inters_send_solution(c.f, c.c);
  case DM_CALCULATE_GRADIENT: return hfx_eles_calculate_gradient(c.e);
  case DM_EVALUATE_INVFLUX: return c.i0 ? hfx_eles_evaluate_invFlux_over_int(c.e) : hfx_eles_evaluate_invFlux(c.e);
  case DM_INT_COMMON_INVFLUX: return hfx_int_inters_calculate_common_invFlux(c.f);
  case DM_BDY_INVFLUX: return hfx_bdy_inters_evaluate_boundaryConditions_invFlux(c.f, 0.0);
  case DM_MPI_RECEIVE_SOLUTION: return hfx_mpi_inters_receive_solution(c.f, c.c);
  case DM_MPI_COMMON_INVFLUX: return hfx_mpi_inters_calculate_common_invFlux(c.f);
  case DM_CORRECT_GRADIENT: return hfx_eles_correct_gradient(c.e);
  case DM_MPI_SEND_GRADIENT: return hfx_mpi_inters_send_corrected_gradient(c.f, c.c);
  case DM_EVALUATE_VISCFLUX: return hfx_eles_evaluate_viscFlux(c.e);
  case DM_EXTRAPOLATE_SGSFLUX: return hfx_eles_extrapolate_sgsFlux(c.e);
  case DM_MPI_SEND_SGSF: return hfx_mpi_inters_send_sgsf_fpts(c.f, c.c);
  case DM_EXTRAPOLATE_TOTALFLUX: return hfx_eles_extrapolate_totalFlux(c.e);
  case DM_CALCULATE_DIVERGENCE: return hfx_eles_calculate_divergence(c.e);
  case DM_INT_COMMON_VISCFLUX: return hfx_int_inters_calculate_common_viscFlux(c.f);
  case DM_BDY_VISCFLUX: return hfx_bdy_inters_evaluate_boundaryConditions_viscFlux(c.f, 0.0);
  case DM_MPI_RECEIVE_GRADIENT: return hfx_mpi_inters_receive_corrected_gradient(c.f, c.c);
  case DM_MPI_RECEIVE_SGSF: return hfx_mpi_inters_receive_sgsf_fpts(c.f, c.c);
  case DM_MPI_COMMON_VISCFLUX: return hfx_mpi_inters_calculate_common_viscFlux(c.f);
  case DM_CORRECTED_DIVERGENCE: return hfx_eles_calculate_corrected_divergence(c.e);
  case DM_ADVANCE_SOLUTION: return hfx_eles_AdvanceSolution(c.e, c.i0, c.i1);
  case DM_SHOCK_CAPTURE: return hfx_eles_shock_capture(c.e);
  case DM_SET_RAMP_COUNTER: return hfx_bdy_inters_set_ramp_counter(c.f, c.i0);
  }
  set_error("deferred execution: unknown method %d in the record", c.method);
  return 1;
}

static int run_fused(hfx_ctx *ctx, const DeferPlan &P, int in_step, bool write_div)
{
  const int nst = n_rk_stages(ctx->params);
  HFX_CHECK(in_step >= 0 && in_step < nst, "AdvanceSolution: stage %d out of range for adv_type %d", in_step, ctx->params.adv_type);
  // closures that filter the solution do so at the first stage of a step (src/solver.cpp:55-62); the SVV closure replaces the
  // state, whose flux-point values are then recomputed below
  if (P.sgs_terms)
    for (hfx_eles *e : P.eles)
      if (e->les_ready && e->les.sgs_model >= 2 && hfx_eles_calc_sgs_terms(e)) return 1;
  bool start = false; // partitioned: the flux-point solution of this state has not left yet
  for (hfx_eles *e : P.eles)
  {
    if (!e->fpts_valid && hfx_eles_extrapolate_solution(e)) return 1;
    start = start || !e->fpts_sent;
  }
  hfx_eles *e0 = P.eles[0];
  hfx_inters *const *faces = P.faces.data();
  const int nfb = (int)P.faces.size();
  switch (P.kind)
  {
  case 1:
    if (split_deferred_stage(e0, faces, nfb, in_step, write_div, P.shock)) return 1;
    break;
  case 2:
    if (partitioned_stage_deferred(e0, faces, nfb, P.mpi_faces.data(), (int)P.mpi_faces.size(), P.comm, in_step, start)) return 1;
    break;
  case 3:
    if (general_deferred_stage(P.eles.data(), (int)P.eles.size(), faces, nfb, in_step, write_div)) return 1;
    if (P.shock && general_shock_capture(P.eles.data(), (int)P.eles.size())) return 1;
    break;
  case 4:
    if (general_partitioned_stage(P.eles.data(), (int)P.eles.size(), faces, nfb, P.mpi_faces.data(), (int)P.mpi_faces.size(), P.comm, in_step, start, P.shock))
      return 1;
    break;
  default:
    HFX_CHECK(false, "deferred execution: plan of unknown kind %d", P.kind);
  }
  const unsigned fresh = fresh_after(P, write_div);
  for (hfx_eles *e : P.eles)
  {
    e->fpts_valid = true;
    e->fpts_sent = P.kind == 2 || P.kind == 4;
    e->stale |= ~fresh & ((1u << HFX_N_ARRAYS) - 1u);
    e->stale &= ~fresh;
  }
  return 0;
}

int defer_record(hfx_ctx *ctx, int method, hfx_eles *e, hfx_inters *f, hfx_comm *c, int i0, int i1)
{
  Deferred &d = ctx->defer;
  // a new stage begins behind the previous one's AdvanceSolution / shock_capture: that one is complete, run it
  if (!d.log.empty() && d.log.back().method >= DM_ADVANCE_SOLUTION && method < DM_ADVANCE_SOLUTION)
    if (defer_flush(ctx, 0)) return 1;
  DeferCall k;
  k.method = method; k.e = e; k.f = f; k.c = c; k.i0 = i0; k.i1 = i1;
  d.log.push_back(k);
  // (a caller that never reaches AdvanceSolution -- a residual evaluation in a loop -- must not grow the record for ever)
  if (d.log.size() > 4096) return defer_flush(ctx, ~0u);
  return 0;
}

int defer_flush(hfx_ctx *ctx, unsigned need)
{
  Deferred &d = ctx->defer;
  if (d.busy || d.log.empty()) return 0;
  DeferBusy busy(ctx);
  std::vector<DeferCall> log, after;
  log.swap(d.log);
  // settings recorded behind the stage (ramp counters) take effect once it has run
  while (!log.empty() && log.back().method == DM_SET_RAMP_COUNTER)
  {
    after.insert(after.begin(), log.back());
    log.pop_back();
  }
  struct ApplyAfter
  {
    std::vector<DeferCall> &v;
    ~ApplyAfter()
    {
      for (const DeferCall &c : v) (void)hfx_bdy_inters_set_ramp_counter(c.f, c.i0);
    }
  } apply_after{after};
  // the plan for this record (a run sees one or two distinct records)
  DeferPlan *plan = nullptr;
  for (DeferPlan &p : d.plans)
    if (p.signature.size() == log.size() && std::equal(log.begin(), log.end(), p.signature.begin(), same_call))
    {
      plan = &p;
      break;
    }
  if (!plan)
  {
    if (d.plans.size() >= 8) d.plans.erase(d.plans.begin());
    d.plans.emplace_back();
    plan = &d.plans.back();
    make_plan(ctx, log, *plan);
  }
  if (plan->kind != 0)
  {
    int in_step = 0;
    for (const DeferCall &c : log)
      if (c.method == DM_ADVANCE_SOLUTION) in_step = c.i0;
    // div_tconf_upts is stored at the last stage of a step (where the monitors read it) or when this flush is for it
    // (the partitioned stage stores it at the last stage only)
    const bool write_div = in_step == n_rk_stages(ctx->params) - 1 || (plan->kind != 2 && plan->kind != 4 && (need & (1u << HFX_DIV_TCONF_UPTS)) != 0);
    if ((need & ~fresh_after(*plan, write_div)) == 0)
    {
      d.n_fused++;
      return run_fused(ctx, *plan, in_step, write_div);
    }
    d.last_why = "the caller asked for an array the fused stage keeps on chip";
  }
  else
    d.last_why = plan->why;
  d.n_replayed++;
  for (const DeferCall &c : log)
    if (replay_call(c)) return 1;
  return 0;
}

} // namespace hfx
