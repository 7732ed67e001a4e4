// split_partitioned.hpp -- the split fused stage of a PARTITIONED block in five phases (hfx_stage_partitioned): interior
// pairs by the pairwise kernels, partition faces by the one-sided kernels of kernels_mpi.hpp.  Included at the end of
// fused_hex.hip (it drives that file's launchers).
#pragma once

// ---------------------------------------------------------------------------------------
// split path on a partitioned block: interior pairs by the pairwise kernels, partition faces by the
// one-sided kernels of kernels_mpi.hpp; the caller exchanges the buffers between the phases
// ---------------------------------------------------------------------------------------
template <int ND>
static int mpi_launch(hfx_eles *e, hfx_inters *f, int what, const double *fn_override = nullptr)
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
  a.fn = fn_override ? fn_override : (e->fused ? e->fused->fn_fpts : nullptr);
  a.P = e->ctx->phys();
  if (e->les_ready && split_variant(e) == 2)
  {
    // the split path (variant 2) keeps sgsf_fpts in reference space: the partition-face kernels take it to physical space
    // (variant 3: the SGS flux is part of the projected flux Fn the kernels move anyway)
    if (hfx_mpi_sgsf_buffers_internal(f)) return 1;
    a.sgsf = e->arr[HFX_SGSF_FPTS]; a.jac_fpts = e->Jacobian_fpts; a.detjac_fpts = e->detjac_fpts;
    a.out_sgsf = f->out_sgsf; a.in_sgsf = f->in_sgsf; a.sgs_ref = 1;
  }
  const dim3 g((unsigned)((a.npairs + 255) / 256)), b(256);
  hipStream_t st = e->ctx->mpi_stream ? e->ctx->mpi_stream : e->ctx->stream;
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

// (with an LES closure: 3 where the flux kernel evaluates the closure itself -- needs the block's fused tables, fused_build)
// the one-sided partition-face kernels for a block of the general fused stage (three-dimensional; fn: that block's projected flux)
int mpi_launch_general(hfx_eles *e, hfx_inters *f, int what, const double *fn) { return mpi_launch<3>(e, f, what, fn); }

int split_variant(const hfx_eles *e) { return (e->ctx->fused_mode == 2 || (e->les_ready && !les_in_flux_kernel(e))) ? 2 : 3; }

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
  // ---- the pieces of phases 2 and 4 of variant 3 on their own (hfx_run_steps_partitioned puts the one-sided partition-face
  // kernels on the communication stream, ctx->mpi_stream, beside the interior face kernels of phases 1 and 3)
  case 5: return (variant == 3 && p.viscous) ? mpi_all(1) : 1; // LDG common solution at the partition faces
  case 6: return variant == 3 ? split_stage(e, int_faces, n_int, in_step, false, 2, 3) : 1;
  case 7: return (variant == 3 && p.viscous) ? mpi_all(5) : 1; // pack the projected viscous flux
  case 8: return variant == 3 ? mpi_all(6) : 1;                // common fluxes at the partition faces
  case 9:
    if (variant != 3) return 1;
    if (split_stage(e, int_faces, n_int, in_step, last, 4, variant)) return 1;
    if (e->shock_ready && shock_capture_keep_fpts(e)) return 1;
    return 0;
  case 10: return mpi_all(0); // pack the new flux-point solution
  // the update in two launches (no shock capturing: its filter follows the WHOLE update): the elements with partition-face
  // points, whose new flux-point solution then leaves while the others are updated
  // the flux kernel in three launches: the first half of the elements without partition-face points (needs nothing from the
  // neighbours), the elements with (behind the LDG corrections of the partition faces), the second half
  case 13: return (variant == 3 && !e->over_int_ready) ? split_stage(e, int_faces, n_int, in_step, false, 21, 3) : 1;
  case 14: return (variant == 3 && !e->over_int_ready) ? split_stage(e, int_faces, n_int, in_step, false, 22, 3) : 1;
  case 15: return (variant == 3 && !e->over_int_ready) ? split_stage(e, int_faces, n_int, in_step, false, 23, 3) : 1;
  case 11: return (variant == 3 && !e->shock_ready) ? split_stage(e, int_faces, n_int, in_step, last, 41, variant) : 1;
  case 12: return (variant == 3 && !e->shock_ready) ? split_stage(e, int_faces, n_int, in_step, last, 42, variant) : 1;
  default:
    HFX_CHECK(false, "hfx_stage_partitioned: phase %d out of range", phase);
  }
  return 0;
}

