// general.hpp -- the fused RK stage for general (non-tensor-product) element classes (declarations; general.hip).
#pragma once
#include "hfx_internal.hpp"

namespace hfx
{
// drop / release the tables derived from the block's operators and face registration
void general_invalidate(hfx_eles *e);
void general_destroy(hfx_eles *e);
// n_steps time steps over several element blocks (a mixed mesh) and the face blocks between them; fails loudly when a
// block does not qualify (2-D, LES, over-integration); shock capturing follows every stage (general_shock_capture)
int general_run_steps(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int n_steps);
// average duration (ms, HIP events on the context stream) of the stage's four parts over `reps` stages: face_delta,
// flux kernels (all blocks), face_flux, update kernels (all blocks)
int general_time_kernels(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int reps, double *ms);
// algorithmic HBM bytes per launch group, same order
void general_kernel_bytes(hfx_eles *const *eles, int neb, double *bytes);
// ---- the deferred scheduler's pieces (deferred.hip): tables for these blocks (non-zero when a block does not qualify), and
// ONE stage on blocks whose disu_fpts belong to the current state
int general_deferred_prepare(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb);
// pieces for the partitioned driver (comm.hip): one of the four parts of a stage; the projected viscous flux array of a block
int general_stage_part(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int in_step, bool write_div, int which);
const double *general_fn_fpts(const hfx_eles *e);
int general_deferred_stage(hfx_eles *const *eles, int neb, hfx_inters *const *faces, int nfb, int in_step, bool write_div);
// eles::shock_capture of the blocks that registered it, and the flux-point values of the filtered state
int general_shock_capture(hfx_eles *const *eles, int neb);
} // namespace hfx
