// fused_hex.hpp -- the split fused stage for tensor-product elements (declarations).
#pragma once
#include "hfx_internal.hpp"

namespace hfx
{
// drop any fused-path tables derived from the block's face registration
void fused_invalidate(hfx_eles *e);
void fused_destroy(hfx_eles *e);
// the split fused stage (variants 2 and 3): pairwise face kernels + per-element kernels, three or four launches per stage;
// n_steps time steps, fails loudly when the block does not qualify
int split_run_steps(hfx_eles *e, hfx_inters *const *faces, int nfb, int n_steps, int variant = 2);
// average duration (ms, HIP events on the context stream) of each kernel of the stage over `reps` stages
int split_time_kernels(hfx_eles *e, hfx_inters *const *faces, int nfb, int reps, double *ms, char *names, int names_len,
                       int variant = 2);
// algorithmic HBM bytes per launch of each kernel, same order
void split_kernel_bytes(const hfx_eles *e, double *bytes, int variant = 2);
// which split variant a partitioned block runs: 2 (the reference's gradient arrays kept) when the context asks for it or the
// block has an LES closure, else 3
int split_variant(const hfx_eles *e);
// an LES closure evaluated inside the flux kernel of variant 3 (needs the block's fused tables)
bool les_in_flux_kernel(const hfx_eles *e);
// one phase of a split-path stage on a partitioned block (see hfx_stage_partitioned)
int split_stage_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                            int phase, int in_step, int first);
// one of the one-sided partition-face kernels (kernels_mpi.hpp: 0 pack the solution, 1 LDG corrections, 5 pack Fn, 6 common flux
// from u and Fn) on a block of the GENERAL fused stage, whose projected viscous flux is `fn`
int mpi_launch_general(hfx_eles *e, hfx_inters *f, int what, const double *fn);
// ---- the deferred scheduler's pieces (deferred.hip) ----
// builds the block's fused tables for these face blocks unless they exist; non-zero (message in hfx_last_error) when the
// block does not qualify for the split fused stage.  partitioned: flux points without a registered face are partition-face points
int split_deferred_prepare(hfx_eles *e, hfx_inters *const *faces, int nfb, bool partitioned);
// ONE stage of the split fused path (the variant split_variant(e) names) on a block whose disu_fpts belongs to the current
// state; write_div: store div_tconf_upts; shock: shock_capture follows AdvanceSolution (src/HiFiLES.cpp:214-216)
int split_deferred_stage(hfx_eles *e, hfx_inters *const *faces, int nfb, int in_step, bool write_div, bool shock);
// the same on a partitioned block with the library's transport (comm.hip); start: this state's flux-point solution has not
// been sent yet
int partitioned_stage_deferred(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                               hfx_comm *comm, int rk, bool start);
// the general fused stage on partitioned element blocks (comm.hip)
int general_partitioned_stage(hfx_eles *const *eles, int neb, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces,
                              int n_mpi, hfx_comm *comm, int rk, bool start, bool shock);
} // namespace hfx
