// fused_hex.hpp -- fused per-stage kernels for tensor-product elements (declarations).
#pragma once
#include "hfx_internal.hpp"

namespace hfx
{
// drop any fused-path tables derived from the block's face registration
void fused_invalidate(hfx_eles *e);
void fused_destroy(hfx_eles *e);
// n_steps time steps with the fused kernels; fails loudly when the block does not qualify
int fused_run_steps(hfx_eles *e, hfx_inters *const *faces, int nfb, int n_steps);
// average duration (ms, HIP events on the context stream) of each fused kernel over `reps` stages
int fused_time_kernels(hfx_eles *e, hfx_inters *const *faces, int nfb, int reps, double *ms, char *names, int names_len);
// algorithmic HBM bytes per launch of each fused kernel, same order as fused_time_kernels
void fused_kernel_bytes(const hfx_eles *e, double *bytes);
// the split variant (mode 2): pairwise face kernels + per-element kernels, four launches per stage
int split_run_steps(hfx_eles *e, hfx_inters *const *faces, int nfb, int n_steps, int variant = 2);
int split_time_kernels(hfx_eles *e, hfx_inters *const *faces, int nfb, int reps, double *ms, char *names, int names_len,
                       int variant = 2);
void split_kernel_bytes(const hfx_eles *e, double *bytes, int variant = 2);
// one phase of a split-path stage on a partitioned block (see hfx_stage_partitioned)
int split_stage_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                            int phase, int in_step, int first);
} // namespace hfx
