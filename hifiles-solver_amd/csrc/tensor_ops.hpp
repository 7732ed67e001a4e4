// tensor_ops.hpp -- sum-factorised over-integration and shock capturing (tensor_ops.hip).
#pragma once
#include <vector>

#include "hfx_internal.hpp"

namespace hfx
{
// M (nr1^nd x nc1^nd, column-major; first direction fastest) == A (x) ... (x) A to tol * max|M| ?  A: nr1 x nc1
bool kron_factor(const double *M, int nd, int nr1, int nc1, std::vector<double> &A, double tol);
// recover and verify the 1-D factors of the registered dense matrices; leaves the tensor path off (and returns 0) when
// the element class or the matrices do not qualify -- the dense MFMA contractions then stay in charge
int tensor_over_int_setup(hfx_eles *e, int n_cubpts, const double *opp_over_int_cubpts, const double *over_int_filter);
int tensor_shock_setup(hfx_eles *e, const double *inv_vandermonde, const double *exp_filter, const double *norm_basis_persson,
                       const int *high_modes);
bool tensor_over_int_available(const hfx_eles *e);
bool tensor_shock_available(const hfx_eles *e);
// disu_upts(0) -> tdisf_upts (the de-aliased transformed inviscid flux); folded: -> sum_l Dc[l]-along-l of it, n_fields values
// per solution point in the first n_fields planes of tdisf_upts (tensor_over_int_set_fold gave the 1-D operators Dc[l], N x N
// row-major each -- the fused stage's divergence-minus-correction matrices)
int tensor_over_int_launch(hfx_eles *e, bool folded = false);
int tensor_over_int_set_fold(hfx_eles *e, const double *Dc);
bool tensor_over_int_folded(const hfx_eles *e);
// sensor, and the filtered state where sensor >= s0; refresh_disu_fpts: also the flux-point solution of those elements
int tensor_shock_launch(hfx_eles *e, bool refresh_disu_fpts = false);
void tensor_ops_destroy(hfx_eles *e);
} // namespace hfx
