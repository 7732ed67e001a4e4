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
} // namespace hfx
