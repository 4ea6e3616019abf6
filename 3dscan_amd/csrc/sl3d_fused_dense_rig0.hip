// the 3-step timed dense kernels of rig class 0 (sl3d_fused.h): N = 6..12 exact, NMAX = 6..12 padded (unequal axes) + the unroll bound 16, each with and
// without the LDS reciprocal table
#include "sl3d_fused.h"
namespace sl3d {
void fused_dense_rig0(SL3D_FUSED_FAMILY_ARGS) { launch_fused_n<false, false, 0, 0>(nv, nh, grid, st, P, C, first_view, n_views, vpt); }
}  // namespace sl3d
