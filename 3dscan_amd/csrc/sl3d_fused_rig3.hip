// the 3-step timed kernels of rig class 3 (sl3d_fused.h: plain projector K, purely radial distortion, the table in LDS), dense and
// segmented clouds
#include "sl3d_fused.h"
namespace sl3d {
void fused_dense_rig3(SL3D_FUSED_FAMILY_ARGS) { launch_fused_n<false, false, 3, 0>(nv, nh, grid, st, P, C, first_view, n_views, vpt); }
void fused_clouds_rig3(SL3D_FUSED_FAMILY_ARGS) { launch_fused_n<false, false, 3, 2>(nv, nh, grid, st, P, C, first_view, n_views, vpt); }
}  // namespace sl3d
