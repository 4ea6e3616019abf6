// the 3-step timed kernels that write segmented ordered clouds (CMODE 2, sl3d_run_clouds), all three rig classes (sl3d_fused.h)
#include "sl3d_fused.h"
namespace sl3d {
void fused_clouds_rig0(SL3D_FUSED_FAMILY_ARGS) { launch_fused_n<false, false, 0, 2>(nv, nh, grid, st, P, C, first_view, n_views, vpt); }
void fused_clouds_rig1(SL3D_FUSED_FAMILY_ARGS) { launch_fused_n<false, false, 1, 2>(nv, nh, grid, st, P, C, first_view, n_views, vpt); }
void fused_clouds_rig2(SL3D_FUSED_FAMILY_ARGS) { launch_fused_n<false, false, 2, 2>(nv, nh, grid, st, P, C, first_view, n_views, vpt); }
}  // namespace sl3d
