// the MASKIN launches of rig class 3 (sl3d_fused.h: the fused kernel evaluates the views' raw selection itself -- new mask + one view
// in ONE launch): the pipelined small-launch instantiation of every N = 6..12, exact and padded, dense and segmented clouds
#include "sl3d_fused.h"
namespace sl3d {
void fused_maskin_rig3(int cmode, SL3D_FUSED_FAMILY_ARGS)
{
    if (cmode & 2) launch_fused_maskin_n<3, 6>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
    else launch_fused_maskin_n<3, 4>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
}
}  // namespace sl3d
