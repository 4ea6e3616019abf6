// the MASKIN launches of rig class 2 (sl3d_fused.h: the fused kernel evaluates the views' raw selection itself -- new mask + one view
// in ONE launch): the pipelined small-launch instantiation of every N = 6..12, exact and padded, dense and segmented clouds, and the
// gated large-launch one for views known to be sparsely selected
#include "sl3d_fused.h"
namespace sl3d {
void fused_maskin_rig2(int cmode, bool gated, SL3D_FUSED_FAMILY_ARGS)
{
    if (gated) {
        if (cmode & 2) launch_fused_maskin_n<2, 6, true>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
        else launch_fused_maskin_n<2, 4, true>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
    } else {
        if (cmode & 2) launch_fused_maskin_n<2, 6, false>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
        else launch_fused_maskin_n<2, 4, false>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
    }
}
}  // namespace sl3d
