// the instantiations off the benchmarked path (sl3d_fused.h): the parity mode (KEEP: every stage-boundary plane of the reference,
// rig class 0) and the 4-step / all-invalid 5-step fringes (FGEN: 3/wrapped_phase.cpp:188-229), dense and segmented
#include "sl3d_fused.h"
namespace sl3d {
void fused_parity(bool fgen, SL3D_FUSED_FAMILY_ARGS)
{
    if (fgen) launch_fused_n<true, true, 0, 0>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
    else launch_fused_n<true, false, 0, 0>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
}
template <int CMODE>
static void fgen_rig(int rig, SL3D_FUSED_FAMILY_ARGS)
{
    if (rig == 1) launch_fused_n<false, true, 1, CMODE>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
    else if (rig == 2) launch_fused_n<false, true, 2, CMODE>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
    else launch_fused_n<false, true, 0, CMODE>(nv, nh, grid, st, P, C, first_view, n_views, vpt);
}
void fused_fgen(int rig, int cmode, SL3D_FUSED_FAMILY_ARGS)
{
    if (cmode == 2) fgen_rig<2>(rig, nv, nh, grid, st, P, C, first_view, n_views, vpt);
    else fgen_rig<0>(rig, nv, nh, grid, st, P, C, first_view, n_views, vpt);
}
}  // namespace sl3d
