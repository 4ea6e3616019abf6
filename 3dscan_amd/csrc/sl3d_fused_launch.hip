// sl3d_fused_launch.hip -- launch_fused: grid shape, views per lane and the choice of the k_fused instantiation.  The
// instantiations themselves live in the sl3d_fused_*.hip translation units (one family each; sl3d_fused.h).
#include <stdio.h>
#include <stdlib.h>

#include "sl3d_fused.h"

namespace sl3d {

// number of 1024-pixel tiles (= blocks along x that own pixels) of one view
int fused_tiles(const KParams &P)
{
    const long quads = (long)(P.pitch >> 2) * P.H;
    return (int)((quads + SL3D_BLOCK - 1) / SL3D_BLOCK);
}

// views per lane: as many as possible up to SL3D_VPT_MAX (amortises the set-up of a block and the camera table entries) while
// the grid still has >= ~16 blocks per CU to balance the tail (8 until the end of round 4: with the repaired pipeline 3 / 4 views of
// 1080p run 1 % / 0.4 % faster at one view per lane than at two, 8 views the same at two as at four -- profiles/r04_vpt_final.txt).
// (Rounds 1-2: at most 8 views per lane.  With the stores streaming past the L2 the optimum moved: 16 views per launch 354.5 us at 4 against 359.6 at 8 and 359.3 at 2, 372.7 at 16; 32 views 700 against 708,
// profiles/r03_vpt_sweep4.txt; configs[2], 3 views of 12 Mpx: 1 / 2 / 3 views per lane 0.670 / 0.681 / 0.685,
// profiles/c2_r04_views_vpt_sweep.txt.)
#define SL3D_VPT_MAX 4
#define SL3D_BLOCK_SLOTS 1024 /* blocks of the fused kernel resident at once: 256 CUs x SL3D_OCC */
static int views_per_lane(unsigned bx, int n_views, int cam_table_kind, bool small)
{
    int vpt = 1;
    // (a two-double camera table -- tangential terms -- costs a block 16 B/px: those rigs keep 8 views per lane, measured -0.6 % at 4)
    const int cap = cam_table_kind == 2 ? 8 : SL3D_VPT_MAX;
    // (large launches -- early requests, the block's LDS tables filled under them -- want their 4 views per lane as soon as one round
    // of blocks is left: 5 / 6 / 8 / 12 views of 1080p +1.3...1.7 % at 4 views per lane against 2, profiles/r04_vpt_final.txt)
    const long min_blocks = small ? 4096 : SL3D_BLOCK_SLOTS;
    while (vpt < cap && vpt < n_views && (long)bx * ((n_views + 2 * vpt - 1) / (2 * vpt)) >= min_blocks) vpt *= 2;
#ifdef SL3D_MEASURE
    if (getenv("SL3D_VPT") && atoi(getenv("SL3D_VPT")) >= 1) vpt = atoi(getenv("SL3D_VPT"));
#endif
    return vpt;
}

// (more than 12 Gray planes on an axis: the general kernel whatever the calibration is -- it evaluates any rig, and it is the only
// one whose per-plane-test form does not spill)
static int timed_rig(const KParams &P, int rig)
{
    if (P.Nv > 12 || P.Nh > 12 || P.Nv == 0 || P.Nh == 0) return 0;  // (an axis without Gray planes: no plane to pad the straight-line kernels with)
    return rig == 1 ? 1 : (rig == 2 && P.proj_disp) ? 2 : (rig == 3 && P.proj_rad && P.F == 3) ? 3 : 0;
}

// rig: 0 / 1 / 2 / 3 (sl3d_fused.h; the host knows the calibration, the timed kernels fold it at compile time).
// cmode: 0 = dense xyz + valid planes, 2 = segmented clouds (KParams::clouds / seg_counts must be set).
// Returns the hipError_t of THIS launch.
bool fused_maskin_available(const KParams &P, int rig, int n_views, bool keep)
{
    // (a MASKIN kernel compiles the one-double camera table only: its mask words live in the registers of the two-double kind)
    if (keep || P.F != 3 || n_views > SL3D_SMALL_LAUNCH_VIEWS || timed_rig(P, rig) == 0 || (P.cam_tab != nullptr && P.cam_tab_kind == 2)) return false;
    // the small-launch form with early requests, or (views known to be sparsely selected) the gated large-launch form: both exist for
    // every pattern set the pipelined kernels take
    const FusedChoice c = choose_fused(false, false, 0, P.Nv, P.Nh, n_views, false, timed_rig(P, rig));
    return c.small && c.early;
}

unsigned fused_maskin_part_stride(const KParams &P)
{
    const long quads = (long)(P.pitch >> 2) * P.H;
    return 4u * (((unsigned)((quads + SL3D_SMALL_BLOCK - 1) / SL3D_SMALL_BLOCK) + 7u) & ~7u);
}
unsigned fused_maskin_part_words(const KParams &P) { return (unsigned)(((long)(P.pitch >> 2) * P.H + 63) / 64); }

int launch_fused(const KParams &P_, const DevCal *d_cal, int rig, int first_view, int n_views, bool keep, int cmode, void *stream, bool prefer_gated,
                 const MaskIn *mi)
{
    KParams P = P_;
    P.prefer_gated = prefer_gated ? 1 : 0;
    if (mi) {
        if (!fused_maskin_available(P, rig, n_views, keep)) return (int)hipErrorInvalidValue;
        P.mi = *mi;
    }
    const long quads = (long)(P.pitch >> 2) * P.H;
    const unsigned bx = ((unsigned)((quads + SL3D_BLOCK - 1) / SL3D_BLOCK) + 7u) & ~7u;  // a multiple of 8: consecutive tiles go round the 8 XCDs
    // (`small` is the SIZE of the launch, not the kernel it takes: a launch of up to 4 sparsely selected views runs the large-launch
    // kernel (prefer_gated) but keeps the views-per-lane rule of small launches -- the A/B that chose that kernel for sparse
    // selections, profiles/r04_sparse_mask.txt, was measured with it; following the kernel instead is 5-12 % slower at 19 % / 5 %
    // coverage with 4 views per launch, 1-4 % faster at 50 %: profiles/r05_sparse_small_launch_vpt_ab.txt)
    // (a MASKIN launch: one view per item -- nothing of a next view is in flight beside the selection bytes)
    const int vpt = mi ? 1 : views_per_lane(bx, n_views, P.cam_tab != nullptr ? P.cam_tab_kind : 0, !keep && P.F == 3 && n_views <= SL3D_SMALL_LAUNCH_VIEWS);
    const dim3 grid(bx, (unsigned)((n_views + vpt - 1) / vpt), 1);
    // the timed kernels read the camera-side T1 from the per-calibration table whatever the batch is: with 8 views per lane it
    // costs nothing (1 B/px/view), with 1..4 it saves the iteration (+2..13 %), and a view's result does not depend on the
    // batch it was launched in
    P.use_cam_table = P.cam_tab != nullptr ? P.cam_tab_kind : 0;
    const int r = timed_rig(P, rig);
#ifdef SL3D_MEASURE
    if (getenv("SL3D_CAMTAB") && atoi(getenv("SL3D_CAMTAB")) == 0) P.use_cam_table = 0;
#endif
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();  // an earlier sticky error of another library is not this launch's
    if (mi) {
        (r == 1 ? fused_maskin_rig1 : r == 2 ? fused_maskin_rig2 : fused_maskin_rig3)(cmode, prefer_gated, P.Nv, P.Nh, grid, st, P, d_cal, first_view, n_views, vpt);
    } else if (keep) {
        fused_parity(P.F != 3, P.Nv, P.Nh, grid, st, P, d_cal, first_view, n_views, vpt);
    } else if (P.F != 3) {
        fused_fgen(r, cmode, P.Nv, P.Nh, grid, st, P, d_cal, first_view, n_views, vpt);
    } else if (cmode == 2) {
        (r == 1 ? fused_clouds_rig1 : r == 2 ? fused_clouds_rig2 : r == 3 ? fused_clouds_rig3 : fused_clouds_rig0)(P.Nv, P.Nh, grid, st, P, d_cal, first_view, n_views, vpt);
    } else {
        (r == 1 ? fused_dense_rig1 : r == 2 ? fused_dense_rig2 : r == 3 ? fused_dense_rig3 : fused_dense_rig0)(P.Nv, P.Nh, grid, st, P, d_cal, first_view, n_views, vpt);
    }
    return (int)hipGetLastError();
}

// the instantiation launch_fused picks for such a launch, spelled as rocprofv3 prints it (bench.py names the kernel its roofline
// figure is about; derived from the same choose_fused / timed_rig the launch uses, so it cannot go stale)
int fused_kernel_name(const KParams &P, int rig, int n_views, bool keep, int cmode, char *buf, size_t cap, bool prefer_gated, bool maskin)
{
    const bool fgen = P.F != 3;
    const int r = keep ? 0 : timed_rig(P, rig);
    if (maskin) cmode |= 4;
    const FusedChoice c = choose_fused(keep, fgen, cmode & 2, P.Nv, P.Nh, n_views, prefer_gated, r);
    auto b = [](bool v) { return v ? "true" : "false"; };
    return snprintf(buf, cap, "sl3d::k_fused<%s, %d, %s, %s, %d, %d, %s, %s>", b(keep), c.nmax, b(fgen), b(c.exact), r, keep ? 0 : cmode, b(!c.small), b(c.early));
}

}  // namespace sl3d
