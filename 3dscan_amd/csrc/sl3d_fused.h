// sl3d_fused.h -- k_fused: stages 3(v) 3(h) 4(v) 4(h) 5 7 + the f32 cast of stage 8 in ONE pass over the frames (gfx950, wave64).
//
// One lane owns 4 horizontally adjacent pixels = one dword of every 8-bit plane, so a wave reads 256 contiguous bytes of each of
// the 2F+2Nv+2Nh planes (coalesced) and writes 3 KiB of xyz + 256 B of valid.  No MFMA: this is a per-pixel map bounded by HBM
// bandwidth and by fp64 VALU.  The kernel is a template; its instantiations are compiled by the sl3d_fused_*.hip translation units
// (one family each, in parallel) and picked by launch_fused (sl3d_fused_launch.hip).
//
// The body of k_fused is a sequence of named phases, each a __device__ __forceinline__ function below:
//   item_begin        what depends on the pixel only: tile -> row / column, the first valid-map dword, camera-side T1 (cam_table_*)
//   issue_fringe/gray all plane dwords of a view requested back to back (memory-level parallelism: ~12 KiB in flight per wave)
//   decode_gray       S4b, byte-parallel over the lane's 4 pixels
//   phase_A           S3c + S4c + C2 of the 4 pixels -> correspondences parked in the LDS staging area
//   gather_B/phase_B  T1 (projector table) + T2 + T3 + O1 -> xyz in the LDS staging area
//   store_quad        dense results: whole 1-KiB runs per store instruction, non-temporal
//   store_segment     segmented clouds (CMODE 2): the wave's valid points compacted in scan order into its own slot
//   parity_pixels     the parity mode (KEEP): every stage-boundary plane the reference keeps in globals
// What was measured and rejected on the way (XCD-banded tile order, buffer loads with scope bits, non-temporal 16-byte pieces,
// start stagger, un-split pixel loop, the single-pass look-back compaction, other block sizes / occupancies) is recorded in
// DESIGN.md section 4 and profiles/README.md; git history has the code.
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include "sl3d_device.h"
#include "sl3d_maskbits.h"

#define SL3D_BLOCK 256 /* threads per block: a block is a 1024-pixel tile of the scan, 4 waves = 4 segments of 256 pixels */
#define SL3D_OCC 4     /* waves per SIMD the fused kernel is compiled for (128 VGPRs) */
#define SL3D_SMALL_BLOCK SL3D_BLOCK /* the small-launch instantiation keeps 256-thread blocks too (round 4: 128 / 64 threads +-0.5 %) */

// measurement only (tools/ab.sh builds with -DSL3D_MEASURE -DSL3D_ABLATE=n): 2 = no mask reads, 4 = no xyz stores.  Results are wrong
// by construction; the shipped build has neither the compile-time switch nor the run-time hooks (SL3D_VPT / SL3D_CAMTAB
// environment variables, KParams::ablate).
#if !defined(SL3D_MEASURE) || !defined(SL3D_ABLATE)
#undef SL3D_ABLATE
#define SL3D_ABLATE 0
#endif
// measurement builds only (-DSL3D_MEASURE -DSL3D_TRACE): wall-clock stamps (100 MHz) of every wave of the dense timed kernel at
// its phase boundaries, first view of the item: 0 entry, 1 reciprocal table filled, 2 item set up (camera table entries, first
// mask dword requested), 3 plane loads issued, 4 planes landed + decoded, 5 phase A done, 6 phase B done, 7 stores issued
// -> KParams::dbg [block][wave][8] (tools/phase_trace.py)
#if defined(SL3D_MEASURE) && defined(SL3D_TRACE)
#define SL3D_STAMP(k)                                                                                                               \
    do {                                                                                                                            \
        /* every lane of the wave stores the same (scalar) clock to the same word: no divergent branch in the instrumented code */   \
        if ((CMODE & 2) == 0 && !KEEP && P.dbg)                                                                                           \
            P.dbg[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64();              \
    } while (0)
#else
#define SL3D_STAMP(k)
#endif
#ifdef SL3D_MEASURE
#define SL3D_ABLATE_RT(P) ((P).ablate)
#else
#define SL3D_ABLATE_RT(P) 0
#endif

namespace sl3d {

// RIG (stage 7 of the timed fused kernel, chosen by launch_fused from the calibration):
//   0  general: any K, any distortion, everything evaluated in the kernel with the reference's operation order
//      (also what the parity mode and the per-stage kernels run)
//   1  camera K upper triangular + affine (fx, skew, fy, cx, cy), projector without distortion and with a plain K (the reference's
//      own calibration): camera-frame least squares, the projector point is the correspondence itself
//   2  the same camera, any other projector: camera-frame least squares; the undistorted projector point comes from the
//      per-calibration table KParams::proj_disp (one float2 displacement per projector pixel, built by k_proj_table with
//      the same 5-iteration undistortion) -- the reference also tabulates it (7/triangulation.cpp:363-378), per scan
//   3  the same camera, a projector with a plain K and a purely RADIAL distortion model (both of the reference's own OpenCV projector
//      calibrations are): the undistorted point is pixel + (pixel - principal point) * s(r0^2), s from a 4-KB table of node
//      quadratics over r0^2 (RadEntry, k_radial_table) that every block copies into LDS -- no global gather in stage 7, so it runs
//      under the next view's plane loads like rig 1 (rig 2's four gathers sit BEHIND those loads in the in-order vmcnt queue)

// ---- parity mode: one pixel, everything after the byte loads (stage 4 unwrap, stage 5, stage 7, stage 8 cast), every
// stage-boundary value stored where the reference keeps it.  (cu,cv) = undistorted camera pixel coordinates of this pixel
// (T1, depends on the pixel only); (wv,wh) = wrapped phases, already shifted by +Pi where stage 4 shifts them.
struct PixelResult {
    float x, y, z;
    bool valid;
};

template <typename CalP>
__device__ __forceinline__ PixelResult parity_chain(const KParams &P, CalP Cp, const PinnedRows &PR, int gx, int gy, double cu, double cv, float wv,
                                                    float wh, int code_v, int code_h, size_t keep_off)
{
    PixelResult R;
    const float nanv = __builtin_nanf("");
    R.x = R.y = R.z = nanv;
    // stage 4: the unwrap skips the first/last column (v) or row (h) of the frame; unwrapped stays unset (0 here)
    const bool in_v = gx >= 1 && gx <= P.fullW - 2;  // 4/phase_unwrap.cpp:285
    const bool in_h = gy >= 1 && gy <= P.fullH - 2;  // 4/phase_unwrap.cpp:304
    const float uvv = unwrap_value(wv, code_v), uhv = unwrap_value(wh, code_h);
    const float uv = in_v ? uvv : 0.0f;
    const float uh = in_h ? uhv : 0.0f;
    long cx, cy;
    double cxd, cyd;
    const bool okx = correspond(uv, P.fwv, P.PW, cx, cxd);
    const bool oky = correspond(uh, P.fwh, P.PH, cy, cyd);
    R.valid = okx && oky;
    P.wrapped[0][keep_off] = wv;
    P.wrapped[1][keep_off] = wh;
    P.unwrapped[0][keep_off] = uv;
    P.unwrapped[1][keep_off] = uh;
    P.code[0][keep_off] = code_v;
    P.code[1][keep_off] = code_h;
    // rejected pixels are never compared; store 0 for those
    P.cpmap[2 * keep_off + 0] = R.valid ? cx : 0;
    P.cpmap[2 * keep_off + 1] = R.valid ? cy : 0;
    if (R.valid) {
        double up, vp, X[3];
        const auto &C = *Cp;
        if (C.proj.identity) {
            up = cxd;
            vp = cyd;
        } else {
            undistort_reproject(cxd, cyd, C.proj, up, vp);
        }
        triangulate_px(C, PR, cu, cv, up, vp, X);
        R.x = (float)X[0];  // 8/save_point_cloud.cpp:100-102
        R.y = (float)X[1];
        R.z = (float)X[2];
        P.ipoints[3 * keep_off + 0] = X[0];
        P.ipoints[3 * keep_off + 1] = X[1];
        P.ipoints[3 * keep_off + 2] = X[2];
    }
    return R;
}

// ---- timed mode: the same chain in two halves.  Stages 4 + 5 of one pixel -> its correspondence (bit-exact chain, both axes in
// one basic block), and stage 7 + the cast of stage 8 from that correspondence.
__device__ __forceinline__ bool correspond_px(const KParams &P, int gx, int gy, float wv, float wh, int code_v, int code_h, int &cx, int &cy)
{
    const bool in_v = gx >= 1 && gx <= P.fullW - 2;  // 4/phase_unwrap.cpp:285
    const bool in_h = gy >= 1 && gy <= P.fullH - 2;  // 4/phase_unwrap.cpp:304
    // keep the two phase chains out of divergent branches (the optimiser would sink each atan2 into its own `if (in range)` block
    // and serialise them) so that they interleave in one basic block
    float uvv = unwrap_value(wv, code_v), uhv = unwrap_value(wh, code_h);
    asm volatile("" : "+v"(uvv));
    asm volatile("" : "+v"(uhv));
    // select the 32-bit value (the optimiser would move the select behind the conversion to double: 2 ops each)
    float uv = in_v ? uvv : 0.0f;
    float uh = in_h ? uhv : 0.0f;
    asm volatile("" : "+v"(uv));
    asm volatile("" : "+v"(uh));
    long lx, ly;
    double dxd, dyd;
    const bool okx = correspond(uv, P.fwv, P.PW, lx, dxd);
    const bool oky = correspond(uh, P.fwh, P.PH, ly, dyd);
    cx = (int)lx;
    cy = (int)ly;
    return okx && oky;
}

template <int RIG, typename CalP>
__device__ __forceinline__ void triangulate_from(const KParams &P, CalP Cp, const PinnedRows &PR, double cu, double cv, int cx, int cy, float2 d, bool table,
                                                 const RadEntry *s_rad, float &x, float &y, float &z)
{
    const auto &C = *Cp;
    const double cxd = (double)cx, cyd = (double)cy;
    double X[3];
    bool singular = false;
    if (RIG == 3) {
        // K * ((x0, y0) * (1 + s)) with a plain K: fx*x0 + cx is the pixel itself, so the undistorted point is pixel + (pixel - c) * s
        const double dx = cxd - C.proj.cx, dy = cyd - C.proj.cy;
        const double x0 = dx * C.proj.ifx, y0 = dy * C.proj.ify;
        const double s = radial_lookup(s_rad, P.proj_rad_scale, fma(x0, x0, y0 * y0));
        triangulate_camframe(C, PR, cu, cv, fma(dx, s, cxd), fma(dy, s, cyd), X, singular);
    } else if (RIG == 1) {
        // the projector's undistort + re-project is fx*((x-cx)*(1/fx)) + cx, i.e. x itself up to 2-3 ulp (1e-13 px),
        // and (cu,cv) are the camera's undistorted NORMALISED coordinates for the camera-frame solve
        triangulate_camframe(C, PR, cu, cv, cxd, cyd, X, singular);
    } else if (RIG == 2) {
        triangulate_camframe(C, PR, cu, cv, cxd + (double)d.x, cyd + (double)d.y, X, singular);
    } else {
        double up = cxd, vp = cyd;
        if (table) {  // timed mode: the per-calibration table of the same values (see RIG 2)
            up = cxd + (double)d.x;
            vp = cyd + (double)d.y;
        } else if (!C.proj.identity) {
            undistort_reproject(cxd, cyd, C.proj, up, vp);
        }
        triangulate_px(C, PR, cu, cv, up, vp, X);
    }
    x = (float)X[0];  // 8/save_point_cloud.cpp:100-102
    y = (float)X[1];
    z = (float)X[2];
    if (RIG != 0 && singular) x = y = z = 0.0f;  // cvInvert's zero matrix: V = 0
}

// ---- the lane's work item: one quad (4 pixels of a row) of a 1024-pixel tile, for one group of `vpt` views -----------------------
// grid.x covers the quads of one window, grid.y covers groups of `vpt` views: a lane keeps its 4 pixels and walks through the
// views of its group, so whatever depends on the pixel only (the camera-side undistortion: the most expensive per-pixel constant
// of stage 7) is set up once per pixel, not once per pixel per view.
struct Item {
    unsigned tile;      // 1024-pixel tile of the scan = blockIdx.x
    int cq, row;        // quad column, window row
    int gx0, gy;        // frame coordinates of the quad's first pixel
    int v_begin, v_end; // views of this item
    bool alive;         // the lane owns pixels (segmented clouds keep the lanes past the last row: a wave stores its segment whole)
    unsigned lane_off;  // byte offset of the quad inside any plane
};

// the camera-table entries of the lane's 4 pixels: kind 1 = one double per pixel (factor of the last undistortion iteration of a
// radial model), kind 2 = the normalised point itself (tangential terms).  Request and use are separate so that the small-launch
// instantiation can put its plane requests in between.
// KIND2 = false (the MASKIN instantiations): the two-double kind is not compiled in -- its 16 registers between request and use are
// what the launch's mask words live in; a calibration with tangential camera terms keeps the two-kernel route (fused_maskin_available).
template <bool KIND2 = true>
__device__ __forceinline__ void cam_table_request(const KParams &P, const Item &it, double (&t)[8])
{
    const size_t i0 = (size_t)it.row * P.pitch + (size_t)it.cq * 4;
    if (!KIND2 || P.use_cam_table == 1) {
        const double2 *tp = (const double2 *)(P.cam_tab + i0);
        const double2 a = tp[0], b = tp[1];
        t[0] = a.x; t[1] = a.y; t[2] = b.x; t[3] = b.y;
    } else {
        const double2 *tp = (const double2 *)(P.cam_tab + 2 * i0);
        const double2 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
        t[0] = a.x; t[1] = a.y; t[2] = b.x; t[3] = b.y; t[4] = c.x; t[5] = c.y; t[6] = d.x; t[7] = d.y;
    }
}

// table entries -> the camera coordinates stage 7 uses (normalised for the camera-frame rigs, re-projected pixels for RIG 0), kept
// in LDS so that the rolled pixel loops can index them (each lane reads back only what it wrote: no barrier).  The doubles are the
// ones the in-kernel iteration produces.
template <int RIG, bool KIND2 = true>
__device__ __forceinline__ void cam_table_finish(const KParams &P, const DevCal *Cglobal, const Item &it, const double (&t)[8], double *my_cam)
{
    const auto &I = opaque_const(Cglobal)->cam;
    if (!KIND2 || P.use_cam_table == 1) {
        const double y0 = ((double)it.gy - I.cy) * I.ify;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double xn = (((double)(it.gx0 + k)) - I.cx) * I.ifx * t[k], yn = y0 * t[k];
            if (RIG == 0) reproject(xn, yn, I, xn, yn);
            my_cam[2 * k] = xn;
            my_cam[2 * k + 1] = yn;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double xn = t[2 * k], yn = t[2 * k + 1];
            if (RIG == 0) reproject(xn, yn, I, xn, yn);
            my_cam[2 * k] = xn;
            my_cam[2 * k + 1] = yn;
        }
    }
}

// ---- MASKIN (CMODE bit 4): H0 / S3b / S3d inside the fused launch -----------------------------------------------------------------
// image_scissor produces a new selection every scan (m_tech_project_console.cpp:366) and the reference's loop runs ONE scan per
// iteration: k_mask_prepare (5.5 us at 1080p, all of it launch latency and one load -> compute -> store chain) in front of a 25-us
// one-view launch was a fifth of the per-scan path.  A MASKIN launch takes the views' RAW selection instead (the staging plane behind
// sl3d_set_mask's copy, or the caller's own device-resident mask: KParams::mi) and evaluates the closed form of the boundary
// removal (3/wrapped_phase.cpp:106-115, :253-279) itself, with the very bit-plane arithmetic of k_mask_prepare (sl3d_maskbits.h, one
// row and one quad per lane) -- so "new mask + one view" is ONE kernel.
//   * Nothing waits for it.  The 3 x 8 selection bytes around the lane's quad (rows y-1 .. y+1, plane bytes own-2 .. own+5; the
//     neighbours' bytes are L2 hits; row y-2 matters to few lanes, which ask for it when they need it) are the wave's FIRST requests,
//     in front of its camera-table entries and of the 46 planes the small-launch kernels request before they know the mask anyway:
//     loads return in order, so the selection is there before anything else is, and it is evaluated while the planes travel.
//     (Round 6's first build asked for it behind the Gray decode, where 40 registers are free: one more exposed round trip per wave,
//     30.8 us per scan against 30.4 for the two kernels.)  The six registers come from the camera table: a MASKIN kernel compiles
//     only the one-double kind (KIND2 = false); one view per item, so that nothing of a next view is in flight beside them.
//   * It leaves everything k_mask_prepare would have left: the view's `band` dword (later launches over the view are ordinary ones),
//     the normalised 0/1 plane with its 2-pixel halo (the lanes of the window's first / last row and column also write the halo
//     rows / columns beside them), and per wave the number of quads that hold a valid pixel, as {seq << 8 | count} in host memory
//     mapped into the device (what sparse_views reads).
static_assert(SL3D_MASK_HALO == 2, "rows y-2 .. y+2 of the selection are plane rows row .. row + 4");
// The launch's MaskIn, read through an opaque pointer into the kernel-argument segment (KParams is the kernel's first argument): as plain
// kernel arguments its 20 dwords would be loaded once and held in SGPRs the pixel loop has not got.  ONE read per use site (a burst of
// scalar loads, one wait): the first build re-read field by field -- two dozen dependent scalar round trips in front of the wave's first
// plane request (read in the ISA; ~2 us per one-view launch).
typedef const CONST_AS MaskIn *MaskInP;
__device__ __forceinline__ MaskInP maskin_args()
{
    const CONST_AS char *ka = (const CONST_AS char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return (MaskInP)(ka + offsetof(KParams, mi));
}
struct MaskInArgs {
    const GLOBAL_AS uint8_t *src;  // the selection of this view: plane row 0, byte 0
    unsigned stride;
    int bx0, bx1, r0, r1, lo, hi;
};
__device__ __forceinline__ MaskInArgs maskin_read_args(int slot)
{
    const MaskInP M = maskin_args();
    MaskInArgs A;
    // (wave-uniform by construction; said explicitly -- the request sits under `if (alive)`, and the code generator otherwise meets a
    // uniform value in a vector register where the scalar-base load form wants it: "illegal VGPR to SGPR copy")
    const unsigned long long o = (unsigned long long)M->origin[first_lane_u32((unsigned)slot)];
    A.src = (const GLOBAL_AS uint8_t *)(((unsigned long long)first_lane_u32((unsigned)(o >> 32)) << 32) | (unsigned long long)first_lane_u32((unsigned)o));
    A.stride = first_lane_u32((unsigned)M->stride);
    A.bx0 = M->bx0; A.bx1 = M->bx1; A.r0 = M->r0; A.r1 = M->r1; A.lo = M->lo; A.hi = M->hi;
    return A;
}

// the lane's place in the mask plane
struct MaskInLane {
    int row, cq, delta;
    bool plain;  // mb_quad_plain: the quad's neighbourhood is interior to frame and region
    MbCols c;    // (column constants: the lanes that are not plain)
};
template <bool COLS>
__device__ __forceinline__ MaskInLane maskin_lane(const KParams &P, const MaskInArgs &A, const Item &it)
{
    MaskInLane m;
    m.row = it.row;
    m.cq = it.cq;
    asm volatile("" : "+v"(m.row), "+v"(m.cq));  // (recomputed where it is used, not kept from the request to the evaluation)
    const int own = SL3D_MASK_LPAD + m.cq * 4;
    m.plain = mb_quad_plain(P.col0 + m.cq * 4, P.row0 + m.row, P.fullW, P.fullH, P.col0 + A.bx0 - SL3D_MASK_LPAD, P.col0 + A.bx1 - SL3D_MASK_LPAD) &&
              own - 2 >= A.lo && own + 6 <= A.hi;
    if (COLS) {
        m.c = mb_cols(own, 4, P.col0, SL3D_MASK_LPAD, P.fullW, A.bx0, A.bx1);
        m.delta = mb_quad_delta(own, A.lo, A.hi);
    } else {
        m.c.REG = m.c.INF = m.c.INTC = 0xfffu;
        m.delta = -2;
    }
    return m;
}

// The 8 selection bytes of a plane row around the lane's quad: ONE 8-byte load at own + delta (sl3d_maskbits.h: mb_quad_delta).  (Three
// aligned dwords -- left, own, right -- were measured: 0.4 us per one-view launch slower.)
typedef unsigned mrow_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mrow_t maskin_load8(const GLOBAL_AS uint8_t *src, unsigned off)
{
    asm volatile("" : "+s"(src));
    typedef unsigned u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));
    return *(const GLOBAL_AS u32x2_unaligned *)(src + (size_t)off);
}
// ... of any lane: 0 for a row that holds no source pixels, and for a lane whose own dword holds none (the pitch padding beyond the
// region needs nothing -- its pixels lie outside the window, the dword beside it farther out still -- and beyond the frame's last column
// its load would leave a caller's mask)
__device__ __forceinline__ mrow_t maskin_row(const MaskInArgs &A, const MaskInLane &m, int pr)
{
    mrow_t w = {0u, 0u};
    if (pr >= A.r0 && pr < A.r1 && (m.c.REG & 0x0f0u)) w = maskin_load8(A.src, (unsigned)pr * A.stride + (unsigned)(SL3D_MASK_LPAD + m.cq * 4 + m.delta));
    return w;
}
__device__ __forceinline__ unsigned mrow_word(const MaskInLane &m, mrow_t w) { return mb_quad_word(w.x, w.y, m.delta) & m.c.REG; }
__device__ __forceinline__ unsigned mrow_own(const MaskInLane &m, mrow_t w) { return mb_quad_own(w.x, w.y, m.delta); }
__device__ __forceinline__ unsigned mrow_left(const MaskInLane &m, mrow_t w) { return mb_quad_left(w.x, w.y, m.delta); }
__device__ __forceinline__ unsigned mrow_right(const MaskInLane &m, mrow_t w) { return mb_quad_right(w.x, w.y, m.delta); }

// The selection bytes the lane needs of view slot `slot` of the launch: rows y-1, y, y+1 in w1..w3; row y-2 (w0) and row y+2 (w4)
// only where they matter -- the window's first / last row (halo duties), frame row 2 and the frame's first / last column (mb_quad_top_needed).
// Everything is asked for HERE, in front of the planes: a load behind them would have to wait for all 46.  Returns true (wave-uniform) if
// every lane of the wave is plain and without halo duties: such a wave asked without a predicate (every byte is a source pixel) and
// evaluates the short form (maskin_finish).
__device__ __forceinline__ bool maskin_request(const KParams &P, const Item &it, int slot, mrow_t &w0, mrow_t &w1, mrow_t &w2, mrow_t &w3, mrow_t &w4)
{
    const MaskInArgs A = maskin_read_args(slot);
    {
        const MaskInLane m = maskin_lane<false>(P, A, it);
        const bool own_duty = m.row == 0 || m.row == P.H - 1 || m.cq == 0 || m.cq == (P.pitch >> 2) - 1;
        if (__ballot(!m.plain || own_duty) == 0ull) {
            const unsigned off = (unsigned)(m.row + 1) * A.stride + (unsigned)(SL3D_MASK_LPAD + m.cq * 4 - 2);
            w1 = maskin_load8(A.src, off);
            w2 = maskin_load8(A.src + A.stride, off);
            w3 = maskin_load8(A.src + 2 * (size_t)A.stride, off);
            return true;
        }
    }
    const MaskInLane m = maskin_lane<true>(P, A, it);
    if (m.row == 0 || mb_quad_top_needed(m.c, P.row0 + m.row, P.fullH)) w0 = maskin_row(A, m, m.row);
    w1 = maskin_row(A, m, m.row + 1);
    w2 = maskin_row(A, m, m.row + 2);
    w3 = maskin_row(A, m, m.row + 3);
    if (m.row == P.H - 1) w4 = maskin_row(A, m, m.row + 4);
    return false;
}

// the normalised 0/1 bytes of plane row `pr` beside the lane's own dword: at the window's first / last quad the dword to its left / right
__device__ __forceinline__ void maskin_emit_sides(const KParams &P, uint8_t *mask_view, const MaskInLane &m, int pr, mrow_t w)
{
    unsigned *q = (unsigned *)(mask_view + (size_t)pr * P.mpitch + SL3D_MASK_LPAD + m.cq * 4);
    if (m.cq == 0) q[-1] = mb_eq1_bytes(mrow_left(m, w)) & (mb_expand_nibble(m.c.REG) * 0xffu);
    if (m.cq == (P.pitch >> 2) - 1) q[1] = mb_eq1_bytes(mrow_right(m, w)) & (mb_expand_nibble(m.c.REG >> 8) * 0xffu);
}
__device__ __forceinline__ unsigned maskin_own(const MaskInLane &m, mrow_t w) { return mb_eq1_bytes(mrow_own(m, w)) & (mb_expand_nibble(m.c.REG >> 4) * 0xffu); }
__device__ __forceinline__ void maskin_emit(const KParams &P, uint8_t *mask_view, const MaskInLane &m, int pr, mrow_t w)
{
    *(unsigned *)(mask_view + (size_t)pr * P.mpitch + SL3D_MASK_LPAD + m.cq * 4) = maskin_own(m, w);
    maskin_emit_sides(P, mask_view, m, pr, w);
}

// -> bits 0..3: the valid bits of the lane's 4 pixels (bit k = pixel k), exactly what k_mask_prepare leaves in the band plane;
// bits 4..7: the lane's own selection bits (byte == 1).  The halo rows / columns of the 0/1 plane beside the lane's own dword are
// written here (few lanes); the band dword and the own dword of the 0/1 plane leave with the view's results (maskin_store).
// fast: maskin_request's verdict on the wave -- the short form of the plain interior (sl3d_maskbits.h): ~70 instead of ~250 integer
// instructions per quad.  (The fused kernel is as much VALU- as memory-bound: the general form for every lane cost a one-view launch ~1 us.
// The same form on 64-bit lane masks -- 4 ballots per row, the recurrences on the scalar unit -- was built too, bit-identical and 1.4 us
// SLOWER: all 16 waves of a CU run this prologue at the same time and share ONE scalar unit for ~140 dependent instructions each,
// where the vector form spreads over four SIMDs; profiles/r06_fused_mask_ab.txt.)
__device__ __forceinline__ unsigned maskin_finish(const KParams &P, const Item &it, int view, int slot, mrow_t w0, mrow_t w1, mrow_t w2, mrow_t w3, mrow_t w4, bool fast)
{
    if (fast) {
        // ... and most plain waves need no arithmetic at all: a wave whose 3 x 8 bytes are ALL 1 lies inside the selection (every pixel
        // valid), one whose bytes are all 0 outside it (none is) -- 14 instructions instead of ~70; only the waves the selection's
        // outline crosses evaluate the closed form
        const unsigned all_and = w1.x & w1.y & w2.x & w2.y & w3.x & w3.y, all_or = w1.x | w1.y | w2.x | w2.y | w3.x | w3.y;
        if (__ballot(all_and != 0x01010101u || all_or != 0x01010101u) == 0ull) return 0xffu;
        if (__ballot(all_or != 0u) == 0ull) return 0u;
        const unsigned v = mb_quad_valid_plain(mb_quad_word(w1.x, w1.y, -2), mb_quad_word(w2.x, w2.y, -2), mb_quad_word(w3.x, w3.y, -2));
        return v | (mb_pack_nibble(mb_eq1_bytes(mb_quad_own(w2.x, w2.y, -2))) << 4);
    }
    const MaskInArgs A = maskin_read_args(slot);
    const MaskInLane m = maskin_lane<true>(P, A, it);
    const int gy = P.row0 + m.row;
    const bool first = m.row == 0, last = m.row == P.H - 1;
    uint8_t *mask_view = (uint8_t *)P.mask + (size_t)view * P.mask_view_stride;
    if (first) {  // the halo rows above the window ...
        maskin_emit(P, mask_view, m, m.row, w0);
        maskin_emit(P, mask_view, m, m.row + 1, w1);
    }
    maskin_emit_sides(P, mask_view, m, m.row + 2, w2);
    if (last) {  // ... and below it
        maskin_emit(P, mask_view, m, m.row + 3, w3);
        maskin_emit(P, mask_view, m, m.row + 4, w4);
    }
    const unsigned v = mb_quad_valid(mrow_word(m, w0), mrow_word(m, w1), mrow_word(m, w2), mrow_word(m, w3), m.c, gy, P.fullH) & mb_range_bits(0, P.W, m.cq * 4, 4);
    return v | (mb_pack_nibble(maskin_own(m, w2)) << 4);
}

// the view's band dword and the lane's own dword of the 0/1 plane (with the view's results: nothing of the launch waits for them)
__device__ __forceinline__ void maskin_store(const KParams &P, const Item &it, int view, unsigned bits)
{
    stg_nt(opaque_out((uint8_t *)P.band + (size_t)view * P.px_view_stride), it.lane_off, mb_expand_nibble(bits));
    stg_nt(opaque_out((uint8_t *)P.mask + (size_t)view * P.mask_view_stride), (unsigned)(it.row + SL3D_MASK_HALO) * (unsigned)P.mpitch + (unsigned)(SL3D_MASK_LPAD + it.cq * 4),
           mb_expand_nibble(bits >> 4));
}

// per wave: {seq << 8 | quads with a valid pixel}; a wave whose first lane owns no pixel owns none at all and stores nothing (the
// host expects one word per wave that owns pixels)
__device__ __forceinline__ void maskin_count(const KParams &P, const Item &it, int view, unsigned v)
{
    const MaskInP M = maskin_args();
    const unsigned cnt = (unsigned)__popcll(__ballot((v & 0xfu) != 0u));
    if ((threadIdx.x & 63u) == 0u && it.alive)
        M->part[(size_t)view * M->part_stride + (size_t)it.tile * 4u + (threadIdx.x >> 6)] = (M->seq << 8) | cnt;
}

// Everything of an item that depends on the pixel only; false if this lane has nothing to do.
// EARLY (the small-launch instantiation): the camera-table entries are only REQUESTED here (camt); the caller issues the first
// view's plane loads right behind them and then calls cam_table_finish -- one round trip instead of two in front of the first
// decode.  (Round 3 measured the other order for large launches -- set-up loads before the reciprocal-table fill, consumed behind
// the plane loads: 16 views +-0, profiles/r03_prologue_ab.txt.)
template <int RIG, bool SEG, bool EARLY, int BLK = SL3D_BLOCK, bool MASKIN = false>
__device__ __forceinline__ bool item_begin(const KParams &P, const DevCal *Cglobal, unsigned tile_, int group, int first_view, int n_views, int vpt, Item &it,
                                           MaskQuad &mq_first, double (&camt)[8], double *my_cam, mrow_t &w0, mrow_t &w1, mrow_t &w2, mrow_t &w3, mrow_t &w4, bool &mfast)
{
    const unsigned qpr = (unsigned)P.pitch >> 2;  // quads per row, pitch padding included
    it.tile = tile_;
    it.v_begin = first_view + group * vpt;  // (block-uniform values first: nothing below may make them look divergent)
    it.v_end = min(it.v_begin + vpt, first_view + n_views);
    // (32-bit unsigned: a window has fewer than 2^32 quads -- sl3d_create checks it -- and a 64-bit division by a run-time
    // value is ~120 instructions at the start of every wave)
    const unsigned q = tile_ * (unsigned)BLK + threadIdx.x;
    const int row_q = (int)(q / qpr);
    it.cq = (int)(q - (unsigned)row_q * qpr);
    // SEG: a wave stores its segment with all 64 lanes (whole 16-byte chunks, lane after lane), so the lanes past the last row
    // stay, without a valid pixel; only the blocks the grid was padded with leave (they own no segment)
    // EARLY keeps them too, until the block's barrier (their requests go to the last row: one code path, one wait count)
    if (SEG && tile_ >= (unsigned)P.n_tiles) return false;
    if (!SEG && !EARLY && row_q >= P.H) return false;
    it.alive = row_q < P.H;
    it.row = (SEG || EARLY) ? min(row_q, P.H - 1) : row_q;
    it.gx0 = P.col0 + it.cq * 4;
    it.gy = P.row0 + it.row;
    it.lane_off = (unsigned)it.row * (unsigned)P.pitch + (unsigned)it.cq * 4u;
    // the valid bits of the item's first view are requested now, so that they travel together with the camera table
    // entries below instead of after them (one round trip less before the first plane loads can leave)
    // (MASKIN: there is no valid-map dword yet -- the selection bytes around the quad are the wave's first requests instead)
    if (MASKIN) {
        mq_first.band = 0u;
        bool f = false;
        if (it.alive) f = maskin_request(P, it, min(it.v_begin, first_view + n_views - 1) - first_view, w0, w1, w2, w3, w4);
        mfast = __ballot(f) != 0ull;  // (wave-uniform by construction: a wave with a lane past the last row is never fast)
    } else {
        mq_first = load_mask_quad(P, min(it.v_begin, first_view + n_views - 1), it.lane_off);
    }
    if (EARLY && P.use_cam_table) {
        cam_table_request<!MASKIN>(P, it, camt);
        return true;
    }
    if (P.use_cam_table) {
        double t[8];
        cam_table_request<!MASKIN>(P, it, t);
        cam_table_finish<RIG, !MASKIN>(P, Cglobal, it, t, my_cam);
        return true;
    }
    // no table (a camera without distortion, or the parity mode): T1 of the camera evaluated here
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        double cu = 0.0, cv = 0.0;
        if (it.cq * 4 < P.W && !(SL3D_ABLATE_RT(P) & 4)) {
            if (RIG != 0) undistort_normalized((double)(it.gx0 + k), (double)it.gy, opaque_const(Cglobal)->cam, cu, cv);  // camera-frame solve
            else undistort_reproject((double)(it.gx0 + k), (double)it.gy, opaque_const(Cglobal)->cam, cu, cv);
        }
        my_cam[2 * k] = cu;
        my_cam[2 * k + 1] = cv;
    }
    return true;
}

// the third rows of the projection matrices and the translation, pinned in VGPRs (see PinnedRows).  They are read through the
// constant address space -- scalar loads, all of them in flight at once, one wait -- and only then copied into VGPRs.  (Until
// round 4 they were read through the plain pointer: six VECTOR loads, each followed by its own s_waitcnt vmcnt(0) because the
// register pin right behind it consumes the value -- six dependent L2 round trips at the start of every wave, in front of its
// first plane request.  The caller now also asks for them AFTER the item's first loads have been issued.)
template <int RIG>
__device__ __forceinline__ PinnedRows pinned_rows(const DevCal *Cglobal)
{
    const auto *C = opaque_const(Cglobal);
    PinnedRows PR;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        PR.c2[j] = RIG != 0 ? 0.0 : C->Ac[8 + j];
        PR.p2[j] = RIG != 0 ? C->Apc[8 + j] : C->Ap[8 + j];
        if (j < 3) PR.t[j] = RIG != 0 ? C->tcn[j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (RIG == 0) asm volatile("" : "+v"(PR.c2[j]));
        asm volatile("" : "+v"(PR.p2[j]));  // stay in VGPRs
        if (j < 3 && RIG != 0) asm volatile("" : "+v"(PR.t[j]));
    }
    return PR;
}

// ---- plane loads ---------------------------------------------------------------------------------------------------------------
// planes of a view: vertical axis (fringe F, gray Nv, inverse Nv), then the horizontal axis.  Plane offsets are added to the scalar
// view base (SALU); every load uses the same 32-bit VGPR offset.  Instruction selection works per basic block: the zero-extension
// of the lane offset has to happen in the block of the loads for them to select the (SGPR base + 32-bit VGPR offset) form, hence
// the per-call copy behind an empty asm.
template <bool FGEN>
__device__ __forceinline__ void issue_fringe(const KParams &P, int view, unsigned lane_off, int F, int Nv, unsigned (&f)[2][4])
{
    const GLOBAL_AS uint8_t *vb = opaque(P.frames + (size_t)view * P.view_stride);
    const unsigned psv = opaque_u32((unsigned)P.plane_stride);  // per-view copy: plane offsets are recomputed, not kept live
    unsigned lo = lane_off;
    asm volatile("" : "+v"(lo));
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const unsigned p0 = a == 0 ? 0u : (unsigned)(F + 2 * Nv) * psv;
        f[a][0] = ldg32(vb + (size_t)p0, lo);
        f[a][1] = ldg32(vb + (size_t)(p0 + psv), lo);
        f[a][2] = ldg32(vb + (size_t)(p0 + 2u * psv), lo);
        f[a][3] = (FGEN && F == 4) ? ldg32(vb + (size_t)(p0 + 3u * psv), lo) : 0u;
    }
}

// NMAX is the compile-time unroll bound of the Gray planes.  PLANES (how an axis with N planes maps onto it):
//   1  exact: both axes have exactly NMAX planes -- clamps and per-plane tests fold away (the benchmarked kernels)
//   2  padded: straight-line code for NMAX planes per axis whatever the axes really have (N <= NMAX).  An axis with fewer planes is
//      padded IN FRONT with NMAX - N virtual planes that decode to G = 0 -- the binary code sum B_i 2^(N-1-i) is unchanged by
//      leading zero bits -- and that cost nothing but scalar arithmetic: logical plane i reads physical plane max(i - pad, 0) (the
//      padded ones re-read plane 0: an L2 hit, never used), and the decode xors them in with an all-zero mask (decode_gray).  No
//      per-plane test, no branch, one instruction stream for every (N_v, N_h) with max(N_v, N_h) <= NMAX.  The reference's own
//      capture set is N_v = 6, N_h = 5 (global_cv.h:49-62); until round 4 unequal axes took the per-plane tests below, whose code
//      spills kilobytes: 3.8 x slower at 6 / 5, 40 x at 10 / 9 (tools/nvnh.py, profiles/r04_unequal_axes.txt).  With equal axes it is
//      1 % behind the exact form (the plane offsets are no longer compile-time multiples), which therefore stays.
//   3  padded, with the pad count taken through v_readfirstlane: the gated MASKIN kernels of rig classes 2 and 3 -- there the code
//      generator holds the (uniform) plane count in a vector register at the empty asm below and stops with "illegal VGPR to SGPR
//      copy"; the other padded kernels keep form 2 (their instruction streams are round 5's)
//   0  per-plane tests (parity mode, more than 12 planes): an axis with fewer planes skips the surplus loads through a wave-uniform
//      test
template <int NMAX, int PLANES>
__device__ __forceinline__ void issue_gray(const KParams &P, int view, unsigned lane_off, int F, int Nv, int Nh, unsigned (&g)[2][NMAX], unsigned (&iv)[2][NMAX])
{
    const GLOBAL_AS uint8_t *vb = opaque(P.frames + (size_t)view * P.view_stride);
    const unsigned psv = opaque_u32((unsigned)P.plane_stride);
    unsigned lo = lane_off;
    asm volatile("" : "+v"(lo));
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int N = a == 0 ? Nv : Nh;
        // (PLANES == 2 needs N >= 1 on both axes: an axis WITHOUT Gray planes has no plane of its own to pad with, and its padded loads
        // would read whatever follows the axis -- for the last axis of the last resident view the first bytes past the frame stack.
        // choose_fused sends such pattern sets to the per-plane-test kernel)
        const unsigned pg = (unsigned)((a == 0 ? 0 : F + 2 * Nv) + F) * psv;
        // (behind an empty asm: everything derived from it is loop-invariant, and 40 hoisted plane offsets + 20 masks are more
        // SGPRs than there are)
        int pad = NMAX - N;
        if (PLANES == 3) pad = __builtin_amdgcn_readfirstlane(pad);
        if (PLANES >= 2) asm volatile("" : "+s"(pad));
#pragma unroll
        for (int i = 0; i < NMAX; i++) {
            if (PLANES >= 2) {
                const unsigned idx = (unsigned)max(i - pad, 0);
                g[a][i] = ldg32(vb + (size_t)(pg + idx * psv), lo);
                iv[a][i] = ldg32(vb + (size_t)(pg + ((unsigned)N + idx) * psv), lo);
                continue;
            }
            g[a][i] = iv[a][i] = 0u;
            if (PLANES == 1 || i < N) {
                g[a][i] = ldg32(vb + (size_t)(pg + (unsigned)i * psv), lo);
                iv[a][i] = ldg32(vb + (size_t)(pg + (unsigned)(N + i) * psv), lo);
            }
        }
    }
}

// ---- S4b: Gray decode, byte-parallel over the 4 pixels of the lane ---------------------------------------------------------------
// G_i = (gray - inverse >= 0) (4/phase_unwrap.cpp:183) for 4 bytes at once: the low 7 bits are compared by a
// borrow-protected subtraction, bit 7 decides unless the top bits are equal (one v_bitop3 on x, y, t).
// B_0 = G_0, B_i = B_{i-1} xor G_i (:187-191) is a running xor of the masks; the code sum B_i 2^(N-1-i) (:193) is
// accumulated per byte, the LAST 8 planes in `lo`, the ones before them in `hi`, so that the 16-bit code of a pixel
// is (hi byte, lo byte) and one v_perm per pixel pair builds it: code[a][j] = codes of pixels 2j (low half), 2j+1.
// PLANES == 2 (padded, see issue_gray): NMAX positions; the NMAX - N padded planes in front are xored in with a zero mask (a scalar
// select feeds the third operand of the v_bitop3 that the plain decode feeds with the constant).
template <int NMAX, int PLANES>
__device__ __forceinline__ void decode_gray(const unsigned (&g)[2][NMAX], const unsigned (&iv)[2][NMAX], int Nv, int Nh, unsigned (&code)[2][2])
{
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int N = a == 0 ? Nv : Nh;
        const unsigned H = 0x80808080u;
        unsigned bacc = 0;  // running binary bit of pixel k at bit 8k+7
        unsigned hi = 0, lo = 0;
        if (PLANES >= 2) {
            int pad = NMAX - N;
            if (PLANES == 3) pad = __builtin_amdgcn_readfirstlane(pad);
            asm volatile("" : "+s"(pad));
#pragma unroll
            for (int i = 0; i < NMAX; i++) {
                const unsigned x = g[a][i], y = iv[a][i];
                const unsigned t = (x | H) - (y & ~H);
                const unsigned ge = __builtin_amdgcn_bitop3_b32(x, y, t, 0xB2);
                const unsigned Hm = i >= pad ? H : 0u;  // (wave-uniform: an SGPR operand)
                bacc = __builtin_amdgcn_bitop3_b32(bacc, ge, Hm, 0x78);
                if (i < NMAX - 8) hi = (hi << 1) | (bacc >> 7);
                else lo = (lo << 1) | (bacc >> 7);
            }
            code[a][0] = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
            code[a][1] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
            continue;
        }
#pragma unroll
        for (int i = 0; i < NMAX; i++) {
            if (i < N) {
                const unsigned x = g[a][i], y = iv[a][i];
                const unsigned t = (x | H) - (y & ~H);                           // bit 8k+7: (x & 0x7f) >= (y & 0x7f)
                const unsigned ge = __builtin_amdgcn_bitop3_b32(x, y, t, 0xB2);  // (x & ~y) | (~(x ^ y) & t): byte x >= byte y
                bacc = __builtin_amdgcn_bitop3_b32(bacc, ge, H, 0x78);           // bacc ^ (ge & H)
                if (i < N - 8) hi = (hi << 1) | (bacc >> 7);
                else lo = (lo << 1) | (bacc >> 7);
            }
        }
        code[a][0] = __builtin_amdgcn_perm(hi, lo, 0x05010400u);  // bytes (lo0, hi0, lo1, hi1); selectors 0-3 = lo, 4-7 = hi
        code[a][1] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);  // bytes (lo2, hi2, lo3, hi3)
    }
}

// ---- phase A: stages 3..5 of the lane's 4 pixels ---------------------------------------------------------------------------------
// The correspondences are parked in the LDS staging area (slots 3k, 3k+1 of pixel k, which its own result overwrites later).
// Between A and B the plane registers are dead -- that is where a table rig asks for its 4 projector-table entries at once (instead
// of one dependent gather inside every pixel's chain), and where the pipelined loop requests the next view's planes.
// `i` (0 or 1, compile-time after inlining) is the pixel's place in the CURRENT pair: the pair loop shifts the fringe dwords, the
// code words and the valid bits down after each pair, so every operand sits at a fixed byte / half-word (static sub-dword selects
// instead of shifts by a loop counter); k = 2*pair + i only addresses LDS and the frame.
template <bool RCP_TAB>
__device__ __forceinline__ unsigned pixel_A(const KParams &P, const Item &it, int F, int i, int k, unsigned vbits, const unsigned (&f)[2][4],
                                            const unsigned (&code)[2][2], const double *s_rcp, int *pair_cp)
{
    const int sh = 8 * i;
    const int code_v = (int)((code[0][0] >> (16 * i)) & 0xffffu);
    const int code_h = (int)((code[1][0] >> (16 * i)) & 0xffffu);
    const AtanK AK = atan_consts<true>();
    float wv = wrapped_phase<RCP_TAB>(F, (f[0][0] >> sh) & 255, (f[0][1] >> sh) & 255, (f[0][2] >> sh) & 255, (f[0][3] >> sh) & 255, s_rcp, AK);
    float wh = wrapped_phase<RCP_TAB>(F, (f[1][0] >> sh) & 255, (f[1][1] >> sh) & 255, (f[1][2] >> sh) & 255, (f[1][3] >> sh) & 255, s_rcp, AK);
    // stage 4 shifts by +Pi only inside its loop range (4/phase_unwrap.cpp:285,290,304,308); outside it the unwrapped value is 0
    // whatever the wrapped one is (correspond_px), and the timed mode does not keep wrapped
    wv = shift_pi(wv);
    wh = shift_pi(wh);
    int cx, cy;
    const bool ok = correspond_px(P, it.gx0 + k, it.gy, wv, wh, code_v, code_h, cx, cy) && ((vbits >> i) & 1u);
    pair_cp[3 * i] = ok ? cx : 0;  // (the pair's 6 staging words) a rejected pixel keeps a harmless table index
    pair_cp[3 * i + 1] = ok ? cy : 0;
    return ok ? 1u : 0u;
}

// the 4 pixels of a lane as two pairs: two independent fp64 dependency chains per iteration for the scheduler to interleave
// (tools/ab.sh: 1 pixel per iteration -3 %, all 4 unrolled +1 % but 12 more VGPRs).
// UNROLL (the small-launch instantiation): both pairs in one basic block = four independent chains.  In a launch of one or two
// views a SIMD is often NOT saturated by its four waves -- the first round's waves all wait for their planes and then all compute,
// the last waves of the launch compute alone -- and there a wave's own latency is what the launch waits for.
template <bool RCP_TAB>
__device__ __forceinline__ void phase_A_pair(const KParams &P, const Item &it, int F, int j, unsigned &vbits, unsigned (&f)[2][4], unsigned (&code)[2][2],
                                             const double *s_rcp, int *pair_cp, unsigned &vout)
{
    const unsigned ok0 = pixel_A<RCP_TAB>(P, it, F, 0, 2 * j, vbits, f, code, s_rcp, pair_cp), ok1 = pixel_A<RCP_TAB>(P, it, F, 1, 2 * j + 1, vbits, f, code, s_rcp, pair_cp);
    vout = (vout >> 16) | (ok0 << 16) | (ok1 << 24);  // after two pairs: valid byte of pixel k at byte k
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int p = 0; p < 4; p++) f[a][p] >>= 16;
        code[a][0] = code[a][1];
    }
    vbits >>= 2;
}

template <bool RCP_TAB, bool UNROLL>
__device__ __forceinline__ unsigned phase_A(const KParams &P, const Item &it, int F, unsigned vbits, unsigned (&f)[2][4], unsigned (&code)[2][2],
                                            const double *s_rcp, int *my_cp)
{
    unsigned vout = 0;
#pragma unroll(UNROLL ? 2 : 1)
    for (int j = 0; j < 2; j++) {
        // the pair's 6 staging words (24 bytes): in the rolled loop the offset is built from shifts of a VGPR copy of j, as in
        // phase_B -- `base + 24 * j` became a v_mad_u64_u32 whose addend pair is (LDS base, the NEXT register: the mask dword of the
        // view after next, still in flight) and put an s_waitcnt vmcnt(0) in the middle of stage 5: the wave waited for the
        // acknowledgement of the stores it had just issued (read in the ISA at the end of round 4)
        int j8 = j * 8;
        if (!UNROLL) asm volatile("" : "+v"(j8));
        phase_A_pair<RCP_TAB>(P, it, F, j, vbits, f, code, s_rcp, my_cp + (UNROLL ? 6 * j : ((j8 << 1) + j8) >> 2), vout);
    }
    return vout;
}

// SPLIT (the pipelined small-launch instantiations): stage 3 of the lane's 4 pixels runs BEFORE the Gray planes are waited for.
// The fringe planes are the first 6 of a view's 46 requests and vmcnt counts in issue order, so the 8 wrapped phases (8 floats) need
// s_waitcnt vmcnt(40) only -- the atan2 arithmetic, most of phase A, runs while the wave's Gray planes are still on their way.  In a
// launch of one view per lane every wave of a round asks for its planes at the same time and would then compute at the same time:
// this is arithmetic moved under the wave's OWN memory wait.
template <bool RCP_TAB>
__device__ __forceinline__ void wrapped_quad(int F, const unsigned (&f)[2][4], const double *s_rcp, float (&w)[2][4])
{
    const AtanK AK = atan_consts<true>();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sh = 8 * k;
#pragma unroll
        for (int a = 0; a < 2; a++)
            w[a][k] = shift_pi(wrapped_phase<RCP_TAB>(F, (f[a][0] >> sh) & 255, (f[a][1] >> sh) & 255, (f[a][2] >> sh) & 255, (f[a][3] >> sh) & 255, s_rcp, AK));
    }
}

// stages 4 + 5 of the 4 pixels from their wrapped phases and Gray codes (the rest of phase A behind wrapped_quad); -> valid byte of pixel k at byte k
__device__ __forceinline__ unsigned correspond_quad(const KParams &P, const Item &it, unsigned vbits, const float (&w)[2][4], const unsigned (&code)[2][2], int *my_cp)
{
    unsigned vout = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int code_v = (int)((code[0][k >> 1] >> (16 * (k & 1))) & 0xffffu);
        const int code_h = (int)((code[1][k >> 1] >> (16 * (k & 1))) & 0xffffu);
        int cx, cy;
        const bool ok = correspond_px(P, it.gx0 + k, it.gy, w[0][k], w[1][k], code_v, code_h, cx, cy) && ((vbits >> k) & 1u);
        my_cp[3 * k] = ok ? cx : 0;
        my_cp[3 * k + 1] = ok ? cy : 0;
        vout |= (ok ? 1u : 0u) << (8 * k);
    }
    return vout;
}

// ---- phase B: stage 7 + the cast of stage 8 ------------------------------------------------------------------------------------
// neighbouring camera pixels see neighbouring projector pixels: the 4 gathers stay within a few cache lines per wave
__device__ __forceinline__ void gather_B(const KParams &P, bool proj_table, const int *my_cp, float2 (&d)[4])
{
#pragma unroll
    for (int k = 0; k < 4; k++) d[k] = make_float2(0.f, 0.f);
    if (proj_table) {
        // (SGPR base + 32-bit byte offset: a projector has fewer than 2^29 pixels -- sl3d_create checks it)
        const GLOBAL_AS uint8_t *tab = opaque((const uint8_t *)P.proj_disp);
        const unsigned pw = (unsigned)P.PW;
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = ldg_f2(tab, ((unsigned)my_cp[3 * k + 1] * pw + (unsigned)my_cp[3 * k]) * 8u);
    }
}

template <int RIG, bool UNROLL>
__device__ __forceinline__ void phase_B(const KParams &P, const DevCal *Cglobal, const PinnedRows &PR, bool proj_table, unsigned vout, float2 (&d)[4],
                                        const double *my_cam, const int *my_cp, float *my_xyz, const RadEntry *s_rad)
{
    const float nanv = __builtin_nanf("");
    unsigned vb = vout;
    // The pair index reaches the staging addresses through a VGPR.  With the loop counter in an SGPR the rolled loop's address
    // `base + 24*j` became a v_mad_u64_u32 whose 64-bit addend pairs the LDS base with WHATEVER sits in the next VGPR -- the mask
    // dword just requested -- and that false read of a pending load put an s_waitcnt vmcnt(0) at the top of this loop: stage 7
    // waited for every plane of the next view (round 4, read in the ISA).
    // (so the offset of a pair inside the lane's 48-byte staging slot, 24*j, is built from shifts of a VGPR copy of j, with an
    // empty asm in between that keeps the optimiser from folding them back into a multiply)
#pragma unroll(UNROLL ? 2 : 1)
    for (int j = 0; j < 2; j++) {
        int j8 = j * 8;  // bytes
        if (!UNROLL) asm volatile("" : "+v"(j8));
        const int pair_words = UNROLL ? 6 * j : ((j8 << 1) + j8) >> 2;  // 6 floats per pair
        const int *cp = my_cp + pair_words;
        float *xyz = my_xyz + pair_words;
        const double *cam = my_cam + (UNROLL ? 4 * j : j8 >> 1);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float x, y, z;
            triangulate_from<RIG>(P, opaque_const(Cglobal), PR, cam[2 * i], cam[2 * i + 1], cp[3 * i], cp[3 * i + 1], d[i], proj_table, s_rad, x, y, z);
            const bool ok = ((vb >> (8 * i)) & 1u) != 0u;
            xyz[3 * i + 0] = ok ? x : nanv;
            xyz[3 * i + 1] = ok ? y : nanv;
            xyz[3 * i + 2] = ok ? z : nanv;
        }
        d[0] = d[2];
        d[1] = d[3];
        vb >>= 16;
    }
}

// ---- parity mode (KEEP): the 4 pixels of the lane, one after the other, every stage-boundary plane stored ----------------------------
template <bool RCP_TAB>
__device__ __forceinline__ unsigned parity_pixels(const KParams &P, const DevCal *Cglobal, const PinnedRows &PR, const Item &it, int F, unsigned vbits,
                                                  const unsigned (&f)[2][4], const unsigned (&code)[2][2], const double *s_rcp, const double *my_cam,
                                                  float *my_xyz, size_t px)
{
    unsigned vout = 0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        if ((vbits >> k) & 1u) {
            const int sh = 8 * k;
            const int code_v = (int)((code[0][k >> 1] >> (16 * (k & 1))) & 0xffffu);
            const int code_h = (int)((code[1][k >> 1] >> (16 * (k & 1))) & 0xffffu);
            // stage 3: wrapped phase of both axes; stage 4 shifts it by +Pi inside its loop range
            const AtanK AK = atan_consts<true>();
            float wv = wrapped_phase<RCP_TAB>(F, (f[0][0] >> sh) & 255, (f[0][1] >> sh) & 255, (f[0][2] >> sh) & 255, (f[0][3] >> sh) & 255, s_rcp, AK);
            float wh = wrapped_phase<RCP_TAB>(F, (f[1][0] >> sh) & 255, (f[1][1] >> sh) & 255, (f[1][2] >> sh) & 255, (f[1][3] >> sh) & 255, s_rcp, AK);
            wv = shift_pi_if(wv, it.gx0 + k >= 1 && it.gx0 + k <= P.fullW - 2);  // 4/phase_unwrap.cpp:285,290
            wh = shift_pi_if(wh, it.gy >= 1 && it.gy <= P.fullH - 2);            // 4/phase_unwrap.cpp:304,308
            const double cu = my_cam[2 * k], cv = my_cam[2 * k + 1];
            const PixelResult R = parity_chain(P, opaque_const(Cglobal), PR, it.gx0 + k, it.gy, cu, cv, wv, wh, code_v, code_h, px + k);
            if (R.valid) {
                my_xyz[3 * k + 0] = R.x;
                my_xyz[3 * k + 1] = R.y;
                my_xyz[3 * k + 2] = R.z;
                vout |= 1u << (8 * k);
            }
        }
    }
    return vout;
}

// the planes the reference initialises before its loops (code = -1: 4/phase_unwrap.cpp:143; everything else the zero a fresh
// allocation reads as)
__device__ __forceinline__ void parity_init(const KParams &P, size_t px, unsigned vbits)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        for (int a = 0; a < 2; a++) {
            P.wrapped[a][px + k] = 0.f;
            P.unwrapped[a][px + k] = 0.f;
            P.code[a][px + k] = -1;
            P.valid_axis[a][px + k] = (vbits >> k) & 1u;
        }
        P.cpmap[2 * (px + k)] = 0;
        P.cpmap[2 * (px + k) + 1] = 0;
        P.ipoints[3 * (px + k)] = P.ipoints[3 * (px + k) + 1] = P.ipoints[3 * (px + k) + 2] = 0.0;
    }
}

// ---- results out ---------------------------------------------------------------------------------------------------------------
// A wave's LDS instructions execute in order, but the COMPILER knows nothing of that: every hand-off of staged results from the
// lane that wrote them to another lane of the same wave (store_quad's read-back across lanes, store_segment's in-place
// compaction and its chunk read-back) is fenced at wavefront scope -- a release fence + a wave barrier: no instruction of their
// own, they pin the order of the ds_write / ds_read the optimiser may not disambiguate.
__device__ __forceinline__ void wave_lds_handoff()
{
#ifndef SL3D_NO_WAVE_FENCE /* (A/B only, tools/ab.sh: the fences must cost nothing) */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Dense results.  The 48 B of xyz a lane produces are staged in LDS (the rolled pixel loops index them).  A full wave's 64 x 48 B
// are 3 KB contiguous in LDS AND in the dense plane (quads are consecutive in the pitch-padded layout): the wave reads them back
// ACROSS lanes and every store instruction writes 1 KiB of whole lines, lane after lane -- and because they are whole lines they
// carry the non-temporal hint: nothing is left for the L2 to merge, the lines stream out instead of sitting dirty in the L2 until
// they are evicted or the kernel ends.  Round 3, alternating (profiles/r03_nt_coalesced_ab.txt): 16 views per launch +6...8 %, one
// view 29.2 -> 27.3 us, other rigs +7 %; each half alone LOSES (coalesced without the hint -2.4 %; the hint on 16-byte pieces
// -11 %: every piece becomes a memory write of its own).  A wave that is not whole (the last rows of a window) and the parity mode
// store three 16-B pieces per lane, each lane what it wrote itself.
template <bool KEEP>
__device__ __forceinline__ void store_quad(const KParams &P, const float *s_xyz, const float *my_xyz, size_t px, unsigned vout)
{
    float4 *out_xyz = (float4 *)(P.points + 3 * px);
    const float4 *sx = (const float4 *)my_xyz;
    if (!(SL3D_ABLATE & 4) || KEEP || sx[0].x == 12345.f) {
        if (!KEEP && __ballot(true) == ~0ull) {
            wave_lds_handoff();
            // wave-uniform parts as scalars (the wave's number, the first lane's pixel = the start of the wave's run): the store
            // addresses are an SGPR base + a 32-bit lane offset, nothing of them lives in VGPRs across the view loop
            unsigned lane_ = threadIdx.x & 63u;
            asm volatile("" : "+v"(lane_));  // (recomputed here: hoisted out of the view loop the lane offsets were kept as 64-bit pairs, and spilled)
            const unsigned wave_ = first_lane_u32(threadIdx.x >> 6);
            const size_t px0 = ((size_t)first_lane_u32((unsigned)(px >> 32)) << 32) | (size_t)first_lane_u32((unsigned)px);
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 *w4 = (const f32x4 *)(s_xyz + wave_ * (64u * 12u));
            GLOBAL_AS uint8_t *q = opaque_out(P.points + 3 * px0);
            const unsigned lo = lane_ * 16u;
            stg_nt(q, lo, w4[lane_]);
            stg_nt(q, lo + 1024u, w4[64u + lane_]);
            stg_nt(q, lo + 2048u, w4[128u + lane_]);
            stg_nt(opaque_out(P.valid + px0), lane_ * 4u, vout);
            wave_lds_handoff();  // ... and every lane has read them before the next view's phase A reuses the area
            return;
        }
        out_xyz[0] = sx[0];
        out_xyz[1] = sx[1];
        out_xyz[2] = sx[2];
    }
    *(unsigned *)(P.valid + px) = vout;
}

__device__ __forceinline__ void fill_nan(float *my_xyz)
{
    const float nanv = __builtin_nanf("");
#pragma unroll
    for (int i = 0; i < 12; i++) my_xyz[i] = nanv;
}

// Segmented clouds (CMODE 2; O1 / N2: 8/save_point_cloud.cpp:33-37 counts the valid pixels, :85-104 appends them in row-major scan
// order) -- no dependency between tiles at all.  A wave owns 256 consecutive pixels of the scan; it ranks ITS valid pixels (4
// ballots + mbcnt, no block barrier), compacts their points in place inside its own 3 KB of the LDS staging area (every lane first
// reads its 12 floats, then writes its valid points at their compacted position, which is never above its own) and stores
// ceil(3*count/4) whole 16-byte chunks, lane after lane (1 KiB per store instruction, non-temporal), into its own fixed slot of
// the cloud buffer -- points [256*seg, 256*seg + count) with seg = 4*tile + wave -- plus the count.  Scan order is preserved inside
// a segment and across segments, so the cloud of a view is the concatenation of its segments; k_seg_scan turns the counts into
// offsets, and the consumers that exist anyway close the gaps (k_seg_close into a contiguous device / mapped host buffer, the
// registration, the pack before an RCCL send).  Lanes past the last row take part with no valid pixel.
// (A chunk may run up to 3 floats past the last point: still inside the slot.)
__device__ __forceinline__ void store_segment(const KParams &P, const Item &it, int view, unsigned vout, float *s_xyz, const float *my_xyz)
{
    if (it.alive) stg_nt(opaque_out(P.valid + (size_t)view * P.px_view_stride), it.lane_off, vout);
    const unsigned long long b0 = __ballot((vout & 0x00000001u) != 0u), b1 = __ballot((vout & 0x00000100u) != 0u),
                             b2 = __ballot((vout & 0x00010000u) != 0u), b3 = __ballot((vout & 0x01000000u) != 0u);
    auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
    unsigned rank = below(b0) + below(b1) + below(b2) + below(b3);  // valid pixels of the lanes below this one
    const unsigned total = (unsigned)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
    unsigned lane_ = threadIdx.x & 63u;
    asm volatile("" : "+v"(lane_));  // (see store_quad: the wave's number, its segment and its slot are scalars, the lane index is recomputed)
    const unsigned wave_ = first_lane_u32(threadIdx.x >> 6);
    const unsigned seg = first_lane_u32(it.tile) * 4u + wave_;
    if (lane_ == 0u) P.seg_counts[(size_t)view * (size_t)P.n_segs + seg] = total;
    float *slot = P.clouds + 3 * ((size_t)view * P.px_view_stride + (size_t)seg * SL3D_SEG_POINTS);
    float *wbase = s_xyz + wave_ * (64u * 12u);
    // (a wave whose 256 pixels are all valid -- the usual case away from the selection's edge -- has nothing to compact: every point
    // already sits at its rank)
    if (total != 4u * 64u) {
        const float4 *sx = (const float4 *)my_xyz;
        const float4 a = sx[0], b = sx[1], c = sx[2];
        const float q[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        wave_lds_handoff();  // every lane has read its own 12 floats before any lane overwrites a slot
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((vout >> (8 * k)) & 1u) {
                wbase[3 * rank + 0] = q[3 * k + 0];
                wbase[3 * rank + 1] = q[3 * k + 1];
                wbase[3 * rank + 2] = q[3 * k + 2];
                rank++;
            }
    }
    wave_lds_handoff();  // the (compacted) points are in place before the chunks are read back across lanes
    const unsigned chunks = (3u * total + 3u) >> 2;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 *wb4 = (const f32x4 *)wbase;
    GLOBAL_AS uint8_t *out = opaque_out(slot);
#pragma unroll
    for (int c3 = 0; c3 < 3; c3++) {
        const unsigned i = (unsigned)c3 * 64u + lane_;
        if (i < chunks) stg_nt(out, lane_ * 16u + (unsigned)c3 * 1024u, wb4[i]);
    }
    wave_lds_handoff();  // ... and read before the next view's phase A parks its correspondences in the same area
}

// F == 5: check_I_mod_criteria's assignment is commented out (3/wrapped_phase.cpp:117-129): nothing is valid
template <bool KEEP, bool FGEN, bool SEG>
__device__ __forceinline__ unsigned valid_bits(const Item &it, int F, const MaskQuad &m)
{
    if (SEG && !it.alive) return 0u;
    return (FGEN && F == 5) ? 0u : (!KEEP && (SL3D_ABLATE & 2)) ? 0xfu : mask_quad_bits(m);
}

// the LDS copy of rig 3's radial table (no byte of LDS in the kernels of the other rig classes)
template <int RIG>
struct RadialLds {
    __device__ __forceinline__ static RadEntry *get() { return nullptr; }
};
template <>
struct RadialLds<3> {
    __device__ __forceinline__ static RadEntry *get()
    {
        __shared__ RadEntry tab[SL3D_RAD_NODES];
        return tab;
    }
};

// ---- the kernel ----------------------------------------------------------------------------------------------------------------
// KEEP  parity mode: the stage-boundary planes are written too (RIG 0, dense results)
// NMAX / EXACT  unroll bound of the Gray planes / both axes have exactly NMAX planes
// FGEN  false: 3-step fringes (the reference's configuration) with the F test folded at compile time
// RIG   stage 7, see above
// CMODE 0: dense xyz + valid planes; 2: segmented ordered clouds + the valid plane (sl3d_run_clouds); + 4 (MASKIN, the pipelined
//       small-launch instantiations, and the gated large-launch ones for views known to be sparsely selected): the launch evaluates
//       the views' raw selection itself -- see maskin_request
// RCPT  true: 1/d of the atan2 quotient from an LDS table (6 KB per block, 768 IEEE divisions + a block barrier to fill it);
//       false: the instantiation for SMALL launches (a handful of views: the reference's one scan per call) -- v_rcp_f64 + one
//       Newton step instead of the table, whose fill nothing amortises when a block lives for one or two views, and the first
//       view's planes requested before the mask is known (EARLY).  Both quotients are proven equal to the host's libm on the whole
//       lattice by the device self-check.  Round 3, alternating on one box: 1 view 30.3 against 31.6 us, 2 views -2.7 %, 4 views
//       -1 %; at 16 views per launch the table is as fast (dense) or 1.4 % faster (clouds) -- profiles/r03_rcp_table_ab*.txt.
// EARLY the first view's planes of an item are requested before its mask is known (pipelined kernels only).  Always in the
//       small-launch instantiation.  In the large-launch one: when the launch's views are not known to be sparsely selected
//       (launch_fused) -- a block's prologue is one memory round trip shorter: 16 views +1.4 %, 8 views +1.7 %, clouds +0.8 %
//       (profiles/r04_early_large_ab.txt); EARLY = false is the large-launch kernel for sparse selections, whose every request
//       waits for the valid bits.
template <bool KEEP, int NMAX, bool FGEN, bool EXACT, int RIG, int CMODE, bool RCPT, bool EARLY_>
__global__ __launch_bounds__(RCPT ? SL3D_BLOCK : SL3D_SMALL_BLOCK, SL3D_OCC) void k_fused(const KParams P, const DevCal *__restrict__ Cglobal, int first_view, int n_views, int vpt)
{
    constexpr bool SEG = (CMODE & 2) != 0;
    constexpr bool MASKIN = (CMODE & 4) != 0;
    constexpr int BLK = RCPT ? SL3D_BLOCK : SL3D_SMALL_BLOCK;
    static_assert((CMODE & ~6) == 0, "0 = dense planes, 2 = segmented clouds (1 was round 2's look-back compaction), + 4 = MASKIN");
    static_assert(!MASKIN || (!KEEP && RIG != 0 && (RCPT ? !EARLY_ : EARLY_)),
                  "MASKIN: the pipelined small-launch instantiations, and the gated large-launch ones (views known to be sparsely selected)");
    static_assert(!(KEEP && CMODE != 0), "the parity mode writes dense planes");
    static_assert(!(KEEP && RIG != 0), "the parity mode evaluates everything with the reference's operation order");
    __shared__ __attribute__((aligned(16))) float s_xyz[BLK * 12];  // staging area: correspondences, then xyz, of the lane's 4 pixels
    __shared__ __attribute__((aligned(16))) double s_cam[BLK * 8];  // undistorted camera coordinates of the lane's 4 pixels
    __shared__ __attribute__((aligned(16))) double s_rcp[RCPT ? SL3D_RCP_TAB : 1];  // 1/d for the atan2 quotient
    RadEntry *const s_rad = RadialLds<RIG>::get();                                  // rig 3: the projector's radial table
    static_assert(RIG != 3 || SL3D_BLOCK == SL3D_RAD_NODES, "one table node per thread");
    SL3D_STAMP(0);
    constexpr bool PIPE = !KEEP && RIG != 0;  // (see below)
    static_assert(RCPT || EARLY_ == PIPE, "the small-launch instantiation: early requests iff pipelined");
    constexpr bool EARLY = EARLY_ && PIPE;
    // rig 3: one of the 8 copies of the table (one per XCD, as consecutive blocks go round the XCDs: all blocks of a launch would
    // otherwise start on the same 32 cache lines of one L2 -- what cost the camera-side radial table 2 us per one-view launch).
    // REQ_FIRST (every pipelined kernel): the block's LDS tables -- this one, the reciprocal table -- are filled UNDER the item's first
    // memory requests (mask dword, camera-table entries and, with EARLY, the first view's planes): the node is only requested here.
    // The un-pipelined kernels fill them here and now.
    constexpr bool REQ_FIRST = PIPE;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (a node travels as one 16-byte register quad)
    u32x4 rad_node = {0u, 0u, 0u, 0u};
    if (RIG == 3) rad_node = ((const u32x4 *)P.proj_rad)[(blockIdx.x & 7u) * SL3D_RAD_STRIDE + threadIdx.x];
    if (!REQ_FIRST) {
        if (RIG == 3) ((u32x4 *)s_rad)[threadIdx.x] = rad_node;
        if (RCPT) fill_rcp_table(s_rcp);
        if (RCPT || RIG == 3) __syncthreads();
    }
    SL3D_STAMP(1);
    const int F = FGEN ? P.F : 3;
    float *my_xyz = s_xyz + threadIdx.x * 12;
    double *my_cam = s_cam + threadIdx.x * 8;
    int *my_cp = (int *)my_xyz;
    // PIPE (timed kernels): the planes of view v+1 are requested in the middle of view v -- after phase A, when the plane
    // registers of view v are dead, before stage 7 -- so a wave's own arithmetic runs under its own memory requests
    // (the general rig's stage 7 is too register-hungry for it: 44 bytes of scratch per lane, -9 %)
    // EARLY: the first view's planes of an item are requested UNCONDITIONALLY, right behind the item's mask / camera-table requests
    // and before any of those is waited for, at the price of plane loads for quads that turn out to be masked off
    // (profiles/r03_early_planes_ab.txt, r04_early_large_ab.txt)
    constexpr bool UNROLL = !RCPT;  // the small-launch instantiation: both pixel pairs of phases A and B in one basic block (see phase_A)
#ifdef SL3D_NO_SPLIT
    constexpr bool SPLIT = false;
#else
    constexpr bool SPLIT = EARLY;   // ... and stage 3 ahead of the wait for the Gray planes (wrapped_quad)
#endif

    // EXACT: both axes have exactly NMAX Gray planes (the usual case)
    // how the axes map onto the NMAX unrolled planes (issue_gray): exact; padded (the timed 3-step kernels up to 12 planes); tests
    constexpr int PLANES = EXACT ? 1 : (!KEEP && NMAX <= 12) ? ((MASKIN && RCPT) ? 3 : 2) : 0;
    const int Nv = EXACT ? NMAX : P.Nv, Nh = EXACT ? NMAX : P.Nh;
    const bool proj_table = RIG == 2 || (RIG == 0 && !KEEP && P.proj_disp != nullptr);

    Item it;
    MaskQuad mq;
    double camt[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // EARLY: the lane's camera-table entries between their request and cam_table_finish
    unsigned mbits = 0u;                         // MASKIN: valid bits (0..3) and own selection bits (4..7) of the quad
    mrow_t mw0 = {0u, 0u}, mw1 = mw0, mw2 = mw0, mw3 = mw0, mw4 = mw0;  // MASKIN: the selection bytes around the quad (maskin_request)
    bool mfast = false;                          // MASKIN: the wave evaluates the short form (maskin_request)  // MASKIN: the selection bytes around the quad between their request and maskin_finish
    unsigned f[2][4], g[2][NMAX], iv[2][NMAX], code[2][2];
    // gridDim.x is a multiple of 8 (launch_fused): consecutive tiles go round-robin over the 8 XCDs on purpose (the XCD-banded
    // order was measured at -4 %: DRAM locality across XCDs beats L2 locality for 2 % of shared bytes)
    // (false: a lane past the last row, or a block the grid was padded with.  REQ_FIRST: block-uniform, only the latter -- a lane past
    // the last row stays until the block's barrier, its requests go to the last row.)
    if (!item_begin<RIG, SEG, REQ_FIRST, BLK, MASKIN>(P, Cglobal, blockIdx.x, (int)blockIdx.y, first_view, n_views, vpt, it, mq, camt, my_cam, mw0, mw1, mw2, mw3, mw4, mfast)) return;
    if (MASKIN) it.v_end = it.v_begin + 1;  // (one view per item: launch_fused makes the grid so; a compile-time trip count of the loop below)
    if (REQ_FIRST) {
        // EARLY: planes of the first view right behind the set-up requests.  The block's LDS tables are filled while all of that
        // travels (the radial node requested at the very top is the oldest request: its store waits for nothing else); then the
        // set-up results are consumed
        if (EARLY) {
            issue_fringe<FGEN>(P, it.v_begin, it.lane_off, F, Nv, f);
            issue_gray<NMAX, PLANES>(P, it.v_begin, it.lane_off, F, Nv, Nh, g, iv);
        }
        if (RCPT) fill_rcp_table(s_rcp);
        if (RIG == 3) ((u32x4 *)s_rad)[threadIdx.x] = rad_node;
        if (RCPT || RIG == 3) __syncthreads();
        if (!SEG && !it.alive) return;
        // MASKIN: the selection bytes were requested first, so they are here first: evaluated while the planes travel
        if (MASKIN) {
            if (it.alive) mbits = maskin_finish(P, it, it.v_begin, it.v_begin - first_view, mw0, mw1, mw2, mw3, mw4, mfast);
            maskin_count(P, it, it.v_begin, mbits);
        }
        if (P.use_cam_table) cam_table_finish<RIG, !MASKIN>(P, Cglobal, it, camt, my_cam);
    }
    SL3D_STAMP(2);
    // The mask dword of the NEXT view is requested a view ahead, so a wave never waits a full memory round trip for it before it
    // can ask for its 11.5 KB of planes.  (gfx950 has one in-order vmcnt for loads and stores, and the wait-count pass is
    // conservative wherever a register a load is still writing is touched: round 4 removed, one by one, every s_waitcnt vmcnt(0)
    // of this loop except the decode's -- see the comments at the pipeline point, in phase_A / phase_B and at vb_pre.)
    unsigned vb_next = 0;
    if (PIPE) {
        vb_next = MASKIN ? (mbits & 0xfu) : valid_bits<KEEP, FGEN, SEG>(it, F, mq);
        if (!MASKIN && it.v_begin + 1 < it.v_end) mq = load_mask_quad(P, it.v_begin + 1, it.lane_off);
        if (!EARLY && vb_next != 0) {  // (EARLY: they are in flight already)
            issue_fringe<FGEN>(P, it.v_begin, it.lane_off, F, Nv, f);
            issue_gray<NMAX, PLANES>(P, it.v_begin, it.lane_off, F, Nv, Nh, g, iv);
        }
    }
    const PinnedRows PR = pinned_rows<RIG>(Cglobal);  // (stage 7 needs them; by now the item's first memory requests are on their way)
    // DEFER: the results of view v leave AFTER the decode of view v + 1 instead of right behind stage 7.  gfx950 has ONE in-order
    // vmcnt for loads and stores: stores issued behind the next view's plane loads are what that view's decode ends up waiting for
    // (its last s_waitcnt vmcnt(0) = planes landed AND those stores acknowledged).  Deferred, they are issued once the planes have
    // been consumed and have a whole view's arithmetic to complete: 4 views per launch +0.7 %, 12 Mpx x 3 views +1 %, 16 views
    // and the table rig +0.3 % (profiles/r04_pipeline_point_ab.txt); the staged results wait in LDS, which phase A of the next view
    // only touches after the store.
    constexpr bool DEFER = PIPE;
    unsigned pvout = 0;
    auto store_view = [&](int v, unsigned vo) {
        const size_t p = (size_t)v * P.px_view_stride + (size_t)it.lane_off;
        if (MASKIN && it.alive) maskin_store(P, it, v, mbits);
        if (SEG) store_segment(P, it, v, vo, s_xyz, my_xyz);
        else store_quad<KEEP>(P, s_xyz, my_xyz, p, vo);
    };
    for (int view = it.v_begin; view < it.v_end; view++) {
        unsigned vbits;
        if (PIPE) {
            vbits = vb_next;
        } else {
            vbits = valid_bits<KEEP, FGEN, SEG>(it, F, mq);
            if (view + 1 < it.v_end) mq = load_mask_quad(P, view + 1, it.lane_off);
        }
        const size_t px = (size_t)view * P.px_view_stride + (size_t)it.lane_off;  // first pixel of the quad
        unsigned vout = 0;
        if (KEEP) parity_init(P, px, vbits);
        if (!DEFER && !SEG && (KEEP || vbits == 0)) fill_nan(my_xyz);
        if (!PIPE && vbits != 0) {  // every load of the view is issued before the first one is consumed
            issue_fringe<FGEN>(P, view, it.lane_off, F, Nv, f);
            issue_gray<NMAX, PLANES>(P, view, it.lane_off, F, Nv, Nh, g, iv);
        }
        if (view == it.v_begin) SL3D_STAMP(3);
        float w[2][4];
        if (SPLIT && vbits != 0) {
            wrapped_quad<RCPT>(F, f, s_rcp, w);          // waits for the 6 fringe planes
            __builtin_amdgcn_sched_barrier(0);           // (the decode below is not to be scheduled up in front of this arithmetic)
        }
        if (vbits != 0) decode_gray<NMAX, PLANES>(g, iv, Nv, Nh, code);  // waits for the planes of this view
        if (view == it.v_begin) SL3D_STAMP(4);
        // the NEXT view's valid bits, taken here -- its mask dword is older than the planes just decoded, so it has landed, and no
        // store of this view has been issued yet.  Taken at the pipeline point (where they are needed) they cost an s_waitcnt
        // vmcnt(0) there: one in-order counter, and by then the deferred stores are in it
        unsigned vb_pre = 0;
        if (PIPE) {
            vb_pre = MASKIN ? 0u : valid_bits<KEEP, FGEN, SEG>(it, F, mq);
            asm volatile("" : "+v"(vb_pre));  // (here, not sunk to its use)
        }
        if (DEFER) {
            if (view > it.v_begin) store_view(view - 1, pvout);
            if (!SEG && vbits == 0) fill_nan(my_xyz);
        }
        if (vbits != 0) {
            if (KEEP) vout = parity_pixels<RCPT>(P, Cglobal, PR, it, F, vbits, f, code, s_rcp, my_cam, my_xyz, px);
            else if (SPLIT) vout = correspond_quad(P, it, vbits, w, code, my_cp);
            else vout = phase_A<RCPT, UNROLL>(P, it, F, vbits, f, code, s_rcp, my_cp);
        }
        if (view == it.v_begin) SL3D_STAMP(5);
        // (Round 3 read the ISA of the table rigs: their 4 projector-table entries are requested BEHIND the next view's 46 plane
        // loads, so -- vmcnt counts in issue order -- phase B starts only once those planes have landed.  Requesting them first was
        // built and measured: distorted rig 77.8-78.2 Gpx/s against 79.0-79.5 for this order, profiles/r03_gather_first_ab.txt.)
        if (PIPE && view + 1 < it.v_end) {
            vb_next = vb_pre;
            // UNCONDITIONAL (the index is clamped; the last view's dword is asked for once more): with `if (view + 2 < v_end)` the
            // new value meets the old one in a phi, whose copy the compiler places behind the plane loads below -- and a copy of a
            // loaded value is a use: s_waitcnt vmcnt(0), i.e. stage 7 of this view waited for ALL of the next view's planes to land
            // (round 3 read this wait in the ISA and measured a schedule without it at +-0.3 %; with today's kernel: 4 views per
            // launch +5 %, 16 views +1 %, the table rig +1.5 %, profiles/r04_pipeline_point_ab.txt)
            if (!MASKIN) mq = load_mask_quad(P, min(view + 2, it.v_end - 1), it.lane_off);
            if (vb_next != 0) {
                issue_fringe<FGEN>(P, view + 1, it.lane_off, F, Nv, f);
                issue_gray<NMAX, PLANES>(P, view + 1, it.lane_off, F, Nv, Nh, g, iv);
            }
        }
        if (!KEEP && vbits != 0) {
            float2 d[4];
            gather_B(P, proj_table, my_cp, d);
            phase_B<RIG, UNROLL>(P, Cglobal, PR, proj_table, vout, d, my_cam, my_cp, my_xyz, s_rad);
        }
        if (DEFER) {
            if (view == it.v_begin) SL3D_STAMP(6);
            pvout = vout;
            continue;
        }
        if (SEG) {
            store_segment(P, it, view, vout, s_xyz, my_xyz);
            continue;
        }
        if (view == it.v_begin) SL3D_STAMP(6);
        store_quad<KEEP>(P, s_xyz, my_xyz, px, vout);
        if (view == it.v_begin) SL3D_STAMP(7);
    }
    if (DEFER) {
        store_view(it.v_end - 1, pvout);
        SL3D_STAMP(7);  // (trace builds: with deferred stores, the LAST view's)
    }
}

// ---- launch plumbing -------------------------------------------------------------------------------------------------------------
// Instantiations: the timed 3-step kernel exists for every N = 6..12 with both axes equal (EXACT), as padded straight-line code for
// every NMAX = 6..12 (any other pair of axes up to NMAX planes -- issue_gray; the 4-/5-step fringes have this form only) and with
// the unroll bound 16 beyond; the parity mode uses the bounds 8 / 12 / 16 with per-plane tests.  Dense 3-step launches
// of at most SL3D_SMALL_LAUNCH_VIEWS views take the instantiation without the LDS reciprocal table (re-measured with the streaming
// stores: 8 views 185.6-187.5 us through it against 183.8-184.7, 16 views +-0: stays at 4).
struct FusedChoice {
    int nmax;
    bool exact, small;  // exact: both axes have exactly nmax planes.  (!exact, timed kernels, nmax <= 12: the padded form, issue_gray)
    bool early;         // the last template argument (EARLY): see k_fused
};
// prefer_gated: the views of the launch are sparsely selected -- a small launch then takes the large-launch instantiation, whose
// plane requests wait for the valid bits instead of going out first (one view of 1080p with 19 % of the frame selected, as in the
// reference's real captures: 15.8 us against 22.2; a full frame: 26.9 against 24.6 -- profiles/r04_sparse_mask.txt)
// rig: the rig class the launch runs (0 = the un-pipelined general kernel: no early requests there)
inline FusedChoice choose_fused(bool keep, bool fgen, int cmode, int nv, int nh, int n_views, bool prefer_gated, int rig)
{
    FusedChoice c;
    const int m = nv > nh ? nv : nh;
    c.exact = !keep && !fgen && nv == nh && nv >= 6 && nv <= 12;
    if (c.exact) c.nmax = nv;
    else if (!keep && m <= 12 && nv > 0 && nh > 0) c.nmax = m < 6 ? 6 : m;  // padded (4-/5-step fringes: always)
    else if (!keep) c.nmax = SL3D_MAX_GRAY;  // more than 12 planes, or an axis with NONE (sl3d_config allows 0): the per-plane tests
    else c.nmax = m <= 8 ? 8 : (m <= 12 ? 12 : SL3D_MAX_GRAY);
    c.small = !keep && !fgen && n_views <= SL3D_SMALL_LAUNCH_VIEWS && !prefer_gated;
#ifdef SL3D_MEASURE
    if (getenv("SL3D_NO_SMALL")) c.small = false;
#endif
    const bool pipelined = !keep && rig != 0 && m <= 12 && nv > 0 && nh > 0;
    c.early = c.small ? pipelined : (pipelined && !fgen && !prefer_gated);
    return c;
}

template <bool KEEP, bool FGEN, int RIG, int CMODE>
static void launch_fused_n(int nv, int nh, dim3 grid, hipStream_t st, const KParams &P, const DevCal *C, int first_view, int n_views, int vpt)
{
    const dim3 block(SL3D_BLOCK, 1, 1);
    const FusedChoice c = choose_fused(KEEP, FGEN, CMODE, nv, nh, n_views, P.prefer_gated != 0, RIG);
    constexpr bool HAS_SMALL = !KEEP && !FGEN;  // (the 3-step timed families have the second instantiation, dense and clouds)
    const long quads_ = (long)(P.pitch >> 2) * P.H;
    const dim3 small_grid((((unsigned)((quads_ + SL3D_SMALL_BLOCK - 1) / SL3D_SMALL_BLOCK)) + 7u) & ~7u, grid.y, 1);
#define SL3D_LAUNCH(NM, EX)                                                                                                                \
    do {                                                                                                                                   \
        if constexpr (HAS_SMALL) {                                                                                                         \
            if (c.small) {                                                                                                                 \
                hipLaunchKernelGGL((k_fused<KEEP, NM, FGEN, EX, RIG, CMODE, false, RIG != 0>), small_grid, dim3(SL3D_SMALL_BLOCK), 0, st, P, C, first_view, n_views, vpt); \
                break;                                                                                                                     \
            }                                                                                                                              \
        }                                                                                                                                  \
        if constexpr (HAS_SMALL && RIG != 0 && (NM) <= 12) {                                                                               \
            if (c.early) {                                                                                                                 \
                hipLaunchKernelGGL((k_fused<KEEP, NM, FGEN, EX, RIG, CMODE, true, true>), grid, block, 0, st, P, C, first_view, n_views, vpt); \
                break;                                                                                                                     \
            }                                                                                                                              \
        }                                                                                                                                  \
        hipLaunchKernelGGL((k_fused<KEEP, NM, FGEN, EX, RIG, CMODE, true, false>), grid, block, 0, st, P, C, first_view, n_views, vpt);     \
    } while (0)
    if constexpr (!KEEP) {
        if constexpr (!FGEN) {
            if (c.exact) {
                switch (c.nmax) {
                case 6: SL3D_LAUNCH(6, true); break;
                case 7: SL3D_LAUNCH(7, true); break;
                case 8: SL3D_LAUNCH(8, true); break;
                case 9: SL3D_LAUNCH(9, true); break;
                case 10: SL3D_LAUNCH(10, true); break;
                case 11: SL3D_LAUNCH(11, true); break;
                default: SL3D_LAUNCH(12, true); break;
                }
                return;
            }
        }
        switch (c.nmax) {  // padded (unequal axes, or fewer than 6 planes); more than 12 planes: the per-plane tests
        case 6: SL3D_LAUNCH(6, false); break;
        case 7: SL3D_LAUNCH(7, false); break;
        case 8: SL3D_LAUNCH(8, false); break;
        case 9: SL3D_LAUNCH(9, false); break;
        case 10: SL3D_LAUNCH(10, false); break;
        case 11: SL3D_LAUNCH(11, false); break;
        case 12: SL3D_LAUNCH(12, false); break;
        default:
            // 13..16 planes: per-plane tests, un-pipelined general kernel only (launch_fused routes every rig class there: the
            // pipelined kernels with per-plane tests spill 3.5 KB per lane -- 21 ms per 16-view launch, tools/corners.py)
            if constexpr (RIG == 0) SL3D_LAUNCH(SL3D_MAX_GRAY, false);
            break;
        }
    } else {
        if (c.nmax == 8) SL3D_LAUNCH(8, false);
        else if (c.nmax == 12) SL3D_LAUNCH(12, false);
        else SL3D_LAUNCH(SL3D_MAX_GRAY, false);
    }
#undef SL3D_LAUNCH
}

// MASKIN launches (CMODE | 4): the pipelined small-launch instantiation of every N, exact and padded, nothing else
// GATED: the views are known (by their last counts) to be sparsely selected -- the large-launch form whose plane requests wait for the
// valid bits (k_fused<..., true, false>); the selection is then evaluated under the block's reciprocal-table fill, in front of those requests
template <int RIG, int CMODE, bool GATED>
static void launch_fused_maskin_n(int nv, int nh, dim3 grid, hipStream_t st, const KParams &P, const DevCal *C, int first_view, int n_views, int vpt)
{
    static_assert(RIG != 0 && (CMODE == 4 || CMODE == 6), "MASKIN: rig classes 1..3, dense or segmented clouds");
    const FusedChoice c = choose_fused(false, false, CMODE & 2, nv, nh, n_views, GATED, RIG);
    const long quads_ = (long)(P.pitch >> 2) * P.H;
    const dim3 small_grid((((unsigned)((quads_ + SL3D_SMALL_BLOCK - 1) / SL3D_SMALL_BLOCK)) + 7u) & ~7u, grid.y, 1);
#define SL3D_LAUNCH_MI(NM, EX) hipLaunchKernelGGL((k_fused<false, NM, false, EX, RIG, CMODE, GATED, !GATED>), GATED ? grid : small_grid, dim3(GATED ? SL3D_BLOCK : SL3D_SMALL_BLOCK), 0, st, P, C, first_view, n_views, vpt)
    if (c.exact) {
        switch (c.nmax) {
        case 6: SL3D_LAUNCH_MI(6, true); break;
        case 7: SL3D_LAUNCH_MI(7, true); break;
        case 8: SL3D_LAUNCH_MI(8, true); break;
        case 9: SL3D_LAUNCH_MI(9, true); break;
        case 10: SL3D_LAUNCH_MI(10, true); break;
        case 11: SL3D_LAUNCH_MI(11, true); break;
        default: SL3D_LAUNCH_MI(12, true); break;
        }
        return;
    }
    switch (c.nmax) {
    case 6: SL3D_LAUNCH_MI(6, false); break;
    case 7: SL3D_LAUNCH_MI(7, false); break;
    case 8: SL3D_LAUNCH_MI(8, false); break;
    case 9: SL3D_LAUNCH_MI(9, false); break;
    case 10: SL3D_LAUNCH_MI(10, false); break;
    case 11: SL3D_LAUNCH_MI(11, false); break;
    default: SL3D_LAUNCH_MI(12, false); break;
    }
#undef SL3D_LAUNCH_MI
}

// one family of instantiations per translation unit (sl3d_fused_*.hip; they compile in parallel): 3-step timed kernels per rig,
// dense and segmented; the 4-/5-step timed kernels; the parity mode
#define SL3D_FUSED_FAMILY_ARGS int nv, int nh, dim3 grid, hipStream_t st, const KParams &P, const DevCal *C, int first_view, int n_views, int vpt
void fused_dense_rig0(SL3D_FUSED_FAMILY_ARGS);
void fused_dense_rig1(SL3D_FUSED_FAMILY_ARGS);
void fused_dense_rig2(SL3D_FUSED_FAMILY_ARGS);
void fused_dense_rig3(SL3D_FUSED_FAMILY_ARGS);
void fused_clouds_rig0(SL3D_FUSED_FAMILY_ARGS);
void fused_clouds_rig1(SL3D_FUSED_FAMILY_ARGS);
void fused_clouds_rig2(SL3D_FUSED_FAMILY_ARGS);
void fused_clouds_rig3(SL3D_FUSED_FAMILY_ARGS);
void fused_maskin_rig1(int cmode, bool gated, SL3D_FUSED_FAMILY_ARGS);  // MASKIN launches (cmode 4 / 6), one translation unit per rig class
void fused_maskin_rig2(int cmode, bool gated, SL3D_FUSED_FAMILY_ARGS);
void fused_maskin_rig3(int cmode, bool gated, SL3D_FUSED_FAMILY_ARGS);
void fused_fgen(int rig, int cmode, SL3D_FUSED_FAMILY_ARGS);  // 4-step (and the all-invalid 5-step) fringes: the F test stays a run-time branch
void fused_parity(bool fgen, SL3D_FUSED_FAMILY_ARGS);

}  // namespace sl3d
