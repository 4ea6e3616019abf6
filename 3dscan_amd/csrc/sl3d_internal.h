// sl3d_internal.h -- structures shared by the C-ABI host code (sl3d_capi_*.cpp) and the HIP
// kernels (sl3d_kernels.hip).  Not part of the public ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_vector_types.h>  // float2

#define SL3D_MASK_HALO 2        // rows / columns of selection mask kept around the window
#define SL3D_MASK_LPAD 16       // bytes in front of window column 0 in every mask row
#define SL3D_ATAN_T1 511        // t1 = I0 - I2        in [-255, 255]
#define SL3D_ATAN_T2 1021       // t2 = 2*I1 - I0 - I2 in [-510, 510]
#define SL3D_MAX_GRAY 16
#define SL3D_SEG_POINTS 256     // pixels (point slots) per segment of the segmented clouds = one wave of the fused kernel
#define SL3D_SMALL_LAUNCH_VIEWS 4  // launches of at most this many views take the small-launch instantiation (sl3d_fused.h)

namespace sl3d {

// Intrinsics of one device (camera or projector) as stage 7 uses them (T1).
struct Intr {
    double K[9];
    double ifx, ify, cx, cy;      // cvUndistortPoints normalises with the reciprocal focal lengths
    double k1, k2, p1, p2, k3;
    int has_dist;                 // any distortion coefficient non-zero (else the 5 iterations are an exact no-op)
    int affine;                   // last row of K is (0,0,1): the homogeneous divide is an exact no-op
    int has_tan;                  // p1 or p2 non-zero
    int plain;                    // affine and K[1] == K[3] == 0 (no skew)
    int identity;                 // plain and no distortion: undistort + re-project returns the pixel itself
};

// One node of a radial undistortion table: s(r0^2) = (factor of the last of the 5 fixed-point iterations of cvUndistortPoints) - 1 for a
// purely radial model, as a quadratic around node i of a uniform grid over r0^2 (w = distance to the node in node spacings,
// |w| <= 1/2):  s = c0 + (c1 + c2*w)*w.  The quadratic interpolates s at the node and at both cell boundaries, so neighbouring
// cells agree where they meet; with SL3D_RAD_NODES nodes its remainder is ~1e-10 of the normalised coordinate.
struct __attribute__((aligned(16))) RadEntry {
    double c0;
    float c1, c2;
};
#define SL3D_RAD_NODES 256
#define SL3D_RAD_COPIES 8                      // copies of a table the kernels read (one per XCD)
#define SL3D_RAD_STRIDE (SL3D_RAD_NODES + 16)  // entries between two copies: 4 KB + 256 B, so that the copies start on different channels

// Per-scan constants of stage 7 (T0): A = K*[R|t] for camera and projector.
struct DevCal {
    double Ac[12], Ap[12];
    Intr cam, proj;
    // Camera-frame form of the same least-squares problem (fast path of the fused kernel; valid when the camera matrix
    // is upper triangular and affine, K = [fx s cx; 0 fy cy; 0 0 1]).  With Y = Rc*X + tc the two camera rows of P become
    // fx*(1,0,-xn) + s*(0,1,-yn) and fy*(0,1,-yn) with a zero right-hand side ((xn,yn) = undistorted normalised coordinates --
    // cvUndistortPoints normalises with fx, fy, cx, cy only, the skew enters when K re-projects), so their part of
    // P^T P is a handful of flops instead of 26; the rigid change of variables leaves the minimiser unchanged.
    double Apc[12];       // Ap * [Rc tc; 0 1]^-1 : projector projection matrix acting on camera-frame points
    double Rct[9], tcn[3];  // X = Rct*Y + tcn  (Rct = Rc^T, tcn = -Rc^T tc)
    double fx2, fy2;      // Kc[0]^2, Kc[1]^2 + Kc[4]^2
    double fxs;           // Kc[0]*Kc[1]: the skew term (0 for the usual K); the camera-frame form holds for any upper-triangular affine K
};

// Scene + camera model of the synthetic-capture generator (k_synth).
struct SynthParams {
    double Rc[9], tc[3], Rp[9], tp[3];  // world -> camera / projector
    double Kp[9];
    double z0, a, b;                    // plane Z = z0 + a*X + b*Y
    float gain, offset;
    int noise;                          // uniform integer noise in [-noise, noise]
    unsigned long long seed;
    int view_id;                        // enters the noise hash
};

// Where k_mask_prepare reads a selection mask from: the staging plane behind sl3d_set_mask's copy, or the caller's device memory.
struct MaskSrc {
    uintptr_t origin;      // address of plane row 0, byte 0 (= window row -2, window column -16); only bytes inside the region below are read
    size_t stride;         // bytes between rows
    size_t view_stride;    // bytes between the masks of consecutive views of one call (0: every view gets the same mask)
    int bx0, bx1, r0, r1;  // plane bytes [bx0, bx1) x plane rows [r0, r1) hold source pixels: window + 2-pixel halo, clipped to the frame
};

// A MASKIN launch (k_fused with CMODE bit 4: the valid bits come from the raw selection -- H0 / S3b / S3d inside the fused kernel):
// where the selection of view first_view + k lies (MaskSrc::origin of that view), the region of the plane that holds source bytes,
// what may be read at all, and where every wave leaves {seq << 8 | quads with a valid pixel}.
struct MaskIn {
    uintptr_t origin[SL3D_SMALL_LAUNCH_VIEWS];
    size_t stride;
    int bx0, bx1, r0, r1;  // as in MaskSrc
    int lo, hi;            // plane bytes [lo, hi) of a row may be READ (the staging plane: the whole row; a caller's mask: the frame's columns)
    unsigned *part;        // host memory mapped into the device: [view][part_stride] words, one per wave that owns pixels
    unsigned part_stride;
    unsigned seq;
};

// Everything a kernel needs to address one context's buffers.
struct KParams {
    int W, H;                  // window
    int fullW, fullH;          // camera frame
    int col0, row0;            // window origin in the frame
    int PW, PH;
    int F, Nv, Nh;
    int fwv, fwh;
    int ncodes_v, ncodes_h;
    int ablate;                // measurement builds only (-DSL3D_MEASURE, env SL3D_ABLATE): bit2 skips the camera undistortion; 0 otherwise
    int pitch;                 // bytes per row of every u8 plane (multiple of 16)
    int planes_per_view;
    size_t plane_stride;       // pitch * H
    size_t view_stride;        // planes_per_view * plane_stride
    int mpitch;                // mask row pitch (pitch + 32)
    size_t mask_view_stride;   // mpitch * (H + 2*SL3D_MASK_HALO)
    const uint8_t *frames;
    const uint8_t *mask;       // 0/1 bytes, halo included
    const uint8_t *band;       // [view][row][pitch]: final valid bytes of the quads within 3 px of the frame border (set_mask)
    const float2 *proj_disp;   // [PH][PW] undistorted-minus-raw projector point (set_calibration; NULL unless the projector is distorted)
    const RadEntry *proj_rad;  // rig 3: SL3D_RAD_NODES nodes of a purely radial projector model over r0^2 in [0, (NODES - 1) / proj_rad_scale]
    float proj_rad_scale;      // nodes per unit of r0^2
    const double *cam_tab;     // T1 of the camera per window pixel (set_calibration, timed mode): kind 1 = [H][pitch] factor of the last
                               // undistortion iteration (radial model), kind 2 = [H][pitch][2] normalised point (tangential terms)
    int cam_tab_kind;          // 0 = no table (no distortion: nothing to iterate)
    int use_cam_table;         // set per launch: 0, or cam_tab_kind
    // dense results
    float *points;             // [view][row][pitch][3] f32
    uint8_t *valid;            // [view][row][pitch]    merged valid map
    size_t px_view_stride;     // pitch * H   (elements per view of every per-pixel plane)
    // ordered clouds written by the fused kernel itself (sl3d_run_clouds; NULL until first used)
    float *clouds;             // [view][px_view_stride][3]: per view 4*n_tiles segments of SL3D_SEG_POINTS point slots
    unsigned long long *cloud_totals;  // [view] number of valid points -- HOST memory mapped into the device: k_seg_scan stores the
                                       // count where sl3d_get_cloud_counts reads it after the stream has drained (no copy)
    int n_tiles;               // 1024-pixel tiles per view = blocks of the fused kernel along x that own pixels
    int prefer_gated;          // set per launch (host only): small launch over sparsely selected views -> the large-launch kernel
    // a segment's first seg_counts[view][seg] slots are its valid points in scan order
    unsigned *seg_counts;              // [view][n_segs]
    unsigned long long *seg_offsets;   // [view][n_segs] exclusive scan of the counts (k_seg_scan)
    int n_segs;                        // 4 * n_tiles
    unsigned long long *dbg;   // measurement builds (-DSL3D_TRACE): [block][wave][8] clock stamps of the dense kernel's phases; NULL otherwise
    // stage-boundary planes (NULL unless SL3D_FLAG_KEEP_STAGES)
    float *wrapped[2];
    float *unwrapped[2];
    int32_t *code[2];
    uint8_t *valid_axis[2];
    uint8_t *dbg3[2];
    uint8_t *dbg4[2];
    int64_t *cpmap;            // [view][row][pitch][2]
    double *ipoints;           // [view][row][pitch][3]
    // (appended in round 6: the fields above keep their kernel-argument offsets)
    MaskIn mi;                 // MASKIN launches only (read through the kernel-argument segment, sl3d_fused.h: maskin_args)
};
static_assert(SL3D_SMALL_LAUNCH_VIEWS == 4, "KParams::mi_origin holds one entry per view of a small launch");

// launchers (sl3d_fused_launch.hip, sl3d_kernels.hip); `stream` is a hipStream_t
// cmode: 0 = dense xyz + valid planes, 2 = segmented clouds
// prefer_gated: the views of a small launch are sparsely selected (sl3d_capi_inputs.cpp: sparse_views)
// mi != nullptr: a MASKIN launch (the views' valid bits from their raw selection; only where fused_maskin_available says so)
int launch_fused(const KParams &P, const DevCal *d_cal, int rig, int first_view, int n_views, bool keep, int cmode, void *stream, bool prefer_gated = false,
                 const MaskIn *mi = nullptr);
// the k_fused instantiation such a launch runs, as rocprofv3 spells it; returns snprintf's value
int fused_kernel_name(const KParams &P, int rig, int n_views, bool keep, int cmode, char *buf, size_t cap, bool prefer_gated = false, bool maskin = false);
// can a launch of n_views views of this context evaluate the selection itself?  (3-step fringes, a pipelined rig class, at most
// SL3D_SMALL_LAUNCH_VIEWS views, up to 12 Gray planes per axis; not the parity mode)
bool fused_maskin_available(const KParams &P, int rig, int n_views, bool keep);
// words per view in MaskIn::part (one per wave of the small-launch grid) / how many of them belong to waves that own pixels
unsigned fused_maskin_part_stride(const KParams &P);
unsigned fused_maskin_part_words(const KParams &P);
// segmented clouds: offsets / totals of views [first_view, first_view + n_views) from the counts the fused kernel stored
int launch_seg_scan(const KParams &P, int first_view, int n_views, void *stream);
// segments -> contiguous: view first_view+k's points to dst + 3*k*dst_view_stride_points (dst: device memory or mapped host memory)
int launch_seg_close(const KParams &P, int first_view, int n_views, float *dst, size_t dst_view_stride_points, void *stream);
// the same when the views' counts have not been scanned: the consumer scans on entry (k_seg_close<.., SCAN>), writes at most
// capacity_points points per view and leaves the views' totals in P.cloud_totals
int launch_seg_close_scan(const KParams &P, int first_view, int n_views, float *dst, size_t dst_view_stride_points, unsigned long long capacity_points,
                          void *stream);
// register_point_clouds on segmented input: view first_view+k rotated by R4[4*k..], written at out + 3*(out_base[k] + offset)
int launch_seg_register(const KParams &P, int view, float *out, const float R4[4], float tx, float ty, float tz, void *stream);
int fused_tiles(const KParams &P);  // number of 1024-pixel tiles per view (KParams::n_tiles)
// k_mask_prepare over views [first_view, first_view + n_views): view k reads S.origin + k * S.view_stride; block b of view v stores
// {seq, quads with a valid pixel} at partials[v * mask_prepare_blocks(P) + b] (host memory mapped into the device)
int launch_mask_prepare(const KParams &P, int first_view, int n_views, const MaskSrc &S, unsigned long long *partials, unsigned seq, void *stream);
int mask_prepare_blocks(const KParams &P);
int launch_to_colrow(const KParams &P, int view, int which, void *dst, void *stream);  // a global in the reference's [col][row] layout
int launch_mask_from_colrow(const KParams &P, const int *sel, int gx0, int gy0, int ncols, int nrows, uint8_t *raw, void *stream);
int launch_proj_table(const DevCal *d_cal, int PW, int PH, float2 *out, void *stream);
// SL3D_RAD_COPIES copies (SL3D_RAD_STRIDE entries apart) of the SL3D_RAD_NODES nodes of the radial factor of the camera (which = 0) or
// the projector (1) over r0^2 in [0, r2max]
int launch_radial_table(const DevCal *d_cal, int which, double r2max, RadEntry *out, void *stream);
int launch_cam_table(const KParams &P, const DevCal *d_cal, int kind, double *out, void *stream);
int launch_wrap(const KParams &P, int view, int axis, void *stream);
int launch_unwrap(const KParams &P, int view, int axis, void *stream);
int launch_corr(const KParams &P, int view, void *stream);
int launch_tri(const KParams &P, const DevCal &C, int view, void *stream);
int launch_compact(const KParams &P, int view, unsigned *block_counts, unsigned long long *block_offsets, unsigned long long *total,
                   float *cloud, const uint8_t *texture, uint8_t *rgb_out, void *stream);
int launch_compact_views(const KParams &P, int first_view, int n_views, unsigned *block_counts, unsigned long long *block_offsets,
                         unsigned long long *totals, float *clouds, void *stream);
int launch_register(const float *in, float *out, long n, const float R4[4], float tx, float ty, float tz, void *stream);
int launch_synth(const KParams &P, const DevCal &C, const SynthParams &S, int view, void *stream);
int launch_undistort(const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride, int width, int height, int cn, const double K[9],
                     const double dist[5], short *m1, unsigned short *m2, bool build_map, void *stream);
int launch_undistort_planes(const uint8_t *src, size_t spitch, size_t splane, uint8_t *dst, size_t dpitch, size_t dplane, int width, int height,
                            int n_planes, const double K[9], const double dist[5], short *m1, unsigned short *m2, bool build_map, void *stream);
int launch_pattern(uint8_t *dst, size_t pitch, int PW, int PH, int axis, const uint8_t *profile, void *stream);
int launch_atan_selfcheck(const float *tab_phi, const float *tab_shift, unsigned *mismatches, void *stream);

}  // namespace sl3d
