/*
 * sl3d.h -- C ABI of the MI355X-native decode -> unwrap -> correspond -> triangulate path
 *           of the pranavkantgaur/3dscan structured-light scanner.
 *
 * This is the drop-in boundary for stages 3, 4, 5 and 7 of the reference.  Each entry point
 * names the reference interface it replaces (paths are relative to the reference tree).
 * The reference's own four entry points (C++ linkage, no arguments, global-array results) are
 * re-exported on top of this ABI by include/sl3d_shim.h / csrc/sl3d_shim.cpp.
 *
 * Conventions
 *   - plain C: opaque handle, int status codes (0 = ok, negative = error), no exceptions.
 *   - images are 8-bit single-channel row-major planes passed as (pointer, stride in bytes),
 *     which is exactly what an IplImage holds (imageData, widthStep) after
 *     cvLoadImage(..., CV_LOAD_IMAGE_GRAYSCALE)  [3/wrapped_phase.cpp:44, 4/phase_unwrap.cpp:78,84].
 *   - results are row-major planes; the shim transposes into the reference's [col][row] globals.
 *   - one context = one GPU + one HIP stream; all compute calls are asynchronous on that stream,
 *     getters synchronise.  Calls on one context must be serialised by the caller; different
 *     contexts are independent.
 *   - a context processes a WINDOW (width x height at col0,row0) of a camera frame of
 *     full_width x full_height: the whole frame on one GPU, or a row stripe per GPU.
 *   - the library needs a HIP device: there is no CPU fallback (SL3D_E_NO_DEVICE).
 */
#ifndef SL3D_H
#define SL3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SL3D_VERSION_STRING "0.6.0"

typedef struct sl3d_ctx sl3d_ctx;

enum sl3d_status {
    SL3D_OK = 0,
    SL3D_E_INVALID_ARG = -1,
    SL3D_E_NO_DEVICE = -2,   /* no HIP device / runtime: the product path has no CPU fallback */
    SL3D_E_HIP = -3,         /* a HIP runtime call failed; see sl3d_last_error() */
    SL3D_E_STATE = -4,       /* call order violated (e.g. triangulate before set_calibration) */
    SL3D_E_UNSUPPORTED = -5,
    SL3D_E_NOMEM = -6,       /* device or host memory exhausted */
    SL3D_E_INTERNAL = -7     /* a C++ exception was stopped at the boundary (no exception ever crosses this C ABI); see sl3d_last_error() */
};

/* pattern_type of the reference: 0 = vertical stripes (encode the projector COLUMN),
 * 1 = horizontal stripes (encode the projector ROW)   [intermodule_dependencies.h:10,13] */
enum sl3d_axis { SL3D_AXIS_VERTICAL = 0, SL3D_AXIS_HORIZONTAL = 1 };

/* which valid map: valid_map_vertical / valid_map_horizontal / valid_map  [common_variables.h:18-21] */
enum sl3d_valid_which { SL3D_VALID_VERTICAL = 0, SL3D_VALID_HORIZONTAL = 1, SL3D_VALID_MERGED = 2 };

enum sl3d_flags {
    /* allocate the stage-boundary planes (wrapped/unwrapped phase, code, per-axis valid maps,
     * debug images, c_p_map, intersection_points).  Required for the per-stage entry points and
     * their getters; sl3d_run() then also stores them ("parity mode", +~70 B/px of traffic).
     * Without it only sl3d_run() and the point / valid getters are available (the timed mode). */
    SL3D_FLAG_KEEP_STAGES = 1u,
    /* sl3d_group_create only.  FORCE_RCCL: every stripe but the root's own travels by RCCL send/recv even when it shares the
     * root's GPU (exercises the RCCL path on a single-GPU box); NO_RCCL: (peer) device copies even across GPUs. */
    SL3D_FLAG_GROUP_FORCE_RCCL = 2u,
    SL3D_FLAG_GROUP_NO_RCCL = 4u,
    /* sl3d_group_create only, for tests on a box with fewer GPUs than stripes: every stripe gets its own communication side
     * (stream, event, communicator rank) even where devices repeat, so that the code paths of a stripe on ANOTHER GPU than the
     * root's -- peer copies, the waits between the sides' streams, an N-rank exchange -- run with all stripes on one device.
     * With RCCL this needs a communicator that accepts repeated devices (the test double bound through SL3D_RCCL_LIB). */
    SL3D_FLAG_GROUP_DISTINCT_SIDES = 16u,
    /* A context without KEEP_STAGES normally DEFERS the preparation of a selection mask handed over for at most 4 views: the next
     * sl3d_run / sl3d_run_clouds over exactly such views evaluates the selection inside the fused kernel itself (new mask + one view
     * = ONE launch, the reference's per-scan call shape), any other consumer prepares it first.  With this flag every mask is prepared
     * by its own kernel when it is set (the behaviour up to 0.4; A/B measurements). */
    SL3D_FLAG_EAGER_MASK = 32u,
    /* A context that created its own stream (config.stream == NULL) and has no KEEP_STAGES puts the sl3d_run / sl3d_run_clouds launches
     * of a LONG series of launches of at most 2 views (8 in a row, or fewer if the previous series was that long) on two internal
     * streams in turn, so that the tail of one launch runs under the ramp of the next when they work on different views (a launch over
     * a view a lane still works on goes behind it); every other call first makes the context's stream wait for both, so nothing a
     * caller can observe changes -- only the time: one view per launch from HBM, back to back, 25.8 -> 21.8 us per launch.  With this
     * flag every launch goes to the context's one stream (the behaviour up to 0.5; A/B measurements, and the time of a LONE launch). */
    SL3D_FLAG_SERIAL_LAUNCHES = 64u
    /* (8u was SL3D_FLAG_CLOUDS_LOOKBACK until 0.3: contiguous clouds in one pass by a decoupled look-back between tiles.  Removed
     * in 0.4 -- every tile waited for its predecessors, 0.53 of the roofline against 0.67 for the segmented clouds; a consumer
     * that wants one contiguous device array asks sl3d_get_cloud_counts for it.) */
};

/* Replaces the compile-time macros and initialised globals of the reference:
 * PROJECT_GLOBAL/global_cv.h:49-53 (dimensions) and common_variables.h:6-10,23-24. */
typedef struct sl3d_config {
    int32_t width, height;            /* window processed by this context (pixels)                 */
    int32_t full_width, full_height;  /* Camera_imagewidth/height of the frame; 0 = same as window */
    int32_t col0, row0;               /* window origin inside the frame                            */
    int32_t proj_width, proj_height;  /* Projector_imagewidth / Projector_imageheight (fewer than 2^29 pixels) */
    int32_t n_fringe;                 /* number_of_patterns_fringe: 3 (or 4; 5 yields no valid pixel,
                                         exactly as 3/wrapped_phase.cpp:106-129 does)              */
    int32_t n_gray_v, n_gray_h;       /* number_of_patterns_binary_{vertical,horizontal}: 0..16; they need
                                         not be equal (the reference's own capture set is 6 / 5)   */
    int32_t fringe_width_v, fringe_width_h; /* fringe_width_pixels_{vertical,horizontal}           */
    int32_t n_codes_v, n_codes_h;     /* number_of_codes_* (stage-4 debug image only); 0 = ceil(P/fw) */
    int32_t max_views;                /* batch capacity: views resident in HBM at once (>= 1)      */
    int32_t device;                   /* HIP device ordinal                                        */
    uint32_t flags;                   /* sl3d_flags                                                */
    void *stream;                     /* hipStream_t to run on, or NULL: the context creates one   */
} sl3d_config;

/* Device-resident layout, for callers that produce frames / consume points on the GPU
 * (benchmarks, torch tensors, RCCL).  All pitches are in bytes unless noted. */
typedef struct sl3d_device_buffers {
    /* frames: view-major, then plane, then row.  Plane order inside a view:
     *   vertical   axis: fringe[0..F), gray[0..N_v), inverse_gray[0..N_v)
     *   horizontal axis: fringe[0..F), gray[0..N_h), inverse_gray[0..N_h)                         */
    uint8_t *frames;
    size_t frame_pitch;       /* bytes per row (multiple of 16, >= width)        */
    size_t plane_stride;      /* bytes per plane = frame_pitch * height          */
    size_t view_stride;       /* bytes per view  = planes_per_view * plane_stride */
    int32_t planes_per_view;  /* 2*F + 2*N_v + 2*N_h                              */
    /* selection mask with a 2-pixel halo; byte for window pixel (col,row) of a view is at
     * mask + view*mask_view_stride + (row + 2)*mask_pitch + 16 + col; bytes are 0 or 1.           */
    uint8_t *mask;
    size_t mask_pitch;
    size_t mask_view_stride;
    /* dense results of sl3d_run(): xyz as 3 consecutive floats per pixel (NaN where invalid);
     * pixel (col,row) of a view at points + view*points_view_stride + row*points_pitch + 12*col   */
    float *points;
    size_t points_pitch;        /* bytes per row = 12 * (frame_pitch)            */
    size_t points_view_stride;  /* bytes per view                                */
    uint8_t *valid;             /* merged valid map after stage 5, 0/1           */
    size_t valid_pitch;         /* bytes per row (= frame_pitch)                 */
    size_t valid_view_stride;   /* bytes per view                                */
} sl3d_device_buffers;

/* ---- library ------------------------------------------------------------------------------ */
const char *sl3d_version(void);
const char *sl3d_strerror(int status);
/* text of the last error on this context (HIP error string etc.); never NULL */
const char *sl3d_last_error(const sl3d_ctx *ctx);

/* ---- lifetime ----------------------------------------------------------------------------- */
/* Replaces the `new[]` allocations the reference performs inside every stage and never frees
 * (3/wrapped_phase.cpp:410-424, 4/phase_unwrap.cpp:282,300,373-376, 5/compute_correspondance.cpp:635-640,
 * 7/triangulation.cpp:1513): all buffers are owned by the context and released by sl3d_destroy. */
int sl3d_create(const sl3d_config *cfg, sl3d_ctx **out);
void sl3d_destroy(sl3d_ctx *ctx);

/* ---- inputs ------------------------------------------------------------------------------- */
/* The 8 calibration files read by read_parameters() 7/triangulation.cpp:149-180 and
 * compute_A() :1061-1126: intrinsics K (3x3 row-major), distortion (k1,k2,p1,p2,k3),
 * Rodrigues rotation vector and translation vector (world -> device) for camera and projector.
 * In the timed mode the call also builds, on the device, what the reference tabulates per scan (assign_3d_coordinates,
 * 7/triangulation.cpp:252-307, :352-378): the camera-side undistortion per window pixel and, for a distorted projector, either a
 * displacement per projector pixel or -- purely radial model, plain K: both projector calibrations the reference ships -- a 4-KB
 * table of the radial factor.  It synchronises the context's stream. */
int sl3d_set_calibration(sl3d_ctx *ctx,
                         const double Kc[9], const double dc[5], const double rc[3], const double tc[3],
                         const double Kp[9], const double dp[5], const double rp[3], const double tp[3]);

/* T0 as the library computed it from the last sl3d_set_calibration: the two 3x4 projection matrices A = K*[R|t] of compute_A()
 * (7/triangulation.cpp:1061-1126: cvRodrigues2 :1072,1080, [R|t] :1090-1099,1105-1114, cvMatMul :1101,1116), row-major doubles.
 * Host only.  With K = I the left 3x3 block is cvRodrigues2's matrix itself -- which is how tests/ compare the product's T0 with
 * the reference-held OpenCV answer (the two XML files under Triangulation/Relative_geometry, written by 6/system_calibration.cpp:1488-1516). */
int sl3d_get_projection_matrices(sl3d_ctx *ctx, double A_cam[12], double A_proj[12]);

/* selected_region of image_scissor() m_tech_project_console.cpp:146-238, handed over as a
 * FULL-FRAME row-major u8 plane (full_width x full_height); a pixel is selected iff byte == 1
 * (every consumer in the reference tests `== 1`).  The context copies its window plus a 2-pixel halo in ONE 2-D copy and
 * prepares it on the device (bytes normalised to 0/1, border band of the boundary removal evaluated by a kernel, the selected
 * quads counted -- a launch of a few sparsely selected views picks its kernel by that count): no host pass over the mask.  Pageable memory is consumed before the call returns; PINNED memory (sl3d_host_alloc) is read by
 * asynchronous DMA on the context's stream, so it must stay unchanged until the next synchronising call on the context
 * (any getter, sl3d_synchronize) -- the call then costs a copy and a launch (tens of microseconds at 1080p). */
int sl3d_set_mask(sl3d_ctx *ctx, int view, const uint8_t *full_frame_mask, size_t stride);

/* The same for n_views views [first_view, first_view + n_views) with ONE kernel launch: the mask of view first_view + k starts at
 * full_frame_masks + k * view_stride; view_stride == 0 hands every view the same mask (one copy, one launch).
 * The masks may also live in DEVICE memory of the context's GPU (an acquisition stage that segments on the GPU, a benchmark): with
 * 4-byte aligned rows (pointer, stride, view_stride, col0 and full_width multiples of 4) nothing is copied -- the kernels read the
 * caller's buffer, which must stay unchanged until the next synchronising call on the context (any getter, sl3d_synchronize),
 * like pinned host memory: with at most 4 views per call the selection is evaluated by the next launch over those views
 * (SL3D_FLAG_EAGER_MASK); otherwise the rows go through the staging plane by a device copy.  This is the per-scan device cost of image_scissor()'s result (m_tech_project_console.cpp:366) + stage
 * 3's boundary removal (3/wrapped_phase.cpp:253-279): bench.py `side.per_scan_device`. */
int sl3d_set_masks(sl3d_ctx *ctx, int first_view, int n_views, const uint8_t *full_frame_masks, size_t stride, size_t view_stride);

/* The captured frames of one axis of one view, window-sized planes in host memory: what
 * read_image() 3/wrapped_phase.cpp:29-58 (n_fringe planes) and read_captured_images()
 * 4/phase_unwrap.cpp:51-131 (n_gray planes + n_gray inverse planes) load.
 * planes[] order: fringe[0..F), gray[0..N), inverse_gray[0..N).  Planes that follow each other in host memory (plane i+1
 * at plane i + stride*height) go up as ONE 2-D copy per axis.  Pageable / pinned memory: as for sl3d_set_mask; planes in DEVICE
 * memory (another context's frame stack, sl3d_get_device_buffers) are copied device to device, asynchronously. */
int sl3d_set_frames(sl3d_ctx *ctx, int view, int axis, const uint8_t *const *planes, int n_planes, size_t stride);

/* selected_region exactly as the reference holds it: int [full_width][full_height], indexed [col][row]
 * (m_tech_project_console.cpp:146-238, common_variables.h), selected iff == 1.  The window's columns (+ halo) go up as one 2-D
 * copy and are transposed into the byte mask on the device; no pass of the host over the array. */
int sl3d_set_mask_colrow(sl3d_ctx *ctx, int view, const int32_t *selected_region);

/* n_planes planes of one axis starting at plane index first_plane of sl3d_set_frames' order (fringe[0..F), gray[0..N),
 * inverse_gray[0..N)): stage 3 loads only the F fringe frames (read_image, 3/wrapped_phase.cpp:29-58), stage 4 only the Gray /
 * inverse frames (read_captured_images, 4/phase_unwrap.cpp:51-131). */
int sl3d_set_frames_range(sl3d_ctx *ctx, int view, int axis, int first_plane, const uint8_t *const *planes, int n_planes, size_t stride);

/* Device-to-device duplicate of one resident view (frame stack + mask) into another slot of the batch. */
int sl3d_copy_view(sl3d_ctx *ctx, int src_view, int dst_view);

/* Synthetic capture generated on the device into view slot `view` (inputs for tests and benchmarks; the pattern
 * formulas are the reference generator's, 1/pattern_generator.cpp:80-105,302,313,497, Pi = 22/7): plane[3] = z0,a,b of
 * the scene plane Z = z0 + a*X + b*Y in the world frame of the calibration set with sl3d_set_calibration; camera
 * model I' = clamp(round(gain*I + offset + noise)), noise uniform in [-noise, noise] from a counter hash of
 * (seed, view_id, frame, row, col).  The mask is not touched. */
int sl3d_synth_view(sl3d_ctx *ctx, int view, const double plane[3], uint64_t seed, int view_id, int noise, float gain, float offset);
/* read the resident frames of one axis back (same plane order as sl3d_set_frames) */
int sl3d_get_frames(sl3d_ctx *ctx, int view, int axis, uint8_t *const *planes, int n_planes, size_t stride);

/* ---- the reference's four stage entry points (need SL3D_FLAG_KEEP_STAGES) ------------------- */
/* void compute_wrapped_phase(int pattern_type)   3/wrapped_phase.cpp:402   */
int sl3d_compute_wrapped_phase(sl3d_ctx *ctx, int view, int axis);
/* void unwrap_phase(int pattern_type)            4/phase_unwrap.cpp:367    */
int sl3d_unwrap_phase(sl3d_ctx *ctx, int view, int axis);
/* void compute_c_p_map()                         5/compute_correspondance.cpp:630 */
int sl3d_compute_c_p_map(sl3d_ctx *ctx, int view);
/* void triangulate()                             7/triangulation.cpp:1444  */
int sl3d_triangulate(sl3d_ctx *ctx, int view);

/* ---- the fused hot path ------------------------------------------------------------------- */
/* Stages 3(v) 3(h) 4(v) 4(h) 5 7 in main()'s order (m_tech_project_console.cpp:372-395) plus the
 * float cast of 8/save_point_cloud.cpp:100-102, for views [first_view, first_view+n_views), as ONE
 * kernel launch that reads every frame byte once and writes xyz (f32) + valid (u8). Asynchronous. */
int sl3d_run(sl3d_ctx *ctx, int first_view, int n_views);
/* The same pass with the compaction of 8/save_point_cloud.cpp:33-37,85-104 INSIDE the kernel: instead of the dense xyz plane every
 * view's valid points are written once, compacted in the reference's row-major scan order, plus the valid map:
 * ~47 + 12*valid_fraction + 1 bytes per pixel instead of 60 for the dense pass + 25 for a separate compaction.  Timed mode only
 * (no SL3D_FLAG_KEEP_STAGES).  Asynchronous.
 *   SEGMENTED clouds.  Every wave of the kernel compacts the 256 consecutive scan pixels it owns into its own fixed slot
 *   (points [256*s, 256*s + count_s) of the view's region) and stores count_s.  The counts become offsets by a small scan kernel
 *   right behind a launch of more than 4 views; a launch of up to 4 views -- the reference's one scan per call -- leaves the scan
 *   to its consumer (the gap-closing kernel adds up the counts in front of its segments on entry; sl3d_get_cloud_segments and
 *   sl3d_register_clouds, which want the offsets as an array, run the scan kernel when they are called).
 *   No tile ever waits for another one, scan order is preserved inside and across segments, so a view's cloud is the
 *   concatenation of its segments -- and the consumers that exist anyway close the gaps for free: sl3d_download_clouds (host
 *   copy), sl3d_register_clouds (turntable registration), the pack before a group's RCCL send, or any device consumer through
 *   sl3d_get_cloud_segments.
 * sl3d_get_cloud_counts synchronises and returns, for views [first_view, first_view+n_views), the number of points of each
 * cloud; with device_xyz != NULL also a CONTIGUOUS device copy: view first_view+k's cloud is counts[k] points (3 floats each) at
 * *device_xyz + 3*k*(*view_stride_points), valid until the next sl3d_run_clouds / sl3d_compact_views on this context (produced
 * on demand by one gap-closing launch; pass NULL if only the counts are wanted). */
int sl3d_run_clouds(sl3d_ctx *ctx, int first_view, int n_views);
int sl3d_get_cloud_counts(sl3d_ctx *ctx, int first_view, int n_views, const float **device_xyz, size_t *view_stride_points, int64_t *counts);

/* The segmented clouds themselves, for consumers that stay on the device:
 * segment s of view first_view+k holds counts[k*view_stride_segments + s] points at xyz + 3*(k*view_stride_points +
 * s*segment_points); its first point is point number offsets[k*view_stride_segments + s] of the view's cloud. */
typedef struct sl3d_cloud_segments {
    const float *xyz;
    const uint32_t *counts;
    const uint64_t *offsets;
    int32_t n_segments;            /* per view */
    int32_t segment_points;        /* point slots per segment (256) */
    size_t view_stride_points;
    size_t view_stride_segments;   /* = n_segments */
} sl3d_cloud_segments;
int sl3d_get_cloud_segments(sl3d_ctx *ctx, int first_view, int n_views, sl3d_cloud_segments *out, int64_t *counts);

/* The reference's consumer is the host (8/save_point_cloud.cpp:85-104 fills a host pcl::PointCloud): the clouds of views
 * [first_view, first_view+n_views) of the last sl3d_run_clouds, back to back in host memory (at most `capacity` points in all;
 * xyz may be NULL: counts only).  Pinned memory (sl3d_host_alloc): the gap-closing kernel writes the host buffer directly;
 * otherwise a contiguous device copy is downloaded.  Synchronises. */
int sl3d_download_clouds(sl3d_ctx *ctx, int first_view, int n_views, float *xyz, int64_t capacity, int64_t *counts);

/* sl3d_register_views on the clouds of the last sl3d_run_clouds (no dense planes, no separate compaction): the rotation of
 * 9/register_point_clouds.cpp:83-128 is applied while the segments are concatenated. */
int sl3d_register_clouds(sl3d_ctx *ctx, int first_view, int n_views, float tx, float ty, float tz, float rot_step, float *xyz, int64_t capacity,
                         int64_t *total);
/* Name of the fused-kernel instantiation sl3d_run (clouds = 0) / sl3d_run_clouds (clouds = 1) launches for a batch of n_views views
 * with this context's configuration and calibration, spelled as rocprofv3 --kernel-trace prints it: benchmarks name the kernel
 * their roofline figure is about without restating the library's dispatch rules. */
int sl3d_fused_kernel_name(sl3d_ctx *ctx, int n_views, int clouds, char *buf, size_t capacity);
/* The instantiation the LAST fused launch of this context ran (sl3d_run, sl3d_run_clouds, sl3d_run_timed, sl3d_process_views): the
 * choice is recorded when the launch is made -- it depends on what was known about the views' masks at that moment (a small
 * launch over sparsely selected views takes another kernel; while the count of a selection is still on its way from the device, the
 * last count of that view that did arrive decides), so a later prediction could name a different one.  The choice never changes a
 * bit of the results. */
int sl3d_last_fused_kernel_name(sl3d_ctx *ctx, char *buf, size_t capacity);
/* How many fused launches of this context went to the context's stream and how many to its launch lanes (SL3D_FLAG_SERIAL_LAUNCHES):
 * what a benchmark or a test states beside a per-launch time of a series.  Either pointer may be NULL.  Touches nothing on the device. */
int sl3d_launch_counts(sl3d_ctx *ctx, int64_t *on_stream, int64_t *on_lanes);
/* Bytes per camera pixel a fused launch of n_views views reads from the camera-side table of sl3d_set_calibration (T1 per window pixel,
 * 7/triangulation.cpp:252-307), once per LAUNCH whatever the number of views: 0 = no table (no camera distortion, parity mode),
 * 8 = the radial factor as a double, 16 = the normalised point (tangential terms).  Benchmarks state the bytes a launch moves beside
 * the algorithmic ones with it.  < 0: error. */
int sl3d_camera_table_bytes_per_pixel(sl3d_ctx *ctx, int n_views);
/* sl3d_run bracketed by HIP events on the context's stream; returns the kernel time of this launch */
int sl3d_run_timed(sl3d_ctx *ctx, int first_view, int n_views, float *kernel_ms);
int sl3d_synchronize(sl3d_ctx *ctx);
/* HIP-event stopwatch on the context's stream (the stream the kernels are launched on): start
 * records an event, stop records a second one, waits for it and returns the elapsed device time
 * between the two -- used to time a whole region of back-to-back launches without host syncs. */
int sl3d_timer_start(sl3d_ctx *ctx);
int sl3d_timer_stop(sl3d_ctx *ctx, float *elapsed_ms);

/* ---- outputs (all synchronise the stream first; planes are window-sized, row-major) --------- */
/* valid_map_vertical / _horizontal / valid_map  (int[W][H] in the reference), as 0/1 bytes */
int sl3d_get_valid_map(sl3d_ctx *ctx, int view, int which, uint8_t *out, size_t stride);
/* wrapped_phi_{vertical,horizontal}: after stage 3 the atan2 phase, after stage 4 shifted by
 * +Pi in place on unwrapped pixels, as the reference does (4/phase_unwrap.cpp:290,308) */
int sl3d_get_wrapped_phase(sl3d_ctx *ctx, int view, int axis, float *out, size_t stride_elems);
int sl3d_get_unwrapped_phase(sl3d_ctx *ctx, int view, int axis, float *out, size_t stride_elems);
/* code_{vertical,horizontal}: Gray-decoded period index, -1 where invalid (4/phase_unwrap.cpp:143) */
int sl3d_get_code(sl3d_ctx *ctx, int view, int axis, int32_t *out, size_t stride_elems);
/* the 8-bit debug images of stage 3 (3/wrapped_phase.cpp:178-179,274) and stage 4
 * (4/phase_unwrap.cpp:334-335): the reference's known-answer images are these */
int sl3d_get_debug_image(sl3d_ctx *ctx, int view, int stage, int axis, uint8_t *out, size_t stride);
/* c_p_map: long[W*H][2] indexed [row*W+col] (common_variables.h:15), zeros where invalid */
int sl3d_get_c_p_map(sl3d_ctx *ctx, int view, int64_t *out);
/* intersection_points as row-major double [H][W][3] (the reference stores [col][row][3]) */
int sl3d_get_intersection_points(sl3d_ctx *ctx, int view, double *out);
/* dense f32 points [H][W][3] (NaN where invalid) + merged valid map [H][W]; either may be NULL */
int sl3d_get_points(sl3d_ctx *ctx, int view, float *xyz, uint8_t *valid);
/* One of the reference's image-shaped globals in the reference's OWN layout and types (common_variables.h:12-21,56-62: every one
 * is indexed [col][row]): transposed and converted on the device, then ONE contiguous copy into the caller's array.
 *   which: SL3D_G_VALID_V / _H / SL3D_G_VALID -> int [W][H]; SL3D_G_WRAPPED_V / _H, SL3D_G_UNWRAPPED_V / _H -> float [W][H];
 *          SL3D_G_CODE_V / _H -> int [W][H]; SL3D_G_INTERSECTION_POINTS -> double [W][H][3]
 *   out: element (col, row) of this context's window is written at out[col * out_height + out_row0 + row]: a whole-frame
 *        context passes (height, 0); a row stripe of a taller array passes the array's height and its own first row, and
 *        writes only its rows of every column.  Needs SL3D_FLAG_KEEP_STAGES.  Synchronises. */
enum sl3d_global {
    SL3D_G_VALID_V = 0, SL3D_G_VALID_H = 1, SL3D_G_VALID = 2, SL3D_G_WRAPPED_V = 3, SL3D_G_WRAPPED_H = 4,
    SL3D_G_UNWRAPPED_V = 5, SL3D_G_UNWRAPPED_H = 6, SL3D_G_CODE_V = 7, SL3D_G_CODE_H = 8, SL3D_G_INTERSECTION_POINTS = 9,
    /* double [W][H][3] like SL3D_G_INTERSECTION_POINTS, but the dense f32 result of sl3d_run widened to double (NaN where invalid):
     * what a context WITHOUT SL3D_FLAG_KEEP_STAGES has of intersection_points -- the very values 8/save_point_cloud.cpp:94-96
     * casts them to, i.e. all that the reference's only reader of that global ever sees (like SL3D_G_VALID, no KEEP_STAGES needed) */
    SL3D_G_POINTS_F64 = 10
};
int sl3d_get_global_colrow(sl3d_ctx *ctx, int view, int which, void *out, int out_height, int out_row0);

/* the compacted cloud of 8/save_point_cloud.cpp:85-104: valid points in row-major scan order;
 * writes at most `capacity` points, always returns the total count in *count */
int sl3d_get_cloud(sl3d_ctx *ctx, int view, float *xyz, int64_t capacity, int64_t *count);

/* save_point_cloud() colours every point with the pixel of a camera image (cvLoadImage("Point_cloud/texture.bmp"), split
 * into blue/green/red, 8/save_point_cloud.cpp:46-52,70-72): sl3d_set_texture uploads that image for a view (window-sized,
 * B,G,R interleaved as cvLoadImage returns it, `stride` bytes per row); sl3d_get_cloud_rgb returns the compacted cloud and
 * r,g,b per point, gathered on the device in the same scan order. */
int sl3d_set_texture(sl3d_ctx *ctx, int view, const uint8_t *bgr, size_t stride);
int sl3d_get_cloud_rgb(sl3d_ctx *ctx, int view, float *xyz, uint8_t *rgb, int64_t capacity, int64_t *count);

/* the same compaction left on the device (valid until the next sl3d_compact / sl3d_get_cloud on this context):
 * *device_xyz points at count*3 floats in HBM; the count comes back to the host */
int sl3d_compact(sl3d_ctx *ctx, int view, const float **device_xyz, int64_t *count);

/* the compaction of views [first_view, first_view+n_views) in one go (three launches, one read-back of the counts):
 * view first_view+k's cloud is counts[k] points at *device_xyz + 3*k*(*view_stride_points) floats, valid until the next
 * sl3d_compact_views on this context */
int sl3d_compact_views(sl3d_ctx *ctx, int first_view, int n_views, const float **device_xyz, size_t *view_stride_points, int64_t *counts);
/* the same with a host copy: the clouds back to back in xyz (at most `capacity` points in all; xyz may be NULL) */
int sl3d_get_clouds(sl3d_ctx *ctx, int first_view, int n_views, float *xyz, int64_t capacity, int64_t *counts);

/* register_point_clouds(unsigned, float tx, float ty, float tz, float rot_step)  9/register_point_clouds.cpp:23:
 * the compacted clouds of views [first_view, first_view+n_views) are rotated about the Y axis through (tx,ty,tz)
 * by 0, rot_step, 2*rot_step ... degrees and concatenated in view order; writes at most `capacity` points to xyz
 * (may be NULL) and always returns the total count. */
int sl3d_register_views(sl3d_ctx *ctx, int first_view, int n_views, float tx, float ty, float tz, float rot_step,
                        float *xyz, int64_t capacity, int64_t *total);

/* ---- host-buffer pipeline -------------------------------------------------------------------- */
/* The reference hands every stage host images (IplImage loaded from disk, 3/wrapped_phase.cpp:44, 4/phase_unwrap.cpp:78,84)
 * and reads host arrays back.  For callers that stay on that side of the boundary, sl3d_process_views runs a batch of
 * host-resident views through the context's view slots as a three-stage pipeline on three HIP streams (upload of view k+1,
 * fused kernel of view k, download of view k-1 overlap), so the sustained rate is that of the slowest stage (the PCIe
 * upload), not the sum.  planes: n_views * planes_per_view pointers, view-major, plane order of sl3d_device_buffers;
 * xyz: n_views dense [height][width][3] float images, NaN where invalid (may be NULL); valid: n_views [height][width]
 * byte images (may be NULL).  Masks and calibration must be set (sl3d_set_mask for every slot < max_views).
 * sl3d_host_alloc returns pinned memory: with it the copies are asynchronous DMA (pageable memory works, serialised). */
void *sl3d_host_alloc(size_t bytes);
void sl3d_host_free(void *p);
int sl3d_process_views(sl3d_ctx *ctx, int n_views, const uint8_t *const *planes, size_t stride, float *xyz, uint8_t *valid);

/* ---- capture-side undistortion --------------------------------------------------------------- */
/* cvUndistort2(src, dst, intrinsic_matrix, distortion_coeffs) as the acquisition stage applies it to every captured frame
 * before it is saved for stage 3/4 (2/project_pattern.cpp:220,232,287,...): OpenCV 2.4.0's algorithm on the device --
 * stripe-wise inverse map accumulated along each row in double, positions rounded to 1/32 pixel, INTER_LINEAR in 2^15
 * fixed point, BORDER_CONSTANT 0.  8-bit images of 1 or 3 interleaved channels, any size; src and dst must not overlap.
 * The arithmetic lives in OpenCV (not in the reference tree) and no raw capture is available: parity unpinned. */
int sl3d_undistort(sl3d_ctx *ctx, const uint8_t *src, size_t src_stride, int width, int height, int channels, const double K[9],
                   const double dist[5], uint8_t *dst, size_t dst_stride);

/* sl3d_set_frames for RAW captures: the frames go through cvUndistort2 with the camera calibration of sl3d_set_calibration
 * on their way into the frame stack (one launch for all planes of the axis, the map is built once per calibration), i.e.
 * the step 2/project_pattern.cpp:220,232,287 performs before stage 3/4 read the images.  Whole frames only. */
int sl3d_set_frames_raw(sl3d_ctx *ctx, int view, int axis, const uint8_t *const *planes, int n_planes, size_t stride);

/* ---- projector patterns (1/pattern_generator.cpp) ------------------------------------------- */
/* allocate_memory() 1/pattern_generator.cpp:224-229: number of codes = ceil(extent / fringe_width) and number of
 * Gray / binary bit planes = ceil(logf(codes) / logf(2)) (float arithmetic, as the reference writes it). Host only. */
int sl3d_pattern_counts(int proj_extent, int fringe_width, int *n_codes, int *n_planes);

#define SL3D_PATTERN_FRINGE 0        /* fringe_pattern_generate_3/_4/_5   :291-383  (index = phase step 0..F-1) */
#define SL3D_PATTERN_GRAY 1          /* generate_gray_coded_patterns      :56-197   (index = bit plane, MSB first) */
#define SL3D_PATTERN_INVERSE_GRAY 2  /* generate_inverse_gray_coded_patterns :490-507 */
#define SL3D_PATTERN_BINARY 3        /* binary_pattern_generate           :386-412 */

/* One projector pattern of generate_pattern() (1/pattern_generator.cpp:513), proj_width x proj_height 8-bit, generated
 * on the device from the context's configuration (n_fringe, fringe widths, n_gray_v / n_gray_h bit planes).
 * axis: SL3D_AXIS_VERTICAL = the pattern varies with the column.  index == n_gray (the extra image the reference
 * saves, :433-465, and never fills) is all zeros.  host_dst (may be NULL) receives the image, `stride` bytes per row;
 * device_ptr / device_pitch (may be NULL) return the HBM copy, valid until the next call on this context. */
int sl3d_generate_pattern(sl3d_ctx *ctx, int kind, int axis, int index, uint8_t *host_dst, size_t stride,
                          const uint8_t **device_ptr, size_t *device_pitch);

/* one cloud of register_point_clouds() (9/register_point_clouds.cpp:83-128) given in host memory, as the reference reads
 * it from a PLY file: p -> R_y(theta)*(p - t) + t in the reference's arithmetic; theta in degrees (Pi = 22/7) */
int sl3d_transform_cloud(sl3d_ctx *ctx, const float *xyz_in, int64_t n, float theta_deg, float tx, float ty, float tz, float *xyz_out);

/* ---- device-resident access ---------------------------------------------------------------- */
/* Frames may be produced in place (frames buffer) and points / valid consumed in place.  The mask buffer is exposed for
 * inspection only: set masks through sl3d_set_mask, which also normalises the bytes to 0/1 and evaluates the quads within
 * 3 pixels of the frame border (a second plane the kernels read for those quads). */
int sl3d_get_device_buffers(sl3d_ctx *ctx, sl3d_device_buffers *out);
/* Copy `bytes` from a device address this library handed out (sl3d_get_cloud_counts, sl3d_compact, sl3d_compact_views,
 * sl3d_get_device_buffers) to host memory, ordered after the context's work; synchronises. */
int sl3d_download(sl3d_ctx *ctx, void *host_dst, const void *device_src, size_t bytes);
/* the same for a pitched region (`height` rows of `width_bytes`), ENQUEUED on the context's stream: asynchronous for pinned host
 * memory (the caller waits with sl3d_synchronize), so many regions can be queued behind one wait */
int sl3d_download_2d(sl3d_ctx *ctx, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes, size_t height);

/* ---- several GPUs behind one caller: row-stripe groups -------------------------------------------------------------
 * The reference is ONE process that scans one view after the other (m_tech_project_console.cpp:366-395); a group lets that
 * single caller use the GPUs of a node without becoming a distributed program.  The window of `cfg` is cut into n_stripes
 * contiguous row stripes (stripe i gets rows [row0_i, row0_i + rows_i): height/n_stripes rows each, the first
 * height % n_stripes stripes one more); stripe i is an ordinary context on HIP device devices[i] with its own stream.
 * Every stage is a per-pixel map whose only neighbourhood input is the selection mask, so the stripes need NO exchange
 * while they compute (each keeps 2 halo rows of the INPUT mask).  The only exchange is the assembly of the results on the
 * root (stripe 0's device): sl3d_group_gather = ONE RCCL group of point-to-point sends over xGMI (ncclSend / ncclRecv,
 * a gather) that land in place in the root's dense [view][row] buffers -- rank order = row order, so the row-major scan
 * order of 8/save_point_cloud.cpp:85 is preserved; stripes that live on the root's own GPU are device-to-device copies.
 * Devices may repeat (several stripes on one GPU: tests, or more stripes than GPUs).  cfg->device and cfg->stream are
 * ignored.  All group calls are asynchronous unless stated; compute runs on the stripes' streams, communication on one
 * communication stream per GPU, handed over by events, so sl3d_group_run of the next views overlaps the gather of the
 * previous ones. */
typedef struct sl3d_group sl3d_group;

int sl3d_group_create(const sl3d_config *cfg, const int *devices, int n_stripes, sl3d_group **out);
void sl3d_group_destroy(sl3d_group *g);
/* text of the last error on this group; sl3d_last_error(NULL) after a failed sl3d_group_create */
const char *sl3d_group_last_error(const sl3d_group *g);
int sl3d_group_size(const sl3d_group *g);
/* stripe i: its rows inside the window, its device and its context (owned by the group; any single-context call that
 * does not change the configuration may be used on it, e.g. sl3d_synth_view, sl3d_get_points, sl3d_get_device_buffers) */
int sl3d_group_stripe(sl3d_group *g, int i, int *row0, int *rows, int *device, sl3d_ctx **ctx);
/* "rccl" if stripes on other GPUs travel by RCCL send/recv, "copy" if every transfer is a (peer) device copy */
const char *sl3d_group_transport(const sl3d_group *g);

/* the single-context inputs, applied to every stripe (each uploads only its own byte range of every plane, on its own GPU;
 * planes are WINDOW-sized: `width` x `height` of cfg) */
int sl3d_group_set_calibration(sl3d_group *g, const double Kc[9], const double dc[5], const double rc[3], const double tc[3],
                               const double Kp[9], const double dp[5], const double rp[3], const double tp[3]);
int sl3d_group_set_mask(sl3d_group *g, int view, const uint8_t *full_frame_mask, size_t stride);
int sl3d_group_set_frames(sl3d_group *g, int view, int axis, const uint8_t *const *planes, int n_planes, size_t stride);

/* sl3d_run on every stripe (one launch per GPU stream) */
int sl3d_group_run(sl3d_group *g, int first_view, int n_views);
/* assemble xyz + valid of views [first_view, first_view+n_views) on the root GPU: waits (on the device) for the stripes'
 * kernels of those views, then one grouped exchange on the communication streams */
int sl3d_group_gather(sl3d_group *g, int first_view, int n_views);
/* the assembled dense results of a view (window-sized [height][width][3] f32, NaN where invalid; [height][width] valid);
 * waits for the gather of that view; either pointer may be NULL */
int sl3d_group_get_points(sl3d_group *g, int view, float *xyz, uint8_t *valid);
/* The assembly for the reference's real consumer, the HOST (its results are host globals / a host cloud: common_variables.h:12-21,
 * 56-62, 8/save_point_cloud.cpp:85-104): views [first_view, first_view+n_views) as n_views dense [height][width][3] float images
 * (NaN where invalid) and [height][width] valid bytes; either pointer may be NULL.  Every stripe copies its rows straight into
 * the caller's images on its own stream and over its own GPU's PCIe link -- no gather to a root, no xGMI hop.  Use pinned
 * memory (sl3d_host_alloc) for concurrent DMA.  Waits for the stripes' kernels and for the copies. */
int sl3d_group_download_points(sl3d_group *g, int first_view, int n_views, float *xyz, uint8_t *valid);
/* sl3d_process_views for a group: host-resident views through every stripe's three-stream pipeline (upload of its rows,
 * kernel, download into the caller's dense images); the stripes work concurrently.  planes: n_views * planes_per_view
 * pointers to window-sized planes, view-major. */
int sl3d_group_process_views(sl3d_group *g, int n_views, const uint8_t *const *planes, size_t stride, float *xyz, uint8_t *valid);
/* the assembled buffers in the root GPU's memory, for consumers that stay on the device (layout of sl3d_device_buffers'
 * points / valid with the window's height; frames / mask members are not filled) */
int sl3d_group_get_device_buffers(sl3d_group *g, sl3d_device_buffers *out);

/* the compacted variant: sl3d_run_clouds on every stripe, then the counts come back to the host and every stripe sends
 * exactly its valid points; the root concatenates them in stripe order = the reference's scan order.
 * counts[k] = points of view first_view+k (all stripes).  sl3d_group_get_cloud waits for the transfer. */
int sl3d_group_run_clouds(sl3d_group *g, int first_view, int n_views);
int sl3d_group_gather_clouds(sl3d_group *g, int first_view, int n_views, int64_t *counts);
int sl3d_group_get_cloud(sl3d_group *g, int view, float *xyz, int64_t capacity, int64_t *count);

/* waits for every stripe's stream and every communication stream */
int sl3d_group_synchronize(sl3d_group *g);

#ifdef __cplusplus
}
#endif
#endif /* SL3D_H */
