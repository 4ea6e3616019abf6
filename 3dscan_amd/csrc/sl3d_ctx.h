// sl3d_ctx.h -- the context behind the opaque sl3d_ctx handle, shared by the C-ABI translation units
// (sl3d_capi_*.cpp: single-GPU entry points; sl3d_group.cpp: row-stripe groups over several GPUs).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/sl3d.h"
#include "sl3d_internal.h"
#include "sl3d_lanes.h"

using namespace sl3d;

struct sl3d_ctx {
    sl3d_config cfg{};
    KParams P{};
    DevCal C{};
    SynthParams S{};  // extrinsics kept for the synthetic-capture generator
    bool have_cal = false;
    int rig = 0;
    bool keep = false;
    bool own_stream = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Launch lanes (sl3d_lanes.h: the policy; sl3d_capi_run.cpp turns its plans into events and waits): consecutive small launches of a
    // LONG series run on two internal streams in turn, the tail of one under the ramp of the next.  Every other entry point first makes
    // the context's stream wait for both lanes (ON_DEVICE), so the one-stream ordering the ABI promises holds for everything a caller
    // can observe.  Only on a stream the context created itself.
    bool lanes_ok = false;
    hipStream_t lane[2] = {nullptr, nullptr};
    hipEvent_t ev_lane[2] = {nullptr, nullptr}, ev_main = nullptr;
    LanePolicy lp;
    long long launches_on_stream = 0, launches_on_lanes = 0;  // sl3d_launch_counts
    std::vector<void *> allocs;
    std::string err;
    uint8_t *d_frames = nullptr, *d_mask = nullptr, *d_valid = nullptr, *d_band = nullptr;
    uint8_t *d_mask_raw = nullptr;            // sl3d_set_mask(s): the caller's bytes (window + halo) of mask_raw_slots views, staged for
    int mask_raw_slots = 0;                   // k_mask_prepare; zero outside the copied region (the kernel relies on that)
    // quads with a valid pixel, per view: every block of k_mask_prepare stores {seq, count} into this host array (mapped into the
    // device: d_quad_part is the same memory as the kernels address it); the host adds a view's blocks up on demand and treats the
    // view as unknown until every block carries the sequence number of the view's last preparation (quads_known, sl3d_capi_inputs.cpp)
    volatile unsigned long long *h_quad_part = nullptr;
    unsigned long long *d_quad_part = nullptr;
    int quad_blocks = 0;                      // blocks per view
    std::vector<unsigned> quad_seq;           // [max_views] sequence number of the view's last preparation (0 = never set)
    std::vector<int> quad_src;                // [max_views] the view whose blocks hold this view's count (sl3d_copy_view duplicates masks)
    mutable std::vector<unsigned> quad_sum_seq, quad_sum;  // [max_views] the sum once it was complete, and the preparation it belongs to
    mutable std::vector<unsigned> quad_last;               // [max_views] the last sum that WAS complete, of whichever preparation (~0u: none yet):
                                                           // what routes a launch while the current count is still on its way (sparse_views)
    unsigned mask_seq = 0;
    // Deferred masks (sl3d_set_mask(s) of at most SL3D_SMALL_LAUNCH_VIEWS views on a timed context, unless SL3D_FLAG_EAGER_MASK): the
    // view's selection has been handed over but not prepared -- the next small launch over such views evaluates it inside the fused
    // kernel (a MASKIN launch, sl3d_fused.h), every other consumer of the view's planes prepares it first (flush_masks, sl3d_capi_inputs.cpp).
    struct PendingMask {
        bool pending = false;
        bool ours = false;     // the source is the context's staging plane (else the CALLER's device memory: prepared at the next synchronising call at the latest)
        uintptr_t origin = 0;  // MaskSrc::origin of this view
        size_t stride = 0;
        int lo = 0, hi = 0;    // plane bytes of a row that may be read (MaskIn)
    };
    std::vector<PendingMask> pend;            // [max_views]
    int n_pending = 0;
    bool eager_mask = false;
    volatile unsigned *h_mi_part = nullptr;   // MASKIN launches: per view and wave {seq << 8 | quads with a valid pixel} (mapped host memory)
    unsigned *d_mi_part = nullptr;
    unsigned mi_part_stride = 0, mi_part_words = 0;
    std::vector<uint8_t> quad_kind;           // [max_views] whose words hold the view's count: 0 = k_mask_prepare's blocks, 1 = a MASKIN launch's waves
    // the fused launch made last on this context (sl3d_last_fused_kernel_name: the instantiation that RAN, not a prediction)
    struct { int n_views = 0, cmode = 0; bool keep = false, prefer_gated = false, maskin = false; } last_fused;
    float *d_points = nullptr;
    unsigned *d_blk_cnt = nullptr;            // compaction scratch: per-1024-pixel block counts,
    unsigned long long *d_blk_off = nullptr;  // their exclusive scan, and the total
    unsigned long long *d_total = nullptr;
    float *d_cloud = nullptr;                 // compacted cloud of one view (capacity = window pixels)
    float *d_reg = nullptr;                   // registered clouds of all views (allocated on first use)
    uint8_t *d_raw = nullptr;  // sl3d_set_frames_raw: the raw planes of one axis + the camera's undistortion map
    bool raw_map_valid = false;
    double Kc_raw[9] = {0}, dc_raw[5] = {0};  // the camera intrinsics the raw path undistorts with (set_calibration)
    uint8_t *d_und = nullptr;  // cvUndistort2 scratch: maps, source image, result (grown on demand)
    size_t und_bytes = 0;
    double und_key[16] = {0};  // K, dist, width, height of the map held in d_und (the 46 frames of a view share one map)
    bool und_map_valid = false;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;   // sl3d_process_views: upload and download run beside the compute stream
    std::vector<hipEvent_t> ev_up, ev_done, ev_down;  // per view slot: frames landed / kernel finished / results copied out
    float *d_clouds = nullptr;                // batched compaction: one region of px_view_stride points per view (first use)
    unsigned *d_blk_cnt_all = nullptr;
    unsigned long long *d_blk_off_all = nullptr, *d_totals = nullptr;
    unsigned *d_seg_counts = nullptr;         // sl3d_run_clouds (segmented clouds): [view][n_segs] counts, their exclusive scan,
    unsigned long long *d_seg_offsets = nullptr;
    float *d_packed = nullptr;                // and the contiguous copy made on demand (also the output of sl3d_compact_views)
    // per view, after sl3d_run_clouds: 0 = k_seg_scan has run (offsets and total valid), 1 = not scanned -- a launch of a few views
    // leaves the scan to the consumer (k_seg_close<.., SCAN>), 2 = a scanning consumer has left the total, the offsets are still unset
    std::vector<uint8_t> scan_state;
    bool clouds_ready = false;                // ensure_cloud_buffers ran to its end: every pointer sl3d_run_clouds needs is set
    unsigned long long *h_counts = nullptr;   // pinned + mapped: the per-view counts k_seg_scan stores, sl3d_get_cloud_counts reads
    uint8_t *d_texture = nullptr;             // [view][row][pitch][3] BGR texture of save_point_cloud (allocated by sl3d_set_texture)
    uint8_t *d_cloud_rgb = nullptr;           // r,g,b of the compacted cloud of one view
    std::vector<char> have_texture;
    double *d_cam_tab = nullptr;              // timed mode: camera-side T1 table of the window (sl3d_set_calibration)
    size_t cam_tab_doubles = 0;
    float2 *d_proj_disp = nullptr;            // RIG 2: projector undistortion table (allocated when a distorted projector is set)
    RadEntry *d_proj_rad = nullptr;           // RIG 3: the projector's radial table, SL3D_RAD_COPIES copies
    uint8_t *d_pattern = nullptr, *d_profile = nullptr;  // projector pattern image + its 1-D profile (allocated on first use)
    size_t pattern_pitch = 0;
    uint8_t *d_colrow = nullptr;  // staging of one global in the reference's [col][row] layout (sl3d_get_global_colrow), and of a
    size_t colrow_bytes = 0;      // [col][row] selection mask on its way in (sl3d_set_mask_colrow)
    DevCal *d_cal = nullptr;  // device copy of C for the fused kernel (read through scalar loads)
    size_t mask_rows = 0;
    size_t dbg_words = 0;     // measurement builds (-DSL3D_TRACE): size of the phase-stamp buffer KParams::dbg
};

// records the message on the context (or, without one, as the thread's creation error) and returns `code`
__attribute__((visibility("hidden"))) int sl3d_fail(sl3d_ctx *c, int code, const std::string &msg);
#define fail sl3d_fail
// The exception barrier of the C ABI (no exceptions across the boundary).  Every extern "C" entry point that can allocate
// (new, std::vector, std::string ...) is a function-try-block whose handler ends in one of these: the exception in flight is
// classified (std::bad_alloc -> SL3D_E_NOMEM, anything else -> SL3D_E_INTERNAL), its text becomes the context's last error if that
// still can be stored, and the status is returned.  Nothing in here throws.
__attribute__((visibility("hidden"))) int sl3d_caught(sl3d_ctx *c, std::string *err_of_other_owner = nullptr) noexcept;
#define SL3D_CATCH(ctx) catch (...) { return sl3d_caught(ctx); }
#define SL3D_CATCH_RETURN(value) catch (...) { (void)sl3d_caught(nullptr); return value; }
#define SL3D_CATCH_VOID catch (...) { (void)sl3d_caught(nullptr); }
// sl3d_process_views in two halves (sl3d_capi_run.cpp), shared with sl3d_group_process_views
__attribute__((visibility("hidden"))) int sl3d_process_views_enqueue(sl3d_ctx *x, int n_views, const uint8_t *const *planes, size_t stride, float *xyz,
                                                                     size_t xyz_view_stride, uint8_t *valid, size_t valid_view_stride, size_t out_width);
__attribute__((visibility("hidden"))) int sl3d_process_views_wait(sl3d_ctx *x);

#define HIPCHK(c, call)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail((c), SL3D_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

// the same for asynchronous copies that touch CALLER memory in a loop: an error return first drains the stream, so that no copy
// enqueued earlier in the call is still running against buffers the caller is about to free
#define HIPCHK_DRAIN(c, call)                                                                       \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (void)hipStreamSynchronize((c)->stream);                                                \
            return fail((c), SL3D_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
        }                                                                                           \
    } while (0)

// Every entry point runs on the context's device and hands the caller's current device back on return: the caller may be a
// process that holds contexts on several GPUs, or a torch process whose current device differs.
struct DeviceGuard {
    int prev = -1, dev;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceGuard()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ON_DEVICE_QUIET(x)                 \
    DeviceGuard dev_guard_((x)->cfg.device); \
    HIPCHK((x), dev_guard_.err)
// ... and whatever the call enqueues on the context's stream comes behind the launches the lanes hold (sl3d_run / sl3d_run_clouds
// and the hand-over of a device-resident deferred mask are the entry points that take the QUIET form and join only where they must)
__attribute__((visibility("hidden"))) int sl3d_lanes_join(sl3d_ctx *x);
#define ON_DEVICE(x)                                  \
    ON_DEVICE_QUIET(x);                               \
    do {                                              \
        const int join_rc_ = sl3d_lanes_join(x);      \
        if (join_rc_) return join_rc_;                \
    } while (0)

