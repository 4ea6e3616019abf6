// sl3d_capi_run.cpp -- compute: the per-stage entry points (parity mode), every launch of the fused kernel (run_fused: which instantiation,
// MASKIN or not), timers, the getters of the stage-boundary planes and results, and the host-buffer pipeline (sl3d_process_views).
#include "sl3d_capi_internal.h"

// ---- compute -------------------------------------------------------------------------------------
int need_keep(sl3d_ctx *x)
{
    if (!x->keep) return fail(x, SL3D_E_STATE, "context was created without SL3D_FLAG_KEEP_STAGES");
    return SL3D_OK;
}

int launched(sl3d_ctx *x, int hip_err)
{
    if (hip_err != 0) return fail(x, SL3D_E_HIP, std::string("kernel launch: ") + hipGetErrorString((hipError_t)hip_err));
    return SL3D_OK;
}

extern "C" int sl3d_compute_wrapped_phase(sl3d_ctx *x, int view, int axis)
try {
    int rc = check_view(x, view);
    if (rc || (rc = need_keep(x))) return rc;
    if (axis != 0 && axis != 1) return fail(x, SL3D_E_INVALID_ARG, "axis must be 0 or 1");
    ON_DEVICE(x);
    return launched(x, launch_wrap(x->P, view, axis, x->stream));
}
SL3D_CATCH(x)

extern "C" int sl3d_unwrap_phase(sl3d_ctx *x, int view, int axis)
try {
    int rc = check_view(x, view);
    if (rc || (rc = need_keep(x))) return rc;
    if (axis != 0 && axis != 1) return fail(x, SL3D_E_INVALID_ARG, "axis must be 0 or 1");
    ON_DEVICE(x);
    return launched(x, launch_unwrap(x->P, view, axis, x->stream));
}
SL3D_CATCH(x)

extern "C" int sl3d_compute_c_p_map(sl3d_ctx *x, int view)
try {
    int rc = check_view(x, view);
    if (rc || (rc = need_keep(x))) return rc;
    ON_DEVICE(x);
    return launched(x, launch_corr(x->P, view, x->stream));
}
SL3D_CATCH(x)

extern "C" int sl3d_triangulate(sl3d_ctx *x, int view)
try {
    int rc = check_view(x, view);
    if (rc || (rc = need_keep(x))) return rc;
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called");
    ON_DEVICE(x);
    return launched(x, launch_tri(x->P, x->C, view, x->stream));
}
SL3D_CATCH(x)

// every fused launch of the library goes through here: the kernel is chosen by what is known about the views' masks NOW, and the
// choice is recorded (sl3d_last_fused_kernel_name reports the instantiation that ran, not a later prediction)
// can the launch over views [first_view, first_view + n_views) evaluate their (deferred) selections itself?  Every view's mask is
// deferred, in one layout, and the launch has a MASKIN instantiation.  (Views known -- by their LAST counts -- to be sparsely selected take
// the gated MASKIN form, whose plane requests wait for the valid bits the launch has just evaluated: launch_fused.)
bool maskin_launch(const sl3d_ctx *x, int first_view, int n_views, bool keep)
{
    if (x->n_pending == 0 || !fused_maskin_available(x->P, x->rig, n_views, keep)) return false;
    const sl3d_ctx::PendingMask &p0 = x->pend[(size_t)first_view];
    for (int v = first_view; v < first_view + n_views; v++) {
        const sl3d_ctx::PendingMask &pm = x->pend[(size_t)v];
        if (!pm.pending || pm.stride != p0.stride || pm.lo != p0.lo || pm.hi != p0.hi) return false;
    }
    return true;
}

// ---- launch lanes (sl3d_lanes.h: the policy; here its plans become events and waits) -------------------------------------------------
// everything the lanes hold comes in front of whatever the context's stream is given next
int lanes_wait(sl3d_ctx *x)
{
    const unsigned busy = x->lp.stream_gets_work();
    for (int l = 0; l < 2; l++) {
        if (!(busy >> l & 1u)) continue;
        // (the lane's event is recorded HERE, once per join, not behind every launch: an event between two kernels of a stream costs
        // the later one ~4 us -- measured on launches tied to one lane)
        HIPCHK(x, hipEventRecord(x->ev_lane[l], x->lane[l]));
        HIPCHK(x, hipStreamWaitEvent(x->stream, x->ev_lane[l], 0));
    }
    return SL3D_OK;
}

// ... and a series of small launches ends here (every entry point but sl3d_run / sl3d_run_clouds: ON_DEVICE)
int sl3d_lanes_join(sl3d_ctx *x)
{
    x->lp.series_ends();
    return lanes_wait(x);
}

// the lane the next small launch over views [first_view, first_view + n_views) goes to, made to wait for what it depends on: everything
// the context's stream was given before (uploads, masks, tables), and the launches that still work on one of these views
static int lane_begin(sl3d_ctx *x, int first_view, int n_views, int *lane)
{
    if (!x->ev_main) {  // (ev_main is created last: a creation that failed half way is taken up where it stopped)
        // Two launches overlap only if their streams sit on different HARDWARE queues, and the runtime spreads a process's streams over
        // a handful of them (GPU_MAX_HW_QUEUES, 4 by default: with one queue the series ran at 25.9 us per launch instead of 21.6, and in
        // a process with many streams -- bench.py's -- the two lanes did land on one).  Streams of another priority have their own pool
        // of queues, which nothing else in a process normally touches: both lanes a step above the default, created back to back, get
        // two different queues of that pool.  (One lane at the default and one above it: always different queues too, but the pair is
        // lopsided -- 23.5 us per launch.)
        int least = 0, greatest = 0;
        HIPCHK(x, hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (int l = 0; l < 2; l++) {
            if (!x->lane[l]) HIPCHK(x, hipStreamCreateWithPriority(&x->lane[l], hipStreamNonBlocking, greatest < 0 ? greatest : 0));
            if (!x->ev_lane[l]) HIPCHK(x, hipEventCreateWithFlags(&x->ev_lane[l], hipEventDisableTiming));
        }
        x->lp.reset(x->cfg.max_views);
        HIPCHK(x, hipEventCreateWithFlags(&x->ev_main, hipEventDisableTiming));
    }
    const LanePlan p = x->lp.begin(first_view, n_views);
    if (p.wait_main) {
        HIPCHK(x, hipEventRecord(x->ev_main, x->stream));
        HIPCHK(x, hipStreamWaitEvent(x->lane[p.lane], x->ev_main, 0));
    }
    if (p.wait_other) {  // views of both lanes: behind everything the other lane holds
        HIPCHK(x, hipEventRecord(x->ev_lane[p.lane ^ 1], x->lane[p.lane ^ 1]));
        HIPCHK(x, hipStreamWaitEvent(x->lane[p.lane], x->ev_lane[p.lane ^ 1], 0));
    }
    *lane = p.lane;
    return SL3D_OK;
}

// a small launch of a context with lanes: does it go beside the one before it?  If not it goes to the stream itself, behind the lanes
// (the series goes on).  Shared by sl3d_run and sl3d_run_clouds.
int small_launch_overlaps(sl3d_ctx *x, int first_view, int n_views, bool *overlap)
{
    *overlap = x->lp.small_launch_pays(first_view, n_views);
    return *overlap ? SL3D_OK : lanes_wait(x);
}

// may_overlap: the caller (sl3d_run, sl3d_run_clouds) took the QUIET form of ON_DEVICE for a small launch on a context with lanes
int run_fused(sl3d_ctx *x, int first_view, int n_views, bool keep, int cmode, bool may_overlap)
{
    const bool prefer_gated = sparse_views(x, first_view, n_views);
    const bool maskin = maskin_launch(x, first_view, n_views, keep);
    x->last_fused.n_views = n_views;
    x->last_fused.cmode = cmode;
    x->last_fused.keep = keep;
    x->last_fused.prefer_gated = prefer_gated;
    x->last_fused.maskin = maskin;
    hipStream_t st = x->stream;
    int lane = -1;
    struct Count {  // (where the launch went, whichever return below is taken)
        sl3d_ctx *x;
        const int &lane;
        ~Count() { (lane < 0 ? x->launches_on_stream : x->launches_on_lanes)++; }
    } count{x, lane};
    if (!maskin) {
        const unsigned epoch = x->lp.main_epoch;
        int rc = flush_masks(x, first_view, n_views);  // (k_mask_prepare on the context's stream: joins the lanes itself)
        if (rc) return rc;
        if (x->lp.main_epoch != epoch) may_overlap = false;  // (... and the launch stays behind it on that stream: no hand-over)
        if (may_overlap) {
            if ((rc = lane_begin(x, first_view, n_views, &lane))) return rc;
            st = x->lane[lane];
        }
        rc = launched(x, launch_fused(x->P, x->d_cal, x->rig, first_view, n_views, keep, cmode, st, prefer_gated));
        if (!rc && lane >= 0) x->lp.end(lane, first_view, n_views);
        return rc;
    }
    if (may_overlap) {
        const int rc = lane_begin(x, first_view, n_views, &lane);
        if (rc) return rc;
        st = x->lane[lane];
    }
    MaskIn mi;
    MaskSrc S;
    (void)mask_region(x->P, S);
    memset(&mi, 0, sizeof mi);
    const unsigned seq = ++x->mask_seq;
    for (int k = 0; k < n_views; k++) {
        sl3d_ctx::PendingMask &pm = x->pend[(size_t)(first_view + k)];
        mi.origin[k] = pm.origin;
        mi.stride = pm.stride;
        mi.lo = pm.lo;
        mi.hi = pm.hi;
        pm.pending = false;
        x->n_pending--;
        x->quad_seq[(size_t)(first_view + k)] = seq;  // the launch leaves the views' counts under a new sequence number
        x->quad_src[(size_t)(first_view + k)] = first_view + k;
        x->quad_kind[(size_t)(first_view + k)] = 1;
    }
    mi.bx0 = S.bx0; mi.bx1 = S.bx1; mi.r0 = S.r0; mi.r1 = S.r1;
    mi.part = x->d_mi_part;
    mi.part_stride = x->mi_part_stride;
    mi.seq = seq & 0xffffffu;
    const int rc = launched(x, launch_fused(x->P, x->d_cal, x->rig, first_view, n_views, keep, cmode, st, prefer_gated, &mi));
    if (!rc && lane >= 0) x->lp.end(lane, first_view, n_views);
    return rc;
}

extern "C" int sl3d_launch_counts(sl3d_ctx *x, int64_t *on_stream, int64_t *on_lanes)
{
    if (!x) return SL3D_E_INVALID_ARG;
    if (on_stream) *on_stream = x->launches_on_stream;
    if (on_lanes) *on_lanes = x->launches_on_lanes;
    return SL3D_OK;
}

extern "C" int sl3d_last_fused_kernel_name(sl3d_ctx *x, char *buf, size_t capacity)
try {
    if (!x || !buf || capacity == 0) return fail(x, SL3D_E_INVALID_ARG, "last_fused_kernel_name: null argument");
    if (x->last_fused.n_views < 1) return fail(x, SL3D_E_STATE, "no fused launch has been made on this context");
    const int n = fused_kernel_name(x->P, x->rig, x->last_fused.n_views, x->last_fused.keep, x->last_fused.cmode, buf, capacity, x->last_fused.prefer_gated,
                                    x->last_fused.maskin);
    return n > 0 && (size_t)n < capacity ? SL3D_OK : fail(x, SL3D_E_INVALID_ARG, "last_fused_kernel_name: buffer too small");
}
SL3D_CATCH(x)

extern "C" int sl3d_run(sl3d_ctx *x, int first_view, int n_views)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called");
    ON_DEVICE_QUIET(x);
    // a small launch of a long series goes beside the one before it, on a lane (sl3d_lanes.h); anything else behind everything, on the stream
    bool overlap = false;
    if (x->lanes_ok && n_views <= LanePolicy::MAX_VIEWS) rc = small_launch_overlaps(x, first_view, n_views, &overlap);
    else rc = sl3d_lanes_join(x);  // (a large launch ends a series of small ones)
    if (rc) return rc;
    return run_fused(x, first_view, n_views, x->keep, 0, overlap);
}
SL3D_CATCH(x)

// the k_fused instantiation sl3d_run / sl3d_run_clouds launches for a batch of n_views views of this context, as rocprofv3 spells it
extern "C" int sl3d_fused_kernel_name(sl3d_ctx *x, int n_views, int clouds, char *buf, size_t capacity)
try {
    if (!x || !buf || capacity == 0 || n_views < 1) return fail(x, SL3D_E_INVALID_ARG, "fused_kernel_name: null argument");
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called (the rig class is part of the name)");
    const bool fits = n_views <= x->cfg.max_views, gated = fits && sparse_views(x, 0, n_views);
    const int n = fused_kernel_name(x->P, x->rig, n_views, x->keep, clouds ? 2 : 0, buf, capacity, gated, fits && maskin_launch(x, 0, n_views, x->keep));
    return n > 0 && (size_t)n < capacity ? SL3D_OK : fail(x, SL3D_E_INVALID_ARG, "fused_kernel_name: buffer too small");
}
SL3D_CATCH(x)

extern "C" int sl3d_camera_table_bytes_per_pixel(sl3d_ctx *x, int n_views)
try {
    if (!x || n_views < 1) return fail(x, SL3D_E_INVALID_ARG, "camera_table_bytes_per_pixel: null context or no views");
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called");
    if (!x->P.cam_tab) return 0;
    if (x->P.cam_tab_kind == 2) return 16;
    (void)n_views;
    return 8;
}
SL3D_CATCH(x)

extern "C" int sl3d_run_timed(sl3d_ctx *x, int first_view, int n_views, float *ms)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called");
    ON_DEVICE(x);
    HIPCHK(x, hipEventRecord(x->ev0, x->stream));
    rc = run_fused(x, first_view, n_views, x->keep, 0);
    if (rc) return rc;
    HIPCHK(x, hipEventRecord(x->ev1, x->stream));
    HIPCHK(x, hipEventSynchronize(x->ev1));
    float t = 0;
    HIPCHK(x, hipEventElapsedTime(&t, x->ev0, x->ev1));
    if (ms) *ms = t;
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_timer_start(sl3d_ctx *x)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    ON_DEVICE(x);
    HIPCHK(x, hipEventRecord(x->ev0, x->stream));
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_timer_stop(sl3d_ctx *x, float *ms)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    ON_DEVICE(x);
    HIPCHK(x, hipEventRecord(x->ev1, x->stream));
    HIPCHK(x, hipEventSynchronize(x->ev1));
    float t = 0;
    HIPCHK(x, hipEventElapsedTime(&t, x->ev0, x->ev1));
    if (ms) *ms = t;
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_synchronize(sl3d_ctx *x)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    ON_DEVICE(x);
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)

// ---- getters -------------------------------------------------------------------------------------
template <typename T>
static int get_plane(sl3d_ctx *x, int view, const T *dev_base, int comps, T *out, size_t out_stride_elems)
{
    int rc = check_view(x, view);
    if (rc) return rc;
    if (!out) return fail(x, SL3D_E_INVALID_ARG, "null output");
    if (!dev_base) return fail(x, SL3D_E_STATE, "plane not available (SL3D_FLAG_KEEP_STAGES not set?)");
    const KParams &P = x->P;
    if (out_stride_elems < (size_t)P.W * comps) return fail(x, SL3D_E_INVALID_ARG, "output stride too small");
    ON_DEVICE(x);
    const T *src = dev_base + (size_t)view * P.px_view_stride * comps;
    HIPCHK(x, hipMemcpy2DAsync(out, out_stride_elems * sizeof(T), src, (size_t)P.pitch * comps * sizeof(T), (size_t)P.W * comps * sizeof(T),
                               P.H, hipMemcpyDeviceToHost, x->stream));
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}

extern "C" int sl3d_get_valid_map(sl3d_ctx *x, int view, int which, uint8_t *out, size_t stride)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    const uint8_t *src = which == SL3D_VALID_MERGED ? x->P.valid : (which == 0 || which == 1) ? x->P.valid_axis[which] : nullptr;
    if (which < 0 || which > 2) return fail(x, SL3D_E_INVALID_ARG, "which must be 0, 1 or 2");
    return get_plane<uint8_t>(x, view, src, 1, out, stride);
}
SL3D_CATCH(x)

extern "C" int sl3d_get_wrapped_phase(sl3d_ctx *x, int view, int axis, float *out, size_t stride)
try {
    if (!x || (axis != 0 && axis != 1)) return fail(x, SL3D_E_INVALID_ARG, "axis must be 0 or 1");
    return get_plane<float>(x, view, x->P.wrapped[axis], 1, out, stride);
}
SL3D_CATCH(x)

extern "C" int sl3d_get_unwrapped_phase(sl3d_ctx *x, int view, int axis, float *out, size_t stride)
try {
    if (!x || (axis != 0 && axis != 1)) return fail(x, SL3D_E_INVALID_ARG, "axis must be 0 or 1");
    return get_plane<float>(x, view, x->P.unwrapped[axis], 1, out, stride);
}
SL3D_CATCH(x)

extern "C" int sl3d_get_code(sl3d_ctx *x, int view, int axis, int32_t *out, size_t stride)
try {
    if (!x || (axis != 0 && axis != 1)) return fail(x, SL3D_E_INVALID_ARG, "axis must be 0 or 1");
    return get_plane<int32_t>(x, view, x->P.code[axis], 1, out, stride);
}
SL3D_CATCH(x)

extern "C" int sl3d_get_debug_image(sl3d_ctx *x, int view, int stage, int axis, uint8_t *out, size_t stride)
try {
    if (!x || (axis != 0 && axis != 1) || (stage != 3 && stage != 4)) return fail(x, SL3D_E_INVALID_ARG, "stage must be 3 or 4, axis 0 or 1");
    return get_plane<uint8_t>(x, view, stage == 3 ? x->P.dbg3[axis] : x->P.dbg4[axis], 1, out, stride);
}
SL3D_CATCH(x)

extern "C" int sl3d_get_c_p_map(sl3d_ctx *x, int view, int64_t *out)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    return get_plane<int64_t>(x, view, (const int64_t *)x->P.cpmap, 2, out, (size_t)x->P.W * 2);
}
SL3D_CATCH(x)

extern "C" int sl3d_get_intersection_points(sl3d_ctx *x, int view, double *out)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    return get_plane<double>(x, view, x->P.ipoints, 3, out, (size_t)x->P.W * 3);
}
SL3D_CATCH(x)

extern "C" int sl3d_get_points(sl3d_ctx *x, int view, float *xyz, uint8_t *valid)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    int rc = SL3D_OK;
    if (xyz) rc = get_plane<float>(x, view, x->P.points, 3, xyz, (size_t)x->P.W * 3);
    if (rc == SL3D_OK && valid) rc = get_plane<uint8_t>(x, view, x->P.valid, 1, valid, (size_t)x->P.W);
    return rc;
}
SL3D_CATCH(x)

// ---- host-buffer pipeline -------------------------------------------------------------------------------------------
// Pinned host memory for frames and results: with it the uploads and downloads of sl3d_process_views are true asynchronous
// DMA (pageable memory still works, but every copy is then staged and serialised by the runtime).
extern "C" void *sl3d_host_alloc(size_t bytes)
try {
    void *p = nullptr;
    // portable + mapped, explicitly: the buffers of a group are read / written by EVERY GPU of the group (per-stripe uploads and
    // downloads over each GPU's own PCIe link), and the zero-copy cloud download stores into them from a kernel
    return hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess ? p : nullptr;
}
SL3D_CATCH_RETURN(nullptr)
extern "C" void sl3d_host_free(void *p)
try {
    if (p) (void)hipHostFree(p);
}
SL3D_CATCH_VOID

// A batch of views that live in HOST memory, through the view slots of the context as a three-stage pipeline on three
// HIP streams: upload of view k+1 (46 plane copies), fused kernel of view k, download of the xyz / valid planes of view
// k-1 overlap; events hand a slot from stage to stage.  What the C ABI sustains when the boundary hands over host buffers
// is then the slowest stage (the upload: PCIe), not the sum of the three.  The mask of every slot must have been set.
//   planes: n_views * planes_per_view pointers, view-major, plane order as in sl3d_device_buffers; `stride` bytes per row
//   xyz:    n_views dense [height][width][3] float images (may be NULL);  valid: n_views [height][width] bytes (may be NULL)
// The pipeline itself, ENQUEUED only (no host wait when the buffers are pinned): xyz / valid of view v go to
// xyz + v*xyz_view_stride (floats) / valid + v*valid_view_stride (bytes) with `out_width` pixels per destination row -- a
// whole-frame context passes its own width and W*H strides, a row stripe of a group passes the frame's.  sl3d_process_views_wait
// drains the three streams.  (Shared with sl3d_group_process_views: every stripe's pipeline is enqueued before any is waited for,
// so the GPUs -- and their PCIe links -- work concurrently behind one host thread.)
int sl3d_process_views_enqueue(sl3d_ctx *x, int n_views, const uint8_t *const *planes, size_t stride, float *xyz, size_t xyz_view_stride, uint8_t *valid,
                               size_t valid_view_stride, size_t out_width)
{
    if (!x || n_views < 1 || !planes) return fail(x, SL3D_E_INVALID_ARG, "process_views: null argument");
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "process_views before set_calibration");
    const KParams &P = x->P;
    if (stride < (size_t)P.W) return fail(x, SL3D_E_INVALID_ARG, "process_views: stride < width");
    ON_DEVICE(x);
    const int S = x->cfg.max_views;  // slots
    if (!x->s_h2d) {
        HIPCHK(x, hipStreamCreateWithFlags(&x->s_h2d, hipStreamNonBlocking));
        HIPCHK(x, hipStreamCreateWithFlags(&x->s_d2h, hipStreamNonBlocking));
        x->ev_up.resize((size_t)S); x->ev_done.resize((size_t)S); x->ev_down.resize((size_t)S);
        for (int i = 0; i < S; i++) {
            HIPCHK(x, hipEventCreateWithFlags(&x->ev_up[(size_t)i], hipEventDisableTiming));
            HIPCHK(x, hipEventCreateWithFlags(&x->ev_done[(size_t)i], hipEventDisableTiming));
            HIPCHK(x, hipEventCreateWithFlags(&x->ev_down[(size_t)i], hipEventDisableTiming));
        }
    }
    HIPCHK(x, hipStreamSynchronize(x->stream));  // earlier work on the context's own stream is done before the slots are reused
    const int ppv = P.planes_per_view;
    // every pointer is checked BEFORE anything is enqueued: an error return must never leave copies running against
    // buffers the caller is about to free
    for (size_t i = 0; i < (size_t)n_views * (size_t)ppv; i++)
        if (!planes[i]) return fail(x, SL3D_E_INVALID_ARG, "process_views: null plane");
    for (int v = 0; v < n_views; v++) {
        const int slot = v % S;
        // the slot's previous occupant must have been computed (frames free) and downloaded (results free)
        if (v >= S) {
            HIPCHK(x, hipStreamWaitEvent(x->s_h2d, x->ev_done[(size_t)slot], 0));
            HIPCHK(x, hipStreamWaitEvent(x->stream, x->ev_down[(size_t)slot], 0));
        }
        // a view whose planes are back to back in host memory, in the device's own pitch, goes up as ONE copy
        bool contiguous = stride == (size_t)P.pitch && (size_t)P.W == (size_t)P.pitch;
        for (int p = 1; p < ppv; p++)
            if (planes[(size_t)v * ppv + p] != planes[(size_t)v * ppv + p - 1] + P.plane_stride) contiguous = false;
        if (contiguous) {
            HIPCHK(x, hipMemcpyAsync(x->d_frames + (size_t)slot * P.view_stride, planes[(size_t)v * ppv], P.view_stride, hipMemcpyHostToDevice, x->s_h2d));
        } else {
            for (int p = 0; p < ppv; p++)
                HIPCHK(x, hipMemcpy2DAsync(x->d_frames + (size_t)slot * P.view_stride + (size_t)p * P.plane_stride, P.pitch,
                                           planes[(size_t)v * ppv + p], stride, P.W, P.H, hipMemcpyHostToDevice, x->s_h2d));
        }
        HIPCHK(x, hipEventRecord(x->ev_up[(size_t)slot], x->s_h2d));
        HIPCHK(x, hipStreamWaitEvent(x->stream, x->ev_up[(size_t)slot], 0));
        const int rc = run_fused(x, slot, 1, x->keep, 0);
        if (rc) return rc;
        HIPCHK(x, hipEventRecord(x->ev_done[(size_t)slot], x->stream));
        HIPCHK(x, hipStreamWaitEvent(x->s_d2h, x->ev_done[(size_t)slot], 0));
        if (xyz)
            HIPCHK(x, hipMemcpy2DAsync(xyz + (size_t)v * xyz_view_stride, out_width * 12, P.points + (size_t)slot * P.px_view_stride * 3,
                                       (size_t)P.pitch * 12, (size_t)P.W * 12, P.H, hipMemcpyDeviceToHost, x->s_d2h));
        if (valid)
            HIPCHK(x, hipMemcpy2DAsync(valid + (size_t)v * valid_view_stride, out_width, P.valid + (size_t)slot * P.px_view_stride, P.pitch, P.W, P.H,
                                       hipMemcpyDeviceToHost, x->s_d2h));
        HIPCHK(x, hipEventRecord(x->ev_down[(size_t)slot], x->s_d2h));
    }
    return SL3D_OK;
}

// drains the three streams of the pipeline; returns the first HIP error met
int sl3d_process_views_wait(sl3d_ctx *x)
{
    if (!x) return SL3D_E_INVALID_ARG;
    if (!x->s_h2d) return SL3D_OK;
    ON_DEVICE(x);
    const hipError_t e1 = hipStreamSynchronize(x->s_h2d), e2 = hipStreamSynchronize(x->stream), e3 = hipStreamSynchronize(x->s_d2h);
    HIPCHK(x, e1);
    HIPCHK(x, e2);
    HIPCHK(x, e3);
    return SL3D_OK;
}

extern "C" int sl3d_process_views(sl3d_ctx *x, int n_views, const uint8_t *const *planes, size_t stride, float *xyz, uint8_t *valid)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    const size_t px = (size_t)x->P.W * x->P.H;
    const int rc = sl3d_process_views_enqueue(x, n_views, planes, stride, xyz, px * 3, valid, px, (size_t)x->P.W);
    // success or not, nothing may still be running against the caller's buffers when this returns, and the three streams are
    // left drained for the next call
    const std::string first_err = x->err;
    const int rc2 = sl3d_process_views_wait(x);
    if (rc) {
        x->err = first_err;
        return rc;
    }
    return rc2;
}
SL3D_CATCH(x)
