// sl3d_capi_clouds.cpp -- O1 / N2 / N3: ordered clouds straight from the fused kernel (segmented), their consumers (contiguous copy, host
// downloads, registration), the compaction of a dense result with colour, turntable registration.
#include "sl3d_capi_internal.h"

// ---- compacted clouds straight from the fused kernel ----------------------------------------------------------------
static int ensure_cloud_buffers(sl3d_ctx *x)
{
    // one flag, set at the very end: a set-up that failed half way (out of memory on a later buffer) is retried by the next
    // call instead of being mistaken for a finished one (every step below skips what an earlier attempt already allocated)
    if (x->clouds_ready) return SL3D_OK;
    KParams &P = x->P;
    const size_t mv = (size_t)x->cfg.max_views;
    int rc = SL3D_OK;
    const size_t nb = (P.px_view_stride + 1023) / 1024;
    if (!x->d_clouds) rc = dev_alloc(x, &x->d_clouds, mv * P.px_view_stride * 3);
    if (!rc && !x->d_blk_cnt_all) rc = dev_alloc(x, &x->d_blk_cnt_all, mv * nb);
    if (!rc && !x->d_blk_off_all) rc = dev_alloc(x, &x->d_blk_off_all, mv * nb);
    if (!rc && !x->d_totals) rc = dev_alloc(x, &x->d_totals, mv);
    if (rc) return rc;
    P.n_tiles = fused_tiles(P);
    P.n_segs = 4 * P.n_tiles;
    if (!x->d_seg_counts) rc = dev_alloc(x, &x->d_seg_counts, mv * (size_t)P.n_segs);
    if (!rc && !x->d_seg_offsets) rc = dev_alloc(x, &x->d_seg_offsets, mv * (size_t)P.n_segs);
    if (rc) return rc;
    // a wave of the last tile that owns no row never stores its count: zero once, for good
    HIPCHK(x, hipMemsetAsync(x->d_seg_counts, 0, mv * (size_t)P.n_segs * sizeof(unsigned), x->stream));
    HIPCHK(x, hipMemsetAsync(x->d_seg_offsets, 0, mv * (size_t)P.n_segs * sizeof(unsigned long long), x->stream));
    P.seg_counts = x->d_seg_counts;
    P.seg_offsets = x->d_seg_offsets;
    P.clouds = x->d_clouds;
    // the per-view counts live in pinned HOST memory the scan kernel writes directly (one 8-byte store per view):
    // sl3d_get_cloud_counts then only has to wait for the stream, no device-to-host copy in the launch -> counts path
    if (!x->h_counts) {
        HIPCHK(x, hipHostMalloc((void **)&x->h_counts, mv * sizeof(unsigned long long), hipHostMallocMapped));
        memset(x->h_counts, 0, mv * sizeof(unsigned long long));
    }
    void *mapped = nullptr;
    HIPCHK(x, hipHostGetDevicePointer(&mapped, x->h_counts, 0));
    P.cloud_totals = (unsigned long long *)mapped;
    x->scan_state.assign(mv, 0);
    x->clouds_ready = true;
    return SL3D_OK;
}

// The fused kernel with the compaction of 8/save_point_cloud.cpp:85-104 inside it (k_fused<..., CMODE = 2>: segmented ordered
// clouds): one launch reads every frame byte once and writes the valid map and the compacted points of every view -- no dense xyz
// plane, no second pass over the results -- then one small scan launch turns the segment counts into offsets and totals.
extern "C" int sl3d_run_clouds(sl3d_ctx *x, int first_view, int n_views)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called");
    if (x->keep) return fail(x, SL3D_E_STATE, "sl3d_run_clouds is the timed mode: create the context without SL3D_FLAG_KEEP_STAGES");
    ON_DEVICE_QUIET(x);
    // (a small launch goes beside the one before it, sl3d_ctx.h: launch lanes -- once the cloud buffers exist; its consumers join)
    bool overlap = false;
    if (x->lanes_ok && x->clouds_ready && n_views <= LanePolicy::MAX_VIEWS) rc = small_launch_overlaps(x, first_view, n_views, &overlap);
    else rc = sl3d_lanes_join(x);
    if (rc) return rc;
    rc = ensure_cloud_buffers(x);
    if (rc) return rc;
    rc = run_fused(x, first_view, n_views, false, 2, overlap);
    if (rc) return rc;
    // A launch of a few views (the reference's one scan per call) leaves the scan of the segment counts to whoever consumes the
    // clouds: the gap-closing kernel adds up the counts in front of its segments itself, so there is no scan launch -- 4.8 us + a
    // kernel boundary behind a 26-us kernel -- between the fused kernel and its consumer; a consumer that wants the offsets as an
    // array (sl3d_get_cloud_segments) gets the scan then.  Large launches scan here, as before: one launch for all views.
    if (n_views <= SL3D_SMALL_LAUNCH_VIEWS) {
        for (int v = first_view; v < first_view + n_views; v++) x->scan_state[v] = 1;
        return SL3D_OK;
    }
    for (int v = first_view; v < first_view + n_views; v++) x->scan_state[v] = 0;
    return launched(x, launch_seg_scan(x->P, first_view, n_views, x->stream));
}
SL3D_CATCH(x)

// offsets and totals of views [first_view, first_view + n_views) are (being) computed: k_seg_scan for the views that still lack them
static int ensure_scanned(sl3d_ctx *x, int first_view, int n_views)
{
    for (int v = first_view; v < first_view + n_views;) {
        if (x->scan_state[v] == 0) { v++; continue; }
        int e = v;
        while (e < first_view + n_views && x->scan_state[e] != 0) x->scan_state[e++] = 0;
        const int rc = launched(x, launch_seg_scan(x->P, v, e - v, x->stream));
        if (rc) return rc;
        v = e;
    }
    return SL3D_OK;
}

static int ensure_packed(sl3d_ctx *x)
{
    if (x->d_packed) return SL3D_OK;
    return dev_alloc(x, &x->d_packed, (size_t)x->cfg.max_views * x->P.px_view_stride * 3);
}

// counts (and the device address) of the clouds the last sl3d_run_clouds over these views produced; synchronises
extern "C" int sl3d_get_cloud_counts(sl3d_ctx *x, int first_view, int n_views, const float **device_xyz, size_t *view_stride_points, int64_t *counts)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!counts) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    if (!x->clouds_ready) return fail(x, SL3D_E_STATE, "sl3d_run_clouds has not been called");
    ON_DEVICE(x);
    bool unscanned = false, no_total = false;
    for (int v = first_view; v < first_view + n_views; v++) {
        unscanned |= x->scan_state[v] != 0;
        no_total |= x->scan_state[v] == 1;
    }
    volatile unsigned long long *t = x->h_counts;
    if (device_xyz && unscanned) {
        // the contiguous copy by the gap-closing kernel that scans on entry: it leaves the totals too -- ONE launch, one wait
        rc = ensure_packed(x);
        if (rc) return rc;
        float *dst = x->d_packed + 3 * (size_t)first_view * x->P.px_view_stride;
        rc = launched(x, launch_seg_close_scan(x->P, first_view, n_views, dst, x->P.px_view_stride, ~0ull, x->stream));
        if (rc) return rc;
        SYNC_FOR_CALLER(x);
        for (int v = 0; v < n_views; v++) {
            counts[v] = (int64_t)t[first_view + v];
            if (x->scan_state[first_view + v] == 1) x->scan_state[first_view + v] = 2;
        }
        *device_xyz = dst;
        if (view_stride_points) *view_stride_points = x->P.px_view_stride;
        return SL3D_OK;
    }
    if (no_total && (rc = ensure_scanned(x, first_view, n_views))) return rc;
    // the scan kernel (or a scanning consumer) stored the counts into pinned host memory itself: wait for it, read them
    SYNC_FOR_CALLER(x);
    for (int v = 0; v < n_views; v++) counts[v] = (int64_t)t[first_view + v];
    if (device_xyz) {  // the contiguous copy is made now, by one gap-closing launch over these views
        rc = ensure_packed(x);
        if (rc) return rc;
        float *dst = x->d_packed + 3 * (size_t)first_view * x->P.px_view_stride;
        rc = launched(x, launch_seg_close(x->P, first_view, n_views, dst, x->P.px_view_stride, x->stream));
        if (rc) return rc;
        // the copy is handed to consumers on OTHER streams too (a group's communication stream, a caller's RCCL stream):
        // like the counts, it is complete when this call returns
        SYNC_FOR_CALLER(x);
        *device_xyz = dst;
    }
    if (view_stride_points) *view_stride_points = x->P.px_view_stride;
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_get_cloud_segments(sl3d_ctx *x, int first_view, int n_views, sl3d_cloud_segments *out, int64_t *counts)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!out) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    if (!x->clouds_ready) return fail(x, SL3D_E_STATE, "sl3d_run_clouds has not been called");
    {   // this consumer wants the offsets as an array: the scan runs now if the launch left it out
        ON_DEVICE(x);
        rc = ensure_scanned(x, first_view, n_views);
        if (rc) return rc;
    }
    if (counts) {
        rc = sl3d_get_cloud_counts(x, first_view, n_views, nullptr, nullptr, counts);
        if (rc) return rc;
    }
    const KParams &P = x->P;
    out->xyz = x->d_clouds + 3 * (size_t)first_view * P.px_view_stride;
    out->counts = x->d_seg_counts + (size_t)first_view * P.n_segs;
    out->offsets = (const uint64_t *)(x->d_seg_offsets + (size_t)first_view * P.n_segs);
    out->n_segments = P.n_segs;
    out->segment_points = SL3D_SEG_POINTS;
    out->view_stride_points = P.px_view_stride;
    out->view_stride_segments = (size_t)P.n_segs;
    return SL3D_OK;
}
SL3D_CATCH(x)

// The host copy of the clouds of the last sl3d_run_clouds, back to back (8/save_point_cloud.cpp:85-104 fills a host cloud).
// Segmented clouds + pinned destination: the gap-closing kernel stores straight into the (mapped) host buffer -- the PCIe link is
// the bound either way, so closing the gaps costs nothing; pageable destination: a contiguous device copy goes down by DMA.
extern "C" int sl3d_download_clouds(sl3d_ctx *x, int first_view, int n_views, float *xyz, int64_t capacity, int64_t *counts)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!counts) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    if (!x->clouds_ready) return fail(x, SL3D_E_STATE, "sl3d_run_clouds has not been called");
    if (n_views == 1 && xyz && capacity > 0 && x->scan_state[first_view] != 0) {
        // ONE unscanned view into pinned host memory -- the reference's own consumer (8/save_point_cloud.cpp:85-104 fills a host cloud
        // per scan): the gap-closing kernel scans on entry, stores straight into the mapped host buffer (clamped to its capacity)
        // and leaves the count -- fused kernel, this kernel, one wait; no scan launch, no wait for the count in between
        ON_DEVICE(x);
        void *mapped = nullptr;
        const char *zc = getenv("SL3D_ZEROCOPY");
        if (!(zc && atoi(zc) == 0) && is_pinned_host(xyz) && hipHostGetDevicePointer(&mapped, xyz, 0) == hipSuccess && mapped) {
            rc = launched(x, launch_seg_close_scan(x->P, first_view, 1, (float *)mapped, 0, (unsigned long long)capacity, x->stream));
            if (rc) return rc;
            SYNC_FOR_CALLER(x);
            counts[0] = (int64_t)((volatile unsigned long long *)x->h_counts)[first_view];
            if (x->scan_state[first_view] == 1) x->scan_state[first_view] = 2;
            return SL3D_OK;
        }
        (void)hipGetLastError();
    }
    rc = sl3d_get_cloud_counts(x, first_view, n_views, nullptr, nullptr, counts);
    if (rc || !xyz) return rc;
    ON_DEVICE(x);
    const KParams &P = x->P;
    int64_t total = 0;
    for (int v = 0; v < n_views; v++) total += counts[v];
    void *mapped = nullptr;
    const char *zc = getenv("SL3D_ZEROCOPY");
    const bool zero_copy = total <= capacity && !(zc && atoi(zc) == 0) && is_pinned_host(xyz) &&
                           hipHostGetDevicePointer(&mapped, xyz, 0) == hipSuccess && mapped;
    if (!zero_copy) (void)hipGetLastError();
    if (zero_copy) {
        rc = ensure_scanned(x, first_view, n_views);  // (k_seg_close reads the offsets array)
        if (rc) return rc;
        int64_t off = 0;
        for (int v = 0; v < n_views; v++) {
            if (counts[v] > 0) {
                rc = launched(x, launch_seg_close(P, first_view + v, 1, (float *)mapped + 3 * off, 0, x->stream));
                if (rc) return rc;
            }
            off += counts[v];
        }
    } else {
        const float *dev = nullptr;
        size_t stride = 0;
        rc = sl3d_get_cloud_counts(x, first_view, n_views, &dev, &stride, counts);
        if (rc) return rc;
        int64_t off = 0;
        for (int v = 0; v < n_views && off < capacity; v++) {
            const int64_t n = std::min<int64_t>(counts[v], capacity - off);
            if (n > 0) HIPCHK_DRAIN(x, hipMemcpyAsync(xyz + 3 * off, dev + 3 * (size_t)v * stride, (size_t)n * 12, hipMemcpyDeviceToHost, x->stream));
            off += n;
        }
    }
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_compact(sl3d_ctx *x, int view, const float **device_xyz, int64_t *count)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    if (!count) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    ON_DEVICE(x);
    const bool tex = x->d_texture && view < (int)x->have_texture.size() && x->have_texture[view];
    rc = launched(x, launch_compact(x->P, view, x->d_blk_cnt, x->d_blk_off, x->d_total, x->d_cloud,
                                    tex ? x->d_texture + (size_t)view * x->P.px_view_stride * 3 : nullptr, x->d_cloud_rgb, x->stream));
    if (rc) return rc;
    unsigned long long n = 0;
    HIPCHK(x, hipMemcpyAsync(&n, x->d_total, sizeof n, hipMemcpyDeviceToHost, x->stream));
    SYNC_FOR_CALLER(x);
    *count = (int64_t)n;
    if (device_xyz) *device_xyz = x->d_cloud;
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_get_cloud(sl3d_ctx *x, int view, float *xyz, int64_t capacity, int64_t *count)
try {
    if (!x || !count) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    ON_DEVICE(x);
    // row-major scan, valid pixels only (8/save_point_cloud.cpp:85-104), compacted on the device
    const float *dev = nullptr;
    int rc = sl3d_compact(x, view, &dev, count);
    if (rc) return rc;
    const int64_t n = *count < capacity ? *count : capacity;
    if (xyz && n > 0) {
        HIPCHK(x, hipMemcpyAsync(xyz, dev, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, x->stream));
        SYNC_FOR_CALLER(x);
    }
    return SL3D_OK;
}
SL3D_CATCH(x)

// The compaction of a whole batch of views in three launches and one read-back: what a pipeline that goes from
// device-resident frames to compacted clouds runs after sl3d_run (bench.py reports it as `to_compacted_clouds`).
extern "C" int sl3d_compact_views(sl3d_ctx *x, int first_view, int n_views, const float **device_xyz, size_t *view_stride_points, int64_t *counts)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!counts) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    ON_DEVICE(x);
    const KParams &P = x->P;
    rc = ensure_cloud_buffers(x);
    if (rc) return rc;
    rc = ensure_packed(x);  // (the region sl3d_run_clouds writes is left alone)
    if (rc) return rc;
    rc = launched(x, launch_compact_views(P, first_view, n_views, x->d_blk_cnt_all, x->d_blk_off_all, x->d_totals + first_view,
                                          x->d_packed + 3 * (size_t)first_view * P.px_view_stride, x->stream));
    if (rc) return rc;
    std::vector<unsigned long long> t((size_t)n_views);
    HIPCHK(x, hipMemcpyAsync(t.data(), x->d_totals + first_view, sizeof(unsigned long long) * (size_t)n_views, hipMemcpyDeviceToHost, x->stream));
    SYNC_FOR_CALLER(x);
    for (int v = 0; v < n_views; v++) counts[v] = (int64_t)t[(size_t)v];
    if (device_xyz) *device_xyz = x->d_packed + 3 * (size_t)first_view * P.px_view_stride;
    if (view_stride_points) *view_stride_points = P.px_view_stride;
    return SL3D_OK;
}
SL3D_CATCH(x)

// host copy of the batched compaction: the clouds of the views back to back in xyz (at most `capacity` points in all)
extern "C" int sl3d_get_clouds(sl3d_ctx *x, int first_view, int n_views, float *xyz, int64_t capacity, int64_t *counts)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    ON_DEVICE(x);
    const float *dev = nullptr;
    size_t stride = 0;
    int rc = sl3d_compact_views(x, first_view, n_views, &dev, &stride, counts);
    if (rc) return rc;
    int64_t off = 0;
    for (int v = 0; v < n_views && xyz; v++) {
        const int64_t n = counts[v] < capacity - off ? counts[v] : capacity - off;
        if (n > 0) HIPCHK_DRAIN(x, hipMemcpyAsync(xyz + 3 * off, dev + 3 * (size_t)v * stride, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, x->stream));
        off += n > 0 ? n : 0;
    }
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)

// the colour image save_point_cloud() takes the r,g,b of every valid pixel from (8/save_point_cloud.cpp:46-52: cvLoadImage
// of Point_cloud/texture.bmp, split into blue / green / red planes)
extern "C" int sl3d_set_texture(sl3d_ctx *x, int view, const uint8_t *bgr, size_t stride)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    const KParams &P = x->P;
    if (!bgr || stride < (size_t)P.W * 3) return fail(x, SL3D_E_INVALID_ARG, "texture: null or stride < 3*width");
    ON_DEVICE(x);
    if (!x->d_texture) {
        rc = dev_alloc(x, &x->d_texture, (size_t)x->cfg.max_views * P.px_view_stride * 3);
        if (rc) return rc;
        rc = dev_alloc(x, &x->d_cloud_rgb, P.px_view_stride * 3);
        if (rc) return rc;
        x->have_texture.assign((size_t)x->cfg.max_views, 0);
    }
    SYNC_FOR_CALLER(x);
    HIPCHK(x, hipMemcpy2D(x->d_texture + (size_t)view * P.px_view_stride * 3, (size_t)P.pitch * 3, bgr, stride, (size_t)P.W * 3, (size_t)P.H,
                          hipMemcpyHostToDevice));
    x->have_texture[view] = 1;
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_get_cloud_rgb(sl3d_ctx *x, int view, float *xyz, uint8_t *rgb, int64_t capacity, int64_t *count)
try {
    if (!x || !count) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    if (!x->d_texture || view < 0 || view >= (int)x->have_texture.size() || !x->have_texture[view])
        return fail(x, SL3D_E_INVALID_ARG, "no texture set for this view (sl3d_set_texture)");
    ON_DEVICE(x);
    const float *dev = nullptr;
    int rc = sl3d_compact(x, view, &dev, count);
    if (rc) return rc;
    const int64_t n = *count < capacity ? *count : capacity;
    if (n > 0) {
        if (xyz) HIPCHK(x, hipMemcpyAsync(xyz, dev, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, x->stream));
        if (rgb) HIPCHK_DRAIN(x, hipMemcpyAsync(rgb, x->d_cloud_rgb, (size_t)n * 3, hipMemcpyDeviceToHost, x->stream));
        SYNC_FOR_CALLER(x);
    }
    return SL3D_OK;
}
SL3D_CATCH(x)

// register_point_clouds(), 9/register_point_clouds.cpp:23-155, without the PLY files: the clouds are the
// compacted clouds of the resident views, view k is rotated about Y by theta_k around (tx,ty,tz), theta_0 = 0,
// theta_{k+1} = theta_k + rot_step in float (:145), angles in degrees converted with Pi = 22/7 (:89-93).
extern "C" int sl3d_register_views(sl3d_ctx *x, int first_view, int n_views, float tx, float ty, float tz, float rot_step, float *xyz,
                                   int64_t capacity, int64_t *total)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!total) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    ON_DEVICE(x);
    const KParams &P = x->P;
    if (!x->d_reg) {
        rc = dev_alloc(x, &x->d_reg, (size_t)x->cfg.max_views * P.px_view_stride * 3);
        if (rc) return rc;
    }
    // one batched compaction (three launches, one read-back), then one transform launch per view, no host sync between them
    std::vector<int64_t> counts((size_t)n_views);
    const float *clouds = nullptr;
    size_t stride = 0;
    rc = sl3d_compact_views(x, first_view, n_views, &clouds, &stride, counts.data());
    if (rc) return rc;
    float theta = 0.0f;
    int64_t off = 0;
    for (int k = 0; k < n_views; k++) {
        const int64_t n = counts[(size_t)k];
        // R entries as the reference stores them: double cos/sin of theta*Pi/180.0 (Pi = 22.0/7.0), rounded to float
        const float R4[4] = {(float)cos(theta * 22.0 / 7.0 / 180.0), (float)(-1.0f * sin(theta * 22.0 / 7.0 / 180.0)),
                             (float)sin(theta * 22.0 / 7.0 / 180.0), (float)cos(theta * 22.0 / 7.0 / 180.0)};
        rc = launched(x, launch_register(clouds + 3 * (size_t)k * stride, x->d_reg + 3 * off, (long)n, R4, tx, ty, tz, x->stream));
        if (rc) return rc;
        off += n;
        theta += rot_step;
    }
    SYNC_FOR_CALLER(x);
    *total = off;
    const int64_t m = off < capacity ? off : capacity;
    if (xyz && m > 0) {
        HIPCHK(x, hipMemcpyAsync(xyz, x->d_reg, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, x->stream));
        SYNC_FOR_CALLER(x);
    }
    return SL3D_OK;
}
SL3D_CATCH(x)

// register_point_clouds() on the clouds of the last sl3d_run_clouds: the segments of view k are rotated by theta_k while they are
// concatenated (k_seg_close<REG>), so neither a dense plane nor a separate compaction nor a gap-closing pass is needed.
extern "C" int sl3d_register_clouds(sl3d_ctx *x, int first_view, int n_views, float tx, float ty, float tz, float rot_step, float *xyz,
                                    int64_t capacity, int64_t *total)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!total) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    std::vector<int64_t> counts((size_t)n_views);
    if (!x->clouds_ready) return fail(x, SL3D_E_STATE, "sl3d_run_clouds has not been called");
    {   // (k_seg_close<REG> reads the offsets array: the scan runs now if the launch left it out)
        ON_DEVICE(x);
        rc = ensure_scanned(x, first_view, n_views);
        if (rc) return rc;
    }
    rc = sl3d_get_cloud_counts(x, first_view, n_views, nullptr, nullptr, counts.data());
    if (rc) return rc;
    ON_DEVICE(x);
    const KParams &P = x->P;
    if (!x->d_reg) {
        rc = dev_alloc(x, &x->d_reg, (size_t)x->cfg.max_views * P.px_view_stride * 3);
        if (rc) return rc;
    }
    float theta = 0.0f;
    int64_t off = 0;
    for (int k = 0; k < n_views; k++) {
        const int64_t n = counts[(size_t)k];
        const float R4[4] = {(float)cos(theta * 22.0 / 7.0 / 180.0), (float)(-1.0f * sin(theta * 22.0 / 7.0 / 180.0)),
                             (float)sin(theta * 22.0 / 7.0 / 180.0), (float)cos(theta * 22.0 / 7.0 / 180.0)};
        if (n > 0) {
            rc = launched(x, launch_seg_register(P, first_view + k, x->d_reg + 3 * off, R4, tx, ty, tz, x->stream));
            if (rc) return rc;
        }
        off += n;
        theta += rot_step;
    }
    *total = off;
    const int64_t m = off < capacity ? off : capacity;
    if (xyz && m > 0) HIPCHK(x, hipMemcpyAsync(xyz, x->d_reg, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, x->stream));
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)
