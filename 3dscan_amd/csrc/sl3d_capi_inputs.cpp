// sl3d_capi_inputs.cpp -- what goes INTO a context: selection masks (H0 / S3b / S3d: prepared by k_mask_prepare when they are set, or deferred to
// the launch that consumes them -- a MASKIN launch of the fused kernel) and frames, in row-major planes or the reference's own [col][row] layouts.
#include "sl3d_capi_internal.h"


// Quads of `view` that hold a valid pixel, as k_mask_prepare counted them: the sum of the view's per-block words, known only once
// every block carries the sequence number of the view's last preparation (the blocks store straight into this host array: no copy
// behind the kernel, no wait here; a word of an earlier preparation cannot pass for a current one).
bool quads_known(const sl3d_ctx *x, int view, unsigned *quads)
{
    const unsigned seq = x->quad_seq[view];
    if (seq == 0u) return false;  // no mask was ever set
    if (x->quad_sum_seq[view] == seq) {
        *quads = x->quad_sum[view];
        return true;
    }
    unsigned sum = 0u;
    if (x->quad_kind[view] == 1) {  // the view's last preparation was a MASKIN launch: one word per wave that owns pixels
        const volatile unsigned *p = x->h_mi_part + (size_t)x->quad_src[view] * x->mi_part_stride;
        for (unsigned i = 0; i < x->mi_part_words; i++) {
            const unsigned w = p[i];
            if ((w >> 8) != (seq & 0xffffffu)) return false;
            sum += w & 0xffu;
        }
    } else {
        const volatile unsigned long long *p = x->h_quad_part + (size_t)x->quad_src[view] * x->quad_blocks;
        for (int b = 0; b < x->quad_blocks; b++) {
            const unsigned long long w = p[b];
            if ((unsigned)(w >> 32) != seq) return false;
            sum += (unsigned)w;
        }
    }
    x->quad_sum_seq[view] = seq;
    x->quad_sum[view] = sum;
    x->quad_last[view] = sum;
    *quads = sum;
    return true;
}

// true if every view of [first, first + n) is known to be sparsely selected (fewer than 65 % of its quads hold a valid pixel): a
// launch over such views takes the instantiation whose every plane request waits for the valid bits (choose_fused: the large-launch
// kernel without early requests, also for a small launch).  A view whose count is still on its way -- the reference's loop sets a new
// selection and launches at once, scan after scan -- is routed by the last count that did arrive (the lasso of one scan is about as
// large as that of the scan before; the route decides time only, never results); a view that never had a complete count is dense:
// the default this library was tuned on.
bool sparse_views(const sl3d_ctx *x, int first, int n)
{
    const double quads = (double)(x->P.pitch >> 2) * (double)x->P.H;
    for (int v = first; v < first + n; v++) {
        unsigned c;
        if (!quads_known(x, v, &c)) c = x->quad_last[(size_t)v];
        if (c == ~0u || (double)c >= 0.65 * quads) return false;
    }
    return true;
}

int check_view(sl3d_ctx *x, int view, int n)
{
    if (!x) return SL3D_E_INVALID_ARG;
    if (view < 0 || n < 1 || view + n > x->cfg.max_views) return fail(x, SL3D_E_INVALID_ARG, "view index out of range");
    return SL3D_OK;
}

// what kind of memory `p` is: 0 = pageable host memory (unknown to the runtime: a copy from it is consumed before the call returns),
// 1 = pinned host memory (hipHostMalloc / hipHostRegister: copies from it are asynchronous DMA, the caller owns the hand-over, see
// include/sl3d.h), 2 = device memory (*device = its ordinal)
int memory_kind(const void *p, int *device)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory is not known to the runtime: not an error of ours
        return 0;
    }
    if (a.type == hipMemoryTypeHost) return 1;
    if (a.type == hipMemoryTypeDevice) {
        if (device) *device = a.device;
        return 2;
    }
    return 0;
}
bool is_pinned_host(const void *p) { return memory_kind(p) == 1; }

// the staging plane(s) of sl3d_set_mask(s): `slots` planes, zero outside the region the copies fill
static int ensure_mask_staging(sl3d_ctx *x, int slots)
{
    if (slots <= x->mask_raw_slots) return SL3D_OK;
    const int frc = flush_masks(x, 0, x->cfg.max_views);  // (deferred masks still lie in the plane that is about to be freed)
    if (frc) return frc;
    HIPCHK(x, hipStreamSynchronize(x->stream));
    (void)hipFree(x->d_mask_raw);
    x->allocs.erase(std::remove(x->allocs.begin(), x->allocs.end(), (void *)x->d_mask_raw), x->allocs.end());
    x->d_mask_raw = nullptr;
    x->mask_raw_slots = 0;
    const int rc = dev_alloc(x, &x->d_mask_raw, (size_t)slots * x->P.mask_view_stride);
    if (rc) return rc;
    HIPCHK(x, hipMemsetAsync(x->d_mask_raw, 0, (size_t)slots * x->P.mask_view_stride, x->stream));
    x->mask_raw_slots = slots;
    return SL3D_OK;
}

MaskRegion mask_region(const KParams &P, MaskSrc &S)
{
    MaskRegion g;
    g.gy0 = std::max(P.row0 - SL3D_MASK_HALO, 0), g.gy1 = std::min(P.row0 + P.H + SL3D_MASK_HALO, P.fullH);
    g.gx0 = std::max(P.col0 - SL3D_MASK_HALO, 0), g.gx1 = std::min(P.col0 + P.W + SL3D_MASK_HALO, P.fullW);
    S.bx0 = SL3D_MASK_LPAD + g.gx0 - P.col0, S.bx1 = SL3D_MASK_LPAD + g.gx1 - P.col0;
    S.r0 = g.gy0 - P.row0 + SL3D_MASK_HALO, S.r1 = g.gy1 - P.row0 + SL3D_MASK_HALO;
    return g;
}

// k_mask_prepare over the views of one call; the views' counts become known under a new sequence number
static int prepare_masks(sl3d_ctx *x, int first_view, int n_views, const MaskSrc &S)
{
    // (the views' mask planes are rewritten on the context's stream: behind any launch a lane still runs over them)
    const int jrc = sl3d_lanes_join(x);
    if (jrc) return jrc;
    const unsigned seq = ++x->mask_seq;
    for (int v = first_view; v < first_view + n_views; v++) {
        unsigned c;
        (void)quads_known(x, v, &c);  // (the count of the selection this one replaces, if it has arrived: quad_last, what sparse_views falls back on)
        x->quad_seq[v] = seq;
        x->quad_src[v] = v;
        x->quad_kind[v] = 0;
    }
    return launched(x, launch_mask_prepare(x->P, first_view, n_views, S, x->d_quad_part, seq, x->stream));
}

// ---- deferred masks ---------------------------------------------------------------------------------------------------------------
// image_scissor hands main() a new selection every scan (m_tech_project_console.cpp:366) and a scan is ONE view: k_mask_prepare in
// front of a one-view launch was a fifth of the per-scan device time.  A timed context therefore only RECORDS where the selection of
// up to SL3D_SMALL_LAUNCH_VIEWS views lies; run_fused hands it to a MASKIN launch (the fused kernel evaluates H0 / S3b / S3d itself
// and leaves every plane and count k_mask_prepare would have left) when the launch qualifies, and anything else that reads the views'
// mask planes prepares them first.
static bool can_defer(const sl3d_ctx *x, int n_views, const MaskSrc &S, int lo, int hi)
{
    const KParams &P = x->P;
    return !x->keep && !x->eager_mask && x->h_mi_part && n_views <= SL3D_SMALL_LAUNCH_VIEWS && P.F == 3 && P.Nv >= 1 && P.Nv <= 12 && P.Nh >= 1 &&
           P.Nh <= 12 && hi - lo >= 8 && (unsigned long long)(P.H + 2 * SL3D_MASK_HALO) * S.stride < (1ull << 32);
}

// the masks of views [first_view, first_view + n_views) that are still deferred go through k_mask_prepare now (callers_only: only
// those whose source is the caller's own memory -- what a synchronising call owes the caller)
int flush_masks(sl3d_ctx *x, int first_view, int n_views, bool callers_only)
{
    if (x->n_pending == 0) return SL3D_OK;
    for (int v = first_view; v < first_view + n_views; v++) {
        sl3d_ctx::PendingMask &pm = x->pend[(size_t)v];
        if (!pm.pending || (callers_only && pm.ours)) continue;
        MaskSrc S;
        (void)mask_region(x->P, S);
        S.origin = pm.origin;
        S.stride = pm.stride;
        S.view_stride = 0;
        pm.pending = false;
        x->n_pending--;
        const int rc = prepare_masks(x, v, 1, S);
        if (rc) return rc;
    }
    return SL3D_OK;
}

// views [first_view, first_view + n_views) are about to get a new mask whose staging overwrites slots [0, slots): their own deferred
// masks are superseded, other views' deferred masks that still lie in those slots are prepared first
int supersede_masks(sl3d_ctx *x, int first_view, int n_views, int slots)
{
    if (x->n_pending == 0) return SL3D_OK;
    const uintptr_t lo = (uintptr_t)x->d_mask_raw, hi = lo + (uintptr_t)slots * x->P.mask_view_stride;
    for (int v = 0; v < x->cfg.max_views; v++) {
        sl3d_ctx::PendingMask &pm = x->pend[(size_t)v];
        if (!pm.pending) continue;
        if (v >= first_view && v < first_view + n_views) {
            pm.pending = false;
            x->n_pending--;
        } else if (pm.ours && pm.origin >= lo && pm.origin < hi) {
            const int rc = flush_masks(x, v, 1);
            if (rc) return rc;
        }
    }
    return SL3D_OK;
}

static void defer_masks(sl3d_ctx *x, int first_view, int n_views, const MaskSrc &S, bool ours, int lo, int hi)
{
    for (int k = 0; k < n_views; k++) {
        sl3d_ctx::PendingMask &pm = x->pend[(size_t)(first_view + k)];
        if (!pm.pending) x->n_pending++;
        pm.pending = true;
        pm.ours = ours;
        pm.origin = S.origin + (uintptr_t)k * S.view_stride;
        pm.stride = S.stride;
        pm.lo = lo;
        pm.hi = hi;
    }
}

// the stream is drained for the caller: what was deferred on the CALLER's memory is prepared first (include/sl3d.h: a device-resident
// mask stays unchanged until the next synchronising call)
int sync_for_caller(sl3d_ctx *x)
{
    const int rc = flush_masks(x, 0, x->cfg.max_views, true);
    if (rc) return rc;
    HIPCHK(x, hipStreamSynchronize(x->stream));
    return SL3D_OK;
}

// selected_region -> the context's mask planes, on the device (H0, S3b, S3d: m_tech_project_console.cpp:366, 3/wrapped_phase.cpp:
// 106-115, :253-279).  Host memory: the rows of the window + 2-pixel halo (clipped to the frame) go up as ONE 2-D copy per distinct
// mask into a staging plane.  Device memory of this context's GPU (4-byte aligned rows): no copy at all -- the kernel reads the
// caller's buffer.  ONE launch of k_mask_prepare then serves every view of the call.  No host pass over a mask, no allocation
// after the first call of a given shape, no stream synchronisation unless the source is pageable host memory.
extern "C" int sl3d_set_masks(sl3d_ctx *x, int first_view, int n_views, const uint8_t *m, size_t stride, size_t view_stride)
try {
    int rc = check_view(x, first_view, n_views);
    if (rc) return rc;
    if (!m || stride < (size_t)x->cfg.full_width) return fail(x, SL3D_E_INVALID_ARG, "mask: null or stride < full_width");
    if (view_stride != 0 && view_stride < stride * (size_t)(x->cfg.full_height - 1) + (size_t)x->cfg.full_width)
        return fail(x, SL3D_E_INVALID_ARG, "masks: view_stride is smaller than one mask (0 = the same mask for every view)");
    const KParams &P = x->P;
    ON_DEVICE_QUIET(x);  // (a device-resident mask that is merely recorded gives the stream nothing: the launches the lanes hold go on)
    MaskSrc S;
    const MaskRegion g = mask_region(P, S);
    int dev = -1;
    // (what kind of memory: asked at the first byte that is READ -- of a window below the frame's first rows the mask's nominal
    // origin `m` may lie in front of the caller's allocation, in somebody else's or in none)
    const int kind = memory_kind(m + (size_t)g.gy0 * stride + (size_t)g.gx0, &dev);
    const int distinct = view_stride ? n_views : 1;
    // dwords of a row must neither straddle the row's end nor start off a 4-byte boundary
    const bool direct = kind == 2 && dev == x->cfg.device && ((uintptr_t)m & 3u) == 0 && (stride & 3u) == 0 && (view_stride & 3u) == 0 && (P.col0 & 3) == 0 &&
                        (P.fullW & 3) == 0;
    if (direct) {
        S.origin = (uintptr_t)((intptr_t)m + ((intptr_t)P.row0 - SL3D_MASK_HALO) * (intptr_t)stride + P.col0 - SL3D_MASK_LPAD);
        S.stride = stride;
        S.view_stride = view_stride;
        rc = supersede_masks(x, first_view, n_views, 0);
        if (rc) return rc;
    } else {
        if ((rc = sl3d_lanes_join(x))) return rc;  // (the staging plane is rewritten: behind any MASKIN launch that still reads it)
        rc = ensure_mask_staging(x, distinct);
        if (rc || (rc = supersede_masks(x, first_view, n_views, distinct))) return rc;
        for (int k = 0; k < distinct; k++) {
            uint8_t *dst = x->d_mask_raw + (size_t)k * P.mask_view_stride + (size_t)S.r0 * P.mpitch + S.bx0;
            HIPCHK_DRAIN(x, hipMemcpy2DAsync(dst, P.mpitch, m + (size_t)k * view_stride + (size_t)g.gy0 * stride + g.gx0, stride, (size_t)(g.gx1 - g.gx0),
                                             (size_t)(g.gy1 - g.gy0), hipMemcpyDefault, x->stream));
        }
        S.origin = (uintptr_t)x->d_mask_raw;
        S.stride = (size_t)P.mpitch;
        S.view_stride = distinct > 1 ? P.mask_view_stride : 0;
    }
    // plane bytes of a row a kernel may read: all of the staging plane's row, the frame's columns of a caller's mask
    const int lo = direct ? SL3D_MASK_LPAD - P.col0 : 0, hi = direct ? SL3D_MASK_LPAD - P.col0 + P.fullW : P.mpitch;
    if (can_defer(x, n_views, S, lo, hi)) {
        defer_masks(x, first_view, n_views, S, !direct, lo, hi);
    } else {
        rc = prepare_masks(x, first_view, n_views, S);
        if (rc) return rc;
    }
    if (kind == 0) HIPCHK(x, hipStreamSynchronize(x->stream));  // pageable source: consumed before we return
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_set_mask(sl3d_ctx *x, int view, const uint8_t *m, size_t stride) { return sl3d_set_masks(x, view, 1, m, stride, 0); }

int ensure_colrow(sl3d_ctx *x, size_t bytes)
{
    if (x->d_colrow && x->colrow_bytes >= bytes) return SL3D_OK;
    HIPCHK(x, hipStreamSynchronize(x->stream));
    if (x->d_colrow) {
        (void)hipFree(x->d_colrow);
        x->allocs.erase(std::remove(x->allocs.begin(), x->allocs.end(), (void *)x->d_colrow), x->allocs.end());
        x->d_colrow = nullptr;
        x->colrow_bytes = 0;
    }
    const int rc = dev_alloc(x, &x->d_colrow, bytes);
    if (rc) return rc;
    x->colrow_bytes = bytes;
    return SL3D_OK;
}

// selected_region in the reference's own [col][row] int layout: the window's columns (+ 2-pixel halo, clipped to the frame) are
// contiguous runs of rows -> one 2-D copy; k_mask_from_colrow transposes them into the byte staging plane, k_mask_prepare does
// the rest (as for sl3d_set_mask).
extern "C" int sl3d_set_mask_colrow(sl3d_ctx *x, int view, const int32_t *sel)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    if (!sel) return fail(x, SL3D_E_INVALID_ARG, "mask: null");
    const KParams &P = x->P;
    ON_DEVICE(x);
    MaskSrc S;
    const MaskRegion g = mask_region(P, S);
    const int gx0 = g.gx0, gy0 = g.gy0, ncols = g.gx1 - g.gx0, nrows = g.gy1 - g.gy0;
    S.origin = (uintptr_t)x->d_mask_raw;
    S.stride = (size_t)P.mpitch;
    S.view_stride = 0;
    rc = ensure_colrow(x, std::max((size_t)ncols * nrows * sizeof(int), (size_t)P.W * P.H * 24));
    if (rc || (rc = supersede_masks(x, view, 1, 1))) return rc;
    HIPCHK(x, hipMemcpy2DAsync(x->d_colrow, (size_t)nrows * sizeof(int), sel + (size_t)gx0 * P.fullH + gy0, (size_t)P.fullH * sizeof(int),
                               (size_t)nrows * sizeof(int), (size_t)ncols, hipMemcpyHostToDevice, x->stream));
    rc = launched(x, launch_mask_from_colrow(P, (const int *)x->d_colrow, gx0, gy0, ncols, nrows, x->d_mask_raw, x->stream));
    if (rc) return rc;
    if (can_defer(x, 1, S, 0, P.mpitch)) {
        defer_masks(x, view, 1, S, true, 0, P.mpitch);
    } else {
        rc = prepare_masks(x, view, 1, S);
        if (rc) return rc;
    }
    if (!is_pinned_host(sel)) HIPCHK(x, hipStreamSynchronize(x->stream));  // pageable source: consumed before we return
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_get_global_colrow(sl3d_ctx *x, int view, int which, void *out, int out_height, int out_row0)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    const KParams &P = x->P;
    if (!out || which < SL3D_G_VALID_V || which > SL3D_G_POINTS_F64 || out_row0 < 0 || out_row0 + P.H > out_height)
        return fail(x, SL3D_E_INVALID_ARG, "get_global_colrow: null output, unknown global, or the window's rows do not fit out_height");
    if (which != SL3D_G_VALID && which != SL3D_G_POINTS_F64 && (rc = need_keep(x))) return rc;
    const size_t elem = (which == SL3D_G_INTERSECTION_POINTS || which == SL3D_G_POINTS_F64) ? 24 : 4;
    ON_DEVICE(x);
    rc = ensure_colrow(x, (size_t)P.W * P.H * 24);
    if (rc) return rc;
    rc = launched(x, launch_to_colrow(P, view, which, x->d_colrow, x->stream));
    if (rc) return rc;
    uint8_t *dst = (uint8_t *)out + (size_t)out_row0 * elem;
    if (out_height == P.H)
        HIPCHK(x, hipMemcpyAsync(dst, x->d_colrow, (size_t)P.W * P.H * elem, hipMemcpyDeviceToHost, x->stream));
    else
        HIPCHK(x, hipMemcpy2DAsync(dst, (size_t)out_height * elem, x->d_colrow, (size_t)P.H * elem, (size_t)P.H * elem, (size_t)P.W, hipMemcpyDeviceToHost,
                                   x->stream));
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_set_frames_range(sl3d_ctx *x, int view, int axis, int first_plane, const uint8_t *const *planes, int n_planes, size_t stride)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    const KParams &P = x->P;
    const int N = axis == 0 ? P.Nv : P.Nh;
    if ((axis != 0 && axis != 1) || !planes || first_plane < 0 || n_planes < 1 || first_plane + n_planes > P.F + 2 * N || stride < (size_t)P.W)
        return fail(x, SL3D_E_INVALID_ARG, "set_frames_range: planes [first, first + n) must lie inside the axis' n_fringe + 2*n_gray planes, stride >= width");
    ON_DEVICE(x);
    const int base = (axis == 0 ? 0 : P.F + 2 * P.Nv) + first_plane;
    bool back_to_back = true;  // the planes follow each other in host memory with the same row stride
    for (int i = 0; i < n_planes; i++) {
        if (!planes[i]) return fail(x, SL3D_E_INVALID_ARG, "set_frames: null plane");
        if (i && planes[i] != planes[i - 1] + stride * (size_t)P.H) back_to_back = false;
    }
    uint8_t *dst0 = x->d_frames + (size_t)view * P.view_stride + (size_t)base * P.plane_stride;
    if (back_to_back) {
        // the device planes of an axis are back to back too (plane_stride = pitch * H): the whole range is ONE 2-D copy
        HIPCHK(x, hipMemcpy2DAsync(dst0, P.pitch, planes[0], stride, P.W, (size_t)P.H * n_planes, hipMemcpyDefault, x->stream));
    } else {
        for (int i = 0; i < n_planes; i++)
            HIPCHK_DRAIN(x, hipMemcpy2DAsync(dst0 + (size_t)i * P.plane_stride, P.pitch, planes[i], stride, P.W, P.H, hipMemcpyDefault, x->stream));
    }
    // a pageable source is consumed before we return; the hand-over is asynchronous only if EVERY plane of the call is pinned host
    // memory or device memory (callers mix sources: a pinned image next to file-decoded pageable frames; planes that already live
    // on a GPU -- another context's frame stack, an acquisition stage on the device -- are copied device to device)
    bool all_pinned = true;
    for (int i = 0; i < n_planes && all_pinned; i++)
        if ((i == 0 || !back_to_back) && memory_kind(planes[i]) == 0) all_pinned = false;
    if (!all_pinned) HIPCHK(x, hipStreamSynchronize(x->stream));
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_set_frames(sl3d_ctx *x, int view, int axis, const uint8_t *const *planes, int n_planes, size_t stride)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    const KParams &P = x->P;
    const int N = axis == 0 ? P.Nv : P.Nh;
    if ((axis != 0 && axis != 1) || !planes || n_planes != P.F + 2 * N || stride < (size_t)P.W)
        return fail(x, SL3D_E_INVALID_ARG, "set_frames: expected n_fringe + 2*n_gray planes (fringe, gray, inverse) and stride >= width");
    return sl3d_set_frames_range(x, view, axis, 0, planes, n_planes, stride);  // the whole axis
}
SL3D_CATCH(x)

// sl3d_set_frames for RAW captures: what the acquisition stage does between the camera and the files stage 3/4 read
// (cvUndistort2 with the camera calibration, 2/project_pattern.cpp:220,232,287,...) happens on the device, one launch for
// all planes of the axis with the camera's map (built once per calibration).  Whole frames only: a window or a row stripe
// would need source rows from outside itself.
extern "C" int sl3d_set_frames_raw(sl3d_ctx *x, int view, int axis, const uint8_t *const *planes, int n_planes, size_t stride)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    const KParams &P = x->P;
    const int N = axis == 0 ? P.Nv : P.Nh;
    if ((axis != 0 && axis != 1) || !planes || n_planes != P.F + 2 * N || stride < (size_t)P.W)
        return fail(x, SL3D_E_INVALID_ARG, "set_frames_raw: expected n_fringe + 2*n_gray planes (fringe, gray, inverse) and stride >= width");
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "set_frames_raw before set_calibration");
    if (P.W != P.fullW || P.H != P.fullH) return fail(x, SL3D_E_UNSUPPORTED, "set_frames_raw: whole frames only (no window / stripe)");
    ON_DEVICE(x);
    const size_t maps = (((size_t)P.W * P.H * 6 + 63) / 64) * 64, max_planes = (size_t)P.F + 2 * (size_t)std::max(P.Nv, P.Nh);
    if (!x->d_raw) {
        rc = dev_alloc(x, &x->d_raw, maps + max_planes * P.plane_stride);
        if (rc) return rc;
    }
    short *m1 = (short *)x->d_raw;
    unsigned short *m2 = (unsigned short *)(x->d_raw + (size_t)P.W * P.H * 4);
    uint8_t *raw = x->d_raw + maps;
    for (int i = 0; i < n_planes; i++)  // every pointer is checked before anything is enqueued
        if (!planes[i]) return fail(x, SL3D_E_INVALID_ARG, "set_frames_raw: null plane");
    for (int i = 0; i < n_planes; i++) {
        HIPCHK_DRAIN(x, hipMemcpy2DAsync(raw + (size_t)i * P.plane_stride, P.pitch, planes[i], stride, P.W, P.H, hipMemcpyHostToDevice, x->stream));
    }
    const int base = axis == 0 ? 0 : P.F + 2 * P.Nv;
    uint8_t *dst = x->d_frames + (size_t)view * P.view_stride + (size_t)base * P.plane_stride;
    rc = launched(x, launch_undistort_planes(raw, P.pitch, P.plane_stride, dst, P.pitch, P.plane_stride, P.W, P.H, n_planes, x->Kc_raw, x->dc_raw, m1, m2,
                                             !x->raw_map_valid, x->stream));
    if (rc) return rc;
    x->raw_map_valid = true;
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_copy_view(sl3d_ctx *x, int src, int dst)
try {
    int rc = check_view(x, src);
    if (rc || (rc = check_view(x, dst))) return rc;
    if (src == dst) return SL3D_OK;
    const KParams &P = x->P;
    ON_DEVICE(x);
    if ((rc = flush_masks(x, src, 1)) || (rc = supersede_masks(x, dst, 1, 0))) return rc;
    HIPCHK(x, hipMemcpyAsync(x->d_frames + (size_t)dst * P.view_stride, x->d_frames + (size_t)src * P.view_stride, P.view_stride,
                             hipMemcpyDeviceToDevice, x->stream));
    HIPCHK(x, hipMemcpyAsync(x->d_mask + (size_t)dst * P.mask_view_stride, x->d_mask + (size_t)src * P.mask_view_stride,
                             P.mask_view_stride, hipMemcpyDeviceToDevice, x->stream));
    HIPCHK(x, hipMemcpyAsync(x->d_band + (size_t)dst * P.px_view_stride, x->d_band + (size_t)src * P.px_view_stride, P.px_view_stride,
                             hipMemcpyDeviceToDevice, x->stream));
    x->quad_seq[dst] = x->quad_seq[src];  // the duplicate's count of selected quads is the source's (until either mask is set again)
    x->quad_src[dst] = x->quad_src[src];
    x->quad_kind[dst] = x->quad_kind[src];
    x->quad_last[dst] = x->quad_last[src];
    x->quad_sum_seq[dst] = 0u;  // (a sum cached for dst under the same sequence number -- one sl3d_set_masks call serves many views -- is not the source's)
    return SL3D_OK;
}
SL3D_CATCH(x)


// Synthetic capture of one view written straight into the resident frame stack (N1; formulas of
// 1/pattern_generator.cpp:80-105,302,313,497 -- see k_synth).  Benchmark / test input, not part of the timed path.
extern "C" int sl3d_synth_view(sl3d_ctx *x, int view, const double plane[3], uint64_t seed, int view_id, int noise, float gain, float offset)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    if (!plane) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called");
    if (x->P.PW > x->P.fwv * (1 << x->P.Nv) || x->P.PH > x->P.fwh * (1 << x->P.Nh))
        return fail(x, SL3D_E_INVALID_ARG, "Gray code too short for the projector size");
    ON_DEVICE(x);
    SynthParams S = x->S;
    S.z0 = plane[0]; S.a = plane[1]; S.b = plane[2];
    S.seed = seed; S.view_id = view_id; S.noise = noise; S.gain = gain; S.offset = offset;
    return launched(x, launch_synth(x->P, x->C, S, view, x->stream));
}
SL3D_CATCH(x)

// the resident frames of one axis of one view, back to host planes (fringe, gray, inverse gray order)
extern "C" int sl3d_get_frames(sl3d_ctx *x, int view, int axis, uint8_t *const *planes, int n_planes, size_t stride)
try {
    int rc = check_view(x, view);
    if (rc) return rc;
    const KParams &P = x->P;
    const int N = axis == 0 ? P.Nv : P.Nh;
    if ((axis != 0 && axis != 1) || !planes || n_planes != P.F + 2 * N || stride < (size_t)P.W)
        return fail(x, SL3D_E_INVALID_ARG, "get_frames: expected n_fringe + 2*n_gray planes and stride >= width");
    ON_DEVICE(x);
    const int base = axis == 0 ? 0 : P.F + 2 * P.Nv;
    for (int i = 0; i < n_planes; i++) {
        const uint8_t *src = x->d_frames + (size_t)view * P.view_stride + (size_t)(base + i) * P.plane_stride;
        HIPCHK_DRAIN(x, hipMemcpy2DAsync(planes[i], stride, src, P.pitch, P.W, P.H, hipMemcpyDeviceToHost, x->stream));
    }
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)
