// sl3d_capi_next.cpp -- the rows either side of the hot path: N4 capture-side cvUndistort2, N3's transform of a host cloud, N1 projector
// patterns; and plain copies out of device buffers the library handed out.
#include "sl3d_capi_internal.h"

// ---- N4: cvUndistort2 (2/project_pattern.cpp:220,232,...) -------------------------------------------------------------
extern "C" int sl3d_undistort(sl3d_ctx *x, const uint8_t *src, size_t src_stride, int width, int height, int channels, const double K[9],
                              const double dist[5], uint8_t *dst, size_t dst_stride)
try {
    if (!x || !src || !dst || !K || !dist || width < 1 || height < 1 || (channels != 1 && channels != 3) || width > 32767 || height > 32767)
        return fail(x, SL3D_E_INVALID_ARG, "undistort: null argument, size, or channels not 1 / 3");
    const size_t row = (size_t)width * channels, img = (((row + 15) / 16) * 16) * (size_t)height;
    if (src_stride < row || dst_stride < row) return fail(x, SL3D_E_INVALID_ARG, "undistort: stride < width*channels");
    const size_t pitch = ((row + 15) / 16) * 16, maps = (size_t)width * height * 6, need = 2 * img + maps + 128;
    ON_DEVICE(x);
    if (need > x->und_bytes) {
        SYNC_FOR_CALLER(x);
        if (x->d_und) {
            (void)hipFree(x->d_und);
            x->allocs.erase(std::remove(x->allocs.begin(), x->allocs.end(), (void *)x->d_und), x->allocs.end());
            x->d_und = nullptr;
            x->und_bytes = 0;
            x->und_map_valid = false;
        }
        int rc = dev_alloc(x, &x->d_und, need);
        if (rc) return rc;
        x->und_bytes = need;
        x->und_map_valid = false;
    }
    // layout: maps first (they survive from call to call), then the source and the result image
    short *m1 = (short *)x->d_und;
    unsigned short *m2 = (unsigned short *)(x->d_und + (size_t)width * height * 4);
    uint8_t *d_src = x->d_und + ((maps + 63) / 64) * 64, *d_dst = d_src + img;
    double key[16];
    for (int k = 0; k < 9; k++) key[k] = K[k];
    for (int k = 0; k < 5; k++) key[9 + k] = dist[k];
    key[14] = width; key[15] = height;
    const bool build = !x->und_map_valid || memcmp(key, x->und_key, sizeof key) != 0;
    HIPCHK(x, hipMemcpy2DAsync(d_src, pitch, src, src_stride, row, (size_t)height, hipMemcpyHostToDevice, x->stream));
    int rc = launched(x, launch_undistort(d_src, pitch, d_dst, pitch, width, height, channels, K, dist, m1, m2, build, x->stream));
    if (rc) return rc;
    memcpy(x->und_key, key, sizeof key);
    x->und_map_valid = true;
    HIPCHK(x, hipMemcpy2DAsync(dst, dst_stride, d_dst, pitch, row, (size_t)height, hipMemcpyDeviceToHost, x->stream));
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)

// One cloud of 9/register_point_clouds.cpp:83-128 that lives in host memory (the reference reads each from a PLY file):
// p -> R_y(theta) * (p - t) + t with the reference's float / double-accumulator arithmetic (k_register), theta in degrees
// converted with Pi = 22/7.  The caller advances theta by rot_step IN FLOAT from cloud to cloud, as :145 does.
extern "C" int sl3d_transform_cloud(sl3d_ctx *x, const float *xyz_in, int64_t n, float theta_deg, float tx, float ty, float tz, float *xyz_out)
try {
    if (!x || n < 0 || (n > 0 && (!xyz_in || !xyz_out))) return fail(x, SL3D_E_INVALID_ARG, "transform_cloud: null argument");
    if (n == 0) return SL3D_OK;
    ON_DEVICE(x);
    float *d = nullptr;
    HIPCHK(x, hipMalloc((void **)&d, (size_t)n * 6 * sizeof(float)));
    const float theta = theta_deg;
    const float R4[4] = {(float)cos(theta * 22.0 / 7.0 / 180.0), (float)(-1.0f * sin(theta * 22.0 / 7.0 / 180.0)),
                         (float)sin(theta * 22.0 / 7.0 / 180.0), (float)cos(theta * 22.0 / 7.0 / 180.0)};
    hipError_t e = hipMemcpyAsync(d, xyz_in, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, x->stream);
    int rc = e == hipSuccess ? launch_register(d, d + 3 * (size_t)n, (long)n, R4, tx, ty, tz, x->stream) : (int)e;
    if (rc == 0) rc = (int)hipMemcpyAsync(xyz_out, d + 3 * (size_t)n, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, x->stream);
    if (rc == 0) rc = (int)hipStreamSynchronize(x->stream);
    (void)hipFree(d);
    if (rc) return fail(x, SL3D_E_HIP, std::string("transform_cloud: ") + hipGetErrorString((hipError_t)rc));
    return SL3D_OK;
}
SL3D_CATCH(x)

// ---- N1: projector patterns (1/pattern_generator.cpp) ---------------------------------------------------------------
#define PI_REF 22.0 / 7.0 /* PROJECT_GLOBAL/global_cv.h:62: unparenthesised on purpose */

extern "C" int sl3d_pattern_counts(int proj_extent, int fringe_width, int *n_codes, int *n_planes)
try {
    if (proj_extent < 1 || fringe_width < 1 || !n_codes || !n_planes) return SL3D_E_INVALID_ARG;
    *n_codes = (int)ceil((float)proj_extent / (float)fringe_width);                 // :224 / :228
    *n_planes = (int)ceil((logf((float)*n_codes) / logf(2.0)));                     // :226 / :229
    return SL3D_OK;
}
SL3D_CATCH(nullptr)

// the values of one pattern along its varying axis, with the reference's expressions and the host libm it calls
static void pattern_profile(int kind, int F, int index, int extent, int fw, int nplanes, uint8_t *out)
{
    memset(out, 0, (size_t)extent);
    if (kind == SL3D_PATTERN_FRINGE) {
        for (int p = 0; p < extent; p++) {
            float t = 0.0;
            if (F == 3) t = 127.0f + 128.0f * cosf(((float)p / (float)fw) * 2.0 * PI_REF - PI_REF - ((PI_REF) / 2.0) + (PI_REF / 2.0) * (float)index);  // :302
            else if (F == 4) t = 127.0 + 128.0 * cosf(((float)p / (float)fw) * (2.0 * PI_REF) - PI_REF + (PI_REF / 2.0) * (float)index);               // :340
            else t = 127.0f + 128.0f * cosf(((float)p / (float)fw) * (2.0 * PI_REF) - PI_REF - 2.0 * ((PI_REF) / 2) + ((PI_REF) / 2) * (float)index);   // :369
            out[p] = (uint8_t)(int)t;  // a float in [-1, 255] through int, as the x86 build of `(unsigned char)t` does (-1 -> 255)
        }
    } else if (index < nplanes && (kind == SL3D_PATTERN_GRAY || kind == SL3D_PATTERN_INVERSE_GRAY)) {
        for (int c = 0, code = 0; c < extent; c += fw, code++) {
            // bit `index` (MSB first, nplanes bits) of the Gray code of `code`: B_{i-1} xor B_i  (:83-101)
            const int sh = nplanes - 1 - index;
            const int b = (code >> sh) & 1, bp = index == 0 ? 0 : (code >> (sh + 1)) & 1;
            const uint8_t g = (uint8_t)((b ^ bp) * 255);
            for (int o = 0; o < fw && c + o < extent; o++) out[c + o] = kind == SL3D_PATTERN_GRAY ? g : (uint8_t)(255 - g);
        }
    } else if (index < nplanes && kind == SL3D_PATTERN_BINARY) {
        for (int p = 0; p < extent; p++) out[p] = ((int)(p / (pow(2, index) * fw)) % 2) == 1 ? 255 : 0;  // :275-283
    }
}

extern "C" int sl3d_generate_pattern(sl3d_ctx *x, int kind, int axis, int index, uint8_t *host_dst, size_t stride,
                                     const uint8_t **device_ptr, size_t *device_pitch)
try {
    if (!x) return SL3D_E_INVALID_ARG;
    const int PW = x->cfg.proj_width, PH = x->cfg.proj_height, F = x->cfg.n_fringe;
    if (kind < SL3D_PATTERN_FRINGE || kind > SL3D_PATTERN_BINARY || (axis != 0 && axis != 1)) return fail(x, SL3D_E_INVALID_ARG, "pattern: kind or axis");
    const int nplanes = axis == 0 ? x->cfg.n_gray_v : x->cfg.n_gray_h, fw = axis == 0 ? x->cfg.fringe_width_v : x->cfg.fringe_width_h;
    if (index < 0 || index >= (kind == SL3D_PATTERN_FRINGE ? F : nplanes + 1)) return fail(x, SL3D_E_INVALID_ARG, "pattern: index out of range");
    if (host_dst && stride < (size_t)PW) return fail(x, SL3D_E_INVALID_ARG, "pattern: stride < proj_width");
    ON_DEVICE(x);
    const size_t pitch = ((size_t)PW + 15) / 16 * 16, extent_max = (size_t)std::max(PW, PH) + 16;
    if (!x->d_pattern) {
        HIPCHK(x, hipMalloc((void **)&x->d_pattern, pitch * (size_t)PH));
        x->allocs.push_back(x->d_pattern);
        HIPCHK(x, hipMalloc((void **)&x->d_profile, extent_max));
        x->allocs.push_back(x->d_profile);
        x->pattern_pitch = pitch;
    }
    std::vector<uint8_t> prof(extent_max, 0);
    pattern_profile(kind, F, index, axis == 0 ? PW : PH, fw, nplanes, prof.data());
    HIPCHK(x, hipStreamSynchronize(x->stream));  // the previous pattern may still be in flight
    HIPCHK(x, hipMemcpy(x->d_profile, prof.data(), extent_max, hipMemcpyHostToDevice));
    int rc = launched(x, launch_pattern(x->d_pattern, pitch, PW, PH, axis, x->d_profile, x->stream));
    if (rc) return rc;
    SYNC_FOR_CALLER(x);
    if (host_dst) HIPCHK(x, hipMemcpy2D(host_dst, stride, x->d_pattern, pitch, (size_t)PW, (size_t)PH, hipMemcpyDeviceToHost));
    if (device_ptr) *device_ptr = x->d_pattern;
    if (device_pitch) *device_pitch = pitch;
    return SL3D_OK;
}
SL3D_CATCH(x)

#ifdef SL3D_MEASURE
// measurement builds (-DSL3D_TRACE): the per-wave phase stamps of the dense kernel (tools/phase_trace.py)
extern "C" int sl3d_debug_buffer(sl3d_ctx *x, const void **dev, size_t *bytes, int *n_tiles)
try {
    if (!x || !x->P.dbg || !x->dbg_words) return SL3D_E_STATE;
    *dev = x->P.dbg;
    *bytes = x->dbg_words * sizeof(unsigned long long);
    *n_tiles = fused_tiles(x->P);
    return SL3D_OK;
}
SL3D_CATCH(x)
#endif

extern "C" int sl3d_download(sl3d_ctx *x, void *host_dst, const void *device_src, size_t bytes)
try {
    if (!x || (bytes && (!host_dst || !device_src))) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    ON_DEVICE(x);
    if (bytes) HIPCHK(x, hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, x->stream));
    SYNC_FOR_CALLER(x);
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_download_2d(sl3d_ctx *x, void *host_dst, size_t dst_pitch, const void *device_src, size_t src_pitch, size_t width_bytes, size_t height)
try {
    if (!x || !host_dst || !device_src || dst_pitch < width_bytes || src_pitch < width_bytes) return fail(x, SL3D_E_INVALID_ARG, "download_2d: null argument or pitch < width");
    ON_DEVICE(x);
    if (width_bytes && height)
        HIPCHK(x, hipMemcpy2DAsync(host_dst, dst_pitch, device_src, src_pitch, width_bytes, height, hipMemcpyDeviceToHost, x->stream));
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_get_device_buffers(sl3d_ctx *x, sl3d_device_buffers *o)
try {
    if (!x || !o) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    const KParams &P = x->P;
    if (x->n_pending) {  // the caller is about to read the mask plane where it lies: deferred masks are prepared first
        ON_DEVICE(x);
        const int rc = flush_masks(x, 0, x->cfg.max_views);
        if (rc) return rc;
    }
    o->frames = x->d_frames;
    o->frame_pitch = P.pitch;
    o->plane_stride = P.plane_stride;
    o->view_stride = P.view_stride;
    o->planes_per_view = P.planes_per_view;
    o->mask = x->d_mask;
    o->mask_pitch = P.mpitch;
    o->mask_view_stride = P.mask_view_stride;
    o->points = x->d_points;
    o->points_pitch = (size_t)P.pitch * 12;
    o->points_view_stride = P.px_view_stride * 12;
    o->valid = x->d_valid;
    o->valid_pitch = P.pitch;
    o->valid_view_stride = P.px_view_stride;
    return SL3D_OK;
}
SL3D_CATCH(x)
