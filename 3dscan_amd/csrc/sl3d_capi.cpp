// sl3d_capi.cpp -- host side of the C ABI declared in include/sl3d.h.
//
// Owns the HBM layout (frame stack, mask with halo, dense results, optional stage planes), the
// per-scan constants of stage 7 (Rodrigues, A = K*[R|t]: 7/triangulation.cpp:1061-1126) and the
// exact atan2 table the kernels look the wrapped phase up in.  All compute happens in the HIP
// kernels of sl3d_kernels.hip; there is no CPU implementation of the path in this library.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <algorithm>
#include <string>
#include <vector>

#include "sl3d_ctx.h"

static thread_local std::string g_create_err;

int sl3d_fail(sl3d_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    else g_create_err = msg;
    return code;
}

// classifies the exception in flight (called from a catch (...) handler only); see sl3d_ctx.h
int sl3d_caught(sl3d_ctx *c, std::string *other) noexcept
{
    int code = SL3D_E_INTERNAL;
    const char *what = "unknown C++ exception";
    char buf[256];
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = SL3D_E_NOMEM;
        what = "out of host memory (std::bad_alloc)";
    } catch (const std::exception &e) {
        snprintf(buf, sizeof buf, "internal error: %s", e.what());
        what = buf;
    } catch (...) {
    }
    try {  // (storing the text allocates: if even that fails the status alone goes back)
        if (other) *other = what;
        else if (c) c->err = what;
        else g_create_err = what;
    } catch (...) {
    }
    return code;
}

extern "C" const char *sl3d_version(void) { return SL3D_VERSION_STRING " (gfx950, hip)"; }

extern "C" const char *sl3d_strerror(int s)
{
    switch (s) {
    case SL3D_OK: return "ok";
    case SL3D_E_INVALID_ARG: return "invalid argument";
    case SL3D_E_NO_DEVICE: return "no HIP device available (this library has no CPU fallback)";
    case SL3D_E_HIP: return "HIP runtime error";
    case SL3D_E_STATE: return "call order violated";
    case SL3D_E_UNSUPPORTED: return "unsupported configuration";
    case SL3D_E_NOMEM: return "out of memory";
    case SL3D_E_INTERNAL: return "internal error (a C++ exception was stopped at the C boundary)";
    default: return "unknown status";
    }
}

extern "C" const char *sl3d_last_error(const sl3d_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

template <typename T>
static int dev_alloc(sl3d_ctx *c, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? SL3D_E_NOMEM : SL3D_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    c->allocs.push_back(q);
    *p = (T *)q;
    return SL3D_OK;
}

// The wrapped phase is a function of two small integers (t1 in [-255,255], t2 in [-510,510]).
// The kernels evaluate it in fp64 (atan2_lattice); this table of the double-precision libm atan2
// the reference calls (3/wrapped_phase.cpp:175) -- table 0 = (float)atan2(t1,t2), table 1 = the
// value after stage 4's in-place `+= Pi` (4/phase_unwrap.cpp:290,308) -- is what the device
// function is verified against, exhaustively, before the first context of a process is handed out.
static void build_atan_tables(std::vector<float> &tab)
{
    const size_t n = (size_t)SL3D_ATAN_T1 * SL3D_ATAN_T2;
    tab.resize(2 * n);
    for (int t1 = -255; t1 <= 255; t1++)
        for (int t2 = -510; t2 <= 510; t2++) {
            const float a = (float)t1, b = (float)t2;  // the reference holds t1,t2 in float
            const float phi = (float)atan2((double)a, (double)b);
            float sh = phi;
            sh += 22.0 / 7.0;  // Pi macro of global_cv.h:62, evaluated in double, rounded on store
            const size_t i = (size_t)(t1 + 255) * SL3D_ATAN_T2 + (size_t)(t2 + 510);
            tab[i] = phi;
            tab[n + i] = sh;
        }
}

extern "C" int sl3d_create(const sl3d_config *cfg, sl3d_ctx **out)
try {
    if (!cfg || !out) return fail(nullptr, SL3D_E_INVALID_ARG, "null argument");
    *out = nullptr;
    sl3d_config c = *cfg;
    if (c.full_width == 0) c.full_width = c.width;
    if (c.full_height == 0) c.full_height = c.height;
    if (c.n_fringe == 0) c.n_fringe = 3;
    if (c.max_views <= 0) c.max_views = 1;
    if (c.width < 1 || c.height < 1 || c.col0 < 0 || c.row0 < 0 || c.col0 + c.width > c.full_width || c.row0 + c.height > c.full_height)
        return fail(nullptr, SL3D_E_INVALID_ARG, "window does not fit the frame");
    if (c.proj_width < 1 || c.proj_height < 1 || c.fringe_width_v < 1 || c.fringe_width_h < 1)
        return fail(nullptr, SL3D_E_INVALID_ARG, "projector size / fringe width must be positive");
    if ((long long)c.proj_width * c.proj_height >= (1ll << 29))  // (the projector table of rig class 2 is addressed with 32-bit byte offsets)
        return fail(nullptr, SL3D_E_UNSUPPORTED, "projector too large: fewer than 2^29 pixels");
    if (c.n_fringe < 3 || c.n_fringe > 5) return fail(nullptr, SL3D_E_UNSUPPORTED, "n_fringe must be 3, 4 or 5");
    if (c.n_gray_v < 0 || c.n_gray_v > SL3D_MAX_GRAY || c.n_gray_h < 0 || c.n_gray_h > SL3D_MAX_GRAY)
        return fail(nullptr, SL3D_E_UNSUPPORTED, "n_gray out of range");
    if (c.n_codes_v <= 0) c.n_codes_v = (c.proj_width + c.fringe_width_v - 1) / c.fringe_width_v;
    if (c.n_codes_h <= 0) c.n_codes_h = (c.proj_height + c.fringe_width_h - 1) / c.fringe_width_h;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, SL3D_E_NO_DEVICE, sl3d_strerror(SL3D_E_NO_DEVICE));
    if (c.device < 0 || c.device >= ndev) return fail(nullptr, SL3D_E_INVALID_ARG, "device ordinal out of range");

    sl3d_ctx *x = new sl3d_ctx();
    // whatever leaves this function early -- an error return or an exception -- releases what exists so far
    struct Unwind {
        sl3d_ctx *p;
        ~Unwind() { if (p) sl3d_destroy(p); }
    } unwind{x};
    x->cfg = c;
    x->keep = (c.flags & SL3D_FLAG_KEEP_STAGES) != 0;
#define CREATE_CHK(call)                                                                 \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            g_create_err = std::string(#call) + ": " + hipGetErrorString(e_);            \
            return e_ == hipErrorOutOfMemory ? SL3D_E_NOMEM : SL3D_E_HIP;                \
        }                                                                                \
    } while (0)
    DeviceGuard dev_guard_(c.device);
    CREATE_CHK(dev_guard_.err);
    if (c.stream) {
        x->stream = (hipStream_t)c.stream;
    } else {
        CREATE_CHK(hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking));
        x->own_stream = true;
    }
    CREATE_CHK(hipEventCreate(&x->ev0));
    CREATE_CHK(hipEventCreate(&x->ev1));

    KParams &P = x->P;
    P.W = c.width; P.H = c.height; P.fullW = c.full_width; P.fullH = c.full_height;
    P.col0 = c.col0; P.row0 = c.row0; P.PW = c.proj_width; P.PH = c.proj_height;
    P.F = c.n_fringe; P.Nv = c.n_gray_v; P.Nh = c.n_gray_h; P.fwv = c.fringe_width_v; P.fwh = c.fringe_width_h;
    P.ncodes_v = c.n_codes_v; P.ncodes_h = c.n_codes_h;
#ifdef SL3D_MEASURE
    P.ablate = getenv("SL3D_ABLATE") ? atoi(getenv("SL3D_ABLATE")) : 0;
#endif
    P.pitch = (c.width + 15) & ~15;
    P.planes_per_view = 2 * P.F + 2 * P.Nv + 2 * P.Nh;
    if ((size_t)P.pitch * (size_t)P.H >= ((size_t)1 << 32)) {  // the kernels address a plane with 32-bit byte offsets
        return fail(nullptr, SL3D_E_UNSUPPORTED, "window too large: a plane must stay below 4 GiB (split it into row stripes)");
    }
    P.plane_stride = (size_t)P.pitch * P.H;
    P.view_stride = (size_t)P.planes_per_view * P.plane_stride;
    P.mpitch = P.pitch + 2 * SL3D_MASK_LPAD;
    x->mask_rows = (size_t)P.H + 2 * SL3D_MASK_HALO;
    P.mask_view_stride = (size_t)P.mpitch * x->mask_rows;
    P.px_view_stride = (size_t)P.pitch * P.H;

    if ((P.view_stride >> 32) != 0) {
        g_create_err = "one view's frame stack exceeds 4 GiB: shard the frame by rows";
        return SL3D_E_UNSUPPORTED;
    }
    const size_t V = (size_t)c.max_views;
    int rc;
#define ALLOC(ptr, count)                                              \
    if ((rc = dev_alloc(x, &(ptr), (count))) != SL3D_OK) {             \
        g_create_err = x->err;                                         \
        return rc;                                                     \
    }
    ALLOC(x->d_frames, V * P.view_stride);
    ALLOC(x->d_mask, V * P.mask_view_stride);
    ALLOC(x->d_points, V * P.px_view_stride * 3);
    ALLOC(x->d_valid, V * P.px_view_stride);
    ALLOC(x->d_cal, 1);
    {
        const size_t nb = (P.px_view_stride + 1023) / 1024;
        ALLOC(x->d_blk_cnt, nb);
        ALLOC(x->d_blk_off, nb);
        ALLOC(x->d_total, 1);
        ALLOC(x->d_cloud, P.px_view_stride * 3);
    }
    ALLOC(x->d_band, V * P.px_view_stride);
    ALLOC(x->d_mask_raw, P.mask_view_stride);
    x->mask_raw_slots = 1;
    CREATE_CHK(hipMemsetAsync(x->d_mask_raw, 0, P.mask_view_stride, x->stream));
    x->quad_blocks = mask_prepare_blocks(P);
    CREATE_CHK(hipHostMalloc((void **)&x->h_quad_part, V * (size_t)x->quad_blocks * sizeof(unsigned long long), hipHostMallocMapped));
    memset((void *)x->h_quad_part, 0, V * (size_t)x->quad_blocks * sizeof(unsigned long long));
    CREATE_CHK(hipHostGetDevicePointer((void **)&x->d_quad_part, (void *)x->h_quad_part, 0));
    x->quad_seq.assign(V, 0u);
    x->quad_sum_seq.assign(V, 0u);
    x->quad_sum.assign(V, 0u);
    x->quad_src.resize(V);
    for (size_t v = 0; v < V; v++) x->quad_src[v] = (int)v;
    x->quad_kind.assign(V, 0);
    x->pend.assign(V, sl3d_ctx::PendingMask());
    x->eager_mask = (c.flags & SL3D_FLAG_EAGER_MASK) != 0;
    if (!x->keep && P.F == 3) {  // what a MASKIN launch leaves per wave (sl3d_fused.h: maskin_count)
        x->mi_part_stride = fused_maskin_part_stride(P);
        x->mi_part_words = fused_maskin_part_words(P);
        CREATE_CHK(hipHostMalloc((void **)&x->h_mi_part, V * (size_t)x->mi_part_stride * sizeof(unsigned), hipHostMallocMapped));
        memset((void *)x->h_mi_part, 0, V * (size_t)x->mi_part_stride * sizeof(unsigned));
        CREATE_CHK(hipHostGetDevicePointer((void **)&x->d_mi_part, (void *)x->h_mi_part, 0));
    }
    CREATE_CHK(hipMemsetAsync(x->d_mask, 0, V * P.mask_view_stride, x->stream));
    CREATE_CHK(hipMemsetAsync(x->d_frames, 0, V * P.view_stride, x->stream));
    CREATE_CHK(hipMemsetAsync(x->d_valid, 0, V * P.px_view_stride, x->stream));
    CREATE_CHK(hipMemsetAsync(x->d_points, 0, V * P.px_view_stride * 3 * sizeof(float), x->stream));
    CREATE_CHK(hipMemsetAsync(x->d_band, 0, V * P.px_view_stride, x->stream));
    P.frames = x->d_frames; P.mask = x->d_mask; P.points = x->d_points; P.valid = x->d_valid; P.band = x->d_band;
    {
        // one-time (per process and device) proof that the device atan2 reproduces the host libm bit for bit
        static std::mutex mu;
        static std::set<int> verified;
        std::lock_guard<std::mutex> lk(mu);
        if (!verified.count(c.device)) {
            std::vector<float> tab;
            build_atan_tables(tab);
            const size_t n = (size_t)SL3D_ATAN_T1 * SL3D_ATAN_T2;
            float *d_tab = nullptr;
            unsigned *d_cnt = nullptr, h_cnt = 0;
            CREATE_CHK(hipMalloc((void **)&d_tab, 2 * n * sizeof(float)));
            CREATE_CHK(hipMalloc((void **)&d_cnt, sizeof(unsigned)));
            CREATE_CHK(hipMemcpyAsync(d_tab, tab.data(), 2 * n * sizeof(float), hipMemcpyHostToDevice, x->stream));
            CREATE_CHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned), x->stream));
            CREATE_CHK((hipError_t)launch_atan_selfcheck(d_tab, d_tab + n, d_cnt, x->stream));
            CREATE_CHK(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(unsigned), hipMemcpyDeviceToHost, x->stream));
            CREATE_CHK(hipStreamSynchronize(x->stream));
            (void)hipFree(d_tab);
            (void)hipFree(d_cnt);
            if (h_cnt != 0) {
                g_create_err = "device atan2 differs from the host libm atan2 on " + std::to_string(h_cnt) +
                               " of 521731 lattice points: bit-exact parity cannot be guaranteed on this host/GPU pair";
                        return SL3D_E_UNSUPPORTED;
            }
            verified.insert(c.device);
        }
    }
    if (x->keep) {
        const size_t n = V * P.px_view_stride;
        for (int a = 0; a < 2; a++) {
            ALLOC(P.wrapped[a], n); ALLOC(P.unwrapped[a], n); ALLOC(P.code[a], n);
            ALLOC(P.valid_axis[a], n); ALLOC(P.dbg3[a], n); ALLOC(P.dbg4[a], n);
            CREATE_CHK(hipMemsetAsync(P.wrapped[a], 0, n * 4, x->stream));
            CREATE_CHK(hipMemsetAsync(P.unwrapped[a], 0, n * 4, x->stream));
            CREATE_CHK(hipMemsetAsync(P.code[a], 0xff, n * 4, x->stream));
            CREATE_CHK(hipMemsetAsync(P.valid_axis[a], 0, n, x->stream));
            CREATE_CHK(hipMemsetAsync(P.dbg3[a], 0, n, x->stream));
            CREATE_CHK(hipMemsetAsync(P.dbg4[a], 0, n, x->stream));
        }
        ALLOC(P.cpmap, n * 2); ALLOC(P.ipoints, n * 3);
        CREATE_CHK(hipMemsetAsync(P.cpmap, 0, n * 16, x->stream));
        CREATE_CHK(hipMemsetAsync(P.ipoints, 0, n * 24, x->stream));
    }
#if defined(SL3D_MEASURE) && defined(SL3D_TRACE)
    {   // phase stamps [view group][block][wave][8] of the dense timed kernel (tools/phase_trace.py)
        const size_t blocks = ((((size_t)(P.pitch >> 2) * P.H + 255) / 256) + 7) & ~(size_t)7;
        x->dbg_words = V * blocks * 4 * 8;
        ALLOC(P.dbg, x->dbg_words);
        CREATE_CHK(hipMemsetAsync(P.dbg, 0, x->dbg_words * sizeof(unsigned long long), x->stream));
    }
#endif
    CREATE_CHK(hipStreamSynchronize(x->stream));
#undef ALLOC
#undef CREATE_CHK
    unwind.p = nullptr;
    *out = x;
    return SL3D_OK;
}
SL3D_CATCH(nullptr)

extern "C" void sl3d_destroy(sl3d_ctx *x)
try {
    if (!x) return;
    DeviceGuard dev_guard_(x->cfg.device);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    for (void *p : x->allocs) (void)hipFree(p);
    if (x->h_counts) (void)hipHostFree(x->h_counts);
    if (x->h_quad_part) (void)hipHostFree((void *)x->h_quad_part);
    if (x->h_mi_part) (void)hipHostFree((void *)x->h_mi_part);
    for (hipEvent_t e : x->ev_up) (void)hipEventDestroy(e);
    for (hipEvent_t e : x->ev_done) (void)hipEventDestroy(e);
    for (hipEvent_t e : x->ev_down) (void)hipEventDestroy(e);
    if (x->s_h2d) (void)hipStreamDestroy(x->s_h2d);
    if (x->s_d2h) (void)hipStreamDestroy(x->s_d2h);
    if (x->ev0) (void)hipEventDestroy(x->ev0);
    if (x->ev1) (void)hipEventDestroy(x->ev1);
    if (x->own_stream && x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}
SL3D_CATCH_VOID

// ---- stage 7 per-scan constants (host, double) ---------------------------------------------------
// cvRodrigues2 (vector -> matrix): theta = |r|; R = cos*I + (1-cos)*rr^T + sin*[r]x   (7/triangulation.cpp:1072,1080)
static void rodrigues(const double r[3], double R[9])
{
    double rx = r[0], ry = r[1], rz = r[2];
    const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
    rx *= it; ry *= it; rz *= it;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * (k % 4 == 0 ? 1.0 : 0.0) + c1 * rrt[k] + s * rx_[k];
}

// A = K * [R|t]   (compute_A, 7/triangulation.cpp:1090-1116)
static void projection_matrix(const double K[9], const double rvec[3], const double tvec[3], double A[12])
{
    double R[9], Rt[12];
    rodrigues(rvec, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rt[i * 4 + j] = R[i * 3 + j];
        Rt[i * 4 + 3] = tvec[i];
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += K[i * 3 + k] * Rt[k * 4 + j];
            A[i * 4 + j] = acc;
        }
}

static void fill_intr(Intr &I, const double K[9], const double d[5])
{
    memcpy(I.K, K, sizeof I.K);
    I.ifx = 1.0 / K[0]; I.ify = 1.0 / K[4]; I.cx = K[2]; I.cy = K[5];
    I.k1 = d[0]; I.k2 = d[1]; I.p1 = d[2]; I.p2 = d[3]; I.k3 = d[4];
    I.has_dist = (d[0] != 0 || d[1] != 0 || d[2] != 0 || d[3] != 0 || d[4] != 0) ? 1 : 0;
    I.affine = (K[6] == 0 && K[7] == 0 && K[8] == 1) ? 1 : 0;
    I.has_tan = (d[2] != 0 || d[3] != 0) ? 1 : 0;
    I.plain = (I.affine && K[1] == 0 && K[3] == 0) ? 1 : 0;
    I.identity = (I.plain && !I.has_dist) ? 1 : 0;
}

extern "C" int sl3d_set_calibration(sl3d_ctx *x, const double Kc[9], const double dc[5], const double rc[3], const double tc[3],
                                    const double Kp[9], const double dp[5], const double rp[3], const double tp[3])
try {
    if (!x || !Kc || !dc || !rc || !tc || !Kp || !dp || !rp || !tp) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    if (Kc[0] == 0 || Kc[4] == 0 || Kp[0] == 0 || Kp[4] == 0) return fail(x, SL3D_E_INVALID_ARG, "zero focal length");
    projection_matrix(Kc, rc, tc, x->C.Ac);
    projection_matrix(Kp, rp, tp, x->C.Ap);
    fill_intr(x->C.cam, Kc, dc);
    memcpy(x->Kc_raw, Kc, sizeof x->Kc_raw);
    memcpy(x->dc_raw, dc, sizeof x->dc_raw);
    x->raw_map_valid = false;
    fill_intr(x->C.proj, Kp, dp);
    rodrigues(rc, x->S.Rc);
    rodrigues(rp, x->S.Rp);
    {   // camera-frame form (DevCal): Apc = [Ap3*Rc^T | ap4 - Ap3*Rc^T*tc], Rct = Rc^T, tcn = -Rc^T*tc
        DevCal &C = x->C;
        const double *R = x->S.Rc;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += C.Ap[i * 4 + k] * R[j * 3 + k];  // (Ap3 * Rc^T)[i][j]
                C.Apc[i * 4 + j] = acc;
            }
            double acc = C.Ap[i * 4 + 3];
            for (int j = 0; j < 3; j++) acc -= C.Apc[i * 4 + j] * tc[j];
            C.Apc[i * 4 + 3] = acc;
        }
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) C.Rct[i * 3 + j] = R[j * 3 + i];
            C.tcn[i] = -(R[0 * 3 + i] * tc[0] + R[1 * 3 + i] * tc[1] + R[2 * 3 + i] * tc[2]);
        }
        C.fx2 = Kc[0] * Kc[0];
        C.fy2 = Kc[1] * Kc[1] + Kc[4] * Kc[4];
        C.fxs = Kc[0] * Kc[1];
    }
    memcpy(x->S.tc, tc, sizeof x->S.tc);
    memcpy(x->S.tp, tp, sizeof x->S.tp);
    memcpy(x->S.Kp, Kp, sizeof x->S.Kp);
    ON_DEVICE(x);
    HIPCHK(x, hipStreamSynchronize(x->stream));  // no launch may still be reading the previous constants
    HIPCHK(x, hipMemcpy(x->d_cal, &x->C, sizeof(DevCal), hipMemcpyHostToDevice));
    // rig class of the timed fused kernel (pixel_chain): 1 = the reference's kind of calibration, 2 = distorted projector
    // behind a per-calibration undistortion table, 3 = a plain projector K with a purely radial model (a 4-KB table of the
    // radial factor, in LDS; 3-step fringes), 0 = everything else, evaluated in the kernel
    // (camera-frame solve: any upper-triangular affine camera matrix -- a skew term included; only a K with a perspective row
    // or a non-zero K[1][0] is left to the general kernel)
    const bool cam_frame_ok = x->C.cam.affine && Kc[3] == 0.0;
    x->rig = !cam_frame_ok ? 0 : x->C.proj.identity ? 1 : 2;
    if (x->rig == 2 && !x->keep && x->P.F == 3 && x->C.proj.plain && !x->C.proj.has_tan) x->rig = 3;
#ifdef SL3D_MEASURE
    if (x->rig == 3 && getenv("SL3D_NO_RIG3")) x->rig = 2;
#endif
    x->P.proj_disp = nullptr;
    x->P.proj_rad = nullptr;
    x->P.cam_tab = nullptr;
    x->P.cam_tab_kind = 0;
    if (!x->keep && x->C.cam.has_dist) {  // timed mode: T1 of the camera per window pixel (k_cam_table)
        const int kind = x->C.cam.has_tan ? 2 : 1;
        const size_t want = (size_t)kind * x->P.px_view_stride;
        if (!x->d_cam_tab || x->cam_tab_doubles < want) {
            if (x->d_cam_tab) {  // a radial-only table that has to grow into a two-double one (no launch reads it: synchronised above)
                (void)hipFree(x->d_cam_tab);
                x->allocs.erase(std::remove(x->allocs.begin(), x->allocs.end(), (void *)x->d_cam_tab), x->allocs.end());
                x->d_cam_tab = nullptr;
                x->cam_tab_doubles = 0;
            }
            const int st = dev_alloc(x, &x->d_cam_tab, want);
            if (st) return st;
            x->cam_tab_doubles = want;
        }
        const int st = launch_cam_table(x->P, x->d_cal, kind, x->d_cam_tab, x->stream);
        if (st) return fail(x, SL3D_E_HIP, std::string("k_cam_table: ") + hipGetErrorString((hipError_t)st));
        x->P.cam_tab = x->d_cam_tab;
        x->P.cam_tab_kind = kind;
    }
    if (x->rig == 3) {
        // the radial factor over r0^2 in [0, r2max]: r2max from the projector pixel farthest from the principal point
        if (!x->d_proj_rad) {
            HIPCHK(x, hipMalloc((void **)&x->d_proj_rad, (size_t)SL3D_RAD_COPIES * SL3D_RAD_STRIDE * sizeof(RadEntry)));
            x->allocs.push_back(x->d_proj_rad);
        }
        const Intr &I = x->C.proj;
        const double ex = std::max(std::fabs(0.0 - I.cx), std::fabs((double)(x->cfg.proj_width - 1) - I.cx)) * std::fabs(I.ifx);
        const double ey = std::max(std::fabs(0.0 - I.cy), std::fabs((double)(x->cfg.proj_height - 1) - I.cy)) * std::fabs(I.ify);
        const double r2max = (ex * ex + ey * ey) * (1.0 + 1e-9) + 1e-300;
        const int st = launch_radial_table(x->d_cal, 1, r2max, x->d_proj_rad, x->stream);
        if (st) return fail(x, SL3D_E_HIP, std::string("k_radial_table: ") + hipGetErrorString((hipError_t)st));
        HIPCHK(x, hipStreamSynchronize(x->stream));
        x->P.proj_rad = x->d_proj_rad;
        x->P.proj_rad_scale = (float)((SL3D_RAD_NODES - 1) / r2max);
    } else if (!x->C.proj.identity && !x->keep) {  // a distorted projector (rig 2, or rig 0 in the timed mode)
        if (!x->d_proj_disp) {
            HIPCHK(x, hipMalloc((void **)&x->d_proj_disp, (size_t)x->cfg.proj_width * x->cfg.proj_height * sizeof(float2)));
            x->allocs.push_back(x->d_proj_disp);
        }
        const int st = launch_proj_table(x->d_cal, x->cfg.proj_width, x->cfg.proj_height, x->d_proj_disp, x->stream);
        if (st) return fail(x, SL3D_E_HIP, std::string("k_proj_table: ") + hipGetErrorString((hipError_t)st));
        HIPCHK(x, hipStreamSynchronize(x->stream));
        x->P.proj_disp = x->d_proj_disp;
    }
    x->have_cal = true;
    return SL3D_OK;
}
SL3D_CATCH(x)

extern "C" int sl3d_get_projection_matrices(sl3d_ctx *x, double A_cam[12], double A_proj[12])
try {
    if (!x || !A_cam || !A_proj) return fail(x, SL3D_E_INVALID_ARG, "null argument");
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called");
    memcpy(A_cam, x->C.Ac, sizeof x->C.Ac);
    memcpy(A_proj, x->C.Ap, sizeof x->C.Ap);
    return SL3D_OK;
}
SL3D_CATCH(x)

static int launched(sl3d_ctx *x, int hip_err);
static int need_keep(sl3d_ctx *x);

// Quads of `view` that hold a valid pixel, as k_mask_prepare counted them: the sum of the view's per-block words, known only once
// every block carries the sequence number of the view's last preparation (the blocks store straight into this host array: no copy
// behind the kernel, no wait here; a word of an earlier preparation cannot pass for a current one).
static bool quads_known(const sl3d_ctx *x, int view, unsigned *quads)
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
    *quads = sum;
    return true;
}

// true if every view of [first, first + n) is KNOWN to be sparsely selected (fewer than 65 % of its quads hold a valid pixel): a
// launch over such views takes the instantiation whose every plane request waits for the valid bits (choose_fused: the large-launch
// kernel without early requests, also for a small launch).  Unknown (the count has not landed, or no mask was ever set) counts as
// dense: that is the default this library was tuned on.
static bool sparse_views(const sl3d_ctx *x, int first, int n)
{
    const double quads = (double)(x->P.pitch >> 2) * (double)x->P.H;
    for (int v = first; v < first + n; v++) {
        unsigned c;
        if (!quads_known(x, v, &c) || (double)c >= 0.65 * quads) return false;
    }
    return true;
}

static int check_view(sl3d_ctx *x, int view, int n = 1)
{
    if (!x) return SL3D_E_INVALID_ARG;
    if (view < 0 || n < 1 || view + n > x->cfg.max_views) return fail(x, SL3D_E_INVALID_ARG, "view index out of range");
    return SL3D_OK;
}

// what kind of memory `p` is: 0 = pageable host memory (unknown to the runtime: a copy from it is consumed before the call returns),
// 1 = pinned host memory (hipHostMalloc / hipHostRegister: copies from it are asynchronous DMA, the caller owns the hand-over, see
// include/sl3d.h), 2 = device memory (*device = its ordinal)
static int memory_kind(const void *p, int *device = nullptr)
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
static bool is_pinned_host(const void *p) { return memory_kind(p) == 1; }

// the staging plane(s) of sl3d_set_mask(s): `slots` planes, zero outside the region the copies fill
static int flush_masks(sl3d_ctx *x, int first_view, int n_views, bool callers_only = false);
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

// the part of the mask plane that holds source pixels: the window + 2-pixel halo, clipped to the frame
struct MaskRegion {
    int gx0, gx1, gy0, gy1;
};
static MaskRegion mask_region(const KParams &P, MaskSrc &S)
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
    const unsigned seq = ++x->mask_seq;
    for (int v = first_view; v < first_view + n_views; v++) {
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
static int flush_masks(sl3d_ctx *x, int first_view, int n_views, bool callers_only)
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
static int supersede_masks(sl3d_ctx *x, int first_view, int n_views, int slots)
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
static int sync_for_caller(sl3d_ctx *x)
{
    const int rc = flush_masks(x, 0, x->cfg.max_views, true);
    if (rc) return rc;
    HIPCHK(x, hipStreamSynchronize(x->stream));
    return SL3D_OK;
}
#define SYNC_FOR_CALLER(x)                     \
    do {                                       \
        const int rc_ = sync_for_caller(x);    \
        if (rc_) return rc_;                   \
    } while (0)

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
    ON_DEVICE(x);
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

static int ensure_colrow(sl3d_ctx *x, size_t bytes)
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

// ---- compute -------------------------------------------------------------------------------------
static int need_keep(sl3d_ctx *x)
{
    if (!x->keep) return fail(x, SL3D_E_STATE, "context was created without SL3D_FLAG_KEEP_STAGES");
    return SL3D_OK;
}

static int launched(sl3d_ctx *x, int hip_err)
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
// deferred, in one layout, the launch has a MASKIN instantiation, and the views are not known -- by their LAST counts -- to be
// sparsely selected (a MASKIN launch requests and computes every pixel of the window before it knows the selection; sparse views keep
// the two-kernel route whose plane requests wait for the valid bits)
static bool maskin_launch(const sl3d_ctx *x, int first_view, int n_views, bool keep, bool prefer_gated)
{
    if (x->n_pending == 0 || prefer_gated || !fused_maskin_available(x->P, x->rig, n_views, keep)) return false;
    const sl3d_ctx::PendingMask &p0 = x->pend[(size_t)first_view];
    for (int v = first_view; v < first_view + n_views; v++) {
        const sl3d_ctx::PendingMask &pm = x->pend[(size_t)v];
        if (!pm.pending || pm.stride != p0.stride || pm.lo != p0.lo || pm.hi != p0.hi) return false;
    }
    return true;
}

static int run_fused(sl3d_ctx *x, int first_view, int n_views, bool keep, int cmode)
{
    const bool prefer_gated = sparse_views(x, first_view, n_views);
    const bool maskin = maskin_launch(x, first_view, n_views, keep, prefer_gated);
    x->last_fused.n_views = n_views;
    x->last_fused.cmode = cmode;
    x->last_fused.keep = keep;
    x->last_fused.prefer_gated = prefer_gated;
    x->last_fused.maskin = maskin;
    if (!maskin) {
        const int rc = flush_masks(x, first_view, n_views);
        if (rc) return rc;
        return launched(x, launch_fused(x->P, x->d_cal, x->rig, first_view, n_views, keep, cmode, x->stream, prefer_gated));
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
    return launched(x, launch_fused(x->P, x->d_cal, x->rig, first_view, n_views, keep, cmode, x->stream, false, &mi));
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
    ON_DEVICE(x);
    return run_fused(x, first_view, n_views, x->keep, 0);
}
SL3D_CATCH(x)

// the k_fused instantiation sl3d_run / sl3d_run_clouds launches for a batch of n_views views of this context, as rocprofv3 spells it
extern "C" int sl3d_fused_kernel_name(sl3d_ctx *x, int n_views, int clouds, char *buf, size_t capacity)
try {
    if (!x || !buf || capacity == 0 || n_views < 1) return fail(x, SL3D_E_INVALID_ARG, "fused_kernel_name: null argument");
    if (!x->have_cal) return fail(x, SL3D_E_STATE, "sl3d_set_calibration has not been called (the rig class is part of the name)");
    const bool fits = n_views <= x->cfg.max_views, gated = fits && sparse_views(x, 0, n_views);
    const int n = fused_kernel_name(x->P, x->rig, n_views, x->keep, clouds ? 2 : 0, buf, capacity, gated, fits && maskin_launch(x, 0, n_views, x->keep, gated));
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
    ON_DEVICE(x);
    rc = ensure_cloud_buffers(x);
    if (rc) return rc;
    rc = run_fused(x, first_view, n_views, false, 2);
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
