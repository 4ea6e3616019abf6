// sl3d_capi_context.cpp -- the context behind the C ABI of include/sl3d.h: errors and the exception barrier, sl3d_create / sl3d_destroy (the HBM
// layout: frame stack, mask with halo, dense results, optional stage planes; the device-side proof of the lattice atan2), and the per-scan
// constants of stage 7 (Rodrigues, A = K*[R|t]: 7/triangulation.cpp:1061-1126) with the per-calibration tables.  All compute happens in the
// HIP kernels; there is no CPU implementation of the path in this library.
#include "sl3d_capi_internal.h"

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
    // (launch lanes: only beside a stream nobody else can see, never in the parity mode; SL3D_NO_LAUNCH_LANES=1 in the environment
    // switches them off for a whole process -- A/B measurements of programs that cannot pass the flag)
    x->lanes_ok = x->own_stream && !(c.flags & (SL3D_FLAG_SERIAL_LAUNCHES | SL3D_FLAG_KEEP_STAGES)) && !getenv("SL3D_NO_LAUNCH_LANES");

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
    x->quad_last.assign(V, ~0u);
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
    for (int l = 0; l < 2; l++)
        if (x->lane[l]) (void)hipStreamSynchronize(x->lane[l]);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    for (int l = 0; l < 2; l++) {
        if (x->ev_lane[l]) (void)hipEventDestroy(x->ev_lane[l]);
        if (x->lane[l]) (void)hipStreamDestroy(x->lane[l]);
    }
    if (x->ev_main) (void)hipEventDestroy(x->ev_main);
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
