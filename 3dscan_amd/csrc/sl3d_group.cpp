// sl3d_group.cpp -- row-stripe groups: several GPUs behind the reference's single-process caller (include/sl3d.h,
// "several GPUs behind one caller").
//
// A group is n ordinary contexts, stripe i on HIP device devices[i], plus the assembled results on the root GPU (stripe 0's
// device).  The stripes never talk to each other while they compute (the 2 halo rows they need are rows of the INPUT mask);
// the assembly is ONE RCCL group of ncclSend / ncclRecv per call -- a gather over xGMI whose messages land in place in the
// root's dense [view][row] planes -- or plain (peer) device copies for stripes that share the root's GPU.  RCCL is bound at
// run time (dlopen of librccl.so.1: the process may already hold torch's copy of it); a group that has to cross GPUs without
// it falls back to hipMemcpyPeerAsync, and says so in sl3d_group_transport().
//
// What a one-GPU box can and cannot execute of this file (DESIGN.md section 7 has the line-by-line list): with
// SL3D_FLAG_GROUP_DISTINCT_SIDES every stripe gets its own GpuSide (communication stream + event) although the devices repeat,
// so every `S.gpu != 0` branch, the peer-copy transport and -- through a test double of librccl selected with SL3D_RCCL_LIB
// (tests/native/fake_rccl.cpp) -- the N-rank send/recv pairing, the 256-message group split and the variable-size cloud gather
// run with N > 1 sides.  The real ncclCommInitAll over several devices, xGMI, and contexts on device != 0 stay unexecuted.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "sl3d_ctx.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool ok = false;
};

// allow_override: the group was created with SL3D_FLAG_GROUP_DISTINCT_SIDES (the flag of the one-GPU tests).  Only such a group honours
// SL3D_RCCL_LIB -- another library with the same seven entry points (a test double that pairs sends with receives itself, so that the
// N-rank exchange runs on one GPU); a production group never loads a library an environment variable names.  TWO bindings, chosen
// per group by its flag: which library a group talks to does not depend on what kind of group the process happened to create first
// (ADVICE r5).
RcclApi &rccl(bool allow_override)
{
    static RcclApi bindings[2];
    static std::once_flag onces[2];
    RcclApi &R = bindings[allow_override ? 1 : 0];
    std::once_flag &once = onces[allow_override ? 1 : 0];
    std::call_once(once, [allow_override, &R] {
        const char *override_lib = allow_override ? getenv("SL3D_RCCL_LIB") : nullptr;
        if (override_lib && *override_lib) {
            R.lib = dlopen(override_lib, RTLD_NOW | RTLD_LOCAL);
            if (!R.lib) {
                R.err = std::string("dlopen(SL3D_RCCL_LIB=") + override_lib + "): " + dlerror();
                return;
            }
        }
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            if (R.lib) break;
            R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!R.lib) {
            R.err = std::string("dlopen(librccl.so.1): ") + dlerror();
            return;
        }
#define SYM(field, name)                                                     \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.lib, name));       \
    if (!R.field) {                                                          \
        R.err = std::string("librccl: missing symbol ") + name;              \
        return;                                                              \
    }
        SYM(CommInitAll, "ncclCommInitAll")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd")
        SYM(Send, "ncclSend")
        SYM(Recv, "ncclRecv")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        R.ok = true;
    });
    return R;
}

struct GpuSide {  // one per distinct device of the group (one per STRIPE with SL3D_FLAG_GROUP_DISTINCT_SIDES)
    int device = 0;
    hipStream_t comm = nullptr;    // communication stream of this GPU
    hipEvent_t ev_comm = nullptr;  // the last exchange enqueued on it
    ncclComm_t nccl = nullptr;     // rank `index in gpus` of the group's communicator
};

struct Stripe {
    sl3d_ctx *ctx = nullptr;
    int row0 = 0, rows = 0, device = 0, gpu = 0;
    hipEvent_t ev_run = nullptr;  // the stripe's last kernel
};

}  // namespace

struct sl3d_group {
    sl3d_config cfg{};
    std::vector<Stripe> st;
    std::vector<GpuSide> gpus;  // gpus[0] = the root's
    bool use_rccl = false, force_rccl = false, distinct_sides = false;
    std::string err;
    int pitch = 0;
    size_t px_view_stride = 0;   // pitch * height: elements per view of the assembled planes
    float *d_points = nullptr;   // root: [view][height][pitch][3]
    uint8_t *d_valid = nullptr;  // root: [view][height][pitch]
    float *d_cloud = nullptr;    // root: [view][px_view_stride][3] compacted clouds (allocated on first use)
    std::vector<int64_t> cloud_count;  // per view, after sl3d_group_gather_clouds
    bool comm_busy = false;
    int busy_first = 0, busy_n = 0;  // the views the communication streams may still be reading
};

static int gfail(sl3d_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    else sl3d_fail(nullptr, code, msg);
    return code;
}

// the exception barrier of the group entry points (sl3d_ctx.h): the text goes to the group's own last error
#define SL3D_GROUP_CATCH(g) catch (...) { sl3d_group *g_ = (g); return sl3d_caught(nullptr, g_ ? &g_->err : nullptr); }

#define GHIP(g, call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return gfail((g), SL3D_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define GNCCL(g, call)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) return gfail((g), SL3D_E_HIP, std::string(#call) + ": " + rccl((g)->distinct_sides).GetErrorString(r_)); \
    } while (0)
// a failing member context: its message becomes the group's
#define GCTX(g, s, call)                                                                     \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != SL3D_OK)                                                                  \
            return gfail((g), rc_, "stripe " + std::to_string(s) + ": " + sl3d_last_error((g)->st[(size_t)(s)].ctx)); \
    } while (0)

extern "C" const char *sl3d_group_last_error(const sl3d_group *g) { return g ? g->err.c_str() : sl3d_last_error(nullptr); }
extern "C" int sl3d_group_size(const sl3d_group *g) { return g ? (int)g->st.size() : 0; }
extern "C" const char *sl3d_group_transport(const sl3d_group *g) { return g && g->use_rccl ? "rccl" : "copy"; }

extern "C" void sl3d_group_destroy(sl3d_group *g)
try {
    if (!g) return;
    for (auto &s : g->st)
        if (s.ctx) {
            DeviceGuard dg(s.device);
            (void)hipStreamSynchronize(s.ctx->stream);
        }
    for (auto &u : g->gpus) {
        DeviceGuard dg(u.device);
        if (u.comm) (void)hipStreamSynchronize(u.comm);
        if (u.nccl) (void)rccl(g->distinct_sides).CommDestroy(u.nccl);
        if (u.ev_comm) (void)hipEventDestroy(u.ev_comm);
        if (u.comm) (void)hipStreamDestroy(u.comm);
    }
    for (auto &s : g->st) {
        if (s.ev_run) {
            DeviceGuard dg(s.device);
            (void)hipEventDestroy(s.ev_run);
        }
        if (s.ctx) sl3d_destroy(s.ctx);
    }
    if (!g->gpus.empty()) {
        DeviceGuard dg(g->gpus[0].device);
        if (g->d_points) (void)hipFree(g->d_points);
        if (g->d_valid) (void)hipFree(g->d_valid);
        if (g->d_cloud) (void)hipFree(g->d_cloud);
    }
    delete g;
}
SL3D_CATCH_VOID

extern "C" int sl3d_group_create(const sl3d_config *cfg, const int *devices, int n, sl3d_group **out)
try {
    if (!cfg || !devices || !out || n < 1) return gfail(nullptr, SL3D_E_INVALID_ARG, "group_create: null argument or no stripes");
    *out = nullptr;
    if (cfg->height < n) return gfail(nullptr, SL3D_E_INVALID_ARG, "group_create: more stripes than rows");
    if (cfg->stream) return gfail(nullptr, SL3D_E_INVALID_ARG, "group_create: every stripe creates its own stream (cfg->stream must be NULL)");
    sl3d_group *g = new sl3d_group();
    g->cfg = *cfg;
    if (g->cfg.full_width == 0) g->cfg.full_width = cfg->width;
    if (g->cfg.full_height == 0) g->cfg.full_height = cfg->height;
    if (g->cfg.max_views <= 0) g->cfg.max_views = 1;
    g->force_rccl = (cfg->flags & SL3D_FLAG_GROUP_FORCE_RCCL) != 0;
    g->distinct_sides = (cfg->flags & SL3D_FLAG_GROUP_DISTINCT_SIDES) != 0;
    const int base = cfg->height / n, rem = cfg->height % n;
    auto bail = [&](int code, const std::string &msg) {
        sl3d_group_destroy(g);
        return gfail(nullptr, code, msg);
    };
    g->st.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        Stripe &s = g->st[(size_t)i];
        s.rows = base + (i < rem ? 1 : 0);
        s.row0 = i * base + std::min(i, rem);
        s.device = devices[i];
        int gi = -1;
        for (size_t k = 0; k < g->gpus.size() && !g->distinct_sides; k++)
            if (g->gpus[k].device == s.device) gi = (int)k;
        if (gi < 0) {
            GpuSide u;
            u.device = s.device;
            g->gpus.push_back(u);
            gi = (int)g->gpus.size() - 1;
        }
        s.gpu = gi;
        sl3d_config c = g->cfg;
        c.height = s.rows;
        c.row0 = g->cfg.row0 + s.row0;
        c.device = s.device;
        c.stream = nullptr;
        // (a stripe's launches stay on its one stream: the group orders its communication behind them by events on that stream)
        c.flags = (cfg->flags & SL3D_FLAG_KEEP_STAGES) | SL3D_FLAG_SERIAL_LAUNCHES;
        const int rc = sl3d_create(&c, &s.ctx);
        if (rc != SL3D_OK) return bail(rc, "stripe " + std::to_string(i) + ": " + sl3d_last_error(nullptr));
        DeviceGuard dg(s.device);
        if (hipEventCreateWithFlags(&s.ev_run, hipEventDisableTiming) != hipSuccess) return bail(SL3D_E_HIP, "hipEventCreate failed");
    }
    for (auto &u : g->gpus) {
        DeviceGuard dg(u.device);
        if (dg.err != hipSuccess || hipStreamCreateWithFlags(&u.comm, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&u.ev_comm, hipEventDisableTiming) != hipSuccess)
            return bail(SL3D_E_HIP, "communication stream / event creation failed on device " + std::to_string(u.device));
    }
    // transport: RCCL as soon as a stripe lives on another GPU than the root (or when asked for), unless it cannot be loaded
    const bool wants_rccl = ((g->gpus.size() > 1) || g->force_rccl) && !(cfg->flags & SL3D_FLAG_GROUP_NO_RCCL);
    if (wants_rccl) {
        RcclApi &R = rccl((cfg->flags & SL3D_FLAG_GROUP_DISTINCT_SIDES) != 0);
        if (!R.ok) {
            if (g->force_rccl) return bail(SL3D_E_UNSUPPORTED, "RCCL requested but unavailable: " + R.err);
        } else {
            std::vector<int> devs;
            for (auto &u : g->gpus) devs.push_back(u.device);
            std::vector<ncclComm_t> comms(devs.size(), nullptr);
            (void)hipGetLastError();  // RCCL reports any stale (non-fatal) HIP error of this thread as its own: start clean
            const ncclResult_t r = R.CommInitAll(comms.data(), (int)devs.size(), devs.data());
            if (r != ncclSuccess)
                return bail(SL3D_E_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(r) +
                                            " (a process that also loads torch must import it BEFORE this library: one ROCm stack per process)");
            for (size_t k = 0; k < devs.size(); k++) g->gpus[k].nccl = comms[k];
            g->use_rccl = true;
        }
    }
    if (!g->use_rccl && g->gpus.size() > 1) {  // peer copies: let the root's GPU reach the others where the platform allows
        DeviceGuard dg(g->gpus[0].device);
        for (size_t k = 1; k < g->gpus.size(); k++) {
            int can = 0;
            if (g->gpus[k].device == g->gpus[0].device) continue;  // (distinct sides on one device: nothing to enable)
            if (hipDeviceCanAccessPeer(&can, g->gpus[0].device, g->gpus[k].device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(g->gpus[k].device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    const KParams &P0 = g->st[0].ctx->P;
    g->pitch = P0.pitch;
    g->px_view_stride = (size_t)P0.pitch * (size_t)g->cfg.height;
    {
        DeviceGuard dg(g->gpus[0].device);
        const size_t V = (size_t)g->cfg.max_views;
        hipError_t e = hipMalloc((void **)&g->d_points, V * g->px_view_stride * 3 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&g->d_valid, V * g->px_view_stride);
        if (e != hipSuccess) return bail(e == hipErrorOutOfMemory ? SL3D_E_NOMEM : SL3D_E_HIP, std::string("hipMalloc (assembled planes): ") + hipGetErrorString(e));
    }
    g->cloud_count.assign((size_t)g->cfg.max_views, 0);
    *out = g;
    return SL3D_OK;
}
SL3D_GROUP_CATCH(nullptr)

extern "C" int sl3d_group_stripe(sl3d_group *g, int i, int *row0, int *rows, int *device, sl3d_ctx **ctx)
try {
    if (!g || i < 0 || i >= (int)g->st.size()) return gfail(g, SL3D_E_INVALID_ARG, "group_stripe: index out of range");
    const Stripe &s = g->st[(size_t)i];
    if (row0) *row0 = s.row0;
    if (rows) *rows = s.rows;
    if (device) *device = s.device;
    if (ctx) *ctx = s.ctx;
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_set_calibration(sl3d_group *g, const double Kc[9], const double dc[5], const double rc[3], const double tc[3],
                                          const double Kp[9], const double dp[5], const double rp[3], const double tp[3])
try {
    if (!g) return SL3D_E_INVALID_ARG;
    for (size_t s = 0; s < g->st.size(); s++) GCTX(g, s, sl3d_set_calibration(g->st[s].ctx, Kc, dc, rc, tc, Kp, dp, rp, tp));
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_set_mask(sl3d_group *g, int view, const uint8_t *m, size_t stride)
try {
    if (!g) return SL3D_E_INVALID_ARG;
    for (size_t s = 0; s < g->st.size(); s++) GCTX(g, s, sl3d_set_mask(g->st[s].ctx, view, m, stride));
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_set_frames(sl3d_group *g, int view, int axis, const uint8_t *const *planes, int n_planes, size_t stride)
try {
    if (!g || !planes || n_planes < 1) return gfail(g, SL3D_E_INVALID_ARG, "group_set_frames: null argument");
    std::vector<const uint8_t *> sub((size_t)n_planes);
    for (size_t s = 0; s < g->st.size(); s++) {
        // a stripe's rows are a contiguous byte range of every row-major plane
        for (int i = 0; i < n_planes; i++) {
            if (!planes[i]) return gfail(g, SL3D_E_INVALID_ARG, "group_set_frames: null plane");
            sub[(size_t)i] = planes[i] + (size_t)g->st[s].row0 * stride;
        }
        GCTX(g, s, sl3d_set_frames(g->st[s].ctx, view, axis, sub.data(), n_planes, stride));
    }
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

static bool overlaps(const sl3d_group *g, int first, int n) { return g->comm_busy && first < g->busy_first + g->busy_n && g->busy_first < first + n; }

static int check_views(sl3d_group *g, int first, int n)
{
    if (!g) return SL3D_E_INVALID_ARG;
    if (first < 0 || n < 1 || first + n > g->cfg.max_views) return gfail(g, SL3D_E_INVALID_ARG, "view range out of bounds");
    return SL3D_OK;
}

template <typename RunFn>
static int group_launch(sl3d_group *g, int first, int n, RunFn run)
{
    int rc = check_views(g, first, n);
    if (rc) return rc;
    for (size_t s = 0; s < g->st.size(); s++) {
        Stripe &S = g->st[s];
        DeviceGuard dg(S.device);
        // results the communication streams may still be reading are not overwritten.  Who reads a stripe depends on the
        // transport: an RCCL send sits on the communication stream of the stripe's OWN GPU, a (peer) device copy on the
        // ROOT's (group_exchange) -- the stripe waits for both, so the rule cannot go stale when a transport is added
        if (overlaps(g, first, n)) {
            GHIP(g, hipStreamWaitEvent(S.ctx->stream, g->gpus[(size_t)S.gpu].ev_comm, 0));
            if (S.gpu != 0) GHIP(g, hipStreamWaitEvent(S.ctx->stream, g->gpus[0].ev_comm, 0));
        }
        GCTX(g, s, run(S.ctx));
        GHIP(g, hipEventRecord(S.ev_run, S.ctx->stream));
    }
    return SL3D_OK;
}

extern "C" int sl3d_group_run(sl3d_group *g, int first, int n)
try {
    return group_launch(g, first, n, [&](sl3d_ctx *c) { return sl3d_run(c, first, n); });
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_run_clouds(sl3d_group *g, int first, int n)
try {
    return group_launch(g, first, n, [&](sl3d_ctx *c) { return sl3d_run_clouds(c, first, n); });
}
SL3D_GROUP_CATCH(g)

namespace {
struct Xfer {  // one contiguous message: stripe -> root
    int stripe;
    const void *src;
    void *dst;
    size_t count;  // elements
    ncclDataType_t type;
    size_t bytes() const { return count * (type == ncclFloat ? 4u : 1u); }
};
}  // namespace

// one exchange: every communication stream first waits for the kernels whose results it moves, then all messages go out as
// ONE RCCL group (send on the stripe's GPU, receive on the root's), copies for the stripes that need none
static int group_exchange(sl3d_group *g, const std::vector<Xfer> &xs)
{
    GpuSide &root = g->gpus[0];
    for (size_t s = 0; s < g->st.size(); s++) {
        Stripe &S = g->st[s];
        const bool by_rccl = g->use_rccl && (S.gpu != 0 || (g->force_rccl && s != 0));
        GpuSide &u = g->gpus[(size_t)S.gpu];
        DeviceGuard dg(by_rccl ? u.device : root.device);
        GHIP(g, hipStreamWaitEvent(by_rccl ? u.comm : root.comm, S.ev_run, 0));
    }
    bool any_rccl = false;
    for (const Xfer &x : xs) {
        const Stripe &S = g->st[(size_t)x.stripe];
        if (g->use_rccl && (S.gpu != 0 || (g->force_rccl && x.stripe != 0))) any_rccl = true;
    }
    if (any_rccl) GNCCL(g, rccl(g->distinct_sides).GroupStart());
    // an error between GroupStart and GroupEnd must not leave this thread inside an open RCCL group
    struct GroupCloser {
        bool open, override_;
        ~GroupCloser() { if (open) (void)rccl(override_).GroupEnd(); }
    } closer{any_rccl, g->distinct_sides};
    int in_group = 0;
    for (const Xfer &x : xs) {
        if (x.count == 0) continue;
        const Stripe &S = g->st[(size_t)x.stripe];
        const bool by_rccl = g->use_rccl && (S.gpu != 0 || (g->force_rccl && x.stripe != 0));
        if (by_rccl) {
            GpuSide &u = g->gpus[(size_t)S.gpu];
            if (in_group == 256) {  // a very large batch goes out as several RCCL groups (same order on both sides)
                GNCCL(g, rccl(g->distinct_sides).GroupEnd());
                GNCCL(g, rccl(g->distinct_sides).GroupStart());
                in_group = 0;
            }
            in_group++;
            GNCCL(g, rccl(g->distinct_sides).Send(x.src, x.count, x.type, 0, u.nccl, u.comm));
            GNCCL(g, rccl(g->distinct_sides).Recv(x.dst, x.count, x.type, S.gpu, root.nccl, root.comm));
        } else if (S.gpu == 0) {
            DeviceGuard dg(root.device);
            GHIP(g, hipMemcpyAsync(x.dst, x.src, x.bytes(), hipMemcpyDeviceToDevice, root.comm));
        } else {
            DeviceGuard dg(root.device);
            GHIP(g, hipMemcpyPeerAsync(x.dst, root.device, x.src, S.device, x.bytes(), root.comm));
        }
    }
    closer.open = false;
    if (any_rccl) GNCCL(g, rccl(g->distinct_sides).GroupEnd());
    for (auto &u : g->gpus) {
        DeviceGuard dg(u.device);
        GHIP(g, hipEventRecord(u.ev_comm, u.comm));
    }
    return SL3D_OK;
}

static void mark_busy(sl3d_group *g, int first, int n)
{
    if (g->comm_busy) {  // keep one covering range
        const int a = std::min(g->busy_first, first), b = std::max(g->busy_first + g->busy_n, first + n);
        g->busy_first = a;
        g->busy_n = b - a;
    } else {
        g->comm_busy = true;
        g->busy_first = first;
        g->busy_n = n;
    }
}

extern "C" int sl3d_group_gather(sl3d_group *g, int first, int n)
try {
    int rc = check_views(g, first, n);
    if (rc) return rc;
    std::vector<Xfer> xs;
    xs.reserve(g->st.size() * (size_t)n * 2);
    const size_t pitch = (size_t)g->pitch;
    for (int v = first; v < first + n; v++)
        for (size_t s = 0; s < g->st.size(); s++) {
            const Stripe &S = g->st[s];
            const KParams &P = S.ctx->P;
            // a stripe of a view is a contiguous slab of the root's dense plane of that view
            const size_t dst_px = (size_t)v * g->px_view_stride + (size_t)S.row0 * pitch, cnt = (size_t)S.rows * pitch;
            xs.push_back({(int)s, P.points + 3 * (size_t)v * P.px_view_stride, g->d_points + 3 * dst_px, cnt * 3, ncclFloat});
            xs.push_back({(int)s, P.valid + (size_t)v * P.px_view_stride, g->d_valid + dst_px, cnt, ncclUint8});
        }
    rc = group_exchange(g, xs);
    if (rc) return rc;
    mark_busy(g, first, n);
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

static int root_sync(sl3d_group *g)
{
    DeviceGuard dg(g->gpus[0].device);
    GHIP(g, hipStreamSynchronize(g->gpus[0].comm));
    return SL3D_OK;
}

extern "C" int sl3d_group_get_points(sl3d_group *g, int view, float *xyz, uint8_t *valid)
try {
    int rc = check_views(g, view, 1);
    if (rc || (rc = root_sync(g))) return rc;
    DeviceGuard dg(g->gpus[0].device);
    const size_t W = (size_t)g->cfg.width, H = (size_t)g->cfg.height, pitch = (size_t)g->pitch;
    if (xyz) GHIP(g, hipMemcpy2D(xyz, W * 12, g->d_points + 3 * (size_t)view * g->px_view_stride, pitch * 12, W * 12, H, hipMemcpyDeviceToHost));
    if (valid) GHIP(g, hipMemcpy2D(valid, W, g->d_valid + (size_t)view * g->px_view_stride, pitch, W, H, hipMemcpyDeviceToHost));
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

// The reference's consumer is the HOST (its results are host globals / a host PCL cloud: common_variables.h:12-21,56-62,
// 8/save_point_cloud.cpp:85-104).  Every stripe copies its rows of every view straight into the caller's dense images, on its
// own stream and over its own GPU's PCIe link -- no xGMI hop to a root, no single link that drains everything.  Pinned
// destination memory (sl3d_host_alloc) makes the copies concurrent DMA; pageable memory works, serialised by the runtime.
// Waits for the stripes' kernels of those views (stream order) and for the copies.
extern "C" int sl3d_group_download_points(sl3d_group *g, int first, int n, float *xyz, uint8_t *valid)
try {
    int rc = check_views(g, first, n);
    if (rc) return rc;
    const size_t W = (size_t)g->cfg.width, H = (size_t)g->cfg.height;
    hipError_t e = hipSuccess;
    for (size_t s = 0; s < g->st.size() && e == hipSuccess; s++) {
        Stripe &S = g->st[s];
        const KParams &P = S.ctx->P;
        DeviceGuard dg(S.device);
        for (int v = 0; v < n && e == hipSuccess; v++) {
            if (xyz)
                e = hipMemcpy2DAsync(xyz + ((size_t)v * H + (size_t)S.row0) * W * 3, W * 12, P.points + 3 * (size_t)(first + v) * P.px_view_stride,
                                     (size_t)P.pitch * 12, W * 12, (size_t)S.rows, hipMemcpyDeviceToHost, S.ctx->stream);
            if (valid && e == hipSuccess)
                e = hipMemcpy2DAsync(valid + ((size_t)v * H + (size_t)S.row0) * W, W, P.valid + (size_t)(first + v) * P.px_view_stride, (size_t)P.pitch, W,
                                     (size_t)S.rows, hipMemcpyDeviceToHost, S.ctx->stream);
        }
    }
    // success or not, no copy may still be running against the caller's images when this returns
    int sync_rc = SL3D_OK;
    size_t sync_s = 0;
    for (size_t s = 0; s < g->st.size(); s++) {
        const int r = sl3d_synchronize(g->st[s].ctx);
        if (r != SL3D_OK && sync_rc == SL3D_OK) {
            sync_rc = r;
            sync_s = s;
        }
    }
    if (e != hipSuccess) return gfail(g, SL3D_E_HIP, std::string("group_download_points: hipMemcpy2DAsync: ") + hipGetErrorString(e));
    if (sync_rc != SL3D_OK) return gfail(g, sync_rc, "stripe " + std::to_string(sync_s) + ": " + sl3d_last_error(g->st[sync_s].ctx));
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

// sl3d_process_views for a group: every stripe runs the three-stream pipeline (upload of its rows of view k+1, fused kernel of
// view k, download of view k-1 into the caller's dense images) on its own GPU; all pipelines are enqueued before any is waited
// for, so with pinned buffers the stripes -- and the PCIe links of their GPUs -- work concurrently behind the one calling thread.
// planes: n_views * planes_per_view pointers to WINDOW-sized planes (`stride` bytes per row), view-major, plane order of
// sl3d_device_buffers.  Masks and calibration must be set (every slot < max_views).
extern "C" int sl3d_group_process_views(sl3d_group *g, int n_views, const uint8_t *const *planes, size_t stride, float *xyz, uint8_t *valid)
try {
    if (!g || n_views < 1 || !planes) return gfail(g, SL3D_E_INVALID_ARG, "group_process_views: null argument");
    const size_t W = (size_t)g->cfg.width, H = (size_t)g->cfg.height;
    const size_t ppv = (size_t)g->st[0].ctx->P.planes_per_view, np = (size_t)n_views * ppv;
    for (size_t i = 0; i < np; i++)
        if (!planes[i]) return gfail(g, SL3D_E_INVALID_ARG, "group_process_views: null plane");
    // the pipelines overwrite result slots [0, min(n_views, max_views)) of every stripe: an asynchronous gather may still be
    // reading them (sl3d_group_gather / _gather_clouds only mark the views busy) -- an RCCL send on the communication stream of the
    // stripe's own GPU, a (peer) copy on the root's.  The stripe's stream waits for both, as group_launch does; the enqueue below
    // then drains that stream before it reuses a slot.
    const int slots = std::min(n_views, g->cfg.max_views);
    if (overlaps(g, 0, slots)) {
        for (size_t s = 0; s < g->st.size(); s++) {
            Stripe &S = g->st[s];
            DeviceGuard dg(S.device);
            GHIP(g, hipStreamWaitEvent(S.ctx->stream, g->gpus[(size_t)S.gpu].ev_comm, 0));
            if (S.gpu != 0) GHIP(g, hipStreamWaitEvent(S.ctx->stream, g->gpus[0].ev_comm, 0));
        }
    }
    int first_rc = SL3D_OK;
    std::string first_err;
    std::vector<const uint8_t *> sub(np);
    size_t enq = 0;
    for (; enq < g->st.size(); enq++) {
        const Stripe &S = g->st[enq];
        for (size_t i = 0; i < np; i++) sub[i] = planes[i] + (size_t)S.row0 * stride;
        const int rc = sl3d_process_views_enqueue(S.ctx, n_views, sub.data(), stride, xyz ? xyz + (size_t)S.row0 * W * 3 : nullptr, W * H * 3,
                                                  valid ? valid + (size_t)S.row0 * W : nullptr, W * H, W);
        if (rc != SL3D_OK) {
            first_rc = rc;
            first_err = "stripe " + std::to_string(enq) + ": " + sl3d_last_error(S.ctx);
            enq++;
            break;
        }
    }
    // nothing may still run against the caller's buffers when this returns, whatever happened above
    for (size_t s = 0; s < enq; s++) {
        const int rc = sl3d_process_views_wait(g->st[s].ctx);
        if (rc != SL3D_OK && first_rc == SL3D_OK) {
            first_rc = rc;
            first_err = "stripe " + std::to_string(s) + ": " + sl3d_last_error(g->st[s].ctx);
        }
    }
    if (first_rc != SL3D_OK) return gfail(g, first_rc, first_err);
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_get_device_buffers(sl3d_group *g, sl3d_device_buffers *o)
try {
    if (!g || !o) return gfail(g, SL3D_E_INVALID_ARG, "null argument");
    memset(o, 0, sizeof *o);
    o->frame_pitch = (size_t)g->pitch;
    o->points = g->d_points;
    o->points_pitch = (size_t)g->pitch * 12;
    o->points_view_stride = g->px_view_stride * 12;
    o->valid = g->d_valid;
    o->valid_pitch = (size_t)g->pitch;
    o->valid_view_stride = g->px_view_stride;
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_gather_clouds(sl3d_group *g, int first, int n, int64_t *counts)
try {
    int rc = check_views(g, first, n);
    if (rc) return rc;
    if (!g->d_cloud) {
        DeviceGuard dg(g->gpus[0].device);
        const hipError_t e = hipMalloc((void **)&g->d_cloud, (size_t)g->cfg.max_views * g->px_view_stride * 3 * sizeof(float));
        if (e != hipSuccess) return gfail(g, e == hipErrorOutOfMemory ? SL3D_E_NOMEM : SL3D_E_HIP, std::string("hipMalloc (assembled clouds): ") + hipGetErrorString(e));
    }
    // the counts come back first (this waits for the stripes' kernels); the payload sizes follow from them
    std::vector<int64_t> c((size_t)n);
    std::vector<Xfer> xs;
    std::vector<int64_t> off((size_t)n, 0);
    for (size_t s = 0; s < g->st.size(); s++) {
        const float *dev = nullptr;
        size_t stride = 0;
        GCTX(g, s, sl3d_get_cloud_counts(g->st[s].ctx, first, n, &dev, &stride, c.data()));
        for (int k = 0; k < n; k++) {
            float *dst = g->d_cloud + 3 * ((size_t)(first + k) * g->px_view_stride + (size_t)off[(size_t)k]);
            xs.push_back({(int)s, dev + 3 * (size_t)k * stride, dst, (size_t)c[(size_t)k] * 3, ncclFloat});
            off[(size_t)k] += c[(size_t)k];
        }
    }
    // view-major order inside the RCCL group, as in the dense gather
    std::stable_sort(xs.begin(), xs.end(), [](const Xfer &a, const Xfer &b) { return a.dst < b.dst; });
    rc = group_exchange(g, xs);
    if (rc) return rc;
    mark_busy(g, first, n);
    for (int k = 0; k < n; k++) {
        g->cloud_count[(size_t)(first + k)] = off[(size_t)k];
        if (counts) counts[k] = off[(size_t)k];
    }
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_get_cloud(sl3d_group *g, int view, float *xyz, int64_t capacity, int64_t *count)
try {
    int rc = check_views(g, view, 1);
    if (rc) return rc;
    if (!count) return gfail(g, SL3D_E_INVALID_ARG, "null argument");
    if (!g->d_cloud) return gfail(g, SL3D_E_STATE, "sl3d_group_gather_clouds has not been called");
    if ((rc = root_sync(g))) return rc;
    *count = g->cloud_count[(size_t)view];
    const int64_t m = std::min(*count, capacity);
    if (xyz && m > 0) {
        DeviceGuard dg(g->gpus[0].device);
        GHIP(g, hipMemcpy(xyz, g->d_cloud + 3 * (size_t)view * g->px_view_stride, (size_t)m * 12, hipMemcpyDeviceToHost));
    }
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)

extern "C" int sl3d_group_synchronize(sl3d_group *g)
try {
    if (!g) return SL3D_E_INVALID_ARG;
    for (size_t s = 0; s < g->st.size(); s++) GCTX(g, s, sl3d_synchronize(g->st[s].ctx));
    for (auto &u : g->gpus) {
        DeviceGuard dg(u.device);
        GHIP(g, hipStreamSynchronize(u.comm));
    }
    g->comm_busy = false;
    return SL3D_OK;
}
SL3D_GROUP_CATCH(g)
