// fake_rccl.cpp -- a TEST DOUBLE of the seven librccl entry points sl3d_group.cpp binds (csrc/sl3d_group.cpp: RcclApi), so that the
// N-rank gather of a row-stripe group -- one ncclSend per (view, stripe) slab on the stripe's side, the matching ncclRecv on the
// root's side, all inside ncclGroupStart / ncclGroupEnd -- can execute on a box with ONE GPU.  Real RCCL refuses a communicator
// whose ranks share a device; this one accepts it.  Selected through the environment variable SL3D_RCCL_LIB (read where the
// library is dlopen'ed); never part of the product.
//
// Semantics kept from NCCL: point-to-point operations are only matched when the outermost group closes; a send from rank a to
// peer b pairs with the earliest unmatched recv on rank b from peer a (FIFO per ordered pair); counts and types must agree;
// the transfer is ordered after everything enqueued earlier on the sender's stream and before everything enqueued later on
// either stream.  Loud where NCCL would hang or corrupt: an unmatched send or recv, a size or type mismatch, a point-to-point
// call outside a group, a rank out of range all fail the closing ncclGroupEnd (or the call itself) with a message.
// Counters (fake_rccl_stats) let the tests assert that the exchange really went through here and how it was grouped.
//
// Build (tests/test_gpu_round4.py does it):  hipcc -shared -fPIC -O2 tests/native/fake_rccl.cpp -o <tmp>/libfake_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct World {
    int nranks = 0;
    int alive = 0;
};

struct Op {
    bool send;
    const void *sbuf;
    void *rbuf;
    size_t count;
    ncclDataType_t type;
    int peer;
    ncclComm_t comm;
    hipStream_t stream;
    bool matched = false;
};

std::mutex g_mu;
std::string g_err = "no error";
int g_groups = 0, g_pairs = 0, g_max_pairs = 0, g_comms = 0, g_self_pairs = 0;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

ncclResult_t fail(ncclResult_t code, const std::string &msg)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_err = "fake_rccl: " + msg;
    fprintf(stderr, "%s\n", g_err.c_str());
    return code;
}

}  // namespace

struct ncclComm {  // what the opaque handle points at
    World *world;
    int rank, device;
};

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return fail(ncclInvalidArgument, "ncclCommInitAll: no ranks");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return fail(ncclUnhandledCudaError, "hipGetDeviceCount failed");
    World *w = new World();
    w->nranks = w->alive = ndev;
    for (int r = 0; r < ndev; r++) {
        const int dev = devlist ? devlist[r] : r;
        if (dev < 0 || dev >= count) {
            for (int k = 0; k < r; k++) delete comms[k];
            delete w;
            return fail(ncclInvalidArgument, "ncclCommInitAll: device ordinal out of range");
        }
        comms[r] = new ncclComm{w, r, dev};  // (ranks may share a device: that is the point of this double)
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_comms += ndev;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    if (--c->world->alive == 0) delete c->world;
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    if (t_depth++ == 0) t_ops.clear();
    return ncclSuccess;
}

static ncclResult_t p2p(bool send, const void *sbuf, void *rbuf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm) return fail(ncclInvalidArgument, "null communicator");
    if (peer < 0 || peer >= comm->world->nranks) return fail(ncclInvalidArgument, "peer " + std::to_string(peer) + " out of range");
    if (t_depth == 0) return fail(ncclInvalidUsage, "point-to-point call outside ncclGroupStart / ncclGroupEnd (a lone blocking send would deadlock one thread)");
    if (count > 0 && !(send ? sbuf : (const void *)rbuf)) return fail(ncclInvalidArgument, "null buffer");
    t_ops.push_back(Op{send, sbuf, rbuf, count, type, peer, comm, stream});
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(true, buf, nullptr, count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(false, nullptr, buf, count, type, peer, comm, stream);
}

ncclResult_t ncclGroupEnd()
{
    if (t_depth == 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    // pair every recv with the earliest unmatched send of its (source rank -> this rank) channel
    struct Pair { Op *s, *r; };
    std::vector<Pair> pairs;
    for (Op &r : ops) {
        if (r.send) continue;
        Op *found = nullptr;
        for (Op &s : ops)
            if (s.send && !s.matched && s.comm->world == r.comm->world && s.comm->rank == r.peer && s.peer == r.comm->rank) {
                found = &s;
                break;
            }
        if (!found)
            return fail(ncclInvalidUsage, "recv on rank " + std::to_string(r.comm->rank) + " from peer " + std::to_string(r.peer) + " has no matching send in the group (real NCCL would hang)");
        if (found->count != r.count || found->type != r.type)
            return fail(ncclInvalidUsage, "send " + std::to_string(found->comm->rank) + " -> " + std::to_string(r.comm->rank) + ": " + std::to_string(found->count) +
                                              " elements sent, " + std::to_string(r.count) + " expected (or the types differ)");
        found->matched = r.matched = true;
        pairs.push_back({found, &r});
    }
    for (Op &s : ops)
        if (s.send && !s.matched)
            return fail(ncclInvalidUsage, "send from rank " + std::to_string(s.comm->rank) + " to peer " + std::to_string(s.peer) + " has no matching recv in the group (real NCCL would hang)");
    // ordering: every receiving stream first waits for what the sending streams have enqueued so far, copies, and then every
    // sending stream waits for the receivers (a send buffer may be overwritten by what its stream runs next)
    int prev = -1;
    (void)hipGetDevice(&prev);
    std::map<hipStream_t, std::pair<int, hipEvent_t>> sent, received;  // stream -> (device, event)
    auto record = [&](std::map<hipStream_t, std::pair<int, hipEvent_t>> &m, hipStream_t st, int dev) -> hipError_t {
        if (m.count(st)) return hipSuccess;
        hipError_t e = hipSetDevice(dev);
        hipEvent_t ev = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ev, st);
        m[st] = {dev, ev};
        return e;
    };
    hipError_t e = hipSuccess;
    for (Pair &p : pairs)
        if (e == hipSuccess && p.s->stream != p.r->stream) e = record(sent, p.s->stream, p.s->comm->device);
    int self_pairs = 0;
    for (Pair &p : pairs) {
        if (e != hipSuccess) break;
        if (p.s->comm == p.r->comm) self_pairs++;
        e = hipSetDevice(p.r->comm->device);
        if (e == hipSuccess && p.s->stream != p.r->stream) e = hipStreamWaitEvent(p.r->stream, sent[p.s->stream].second, 0);
        const size_t bytes = p.r->count * type_bytes(p.r->type);
        if (e == hipSuccess && bytes) {
            if (p.s->comm->device == p.r->comm->device) e = hipMemcpyAsync(p.r->rbuf, p.s->sbuf, bytes, hipMemcpyDeviceToDevice, p.r->stream);
            else e = hipMemcpyPeerAsync(p.r->rbuf, p.r->comm->device, p.s->sbuf, p.s->comm->device, bytes, p.r->stream);
        }
    }
    for (Pair &p : pairs)
        if (e == hipSuccess && p.s->stream != p.r->stream) e = record(received, p.r->stream, p.r->comm->device);
    for (Pair &p : pairs)
        if (e == hipSuccess && p.s->stream != p.r->stream) {
            e = hipSetDevice(p.s->comm->device);
            if (e == hipSuccess) e = hipStreamWaitEvent(p.s->stream, received[p.r->stream].second, 0);
        }
    for (auto &kv : sent) (void)hipEventDestroy(kv.second.second);      // (released once the work that uses them has run)
    for (auto &kv : received) (void)hipEventDestroy(kv.second.second);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail(ncclUnhandledCudaError, std::string("HIP error while moving the data: ") + hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_mu);
    g_groups++;
    g_pairs += (int)pairs.size();
    g_self_pairs += self_pairs;
    if ((int)pairs.size() > g_max_pairs) g_max_pairs = (int)pairs.size();
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    if (r == ncclSuccess) return "no error";
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_mu);
    copy = g_err;
    return copy.c_str();
}

// what went through this double since the process started: closed groups, matched send/recv pairs, the largest group, ranks created,
// pairs whose two ends were the same rank
void fake_rccl_stats(int *groups, int *pairs, int *max_pairs_in_group, int *comms, int *self_pairs)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (groups) *groups = g_groups;
    if (pairs) *pairs = g_pairs;
    if (max_pairs_in_group) *max_pairs_in_group = g_max_pairs;
    if (comms) *comms = g_comms;
    if (self_pairs) *self_pairs = g_self_pairs;
}

}  // extern "C"
