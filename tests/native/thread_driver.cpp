// Two host threads at the C ABI at once (threading row: "contexts independent and thread-safe with respect to
// each other").  The reference's callers are single-threaded (intermodule_dependencies.h); the C ABI promises more, so it is
// exercised: both threads start behind one gate, so the FIRST sl3d_create of the process -- the one that runs the one-time
// atan2 self-check under a mutex (sl3d_capi_context.cpp) -- is raced; each thread then creates / uses / destroys its own context
// `rounds` times (different shapes; thread 0 in parity mode through the per-stage entry points, thread 1 in the timed mode
// with two view slots, the fused kernel and the in-kernel compaction), generating its captures on the device.  The last
// round's inputs and results are dumped for tests/test_gpu_round3.py, which replays the dumped frames through the oracle.
//
//   thread_driver <cal.bin: 2 x 40 doubles Kc dc rc tc Kp dp rp tp, one set per thread> <out prefix> <rounds>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sl3d.h"

namespace {

struct Job {
    int W, H, PW, PH, N, fw, views;
    bool keep;
    int rc = 0;
    std::string err;
};

std::atomic<int> gate{0};
double cals[2][40];

#define CK(call)                                                                                   \
    do {                                                                                           \
        const int rc_ = (call);                                                                    \
        if (rc_ != SL3D_OK) {                                                                      \
            j.rc = rc_;                                                                            \
            j.err = std::string(#call) + ": " + sl3d_strerror(rc_) + ": " + sl3d_last_error(ctx);  \
            if (ctx) sl3d_destroy(ctx);                                                            \
            return;                                                                                \
        }                                                                                          \
    } while (0)

template <typename T>
void dump(FILE *f, const std::vector<T> &v) { fwrite(v.data(), sizeof(T), v.size(), f); }

void worker(Job &j, int id, int rounds, const std::string &prefix)
{
    gate.fetch_add(1);
    while (gate.load() < 2) {}  // both threads leave together: the first sl3d_create is raced
    for (int round = 0; round < rounds; round++) {
        sl3d_ctx *ctx = nullptr;
        sl3d_config c;
        memset(&c, 0, sizeof c);
        c.width = j.W; c.height = j.H; c.proj_width = j.PW; c.proj_height = j.PH;
        c.n_fringe = 3; c.n_gray_v = c.n_gray_h = j.N; c.fringe_width_v = c.fringe_width_h = j.fw;
        c.max_views = j.views; c.device = 0; c.flags = j.keep ? SL3D_FLAG_KEEP_STAGES : 0;
        CK(sl3d_create(&c, &ctx));
        const double *cal = cals[id];
        CK(sl3d_set_calibration(ctx, cal, cal + 9, cal + 14, cal + 17, cal + 20, cal + 29, cal + 34, cal + 37));
        std::vector<uint8_t> mask((size_t)j.W * j.H, 0);
        for (int r = 1; r < j.H - 1; r++)
            for (int x = 1; x < j.W - 1; x++) mask[(size_t)r * j.W + x] = (uint8_t)(((r * 7 + x * 3 + id) % 53) != 0);  // a few holes
        const size_t px = (size_t)j.W * j.H;
        const int ppa = 3 + 2 * j.N;  // planes per axis
        std::vector<uint8_t> frames((size_t)j.views * 2 * ppa * px), valid((size_t)j.views * px);
        std::vector<float> xyz((size_t)j.views * px * 3);
        std::vector<int64_t> cpm;
        std::vector<float> cloud;
        std::vector<int64_t> counts((size_t)j.views, 0);
        for (int v = 0; v < j.views; v++) {
            CK(sl3d_set_mask(ctx, v, mask.data(), (size_t)j.W));
            const double plane[3] = {0.5 * v + id, 0.05, 0.05 - 0.01 * v};
            CK(sl3d_synth_view(ctx, v, plane, 0x3D5CA11ull + (uint64_t)id, v + 10 * round, 2, 0.8f, 10.0f));
        }
        if (j.keep) {  // main()'s order through the four stage entry points
            CK(sl3d_compute_wrapped_phase(ctx, 0, 0));
            CK(sl3d_compute_wrapped_phase(ctx, 0, 1));
            CK(sl3d_unwrap_phase(ctx, 0, 0));
            CK(sl3d_unwrap_phase(ctx, 0, 1));
            CK(sl3d_compute_c_p_map(ctx, 0));
            CK(sl3d_triangulate(ctx, 0));
            cpm.resize(px * 2);
            CK(sl3d_get_c_p_map(ctx, 0, cpm.data()));
        } else {
            for (int rep = 0; rep < 5; rep++) CK(sl3d_run(ctx, 0, j.views));
            CK(sl3d_run_clouds(ctx, 0, j.views));
            const float *dev = nullptr;
            size_t stride = 0;
            CK(sl3d_get_cloud_counts(ctx, 0, j.views, &dev, &stride, counts.data()));
            int64_t total = 0;
            for (int v = 0; v < j.views; v++) total += counts[(size_t)v];
            cloud.resize((size_t)total * 3);
            int64_t off = 0;
            for (int v = 0; v < j.views; v++) {
                CK(sl3d_download(ctx, cloud.data() + 3 * off, dev + 3 * (size_t)v * stride, (size_t)counts[(size_t)v] * 12));
                off += counts[(size_t)v];
            }
        }
        for (int v = 0; v < j.views; v++) {
            CK(sl3d_get_points(ctx, v, xyz.data() + (size_t)v * px * 3, valid.data() + (size_t)v * px));
            for (int a = 0; a < 2; a++) {
                std::vector<uint8_t *> pl((size_t)ppa);
                for (int p = 0; p < ppa; p++) pl[(size_t)p] = frames.data() + (((size_t)v * 2 + a) * ppa + p) * px;
                CK(sl3d_get_frames(ctx, v, a, pl.data(), ppa, (size_t)j.W));
            }
        }
        if (round == rounds - 1) {
            FILE *f = fopen((prefix + std::to_string(id) + ".bin").c_str(), "wb");
            if (!f) { j.rc = -100; j.err = "cannot write the dump"; sl3d_destroy(ctx); return; }
            const int32_t hdr[8] = {j.W, j.H, j.PW, j.PH, j.N, j.fw, j.views, j.keep ? 1 : 0};
            fwrite(hdr, sizeof hdr, 1, f);
            dump(f, mask); dump(f, frames); dump(f, valid); dump(f, xyz);
            if (j.keep) dump(f, cpm);
            else { dump(f, counts); dump(f, cloud); }
            fclose(f);
        }
        sl3d_destroy(ctx);
    }
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(cals, sizeof(double), 80, f) != 80) return 3;
    fclose(f);
    const int rounds = atoi(argv[3]);
    Job a{320, 240, 512, 384, 7, 4, 1, true}, b{200, 150, 256, 192, 6, 4, 2, false};
    std::thread ta(worker, std::ref(a), 0, rounds, std::string(argv[2])), tb(worker, std::ref(b), 1, rounds, std::string(argv[2]));
    ta.join();
    tb.join();
    if (a.rc || b.rc) {
        fprintf(stderr, "thread 0: %d %s\nthread 1: %d %s\n", a.rc, a.err.c_str(), b.rc, b.err.c_str());
        return 10;
    }
    return 0;
}
