// CPU check of the launch-lane policy (3dscan_amd/csrc/sl3d_lanes.h, the very header the library compiles): random sequences of calls
// are run through the policy and through a model of three in-order streams (the context's stream and the two lanes) with events; for
// every launch the model knows which earlier operations are ordered before it.
//   * SAFETY: whatever touched one of a launch's views before it -- an earlier launch over the view, or any other call (an upload, a mask,
//     a getter: modelled as touching every view on the context's stream) -- is ordered before the launch; and such a call is ordered
//     behind every launch made before it.
//   * POLICY: a launch goes to a lane only in a long series (LANES_AFTER launches in a row, or one if the series before was that long),
//     never right behind another call, never when it repeats the views of the previous launch on the stream; the number of cross-stream
//     waits stays small (hand-overs cost ~10 us each).
// usage: lanes_policy_check [sequences] [seed] -> prints a summary, exit code 0 iff no violation
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../3dscan_amd/csrc/sl3d_lanes.h"

using sl3d::LanePlan;
using sl3d::LanePolicy;

namespace {

constexpr int STREAM = 2;  // index of the context's stream in the model (0 / 1: the lanes)

struct Model {
    // ops are numbered in issue order; before[s] = the set of ops known to be complete before the next op of stream s starts
    std::vector<std::vector<char>> before;  // [3][ops]
    std::vector<std::vector<char>> event;   // snapshot a recorded event carries: ev_lane[0], ev_lane[1], ev_main
    std::vector<std::vector<int>> touched;  // per op: the views it touches (all views: any other call)
    std::vector<int> where;                 // per op: the stream it ran on
    int n_ops = 0;
    explicit Model(int) : before(3), event(3) {}
    int issue(int s, const std::vector<int> &views)
    {
        const int id = n_ops++;
        for (auto &b : before) b.resize((size_t)n_ops, 0);
        for (auto &e : event) e.resize((size_t)n_ops, 0);
        touched.push_back(views);
        where.push_back(s);
        return id;
    }
    void complete_on(int s, int id) { before[(size_t)s][(size_t)id] = 1; }  // in-order stream: the next op of s starts behind this one
    void record(int ev, int s) { event[(size_t)ev] = before[(size_t)s]; }
    void wait(int s, int ev)
    {
        event[(size_t)ev].resize((size_t)n_ops, 0);
        for (int i = 0; i < n_ops; i++)
            if (event[(size_t)ev][(size_t)i]) before[(size_t)s][(size_t)i] = 1;
    }
};

}  // namespace

int main(int argc, char **argv)
{
    const int sequences = argc > 1 ? atoi(argv[1]) : 300;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
    std::mt19937 rng(seed);
    long long launches = 0, on_lanes = 0, hand_overs = 0, violations = 0, lone = 0, repeats_on_lane = 0;
    for (int q = 0; q < sequences; q++) {
        const int V = 1 + (int)(rng() % 10u);
        LanePolicy lp;
        lp.reset(V);
        Model m(V);
        std::vector<int> all(V);
        for (int v = 0; v < V; v++) all[(size_t)v] = v;
        const int steps = 50 + (int)(rng() % 400u);
        unsigned streak = 0, last_streak = 0;  // the test's own count of the series (what the policy is specified by)
        int prev_first = -1, prev_n = 0, prev_where = -1;
        const unsigned mode = rng() % 4u;      // 0: mixed, 1: long series round robin, 2: the same views over and over, 3: a call before every launch
        for (int t = 0; t < steps; t++) {
            unsigned r = rng() % 100u;
            if (mode == 1) r = r < 3 ? 0u : 50u;
            if (mode == 2) r = r < 2 ? 0u : 50u;
            if (mode == 3) r = (t & 1) ? 50u : 0u;
            auto join = [&]() {  // what sl3d_lanes_join / lanes_wait do with the policy's answer
                const unsigned busy = lp.stream_gets_work();
                for (int l = 0; l < 2; l++)
#ifdef DROP_JOIN
                    if (l == 0)
#endif
                    if (busy >> l & 1u) { m.record(l, l); m.wait(STREAM, l); hand_overs++; }
            };
            if (r < 12) {  // any other entry point: joins, then gives the stream work that touches every view
                lp.series_ends();
                join();
                const int id = m.issue(STREAM, all);
                for (int i = 0; i < id; i++)
                    if (!m.before[STREAM][(size_t)i]) { violations++; fprintf(stderr, "seq %d step %d: a call on the stream is not behind op %d\n", q, t, i); }
                m.complete_on(STREAM, id);
                if (streak) last_streak = streak;
                streak = 0;
                continue;
            }
            if (r < 15) continue;  // the hand-over of a device-resident deferred mask: touches neither policy nor streams
            int first, n;
            if (mode == 2) { first = 0; n = 1 + (int)(V > 1 && (rng() % 3u) == 0); }
            else if (mode == 1) { n = 1; first = t % V; }
            else { n = 1 + (int)(rng() % 4u); if (n > V) n = V; first = (int)(rng() % (unsigned)(V - n + 1)); }
            std::vector<int> views;
            for (int v = first; v < first + n; v++) views.push_back(v);
            if (r >= 92 && mode == 0) {  // a LARGE launch: ends the series, runs on the stream
                lp.series_ends();
                join();
                const int id = m.issue(STREAM, views);
                m.complete_on(STREAM, id);
                if (streak) last_streak = streak;
                streak = 0;
                prev_where = -1;
                launches++;
                continue;
            }
            // a small launch
            const bool spec_series = streak >= LanePolicy::LANES_AFTER || (streak > 0 && last_streak >= LanePolicy::LANES_AFTER);
            const bool spec_repeat = streak > 0 && prev_where == STREAM && first < prev_first + prev_n && prev_first < first + n;
            const bool pays = lp.small_launch_pays(first, n);
            if (pays != (spec_series && !spec_repeat)) { violations++; fprintf(stderr, "seq %d step %d: the policy %s a lane against its specification\n", q, t, pays ? "takes" : "refuses"); }
            int s = STREAM;
            if (!pays) {
                join();
            } else {
                const LanePlan p = lp.begin(first, n);
                s = p.lane;
#ifndef DROP_MAIN_WAIT   // (the test's own teeth: built with one of these, a plan's wait is NOT carried out -- the model must object)
                if (p.wait_main) { m.record(2, STREAM); m.wait(s, 2); hand_overs++; }
#endif
#ifndef DROP_OTHER_WAIT
                if (p.wait_other) { m.record(s ^ 1, s ^ 1); m.wait(s, s ^ 1); hand_overs++; }
#endif
            }
            const int id = m.issue(s, views);
            // SAFETY: every earlier op that touched one of these views is ordered before this launch
            for (int i = 0; i < id; i++) {
                bool shares = false;
                for (int a : m.touched[(size_t)i])
                    for (int b : views) shares |= a == b;
                if (shares && !m.before[(size_t)s][(size_t)i]) {
                    violations++;
                    fprintf(stderr, "seq %d step %d: launch over [%d, %d) on %d is not behind op %d (on %d)\n", q, t, first, first + n, s, i, m.where[(size_t)i]);
                }
            }
            m.complete_on(s, id);
            if (pays) lp.end(s, first, n);
            launches++;
            on_lanes += pays;
            if (pays && streak == 0) lone++;
            if (pays && spec_repeat) repeats_on_lane++;
            streak++;
            prev_first = first; prev_n = n; prev_where = s;
        }
    }
    printf("%d sequences, %lld launches (%lld on lanes), %lld cross-stream hand-overs, %lld violations, %lld lane launches right behind another call, %lld repeats on a lane\n",
           sequences, launches, on_lanes, hand_overs, violations, lone, repeats_on_lane);
    // hand-overs stay rare where the lanes are used at all: at most one in and one out per series plus the ties to both lanes
    return violations == 0 && lone == 0 && repeats_on_lane == 0 ? 0 : 1;
}
