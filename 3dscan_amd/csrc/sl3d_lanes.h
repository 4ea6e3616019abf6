// sl3d_lanes.h -- WHERE a context's fused launches go: the policy of the launch lanes, free of any HIP call so that the same code runs in
// a CPU test against a model of the streams (tests/native/lanes_policy_check.cpp, tests/test_lanes_policy.py).
//
// A one-view launch spends ~3 of its 25 us ramping up and draining.  Consecutive small launches over DIFFERENT views are independent, so a
// context that owns its stream has two internal streams ("lanes") and lets the tail of one launch run under the ramp of the next
// (one view per launch from HBM: 25.7 -> 21.7 us per launch, 0.61 -> 0.72 of the roofline).  What four builds of it taught
// (profiles/r06_lanes_ab.txt) is the policy below:
//   * Handing work from the stream to a lane and back costs ~10 us each way; overlapping saves ~4 us per one-view launch.  The lanes pay
//     only in a LONG series: a small launch goes to a lane when it follows LANES_AFTER small launches of this series, or follows one and
//     the previous series was that long.  The first launch behind anything else the stream was given stays on the stream.
//   * Ties instead of waits: a launch over a view a lane still works on goes to THAT lane (stream order is the dependency); a launch
//     that repeats the views of the previous launch while that one ran on the stream stays on the stream.  Only a launch whose views
//     are tied to BOTH lanes waits across streams.
//   * A lane must have seen what the stream was given before the launch: one event, recorded only when the stream got new work since
//     the lane last looked (`main_epoch`).  The lanes' own events are recorded when somebody joins, not behind every launch.
//   * Every entry point that gives the stream anything joins first (the stream waits for the busy lanes).
// The caller (sl3d_capi_run.cpp) turns a plan into hipEventRecord / hipStreamWaitEvent calls; nothing here touches the device.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace sl3d {

struct LanePlan {
    int lane;         // 0 / 1: the lane the launch goes to
    bool wait_main;   // first: record the stream's event on the stream, make the lane wait for it
    bool wait_other;  // then: record the other lane's event on that lane, make this lane wait for it
};

struct LanePolicy {
    static constexpr unsigned LANES_AFTER = 8u;
    // what overlapping buys shrinks with the launch (tools/two_stream_probe.py: 1 view per launch -12.6 %, 2 views -4.4 %, 4 views -1.4 %,
    // i.e. 3.1 / 2.0 / 1.2 us per launch against ~20 us of hand-overs per series): launches of at most this many views take part
    static constexpr int MAX_VIEWS = 2;
    bool lane_busy[2] = {false, false};                // the lane holds work the stream has not been made to wait for yet
    unsigned main_epoch = 1, lane_epoch[2] = {0, 0};   // what of the stream's work a lane has already been made to wait for
    int next_lane = 0;
    unsigned runs_in_a_row = 0, last_series = 0;       // small launches since the stream was last given anything else / in the series before
    int prev_first = 0, prev_n = 0;                    // the previous small launch of the series ...
    bool prev_on_stream = false;                       // ... if it ran on the stream itself
    std::vector<int8_t> view_lane;                     // [max_views] the lane whose (unjoined) launch last touched the view, -1: none

    void reset(int max_views) { view_lane.assign((size_t)max_views, (int8_t)-1); }

    // The stream is about to be given work: bit l of the result = lane l is busy and the stream has to wait for it (the caller records
    // that lane's event on the lane and makes the stream wait).  Afterwards no lane is busy and no view is tied.
    unsigned stream_gets_work()
    {
        main_epoch++;
        unsigned mask = 0;
        for (int l = 0; l < 2; l++) {
            if (lane_busy[l]) mask |= 1u << l;
            lane_busy[l] = false;
        }
        if (mask) std::fill(view_lane.begin(), view_lane.end(), (int8_t)-1);
        return mask;
    }

    // ... by anything but a small launch: the series of small launches ends
    void series_ends()
    {
        if (runs_in_a_row) last_series = runs_in_a_row;
        runs_in_a_row = 0;
    }

    // does the small launch over views [first, first + n) go to a lane?  (Called once per small launch, before it; counts it.)
    bool small_launch_pays(int first, int n)
    {
        const bool series = runs_in_a_row >= LANES_AFTER || (runs_in_a_row > 0 && last_series >= LANES_AFTER);
        const bool repeats = runs_in_a_row > 0 && prev_on_stream && first < prev_first + prev_n && prev_first < first + n;
        const bool pay = series && !repeats;
        prev_first = first;
        prev_n = n;
        prev_on_stream = !pay;
        runs_in_a_row++;
        return pay;
    }

    // the lane of a launch that pays, and what that lane has to wait for first
    LanePlan begin(int first, int n)
    {
        int tied = -1;
        bool both = false;
        for (int v = first; v < first + n; v++) {
            const int t = view_lane[(size_t)v];
            if (t < 0 || !lane_busy[t]) continue;
            if (tied < 0) tied = t;
            else if (tied != t) both = true;
        }
        LanePlan p;
        p.lane = tied >= 0 ? tied : next_lane;
        if (tied < 0) next_lane = p.lane ^ 1;
        p.wait_main = lane_epoch[p.lane] != main_epoch;
        lane_epoch[p.lane] = main_epoch;
        p.wait_other = both;
        return p;
    }

    // the launch has been enqueued on its lane
    void end(int lane, int first, int n)
    {
        lane_busy[lane] = true;
        for (int v = first; v < first + n; v++) view_lane[(size_t)v] = (int8_t)lane;
    }
};

}  // namespace sl3d
