"""GPU tests (-m gpu): launch lanes (3dscan_amd/csrc/sl3d_ctx.h).  A context that owns its stream puts consecutive small launches on two
internal streams in turn once a series of them is long enough to pay, so that the tail of one launch runs under the ramp of the next;
every other call joins them first (sl3d_launch_counts tells where the launches went).  Nothing a
caller can observe may differ from a context with SL3D_FLAG_SERIAL_LAUNCHES: every sequence below runs on both and is compared bit for
bit -- independent views back to back, the same view over and over, launches whose views overlap, uploads and new masks between launches
(the lane has to wait for what the context's stream was given), the per-scan loop with device-resident deferred masks (MASKIN launches),
clouds and their consumers, the stopwatch."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def _same(a, b, tag):
    assert np.array_equal(a[1], b[1]), tag
    assert np.array_equal(a[0], b[0], equal_nan=True), tag


def _pair(S, W, H, PW, PH, N, fw, V, **kw):
    return S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V, **kw), S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V, serial_launches=True, **kw)


def test_overlapping_small_launches_equal_serial_ones():
    import torch
    S, syn = pkg("scanner"), pkg("synth")
    W, H, N, fw, V = 640, 360, 8, 4, 6
    PW, PH = fw << N, fw << N
    rng = np.random.default_rng(5)
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    masks = np.stack([(rng.random((H, W)) < p).astype(np.uint8) for p in (1.0, 0.97, 0.9, 0.5, 0.97, 1.0)])
    d_masks = torch.from_numpy(masks).cuda()
    torch.cuda.synchronize()
    lanes, serial = _pair(S, W, H, PW, PH, N, fw, V)
    with lanes, serial:
        caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, plane=(1.5 * v, 0.05, 0.04), view=v, noise=3) for v in range(V)]
        for c in (lanes, serial):
            c.set_calibration(*cal)
            for v in range(V):
                c.set_frames(0, caps[v]["planes_v"], view=v)
                c.set_frames(1, caps[v]["planes_h"], view=v)
            c.set_masks(masks)
            c.synchronize()

        def both(f):
            for c in (lanes, serial):
                f(c)

        def check(tag):
            for v in range(V):
                _same(lanes.points(v), serial.points(v), (tag, v))

        # independent views back to back, nothing waited for in between: the lanes take over once the series is 8 launches long
        both(lambda c: [c.run(i % V, 1) for i in range(30)])
        assert lanes.launch_counts() == (8, 22) and serial.launch_counts() == (30, 0)
        check("independent views")
        # ... and from its second launch on when the series before was that long
        both(lambda c: [c.run(i % V, 1) for i in range(10)])
        assert lanes.launch_counts() == (9, 31)
        check("a second series")
        # the same view over and over (stays on the stream: nothing to overlap), and launches whose views overlap (tied to a lane)
        n0 = lanes.launch_counts()
        both(lambda c: [c.run(2, 1) for _ in range(6)])
        assert lanes.launch_counts() == (n0[0] + 6, n0[1])
        both(lambda c: [c.run(f, n) for f, n in ((0, 2), (1, 2), (2, 3), (4, 2), (3, 1), (0, 4), (3, 3), (5, 1), (0, 1))])
        check("overlapping views")
        # uploads between launches: view v gets the frames of view v + 1 and is launched at once -- the lane must see the upload
        def swap(c):
            for i in range(12):
                v, w = i % V, (i + 1) % V
                c.set_frames(0, caps[w]["planes_v"], view=v)
                c.set_frames(1, caps[w]["planes_h"], view=v)
                c.run(v, 1)
        both(swap)
        check("uploads between launches")
        # the per-scan loop: a new device-resident selection (recorded, nothing enqueued) + one view -- MASKIN launches on both lanes
        def scans(c):
            for i in range(36):
                c.set_masks_device(d_masks.data_ptr() + ((i * 5 + 1) % V) * W * H, W, 0, i % V, 1)
                c.run(i % V, 1)
        n0 = lanes.launch_counts()
        both(scans)
        # (a recorded device-resident mask does not end the series; the series before this one were single launches behind uploads, so
        # the lanes take over at the ninth launch again)
        assert lanes.launch_counts() == (n0[0] + 8, n0[1] + 28)
        assert ", 4, " in lanes.last_fused_kernel_name() and ", 4, " in serial.last_fused_kernel_name()
        check("per-scan loop")
        for v in range(V):
            ba, bb = lanes.device_buffers(), serial.device_buffers()
            pa, pb = np.empty((H + 4, ba.mask_pitch), np.uint8), np.empty((H + 4, bb.mask_pitch), np.uint8)
            lanes._d2h(pa, ba.mask + v * ba.mask_view_stride)
            serial._d2h(pb, bb.mask + v * bb.mask_view_stride)
            assert np.array_equal(pa, pb), v
        # host masks between launches (staged: the staging plane is rewritten behind the MASKIN launch that reads it)
        def host_masks(c):
            for i in range(12):
                c.set_mask(masks[(i * 5 + 2) % V], view=i % V)
                c.run(i % V, 1)
        n0 = lanes.launch_counts()
        both(host_masks)
        assert lanes.launch_counts() == (n0[0] + 12, n0[1])           # (every launch behind a copy on the stream: stays there)
        check("host masks between launches")
        # clouds: small launches side by side, then their consumers
        both(lambda c: [c.run_clouds(i % V, 1) for i in range(12)])
        for v in range(V):
            a, b = lanes.fused_clouds(v, 1)[0], serial.fused_clouds(v, 1)[0]
            assert np.array_equal(a, b) and len(a) == int(lanes.points(v)[1].sum()), v
        # a large launch between small ones
        both(lambda c: [c.run(0, 1), c.run(0, V), c.run(1, 1), c.run(2, 2)])
        check("large between small")
        # the stopwatch brackets what the lanes hold
        for c in (lanes, serial):
            c.timer_start()
            for i in range(20):
                c.run(i % V, 1)
            assert c.timer_stop() > 20 * 0.003, "20 launches cannot take less than 60 us"


def test_lanes_are_off_where_the_stream_is_not_the_contexts_own():
    """A caller's stream, the parity mode and a group's stripes keep every launch on the one stream (the caller / the group orders its own
    work behind them by that stream): launches in a row on a caller's stream, an event of the caller behind them, the same results."""
    import torch
    S, syn = pkg("scanner"), pkg("synth")
    W, H, N, fw, V = 640, 360, 8, 4, 3
    PW, PH = fw << N, fw << N
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    st = torch.cuda.Stream()
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V, stream=st.cuda_stream) as sc, S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V) as ref:
        for c in (sc, ref):
            c.set_calibration(*cal)
            c.set_masks(syn.default_mask(W, H), 0, V)
            for v in range(V):
                c.synth_view(v, plane=(1.5 * v, 0.05, 0.04), view_id=v, noise=2)
        for v in range(V):
            sc.run(v, 1)
            ref.run(v, 1)
        ev = torch.cuda.Event()
        ev.record(st)            # the caller's own ordering: behind everything sl3d_run gave ITS stream
        ev.synchronize()
        assert sc.launch_counts()[1] == 0
        for v in range(V):
            _same(sc.points(v), ref.points(v), v)
