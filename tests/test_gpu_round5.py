"""GPU tests (-m gpu) added in round 5: the bit-plane k_mask_prepare (H0 / S3b / S3d) through every way a mask reaches it --
one view, several views in one launch, pageable / pinned / device-resident sources, windows on every frame border -- against the
oracle's literal scan of 3/wrapped_phase.cpp:253-279; the selected-quad count that picks a small launch's kernel; an axis without
Gray planes on the last resident view."""
import numpy as np
import pytest

from conftest import assert_points_close, pkg

pytestmark = pytest.mark.gpu


def _mask(rng, W, H, trial):
    p = rng.choice([0.05, 0.5, 0.9, 0.98])
    m = (rng.random((H, W)) < p).astype(np.uint8)
    if trial % 3 == 0:
        m[:] = 0
        for _ in range(6):
            y, x, h, w = rng.integers(0, H), rng.integers(0, W), rng.integers(1, H), rng.integers(1, W)
            m[y:y + h, x:x + w] = 1
        m ^= (rng.random((H, W)) < 0.01).astype(np.uint8)
    if trial % 2 == 1:
        z = m == 0
        m[z] = rng.integers(2, 256, size=int(z.sum()), dtype=np.uint8)   # selected iff == 1
    if trial % 5 == 4:
        m[:] = 1                                                          # border pixels selected too
    return m


def _oracle_valid(mask):
    from oracle.oracle import Oracle
    H, W = mask.shape
    o = Oracle(W, H, 64, 64, 3, 3, 8, 8)
    o.set_mask(mask)
    o.compute_wrapped_phase(0, [np.zeros((H, W), np.uint8)] * 3)
    return o.valid_map(0).astype(np.uint8)


def _valid_after_stage3(sc, views):
    """valid_map_vertical of every view, once from the fused kernel (reads the band plane) and once from the per-stage kernel
    k_wrap (evaluates the closed form on the 0/1 plane k_mask_prepare wrote)."""
    sc.run(0, views)
    fused = [sc.valid_map(0, view=v) for v in range(views)]
    for v in range(views):
        sc.run_stages(view=v)
    staged = [sc.valid_map(0, view=v) for v in range(views)]
    return fused, staged


@pytest.mark.parametrize("FW,FH", [(150, 90), (203, 61), (64, 7), (1920, 24), (17, 300)])
def test_mask_prepare_bit_planes_against_literal_scan(FW, FH):
    """Every mask through sl3d_set_masks (one launch for all views, distinct masks) and through sl3d_set_mask; full frames and
    windows touching every border; pageable and pinned sources."""
    S, syn = pkg("scanner"), pkg("synth")
    PW, PH, N, fw, V = 256, 192, 6, 8, 5
    rng = np.random.default_rng(FW * 1000 + FH)
    cal = syn.cal_tuple(syn.synth_rig(FW, FH, PW, PH))
    masks = np.stack([_mask(rng, FW, FH, t) for t in range(V)])
    refs = [_oracle_valid(m) for m in masks]
    wins = [(0, 0, FW, FH)]
    for _ in range(4):
        w, h = int(rng.integers(1, FW + 1)), int(rng.integers(1, FH + 1))
        wins.append((int(rng.integers(0, FW - w + 1)), int(rng.integers(0, FH - h + 1)), w, h))
    wins.append((FW - min(FW, 9), FH - min(FH, 3), min(FW, 9), min(FH, 3)))
    for (x0, y0, w, h) in wins:
        with S.Scanner(w, h, PW, PH, N, N, fw, fw, full_size=(FW, FH), origin=(x0, y0), keep_stages=True, max_views=V) as sc:
            sc.set_calibration(*cal)
            pm = sc.pinned(masks.shape, np.uint8)
            pm[:] = masks
            for how in ("batch", "batch_pinned", "single"):
                for v in range(V):
                    sc.set_mask(np.zeros((FH, FW), np.uint8), view=v)   # whatever the previous pass left is gone
                if how == "single":
                    for v in range(V):
                        sc.set_mask(masks[v], view=v)
                else:
                    sc.set_masks(masks if how == "batch" else pm)
                fused, staged = _valid_after_stage3(sc, V)
                for v in range(V):
                    want = refs[v][y0:y0 + h, x0:x0 + w]
                    assert np.array_equal(fused[v], want), (how, v, (x0, y0, w, h))
                    assert np.array_equal(staged[v], want), (how, v, (x0, y0, w, h))
            # one mask for every view: one copy, one launch
            sc.set_masks(masks[2])
            fused, staged = _valid_after_stage3(sc, V)
            for v in range(V):
                assert np.array_equal(fused[v], refs[2][y0:y0 + h, x0:x0 + w]) and np.array_equal(staged[v], fused[v])


def test_device_resident_masks_direct_and_staged():
    """Masks that already live in device memory: with 4-byte aligned rows the kernel reads the caller's buffer (no copy); a buffer
    at an odd address / odd stride, or a window origin off a 4-byte boundary, goes through the staging plane by a device copy.
    Both equal the oracle."""
    torch = pytest.importorskip("torch")
    S, syn = pkg("scanner"), pkg("synth")
    FW, FH, PW, PH, N, fw, V = 200, 120, 256, 192, 6, 8, 3
    rng = np.random.default_rng(8)
    cal = syn.cal_tuple(syn.synth_rig(FW, FH, PW, PH))
    masks = np.stack([_mask(rng, FW, FH, t) for t in range(V)])
    refs = [_oracle_valid(m) for m in masks]
    d_al = torch.from_numpy(masks).cuda()                       # rows of 200 bytes at a 256-byte aligned base: direct
    big = torch.zeros(V * (FH * 203) + 64, dtype=torch.uint8, device="cuda")
    odd = big[1:1 + V * FH * 203].view(V, FH, 203)              # odd base address, odd stride: staged
    odd[:, :, :FW] = d_al
    torch.cuda.synchronize()
    for (x0, y0, w, h) in [(0, 0, FW, FH), (4, 3, 100, 50), (6, 0, 64, 120), (100, 70, 100, 50)]:
        with S.Scanner(w, h, PW, PH, N, N, fw, fw, full_size=(FW, FH), origin=(x0, y0), keep_stages=True, max_views=V) as sc:
            sc.set_calibration(*cal)
            for name, t, stride in (("aligned", d_al, FW), ("odd", odd, 203)):
                sc.set_masks(np.zeros((FH, FW), np.uint8))
                sc.set_masks_device(t.data_ptr(), stride, FH * stride, 0, V)
                fused, staged = _valid_after_stage3(sc, V)
                for v in range(V):
                    want = refs[v][y0:y0 + h, x0:x0 + w]
                    assert np.array_equal(fused[v], want), (name, v, x0, y0)
                    assert np.array_equal(staged[v], want), (name, v, x0, y0)
                # the same device mask for every view
                sc.set_masks_device(t[1].data_ptr(), stride, 0, 0, V)
                sc.run(0, V)
                for v in range(V):
                    assert np.array_equal(sc.valid_map(0, view=v), refs[1][y0:y0 + h, x0:x0 + w]), (name, v)


def test_selected_quad_count_follows_masks_and_copies():
    """The count of selected quads (per-block words in mapped host memory, no atomics, no copy) decides a small launch's kernel: it
    follows sl3d_set_masks, survives sl3d_copy_view, and sl3d_last_fused_kernel_name reports the instantiation that ran."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 960, 540, 1024, 768, 9, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    dense = syn.default_mask(W, H)
    sparse = np.zeros((H, W), np.uint8)
    sparse[150:390, 300:700] = 1
    gated, small = ", 1, 0, true, false>", ", 1, 0, false, true>"
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=4) as sc:
        sc.set_calibration(*cal)
        sc.set_masks(np.stack([sparse, dense, sparse, dense]))
        for v in range(4):
            sc.synth_view(v, plane=(1.5 * v, 0.05, 0.04), view_id=v, noise=2)
        for v, want in ((0, gated), (1, small), (2, gated), (3, small)):
            sc.run(v, 1)
            assert sc.last_fused_kernel_name().endswith(want), (v, sc.last_fused_kernel_name())
        ref0 = sc.points(0)
        sc.copy_view(0, 1)           # view 1 is a duplicate of the sparse view now: frames, mask, and the count
        sc.run(1, 1)
        assert sc.last_fused_kernel_name().endswith(gated), sc.last_fused_kernel_name()
        got = sc.points(1)
        assert np.array_equal(got[1], ref0[1]) and np.array_equal(got[0], ref0[0], equal_nan=True)
        sc.run(0, 2)
        assert sc.last_fused_kernel_name().endswith(gated)
        sc.set_mask(dense, view=0)   # the source of the duplicate gets another mask: the duplicate keeps its own (sparse) mask and count
        sc.run(0, 1)
        assert sc.last_fused_kernel_name().endswith(small), sc.last_fused_kernel_name()
        sc.run(1, 1)
        assert sc.last_fused_kernel_name().endswith(gated), sc.last_fused_kernel_name()
        got = sc.points(1)
        assert np.array_equal(got[1], ref0[1]) and np.array_equal(got[0], ref0[0], equal_nan=True)
        sc.run_clouds(0, 1)
        assert ", 1, 2, " in sc.last_fused_kernel_name()


@pytest.mark.parametrize("Nv,Nh", [(7, 0), (0, 9), (0, 0)])
def test_axis_without_gray_planes_on_the_last_resident_view(Nv, Nh):
    """n_gray = 0 is a valid configuration (include/sl3d.h: 0..16): the code is 0 and the absolute phase is the shifted wrapped
    phase.  The padded straight-line kernels must not read past the frame stack for the empty axis of the LAST view of the context
    (ADVICE r4): every view of a full context, large and small launches, against the oracle."""
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    W, H, V = 168, 60, 6
    PW, PH, fwv, fwh = 64, 48, 64, 48
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    rng = np.random.default_rng(10 * Nv + Nh)
    planes = [([rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * Nv)],
               [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * Nh)]) for _ in range(V)]
    masks = np.stack([(rng.random((H, W)) < 0.9).astype(np.uint8) for _ in range(V)])
    with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, max_views=V) as sc:
        sc.set_calibration(*cal)
        sc.set_masks(masks)
        for v in range(V):
            sc.set_frames(0, planes[v][0], view=v)
            sc.set_frames(1, planes[v][1], view=v)
        sc.run(0, V)
        batch = [sc.points(v) for v in range(V)]
        sc.run(V - 1, 1)
        last = sc.points(V - 1)
        clouds = sc.fused_clouds(0, V)
    assert np.array_equal(last[1], batch[V - 1][1]) and np.array_equal(last[0], batch[V - 1][0], equal_nan=True)
    nvalid = 0
    for v in range(V):
        o = Oracle(W, H, PW, PH, Nv, Nh, fwv, fwh)
        o.set_mask(masks[v])
        o.set_calibration(*cal)
        oxyz, ovalid, _ = o.run_scan_rowmajor(*planes[v])
        assert np.array_equal(batch[v][1], ovalid), v
        assert_points_close(batch[v][0], oxyz, ovalid == 1)
        assert np.array_equal(clouds[v], batch[v][0][ovalid == 1]), v
        nvalid += int(ovalid.sum())
    assert nvalid > 100


# ---- small launches leave the scan of the segment counts to the consumer ---------------------------------------------------------------
@pytest.mark.parametrize("W,H", [(640, 203), (1021, 64), (96, 5)])
def test_consumers_scan_on_entry_after_small_cloud_launches(W, H):
    """sl3d_run_clouds over at most 4 views launches no k_seg_scan; whoever consumes the clouds gets offsets and totals on entry:
    the one-view download into pinned memory (k_seg_close<.., SCAN>: scan + gap closing + count in one launch, also with a
    destination smaller than the cloud), the contiguous device copy (sl3d_get_cloud_counts), the raw segments (the scan runs then),
    the registration, and a large launch afterwards -- every cloud equals xyz[valid] of the dense launch, in scan order."""
    S, syn = pkg("scanner"), pkg("synth")
    PW, PH, N, fw, V = 512, 384, 8, 4, 6
    rng = np.random.default_rng(W + H)
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V) as sc:
        sc.set_calibration(*cal)
        masks = np.stack([_mask(rng, W, H, t) for t in range(V)])
        masks[0] = syn.default_mask(W, H)
        sc.set_masks(masks)
        for v in range(V):
            sc.synth_view(v, plane=(1.5 * v, 0.05, 0.04 - 0.01 * v), view_id=v, noise=2)
        sc.run(0, V)
        dense = [sc.points(v) for v in range(V)]
        want = [xyz[val == 1] for xyz, val in dense]
        pin = sc.pinned((W * H * 3,), np.float32)
        for v in range(V):
            sc.run_clouds(v, 1)
            pin[:] = -7.0
            n = sc.download_cloud_into(v, pin)                  # pinned, roomy: the scanning gap-closer
            assert n == len(want[v]), v
            assert np.array_equal(pin[:3 * n].reshape(n, 3), want[v]), v
            assert (pin[3 * n:3 * n + 3] == -7.0).all()         # nothing past the cloud
            if n > 10:                                          # a destination smaller than the cloud: its first points, the full count
                sc.run_clouds(v, 1)
                small = sc.pinned((3 * (n // 2),), np.float32)
                small[:] = -7.0
                assert sc.download_cloud_into(v, small) == n
                assert np.array_equal(small.reshape(-1, 3), want[v][:n // 2]), v
            sc.run_clouds(v, 1)
            page = np.full(W * H * 3, -7.0, np.float32)         # pageable: counts first (the scan runs), then the copy
            assert sc.download_cloud_into(v, page) == n and np.array_equal(page[:3 * n].reshape(n, 3), want[v])
        # the contiguous device copy of a small launch (scan on entry of the gap-closer), then the raw segments of the same launch
        sc.run_clouds(1, 3)
        ptr, stride, counts = sc.cloud_counts(1, 3)
        assert counts == [len(want[v]) for v in (1, 2, 3)]
        for k, v in enumerate((1, 2, 3)):
            a = np.empty((counts[k], 3), np.float32)
            if counts[k]:
                sc._d2h(a, ptr + 12 * k * stride)
            assert np.array_equal(a, want[v]), v
        seg, counts2 = sc.cloud_segments(1, 3)
        assert counts2 == counts
        offs = np.empty(seg.n_segments, np.uint64)
        cnts = np.empty(seg.n_segments, np.uint32)
        for k in range(3):
            sc._d2h(offs, seg.offsets + 8 * k * seg.view_stride_segments)
            sc._d2h(cnts, seg.counts + 4 * k * seg.view_stride_segments)
            assert int(cnts.sum()) == counts[k] and np.array_equal(offs, np.concatenate([[0], np.cumsum(cnts.astype(np.uint64))[:-1]]).astype(np.uint64))
        # registration straight from a small launch, against the dense route
        sc.run_clouds(2, 2)
        reg = sc.register_clouds(2, 2, 10.0, 0.0, 300.0, 7.5)
        sc.run(2, 2)
        ref = sc.register_views(2, 2, 10.0, 0.0, 300.0, 7.5)
        assert np.array_equal(reg, ref)
        # and a large launch afterwards scans by itself
        got = sc.fused_clouds(0, V)
        for v in range(V):
            assert np.array_equal(got[v], want[v]), v
