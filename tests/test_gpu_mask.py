"""GPU tests (-m gpu): H0 / S3b / S3d on the device -- k_mask_prepare's bit-plane arithmetic through every way a mask reaches it (one view, several
views per launch, pageable / pinned / device-resident sources, windows on every frame border) against the oracle's literal scan of
3/wrapped_phase.cpp:253-279, and the selected-quad count that picks a small launch's kernel.  (The same through the fused kernel itself:
test_gpu_mask_fused.py.)"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_points_close, pkg
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _S():
    return pkg("scanner")


def _random_mask(rng, W, H, holes=6):
    m = np.zeros((H, W), np.uint8)
    m[1:H - 1, 1:W - 1] = 1
    for _ in range(holes):
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        w, h = int(rng.integers(1, max(2, W // 3))), int(rng.integers(1, max(2, H // 3)))
        m[y:y + h, x:x + w] = rng.integers(0, 2)
    m[rng.integers(0, H, 40), rng.integers(0, W, 40)] = 0
    m[rng.integers(0, H, 10), rng.integers(0, W, 10)] = 7   # selected iff == 1
    return m


# ---- mask preparation on the device -------------------------------------------------------------------------------------
def test_device_mask_preparation_windows_and_borders():
    """sl3d_set_mask prepares the mask on the device (normalisation + border band by k_mask_prepare): valid maps of
    windows that touch every frame border, with arbitrary mask bytes, equal the oracle's boundary removal; pinned and
    pageable sources give the same."""
    S, syn = _S(), pkg("synth")
    FW, FH, PW, PH, N, fw = 150, 90, 256, 192, 6, 8
    rng = np.random.default_rng(11)
    cap = syn.make_capture(FW, FH, PW, PH, N, N, fw, fw, noise=1)
    cal = syn.cal_tuple(cap["cal"])
    for trial in range(4):
        mask = _random_mask(rng, FW, FH, holes=10)
        if trial == 0:
            mask[:] = 1   # border pixels selected too
        o = Oracle(FW, FH, PW, PH, N, N, fw, fw)
        o.set_mask(mask)
        o.set_calibration(*cal)
        o.run_scan(cap["planes_v"], cap["planes_h"])
        vo = o.valid_map(2)
        for (x0, y0, w, h) in [(0, 0, FW, FH), (0, 0, 70, 40), (83, 51, 67, 39), (5, 0, 100, 90), (0, 7, 150, 50), (31, 29, 17, 5)]:
            with S.Scanner(w, h, PW, PH, N, N, fw, fw, full_size=(FW, FH), origin=(x0, y0), keep_stages=True) as sc:
                sc.set_calibration(*cal)
                pm = sc.pinned(mask.shape, np.uint8)
                pm[:] = mask
                for src in (mask, pm):
                    sc.set_mask(src)
                    sc.set_frames(0, [p[y0:y0 + h, x0:x0 + w] for p in cap["planes_v"]])
                    sc.set_frames(1, [p[y0:y0 + h, x0:x0 + w] for p in cap["planes_h"]])
                    sc.run()
                    assert np.array_equal(sc.valid_map(0), o.valid_map(0)[y0:y0 + h, x0:x0 + w]), (trial, x0, y0)
                    assert np.array_equal(sc.valid_map(2), vo[y0:y0 + h, x0:x0 + w]), (trial, x0, y0)
                    sc.run_stages()
                    assert np.array_equal(sc.valid_map(2), vo[y0:y0 + h, x0:x0 + w]), (trial, x0, y0)


# ---- sparse selections: a small launch asks the mask first ------------------------------------------------------------------------
def test_small_launch_over_sparse_masks_takes_the_gated_kernel():
    """k_mask_prepare counts the quads of a view that hold a valid pixel; a launch of at most 4 views whose views are ALL known to be
    sparsely selected (< 65 % of the quads; the reference's real captures select 19 % of the frame) takes the large-launch
    instantiation, which requests a view's planes only for quads its valid bits leave standing, instead of the small-launch one,
    which requests them first (one 1080p view at 19 %: 15.8 us against 22.2, profiles/r04_sparse_mask.txt).  Which kernel runs must
    not change a bit of the result: one view alone and in launches of 2 and 4, sparse and dense views mixed, dense results and
    ordered clouds, against the oracle."""
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 960, 540, 1024, 768, 9, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    dense = syn.default_mask(W, H)
    sparse = np.zeros((H, W), np.uint8)
    sparse[150:390, 300:700] = 1                       # 18.5 % of the frame
    rng = np.random.default_rng(3)
    sparse[rng.random((H, W)) < 0.02] = 0
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=5) as sc:
        sc.set_calibration(*cal)
        for v in range(5):
            sc.set_mask(sparse if v < 4 else dense, view=v)     # (pageable source: the call returns after the count has landed)
            sc.synth_view(v, plane=(1.5 * v, 0.05, 0.04), view_id=v, noise=2)
        for n in (1, 2, 4):
            assert sc.fused_kernel_name(n).endswith(", 1, 0, true, false>"), sc.fused_kernel_name(n)          # views 0..n-1: all sparse
            assert sc.fused_kernel_name(n, clouds=True).endswith(", 1, 2, true, false>")
        sc.run(0, 4)
        batch = [sc.points(v) for v in range(4)]
        clouds = sc.fused_clouds(0, 4)
        sc.run(3, 2)                                            # views 3 (sparse) + 4 (dense): the small-launch kernel
        mixed = [sc.points(3), sc.points(4)]
        assert np.array_equal(mixed[0][1], batch[3][1]) and np.array_equal(mixed[0][0], batch[3][0], equal_nan=True)
        for v in range(4):
            sc.run(v, 1)
            one = sc.points(v)
            assert np.array_equal(one[1], batch[v][1]) and np.array_equal(one[0], batch[v][0], equal_nan=True), v
            assert np.array_equal(sc.fused_clouds(v, 1)[0], clouds[v]), v
        frames = [(sc.frames(0, v), sc.frames(1, v)) for v in (0, 3, 4)]
        # a new (dense) selection for view 0.  Since round 6 it is DEFERRED (evaluated by the launch that consumes it), so what is
        # known is still the view's LAST selection: sparse -> the GATED MASKIN form once more (plane requests behind the valid bits it
        # evaluates); that launch counts the new selection, and from then on the view takes the form that requests its planes at once
        sc.set_mask(dense, view=0)
        assert sc.fused_kernel_name(1).endswith(", 1, 4, true, false>"), sc.fused_kernel_name(1)
        sc.run(0, 1)
        sc.synchronize()
        assert sc.fused_kernel_name(1).endswith(", 1, 0, false, true>"), sc.fused_kernel_name(1)               # view 0 is dense now
        sc.set_mask(dense, view=0)
        assert sc.fused_kernel_name(1).endswith(", 1, 4, false, true>"), sc.fused_kernel_name(1)               # ... and its next mask rides along
        assert sc.fused_kernel_name(5).endswith(", 1, 0, true, true>")                                         # a large launch, not all sparse: early requests
    for (v, m, got), fr in zip(((0, sparse, batch[0]), (3, sparse, batch[3]), (4, dense, mixed[1])), frames):
        o = Oracle(W, H, PW, PH, N, N, fw, fw)
        o.set_mask(m)
        o.set_calibration(*cal)
        oxyz, ovalid, _ = o.run_scan_rowmajor(*fr)
        assert int(ovalid.sum()) > 50_000
        assert np.array_equal(got[1], ovalid), v
        assert_points_close(got[0], oxyz, ovalid == 1)


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
    follows sl3d_set_masks, survives sl3d_copy_view, and sl3d_last_fused_kernel_name reports the instantiation that ran.  (An
    SL3D_FLAG_EAGER_MASK context: every mask is prepared -- and counted -- by k_mask_prepare when it is set; what the counts of the
    deferred route decide is test_gpu_mask_fused.py::test_fused_mask_route_counts_selected_quads.)"""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 960, 540, 1024, 768, 9, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    dense = syn.default_mask(W, H)
    sparse = np.zeros((H, W), np.uint8)
    sparse[150:390, 300:700] = 1
    gated, small = ", 1, 0, true, false>", ", 1, 0, false, true>"
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=4, eager_mask=True) as sc:
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
