"""GPU tests (-m gpu): O1 / N2 -- the ordered compaction inside the fused kernel (segmented clouds, sl3d_run_clouds) and every consumer of it
(contiguous copy, zero-copy / DMA / clamped host downloads, registration, colour gather): cloud == xyz[valid] == the oracle's row-major append
(8/save_point_cloud.cpp:85-104)."""
import numpy as np
import pytest

from conftest import assert_points_close, golden_calibration, load_golden, pkg
from oracle.oracle import Oracle
from test_oracle import golden_relative_geometry, relative_from_projection

pytestmark = pytest.mark.gpu


def _scanner():
    return pkg("scanner")


def test_coloured_cloud_gather():
    """N2: save_point_cloud()'s colour gather on the device: r,g,b of the texture pixel of every valid point, in the
    reference's scan order (8/save_point_cloud.cpp:46-52,70-72,85-104); views keep separate textures."""
    syn = pkg("synth")
    S = _scanner()
    W, H, PW, PH, N, fw = 200, 120, 256, 192, 6, 8   # W is not a multiple of 16: the texture has a pitch too
    rng = np.random.default_rng(11)
    with S.Scanner(W, H, PW, PH, N, 5, fw, fw, max_views=2) as sc:
        caps, texs = [], []
        for view in range(2):
            cap = syn.make_capture(W, H, PW, PH, N, 5, fw, fw, plane=(3.0 * view, 0.05, 0.04), view=view, noise=1)
            mask = cap["mask"].copy()
            mask[rng.random((H, W)) < 0.05] = 0
            tex = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)  # B,G,R
            sc.set_calibration(*syn.cal_tuple(cap["cal"]))
            sc.set_mask(mask, view=view)
            sc.set_frames(0, cap["planes_v"], view=view)
            sc.set_frames(1, cap["planes_h"], view=view)
            sc.set_texture(tex, view=view)
            caps.append((cap, mask))
            texs.append(tex)
        with pytest.raises(S.Sl3dError):
            sc.set_texture(texs[0], view=2)
        sc.run(0, 2)
        for view in range(2):
            xyz, valid = sc.points(view)
            v = valid == 1
            cx, crgb = sc.cloud_rgb(view)
            assert np.array_equal(cx, xyz[v])
            assert np.array_equal(crgb, texs[view][v][:, ::-1])   # b,g,r -> r,g,b
            assert np.array_equal(sc.cloud(view), cx)
        # batched compaction of both views: the same clouds, three launches in all
        both = sc.clouds(0, 2)
        assert sc.compact_views(0, 2) == [len(b) for b in both]
        for view in range(2):
            xyz, valid = sc.points(view)
            assert np.array_equal(both[view], xyz[valid == 1])
        assert np.array_equal(sc.clouds(1, 1)[0], both[1])
    with S.Scanner(W, H, PW, PH, N, 5, fw, fw) as sc:
        with pytest.raises(S.Sl3dError):
            sc.cloud_rgb(0)   # no texture set


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


# ---- ordered compaction inside the fused kernel ------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(640, 200, 8, 8, 4), (333, 77, 9, 9, 4), (200, 120, 6, 5, 16), (1021, 64, 7, 7, 4), (64, 3, 6, 6, 8)])
@pytest.mark.parametrize("dist_proj", [False, True])
def test_fused_compaction_equals_dense(shape, dist_proj):
    """sl3d_run_clouds (segmented ordered clouds written by the fused kernel itself): the cloud of every view equals xyz[valid]
    of the dense pass (8/save_point_cloud.cpp:85-104 order), the valid map is the same, repeated launches and sub-ranges too."""
    S, syn = _S(), pkg("synth")
    W, H, Nv, Nh, fw = shape
    PW, PH, NV = 512, 384, 5
    rng = np.random.default_rng(W * 7 + H)
    caps = [syn.make_capture(W, H, PW, PH, Nv, Nh, fw, fw, view=v, noise=2, plane=(2.0 * v, 0.05, 0.03 + 0.01 * v)) for v in range(NV)]
    cal = {k: np.array(v, dtype=np.float64).copy() for k, v in caps[0]["cal"].items()}
    if dist_proj:
        cal["dp"] = np.array([0.04, -0.01, 0.001, -0.0005, 0.0])
    with S.Scanner(W, H, PW, PH, Nv, Nh, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*syn.cal_tuple(cal))
        for v, c in enumerate(caps):
            sc.set_mask(_random_mask(rng, W, H) if v else c["mask"], view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, NV)
        dense = [sc.points(v) for v in range(NV)]
        for rep in range(3):
            first, n = (0, NV) if rep != 1 else (1, NV - 2)   # a sub-range too
            clouds = sc.fused_clouds(first, n)
            for k, cl in enumerate(clouds):
                xyz, val = dense[first + k]
                assert cl.shape[0] == int((val == 1).sum()), (rep, k)
                assert np.array_equal(cl, xyz[val == 1]), (rep, k)
                assert np.array_equal(sc.valid_map(view=first + k), val)
        assert sum(len(c) for c in clouds) > 0
        # the older three-launch compaction still agrees
        for v in range(NV):
            assert np.array_equal(sc.cloud(v), dense[v][0][dense[v][1] == 1])


def test_fused_compaction_full_hd_batch():
    """BASELINE configs[1] shape: 16 views of 1920x1080 in one launch (2025 tiles per view, 4 views per lane): every
    cloud equals xyz[valid]; a sparse mask and an empty mask included.  The dense planes AND the clouds of the segmented kernel
    (k_fused<..., CMODE = 2>) are compared DIRECTLY with the oracle: valid map / point count bit exact, the cloud against the
    oracle's own row-major append (8/save_point_cloud.cpp:85-104) point by point within 1e-5."""
    S, syn = _S(), pkg("synth")
    W, H, N, fw, NV = 1920, 1080, 10, 2, 16
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    rng = np.random.default_rng(5)
    with S.Scanner(W, H, W, H, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v in range(NV):
            m = syn.default_mask(W, H)
            if v == 3:
                m[:] = 0
            elif v == 5:
                m = _random_mask(rng, W, H, holes=30)
            elif v == 7:
                m[:, ::2] = 0
            sc.set_mask(m, view=v)
            sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
        sc.run(0, NV)
        dense = [sc.points(v) for v in range(NV)]
        for rep in range(2):
            clouds = sc.fused_clouds(0, NV)
            for v in range(NV):
                xyz, val = dense[v]
                assert np.array_equal(clouds[v], xyz[val == 1]), v
        assert len(clouds[3]) == 0 and len(clouds[0]) > 1_900_000
        # the batched launch (the benchmark's exact shape) against the oracle, view by view: full, empty, holes, every other column, last
        for v in (0, 3, 5, 7, 15):
            m = syn.default_mask(W, H)
            if v == 3:
                m[:] = 0
            elif v == 5:
                m = _random_mask(np.random.default_rng(5), W, H, holes=30)
            elif v == 7:
                m[:, ::2] = 0
            o = Oracle(W, H, W, H, N, N, fw, fw)
            o.set_mask(m)
            o.set_calibration(*cal)
            oxyz, ovalid, _ = o.run_scan_rowmajor(sc.frames(0, v), sc.frames(1, v))
            assert np.array_equal(dense[v][1], ovalid), v
            assert_points_close(dense[v][0], oxyz, ovalid == 1)
            # the cloud the compacting instantiation wrote, against the oracle's append in scan order (not against the dense pass)
            ocloud = oxyz[ovalid == 1]
            assert clouds[v].shape == ocloud.shape, v
            if len(ocloud):
                assert_points_close(clouds[v], ocloud, np.ones(len(ocloud), bool))


# ---- segmented clouds: the wait-free ordered compaction and its consumers ------------------------------------------------
def _masks(rng, W, H):
    full = np.zeros((H, W), np.uint8)
    full[1:H - 1, 1:W - 1] = 1
    sparse = (rng.random((H, W)) < 0.07).astype(np.uint8) * full
    holes = full.copy()
    holes[H // 3:H // 2, W // 4:W // 2] = 0
    holes[rng.random((H, W)) < 0.02] = 0
    return [full, sparse, np.zeros((H, W), np.uint8), holes]


@pytest.mark.parametrize("shape", [(640, 200, 8, 4), (333, 77, 9, 4), (1021, 64, 7, 4), (64, 3, 6, 8), (1920, 270, 10, 2)])
def test_segmented_clouds_and_their_consumers(shape):
    """sl3d_run_clouds (default, segmented): (1) the segments where they lie -- counts, offsets = their exclusive scan, every
    segment's points = xyz[valid] of its 256 scan pixels; (2) sl3d_get_cloud_counts' contiguous device copy, sl3d_download_clouds
    into pinned memory (the gap-closing kernel writes the host buffer) and into pageable memory, and with SL3D_ZEROCOPY off:
    all == xyz[valid] of the dense pass bit for bit; (3) sl3d_register_clouds == sl3d_register_views == the oracle's
    register_point_clouds (9/register_point_clouds.cpp:83-148) bit for bit."""
    import os
    from oracle import oracle as O
    S, syn = pkg("scanner"), pkg("synth")
    W, H, N, fw = shape
    PW, PH = (512, 384) if W < 1900 else (1920, 1080)
    rng = np.random.default_rng(W + 31 * H)
    masks = _masks(rng, W, H)
    NV = len(masks)
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v, m in enumerate(masks):
            sc.set_mask(m, view=v)
            sc.synth_view(v, plane=(1.5 * v, 0.05, 0.04 + 0.01 * v), view_id=v, noise=2)
        sc.run(0, NV)
        dense = [sc.points(v) for v in range(NV)]
        want = [xyz[val == 1] for xyz, val in dense]
        for rep in range(2):
            sc.run_clouds(0, NV)
            seg, counts = sc.cloud_segments(0, NV)
            assert counts == [len(w) for w in want]
            assert seg.segment_points == 256 and seg.view_stride_segments == seg.n_segments
            pitch = (W + 15) // 16 * 16
            assert seg.n_segments == 4 * ((pitch * H // 4 + 255) // 256) and seg.view_stride_points == pitch * H
            sc_counts = np.empty(NV * seg.n_segments, np.uint32)
            sc_offs = np.empty(NV * seg.n_segments, np.uint64)
            raw = np.empty(NV * seg.view_stride_points * 3, np.float32)
            sc._d2h(sc_counts, seg.counts); sc._d2h(sc_offs, seg.offsets); sc._d2h(raw, seg.xyz)
            sc_counts, sc_offs = sc_counts.reshape(NV, -1), sc_offs.reshape(NV, -1)
            raw = raw.reshape(NV, seg.view_stride_points, 3)
            for v in range(NV):
                assert int(sc_counts[v].sum()) == counts[v]
                assert np.array_equal(sc_offs[v], np.concatenate([[0], np.cumsum(sc_counts[v].astype(np.uint64))[:-1]]))
                vpad = np.zeros((H, pitch), np.uint8)
                vpad[:, :W] = dense[v][1]
                per_seg = np.add.reduceat(vpad.ravel().astype(np.int64), np.arange(0, pitch * H, 256))
                assert np.array_equal(sc_counts[v][:len(per_seg)], per_seg) and not sc_counts[v][len(per_seg):].any()
                got = np.concatenate([raw[v, 256 * s:256 * s + c] for s, c in enumerate(sc_counts[v]) if c] or [np.zeros((0, 3), np.float32)])
                assert np.array_equal(got, want[v]), v
            # the contiguous device copy made on demand
            ptr, stride, c2 = sc.cloud_counts(0, NV)
            assert c2 == counts
            for v in range(NV):
                a = np.empty((counts[v], 3), np.float32)
                if counts[v]:
                    sc._d2h(a, ptr + 12 * v * stride)
                assert np.array_equal(a, want[v])
            # host copies: pinned (zero copy), pageable, zero copy switched off, a sub-range
            total = sum(counts)
            pin = sc.pinned((max(total, 1) * 3,), np.float32)
            pin[:] = -1
            for out, env in ((pin, None), (None, None), (pin, "0")):
                if env is not None:
                    os.environ["SL3D_ZEROCOPY"] = env
                try:
                    got = sc.download_clouds(0, NV, out=out)
                finally:
                    os.environ.pop("SL3D_ZEROCOPY", None)
                for v in range(NV):
                    assert np.array_equal(got[v], want[v]), (v, env)
            got = sc.download_clouds(1, 2)
            assert np.array_equal(got[0], want[1]) and np.array_equal(got[1], want[2])
        # registration straight from the segments
        t = (12.5, -3.25, 310.0)
        reg_seg = sc.register_clouds(0, NV, *t, 17.5)
        reg_dense = sc.register_views(0, NV, *t, 17.5)
        assert np.array_equal(reg_seg, reg_dense)
        assert np.array_equal(reg_seg, O.register_point_clouds(want, *t, 17.5))


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
