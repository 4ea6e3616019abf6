"""GPU tests (-m gpu) added in round 3: the T0 known answer through the product, threads at the boundary, the drop-in
shim's device-side [col][row] layout, per-stripe host assembly of groups, the segmented (wait-free) compaction."""
import numpy as np
import pytest

from conftest import golden_calibration, pkg
from test_oracle import golden_relative_geometry, relative_from_projection

pytestmark = pytest.mark.gpu


def test_T0_product_matches_reference_held_opencv_output():
    """The product's host T0 (sl3d_set_calibration: Rodrigues + K[R|t], 7/triangulation.cpp:1069-1116) against the reference-held
    known answer (Triangulation/Relative_geometry/*.xml = OpenCV 2.4's own cvRodrigues2 / cvTranspose / cvGEMM / cvSub output on
    the rotation and translation vectors stage 7 reads, 6/system_calibration.cpp:1488-1516).  With K = I the product's A is
    [R|t]; Rc*Rp^T and tc - (Rc*Rp^T)*tp formed from it in cvGEMM's summation order must equal the 12 stored doubles to <= 1 ulp
    (observed: bit for bit), and the product's A = K[R|t] with the real intrinsics must equal the pinned oracle's."""
    from oracle.oracle import Oracle
    S = pkg("scanner")
    ct, dims = golden_calibration()
    cal = dict(zip(("Kc", "dc", "rc", "tc", "Kp", "dp", "rp", "tp"), ct))
    kat_R, kat_t = golden_relative_geometry()
    I3, z5 = np.eye(3).ravel(), np.zeros(5)
    with S.Scanner(64, 32, dims["PW"], dims["PH"], dims["N_v"], dims["N_h"], dims["fw_v"], dims["fw_h"]) as sc:
        sc.set_calibration(I3, z5, cal["rc"], cal["tc"], I3, z5, cal["rp"], cal["tp"])
        A_cam, A_proj = sc.projection_matrices()
        R, t = relative_from_projection(A_cam, A_proj)
        ulp_R = np.abs(R.view(np.int64) - kat_R.view(np.int64)).max()
        ulp_t = np.abs(t.view(np.int64) - kat_t.view(np.int64)).max()
        assert ulp_R <= 1 and ulp_t <= 1, (ulp_R, ulp_t)
        # the real intrinsics: A = K [R|t] equals the oracle's (whose R is pinned by the same known answer)
        sc.set_calibration(*ct)
        A_cam, A_proj = sc.projection_matrices()
    o = Oracle(16, 16, dims["PW"], dims["PH"], dims["N_v"], dims["N_h"], dims["fw_v"], dims["fw_h"])
    o.set_calibration(*ct)
    oc, op = (np.array(a).reshape(3, 4) for a in o.projection_matrices())
    assert np.array_equal(A_cam, oc) and np.array_equal(A_proj, op)


def test_two_threads_at_the_boundary(tmp_path):
    """SURVEY 8b, threading row: contexts are independent and thread-safe with respect to each other.  A fresh process
    (tests/native/thread_driver.cpp) starts two std::threads behind one gate, so the first sl3d_create of the process -- the
    one-time device atan2 self-check under its mutex -- is raced; each thread then creates / uses / destroys its own context
    three times, concurrently (thread 0: 320x240, N = 7, parity mode through the four stage entry points; thread 1: 200x150,
    N = 6, two view slots, timed mode: fused kernel + in-kernel compaction).  The dumped results of the last round equal the
    oracle's on the very frames the contexts processed."""
    import os
    import subprocess
    from conftest import ROOT, assert_points_close
    from oracle.oracle import Oracle
    syn = pkg("synth")
    exe = str(tmp_path / "thread_driver")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "thread_driver.cpp"), "-L" + os.path.join(ROOT, "3dscan_amd"), "-lsl3d",
                           "-Wl,-rpath," + os.path.join(ROOT, "3dscan_amd"), "-o", exe])
    shapes = [(320, 240, 512, 384, 7, 4), (200, 150, 256, 192, 6, 4)]
    cals = [syn.cal_tuple(syn.synth_rig(W, H, PW, PH)) for W, H, PW, PH, _, _ in shapes]
    np.concatenate([np.concatenate(c) for c in cals]).astype(np.float64).tofile(str(tmp_path / "cal.bin"))
    subprocess.run([exe, str(tmp_path / "cal.bin"), str(tmp_path / "t"), "3"], check=True, timeout=600)
    for tid, ((W, H, PW, PH, N, fw), cal) in enumerate(zip(shapes, cals)):
        raw = open(tmp_path / f"t{tid}.bin", "rb").read()
        hdr = np.frombuffer(raw, np.int32, 8)
        assert list(hdr[:6]) == [W, H, PW, PH, N, fw]
        views, keep = int(hdr[6]), bool(hdr[7])
        px, ppa, off = W * H, 3 + 2 * N, 32
        def take(dtype, count):
            nonlocal off
            a = np.frombuffer(raw, dtype, count, off)
            off += a.nbytes
            return a
        mask = take(np.uint8, px).reshape(H, W)
        frames = take(np.uint8, views * 2 * ppa * px).reshape(views, 2, ppa, H, W)
        valid = take(np.uint8, views * px).reshape(views, H, W)
        xyz = take(np.float32, views * px * 3).reshape(views, H, W, 3)
        cpm = take(np.int64, px * 2).reshape(H, W, 2) if keep else None
        counts = take(np.int64, views) if not keep else None
        cloud = take(np.float32, int(counts.sum()) * 3).reshape(-1, 3) if not keep else None
        assert off == len(raw)
        o = Oracle(W, H, PW, PH, N, N, fw, fw)
        o.set_mask(mask)
        o.set_calibration(*cal)
        at = 0
        for v in range(views):
            o.run_scan(list(frames[v, 0]), list(frames[v, 1]))
            ov = o.valid_map(2) == 1
            assert ov.sum() > 0.5 * px
            assert np.array_equal(valid[v] == 1, ov), f"thread {tid} view {v}: valid map"
            assert_points_close(xyz[v], o.intersection_points(), ov)
            if keep:
                assert np.array_equal(cpm[ov], o.c_p_map()[ov]), f"thread {tid}: correspondences"
            else:
                n = int(counts[v])
                assert n == int(ov.sum()) and np.array_equal(cloud[at:at + n], xyz[v][ov]), f"thread {tid} view {v}: cloud"
                at += n


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


# ---- the reference's own layouts on the device ------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(160, 120), (333, 77), (65, 33)])
def test_colrow_globals_and_mask(shape):
    """sl3d_get_global_colrow: every image-shaped global in the reference's [col][row] layout and type
    (common_variables.h:12-21,56-62) == the transpose of the row-major getter, for a whole-frame context and for two row stripes
    writing their rows into one array (out_height / out_row0); sl3d_set_mask_colrow (selected_region as int [col][row]) ==
    sl3d_set_mask on the transposed bytes, values other than 1 unselected; sl3d_set_frames_range == sl3d_set_frames."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H = shape
    PW, PH, Nv, Nh, fw = 256, 192, 6, 5, 8
    cap = syn.make_capture(W, H, PW, PH, Nv, Nh, fw, fw, noise=2)
    rng = np.random.default_rng(W)
    mask = cap["mask"].copy()
    mask[H // 3:H // 2, W // 4:W // 2] = 0
    mask[rng.random((H, W)) < 0.02] = 0
    sel = mask.T.astype(np.int32).copy()           # [col][row]
    sel[rng.random(sel.shape) < 0.01] = 257        # not 1: unselected (and its low byte is 1)
    mask_eq = (sel.T == 1).astype(np.uint8)
    cal = syn.cal_tuple(cap["cal"])

    def feed(sc, rows=slice(None), colrow=True, ranged=True):
        sc.set_calibration(*cal)
        if colrow:
            sc.set_mask_colrow(sel)
        else:
            sc.set_mask(mask_eq)
        for a, planes, N in ((0, cap["planes_v"], Nv), (1, cap["planes_h"], Nh)):
            pl = [p[rows] for p in planes]
            if ranged:
                sc.set_frames_range(a, 3, pl[3:])
                sc.set_frames_range(a, 0, pl[:3])
            else:
                sc.set_frames(a, pl)
        sc.run_stages()

    getters = {0: lambda s: s.valid_map(0), 1: lambda s: s.valid_map(1), 2: lambda s: s.valid_map(2), 3: lambda s: s.wrapped_phase(0),
               4: lambda s: s.wrapped_phase(1), 5: lambda s: s.unwrapped_phase(0), 6: lambda s: s.unwrapped_phase(1), 7: lambda s: s.code(0),
               8: lambda s: s.code(1), 9: lambda s: s.intersection_points()}
    with S.Scanner(W, H, PW, PH, Nv, Nh, fw, fw, keep_stages=True) as ref, S.Scanner(W, H, PW, PH, Nv, Nh, fw, fw, keep_stages=True) as sc:
        feed(ref, colrow=False, ranged=False)
        feed(sc)
        whole = {}
        for which, get in getters.items():
            rm = get(ref)
            cr = sc.global_colrow(which)
            assert cr.shape[:2] == (W, H)
            want = rm.transpose(1, 0, 2) if which == 9 else rm.T
            assert cr.dtype == (np.float64 if which == 9 else np.float32 if 3 <= which <= 6 else np.int32)
            assert np.array_equal(cr, want.astype(cr.dtype), equal_nan=True), which
            whole[which] = cr
    # two stripes write their rows of every column into one [W][H] array
    h0 = H // 2 + 1
    outs = {w: np.full((W, H) + ((3,) if w == 9 else ()), -7, dtype=whole[w].dtype) for w in getters}
    for r0, n in ((0, h0), (h0, H - h0)):
        with S.Scanner(W, n, PW, PH, Nv, Nh, fw, fw, keep_stages=True, full_size=(W, H), origin=(0, r0)) as st:
            feed(st, rows=slice(r0, r0 + n))
            for w in getters:
                st.global_colrow(w, out=outs[w], row0=r0)
    for w in getters:
        assert np.array_equal(outs[w], whole[w], equal_nan=True), w


# ---- groups: assembly for the host consumer -----------------------------------------------------------------------------------
@pytest.mark.parametrize("n_stripes", [1, 4, 7, 8])
def test_group_host_assembly_equals_single_context(n_stripes):
    """sl3d_group_download_points (every stripe copies its rows straight into the caller's dense images) and
    sl3d_group_process_views (the three-stream host-buffer pipeline per stripe, all stripes enqueued before any is waited for)
    == one whole-frame context bit for bit: 1 / 4 / 7 / 8 stripes of unequal heights on device 0, pinned and pageable buffers,
    more views than view slots."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw, NV, SLOTS = 320, 203, 512, 384, 7, 4, 5, 2
    rng = np.random.default_rng(n_stripes)
    caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, view=v, noise=2, plane=(1.5 * v, 0.05, 0.04 + 0.01 * v)) for v in range(NV)]
    cal = syn.cal_tuple(caps[0]["cal"])
    mask = caps[0]["mask"].copy()
    mask[H // 3:H // 2, W // 4:W // 2] = 0
    mask[rng.random((H, W)) < 0.02] = 0
    frames = np.stack([np.stack(c["planes_v"] + c["planes_h"]) for c in caps])     # (NV, 34, H, W)
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v, c in enumerate(caps):
            sc.set_mask(mask, view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, NV)
        want = [sc.points(v) for v in range(NV)]
    with S.Group(W, H, PW, PH, N, N, fw, fw, [0] * n_stripes, max_views=NV) as g:
        g.set_calibration(*cal)
        for v, c in enumerate(caps):
            g.set_mask(mask, view=v)
            g.set_frames(0, c["planes_v"], view=v)
            g.set_frames(1, c["planes_h"], view=v)
        g.run(0, NV)
        xyz, valid = g.download_points(0, NV)
        for v in range(NV):
            assert np.array_equal(valid[v], want[v][1]) and np.array_equal(xyz[v], want[v][0], equal_nan=True), v
        xyz2, valid2 = g.download_points(1, 3)                                   # a sub-range
        assert np.array_equal(valid2, valid[1:4]) and np.array_equal(xyz2, xyz[1:4], equal_nan=True)
    # host-resident views through the stripes' pipelines, fewer slots than views; pinned, then pageable buffers
    with S.Group(W, H, PW, PH, N, N, fw, fw, [0] * n_stripes, max_views=SLOTS) as g, S.Scanner(8, 8, 16, 16, 3, 3, 4, 4) as pin:
        g.set_calibration(*cal)
        for v in range(SLOTS):
            g.set_mask(mask, view=v)
        pf = pin.pinned(frames.shape, np.uint8)
        pf[:] = frames
        px, pv = pin.pinned((NV, H, W, 3), np.float32), pin.pinned((NV, H, W), np.uint8)
        for f, ox, ov in ((pf, px, pv), (frames, None, None)):
            xyz, valid = g.process_views(f, xyz=ox, valid=ov)
            for v in range(NV):
                assert np.array_equal(valid[v], want[v][1]) and np.array_equal(xyz[v], want[v][0], equal_nan=True), v


# ---- the small-launch instantiation ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(640, 64, 8, 4), (333, 65, 9, 4), (1021, 127, 7, 4), (64, 201, 6, 8), (1920, 1080, 10, 2), (1920, 271, 10, 2)])
@pytest.mark.parametrize("rig", ["reference", "distorted"])
def test_one_view_launch_equals_the_batch(shape, rig):
    """A view's result does not depend on the batch it was launched in: points and valid map of every view launched alone (the
    small-launch instantiation: planes requested before the mask is known, no reciprocal table, coalesced stores) == the same
    view inside a batch of 6 (the other instantiation) bit for bit -- even and odd heights, masks with holes, empty masks, masks
    that select two middle rows or the last row only; then every one-view launch against the oracle itself.
    Written for the banded one-view launch that round 3 built, measured and dropped (profiles/r03_bands_ab.txt)."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, N, fw = shape
    PW, PH = (512, 384) if W < 1900 else (1920, 1080)
    rng = np.random.default_rng(W + 31 * H)
    masks = _masks(rng, W, H)
    half = np.ones((H, W), np.uint8)
    half[(H + 1) // 2 - 1:(H + 1) // 2 + 1] = 0          # two middle rows
    bottom = np.zeros((H, W), np.uint8)
    bottom[H - 1] = 1                                    # only the last row
    masks += [half, bottom]
    NV = len(masks)
    r = syn.synth_rig(W, H, PW, PH)
    if rig == "distorted":
        r["dp"] = np.array([0.05, -0.02, 0.001, -0.0005, 0.01])
    cal = syn.cal_tuple(r)
    def load(sc):
        sc.set_calibration(*cal)
        for v, m in enumerate(masks):
            sc.set_mask(m, view=v)
            sc.synth_view(v, plane=(1.5 * v, 0.05, 0.04 + 0.01 * v), view_id=v, noise=2)
    # two contexts, so that no launch finds the other's results in its output planes
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sb, S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as ss:
        load(sb)
        load(ss)
        sb.run(0, NV)
        for v in reversed(range(NV)):
            ss.run(v, 1)
        n_valid = 0
        for v in range(NV):
            bx, bv = sb.points(v)
            xyz, val = ss.points(v)
            assert np.array_equal(val, bv), v
            assert np.array_equal(xyz, bx, equal_nan=True), v
            n_valid += int(bv.sum())
        assert n_valid > 0
        assert int(ss.points(NV - 1)[1][:H - 1].sum()) == 0 and int(ss.points(NV - 1)[1][H - 1].sum()) > 0
        # ... and DIRECTLY against the oracle, not only against the product's own batch launch: every view of the one-view launches
        # (at 1080p: the reference's real call pattern, one scan per launch, BASELINE configs[1] literally) -- valid map bit exact,
        # points within 1e-5, on the very frames the context processed
        from conftest import assert_points_close
        from oracle.oracle import Oracle
        for v in range(NV):
            o = Oracle(W, H, PW, PH, N, N, fw, fw)
            o.set_mask(masks[v])
            o.set_calibration(*cal)
            oxyz, ovalid, _ = o.run_scan_rowmajor(ss.frames(0, v), ss.frames(1, v))
            xyz, val = ss.points(v)
            assert np.array_equal(val, ovalid), v
            assert_points_close(xyz, oxyz, ovalid == 1)
