"""GPU tests (-m gpu): H0 / S3b / S3d INSIDE the fused launch.  A timed context defers the preparation of a selection handed over for
at most 4 views; the next small launch over such views is a MASKIN launch (3dscan_amd/csrc/sl3d_fused.h: the fused kernel evaluates
3/wrapped_phase.cpp:106-115, :253-279 itself from the raw selection and leaves every plane and count k_mask_prepare would have left).
Every mask test of test_gpu_mask.py is repeated here through that route: against the oracle's literal scan, against the two-kernel
route of an SL3D_FLAG_EAGER_MASK context byte for byte, through every hand-over (pageable / pinned / device-resident in place /
[col][row]), on windows at every frame border, with several views per lane, dense planes and segmented clouds; and small / large / MASKIN launches against the oracle on ill-conditioned pixels."""
import numpy as np
import pytest

from conftest import assert_points_close, pkg

pytestmark = pytest.mark.gpu

MASKIN_DENSE, MASKIN_CLOUDS = ", 4, false, true>", ", 6, false, true>"
MASKIN_GATED, MASKIN_GATED_CLOUDS = ", 4, true, false>", ", 6, true, false>"   # the form for views known to be sparsely selected


def _mask(rng, W, H, trial):
    p = rng.choice([0.05, 0.5, 0.9, 0.98])
    m = (rng.random((H, W)) < p).astype(np.uint8)
    if trial % 3 == 0:
        m[:] = 0
        for _ in range(6):
            y, x, h, w = rng.integers(0, H), rng.integers(0, W), rng.integers(1, H + 1), rng.integers(1, W + 1)
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


def _mask_plane(sc, view):
    b = sc.device_buffers()
    a = np.empty((sc.H + 4, b.mask_pitch), np.uint8)
    sc._d2h(a, b.mask + view * b.mask_view_stride)
    return a


def _dense_history(sc, FW, FH, V):
    """What is known about a view's LAST selection decides which MASKIN form its next scan takes (sparse: the gated one): give every
    view a fully selected one."""
    sc.set_masks(np.ones((FH, FW), np.uint8), first_view=0, n_views=V)
    sc.run(0, V)
    sc.synchronize()


def _flat_frames(W, H, N):
    """Frames on which stages 3..5 accept EVERY selected pixel (fringes 10, 130, 250 -> a finite phase; every Gray plane above its
    inverse -> code 1010..b -> a projector coordinate inside the projector), so that the merged valid map of a timed context IS the valid
    map after stage 3's boundary removal."""
    f = [np.full((H, W), v, np.uint8) for v in (10, 130, 250)]
    g = [np.full((H, W), 200, np.uint8) for _ in range(N)] + [np.full((H, W), 20, np.uint8) for _ in range(N)]
    return f + g


@pytest.mark.parametrize("FW,FH", [(152, 90), (204, 61), (64, 7), (1920, 24), (20, 300), (8, 5)])
def test_fused_mask_route_against_literal_scan(FW, FH):
    """Random selections through sl3d_set_mask (pageable, pinned) and sl3d_set_masks (several distinct masks in one call) on a timed
    context; one-view and multi-view MASKIN launches, dense and clouds; full frames and windows touching every border.  The valid map
    equals the oracle's literal scan; the 0/1 plane and the results equal the eager context's byte for byte; a second, ordinary launch
    over the same view (it reads the band plane the MASKIN launch left) gives the same again."""
    S, syn = pkg("scanner"), pkg("synth")
    N, fw, V = 6, 8, 4
    PW, PH = fw << N, fw << N
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
        kw = dict(full_size=(FW, FH), origin=(x0, y0), max_views=V)
        # (a window whose pitch padding alone leaves fewer than 65 % of its quads selectable is "sparsely selected" whatever the mask:
        # such views take the gated MASKIN form, whose plane requests wait for the valid bits)
        fused_route = -(-w // 4) >= 0.65 * (((w + 15) & ~15) // 4)
        with S.Scanner(w, h, PW, PH, N, N, fw, fw, **kw) as sc, S.Scanner(w, h, PW, PH, N, N, fw, fw, eager_mask=True, **kw) as eager:
            for c in (sc, eager):
                c.set_calibration(*cal)
                for v in range(V):
                    c.set_frames(0, _flat_frames(w, h, N), view=v)
                    c.set_frames(1, _flat_frames(w, h, N), view=v)
            pm = sc.pinned(masks.shape, np.uint8)
            pm[:] = masks
            for how in ("single", "single_pinned", "batch", "batch_clouds"):
                # what is known about the views' last selections decides the MASKIN form (a view whose last selection was sparse takes
                # the gated one, test_fused_mask_route_counts_selected_quads): every pass starts from densely selected views
                for c in (sc, eager):
                    _dense_history(c, FW, FH, V)
                if how.startswith("single"):   # the reference's loop: a new selection, then the scan (m_tech_project_console.cpp:366-395)
                    for v in range(V):
                        for c in (sc, eager):
                            c.set_mask((pm if how == "single_pinned" else masks)[v], view=v)
                            c.run(v, 1)
                        assert sc.last_fused_kernel_name().endswith(MASKIN_DENSE if fused_route else MASKIN_GATED), sc.last_fused_kernel_name()
                        assert ", 4, " not in eager.last_fused_kernel_name()
                else:
                    for c in (sc, eager):
                        c.set_masks(masks)
                if how.startswith("single"):
                    pass
                elif how == "batch":
                    sc.run(0, V)
                    assert sc.last_fused_kernel_name().endswith(MASKIN_DENSE if fused_route else MASKIN_GATED), sc.last_fused_kernel_name()
                    eager.run(0, V)
                else:
                    clouds = sc.fused_clouds(0, V)
                    assert sc.last_fused_kernel_name().endswith(MASKIN_CLOUDS if fused_route else MASKIN_GATED_CLOUDS), sc.last_fused_kernel_name()
                    eclouds = eager.fused_clouds(0, V)
                for v in range(V):
                    want = refs[v][y0:y0 + h, x0:x0 + w]
                    tag = (how, v, (x0, y0, w, h))
                    xyz, valid = sc.points(v)
                    exyz, evalid = eager.points(v)
                    assert np.array_equal(valid, want), tag
                    assert np.array_equal(evalid, want), tag
                    if how != "batch_clouds":
                        assert np.array_equal(xyz, exyz, equal_nan=True), tag
                    else:
                        assert np.array_equal(clouds[v], eclouds[v]) and len(clouds[v]) == int(want.sum()), tag
                    assert np.array_equal(_mask_plane(sc, v), _mask_plane(eager, v)), tag
                # an ordinary launch over the same views reads the band plane the MASKIN launch wrote
                sc.run(0, V)
                assert ", 4, " not in sc.last_fused_kernel_name()
                for v in range(V):
                    assert np.array_equal(sc.points(v)[1], refs[v][y0:y0 + h, x0:x0 + w]), (how, v, "second launch")


def test_fused_mask_route_device_resident_and_colrow():
    """A caller's device-resident mask read in place by the fused kernel (4-byte aligned rows), the same at an odd address (staged by
    a device copy, still deferred), and the reference's own int [col][row] selected_region (sl3d_set_mask_colrow)."""
    torch = pytest.importorskip("torch")
    S, syn = pkg("scanner"), pkg("synth")
    FW, FH, N, fw, V = 200, 120, 6, 8, 3
    PW, PH = fw << N, fw << N
    rng = np.random.default_rng(8)
    cal = syn.cal_tuple(syn.synth_rig(FW, FH, PW, PH))
    masks = np.stack([_mask(rng, FW, FH, t) for t in range(V)])
    refs = [_oracle_valid(m) for m in masks]
    d_al = torch.from_numpy(masks).cuda()
    big = torch.zeros(V * (FH * 203) + 64, dtype=torch.uint8, device="cuda")
    odd = big[1:1 + V * FH * 203].view(V, FH, 203)
    odd[:, :, :FW] = d_al
    torch.cuda.synchronize()
    for (x0, y0, w, h) in [(0, 0, FW, FH), (4, 3, 100, 50), (6, 0, 64, 120), (8, 0, 64, 120), (100, 70, 100, 50), (184, 0, 16, 120)]:
        with S.Scanner(w, h, PW, PH, N, N, fw, fw, full_size=(FW, FH), origin=(x0, y0), max_views=V) as sc:
            sc.set_calibration(*cal)
            for v in range(V):
                sc.set_frames(0, _flat_frames(w, h, N), view=v)
                sc.set_frames(1, _flat_frames(w, h, N), view=v)
            for name, t, stride in (("aligned", d_al, FW), ("odd", odd, 203)):
                _dense_history(sc, FW, FH, V)
                sc.set_masks_device(t.data_ptr(), stride, FH * stride, 0, V)
                sc.run(0, V)
                assert sc.last_fused_kernel_name().endswith(MASKIN_DENSE), (name, sc.last_fused_kernel_name())
                for v in range(V):
                    assert np.array_equal(sc.points(v)[1], refs[v][y0:y0 + h, x0:x0 + w]), (name, v, x0, y0)
                # one view at a time, each with the mask of ANOTHER view than last time (what bench.py's per_scan_device does); a view
                # whose last selection was sparse would take the gated MASKIN form: every view starts from a dense one
                _dense_history(sc, FW, FH, V)
                for v in range(V):
                    k = (v + 1) % V
                    sc.set_masks_device(t[k].data_ptr(), stride, 0, v, 1)
                    sc.run_clouds(v, 1)
                    assert sc.last_fused_kernel_name().endswith(MASKIN_CLOUDS)
                    assert np.array_equal(sc.points(v)[1], refs[k][y0:y0 + h, x0:x0 + w]), (name, v)
            sel = np.ascontiguousarray(masks[1].T.astype(np.int32))      # int selected_region[col][row]
            _dense_history(sc, FW, FH, V)
            sc.set_mask_colrow(sel, view=2)
            sc.run(2, 1)
            assert sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
            assert np.array_equal(sc.points(2)[1], refs[1][y0:y0 + h, x0:x0 + w])


def test_deferred_masks_are_prepared_for_every_other_consumer():
    """What is deferred is prepared by k_mask_prepare as soon as something else needs it: a launch of more than 4 views, a launch that
    mixes deferred and prepared views, sl3d_copy_view, a new mask in the same staging slot, sl3d_get_device_buffers, and (a caller's
    device memory only) the next synchronising call."""
    torch = pytest.importorskip("torch")
    S, syn = pkg("scanner"), pkg("synth")
    W, H, N, fw, V = 320, 96, 7, 4, 6
    PW, PH = fw << N, fw << N
    rng = np.random.default_rng(77)
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    # (densely selected throughout: a view whose last selection was sparse would take the gated MASKIN form)
    masks = np.stack([(rng.random((H, W)) < 0.97).astype(np.uint8) * (1 if t % 2 == 0 else rng.integers(1, 2, (H, W), dtype=np.uint8)) for t in range(V)])
    masks[1][masks[1] == 0] = 7
    refs = [_oracle_valid(m) for m in masks]
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V) as sc:
        sc.set_calibration(*cal)
        for v in range(V):
            sc.set_frames(0, _flat_frames(W, H, N), view=v)
            sc.set_frames(1, _flat_frames(W, H, N), view=v)
        # view by view (each call re-uses staging slot 0: the previous view's deferred mask is prepared before it is overwritten),
        # then ONE launch of 6 views: nothing is left to a MASKIN launch
        for v in range(V):
            sc.set_mask(masks[v], view=v)
        sc.run(0, V)
        assert not sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
        for v in range(V):
            assert np.array_equal(sc.points(v)[1], refs[v]), v
        # a small launch that mixes a deferred view (1) with prepared ones (0, 2)
        sc.set_mask(masks[4], view=1)
        sc.run(0, 3)
        assert not sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
        assert np.array_equal(sc.points(1)[1], refs[4]) and np.array_equal(sc.points(0)[1], refs[0])
        # copy_view of a deferred source; the duplicate keeps the mask when the source gets another one
        sc.set_mask(masks[5], view=0)
        sc.copy_view(0, 3)
        sc.set_mask(masks[2], view=0)
        sc.run(3, 1)
        assert np.array_equal(sc.points(3)[1], refs[5])
        sc.run(0, 1)
        assert sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
        assert np.array_equal(sc.points(0)[1], refs[2])
        # the 0/1 plane is there when the caller asks where it lies
        sc.set_mask(masks[3], view=2)
        plane = _mask_plane(sc, 2)
        assert np.array_equal(plane[2:2 + H, 16:16 + W], (masks[3] == 1).astype(np.uint8))
        sc.run(2, 1)
        assert not sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
        assert np.array_equal(sc.points(2)[1], refs[3])
        # a caller's device mask may change after the next synchronising call: it has been consumed by then
        t = torch.from_numpy(masks[1]).cuda()
        torch.cuda.synchronize()
        sc.set_masks_device(t.data_ptr(), W, 0, 4, 1)
        sc.synchronize()
        t.zero_()
        torch.cuda.synchronize()
        sc.run(4, 1)
        assert not sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
        assert np.array_equal(sc.points(4)[1], refs[1])


def test_fused_mask_route_counts_selected_quads():
    """A MASKIN launch leaves the view's count of selected quads (one word per wave in mapped host memory): a view whose LAST selection
    was sparse takes the gated form (its plane requests wait for the valid bits the launch evaluates), a dense one the form that requests
    its planes at once.  Either way: one launch, and the eager context's results bit for bit."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 960, 540, 1024, 768, 9, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    dense = syn.default_mask(W, H)
    sparse = np.zeros((H, W), np.uint8)
    sparse[150:390, 300:700] = 1
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=2) as sc, S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=2, eager_mask=True) as eager:
        for c in (sc, eager):
            c.set_calibration(*cal)
            for v in range(2):
                c.synth_view(v, plane=(1.5 * v, 0.05, 0.04), view_id=v, noise=2)
        seq = [sparse, sparse, dense, dense, sparse, sparse]
        want = [MASKIN_DENSE,    # nothing known about the view: counts as dense
                MASKIN_GATED,    # its last selection was sparse: the form whose plane requests wait for the valid bits
                MASKIN_GATED,    # ... still what is known when the dense one arrives
                MASKIN_DENSE, MASKIN_DENSE, MASKIN_GATED]
        for i, (m, k) in enumerate(zip(seq, want)):
            sc.set_mask(m, view=0)
            eager.set_mask(m, view=0)
            sc.run(0, 1)
            eager.run(0, 1)
            assert sc.last_fused_kernel_name().endswith(k), (i, sc.last_fused_kernel_name())
            a, b = sc.points(0), eager.points(0)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0], equal_nan=True), i


def test_route_of_back_to_back_scans_follows_the_last_count_that_arrived():
    """The reference's loop sets a new selection and launches at once, scan after scan (m_tech_project_console.cpp:366-395): the count
    of the selection a scan replaces is usually still on its way when the next launch is routed.  The route then follows the last count
    that DID arrive (sparse_views): lassos of the reference's size keep the gated kernels -- one MASKIN launch on a deferring context,
    k_mask_prepare + the gated fused kernel on an eager one -- without a single synchronisation between scans, and whichever kernel
    runs, the results are the eager context's bit for bit."""
    import torch
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 960, 540, 1024, 768, 9, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    lassos = np.zeros((6, H, W), np.uint8)
    for k in range(6):
        lassos[k, 150 + 3 * k:390 + 3 * k, 300 + 5 * k:700 + 5 * k] = 1
    d_lassos = torch.from_numpy(lassos).cuda()
    d_dense = torch.from_numpy(syn.default_mask(W, H)).cuda()
    torch.cuda.synchronize()
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=2) as sc, S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=2, eager_mask=True) as eager:
        for c in (sc, eager):
            c.set_calibration(*cal)
            for v in range(2):
                c.synth_view(v, plane=(1.5 * v, 0.05, 0.04), view_id=v, noise=2)
                c.set_masks_device(d_lassos.data_ptr(), W, 0, v, 1)
            c.run(0, 2)
            c.synchronize()                       # the views' counts are known now: sparse
        for c, gated in ((sc, MASKIN_GATED), (eager, ", 1, 0, true, false>")):
            for i in range(24):                   # nothing is waited for in this loop
                c.set_masks_device(d_lassos.data_ptr() + (i % 6) * W * H, W, 0, i % 2, 1)
                c.run(i % 2, 1)
                assert c.last_fused_kernel_name().endswith(gated), (i, c.last_fused_kernel_name())
        for v in range(2):
            a, b = sc.points(v), eager.points(v)
            assert b[1].sum() > 0 and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0], equal_nan=True), v
        # a dense selection behind sparse ones: routed as sparse once (all that is known), as dense from then on
        for c, first, then in ((sc, MASKIN_GATED, MASKIN_DENSE), (eager, ", 1, 0, true, false>", ", 1, 0, false, true>")):
            for want in (first, then):
                c.set_masks_device(d_dense.data_ptr(), W, 0, 0, 1)
                c.run(0, 1)
                name = c.last_fused_kernel_name()
                c.synchronize()
                if c is sc or want == then:       # (eager, first launch: its own count may or may not have landed before the launch was routed)
                    assert name.endswith(want), (want, name)
        a, b = sc.points(0), eager.points(0)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0], equal_nan=True)


@pytest.mark.parametrize("rig,rig_class", [("reference", 1), ("tangential", 2), ("radial", 3)])
@pytest.mark.parametrize("Nv,Nh", [(7, 7), (7, 6)])
def test_gated_form_every_rig_class_exact_and_padded(rig, rig_class, Nv, Nh):
    """The gated MASKIN kernels (views whose last selection was sparse) of every rig class that has them, with equal Gray depths (the
    exact instantiation) and unequal ones (the padded one -- for rig classes 2 and 3 the form that takes its pad count through
    v_readfirstlane, sl3d_fused.h: issue_gray), dense planes and clouds, one view and three: the kernel that ran is the gated one, the
    valid map lies inside the oracle's literal scan of the new selection, and valid map, results, clouds and the 0/1 plane equal the eager context's
    (k_mask_prepare + the ordinary gated kernel) bit for bit."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, fw, V = 328, 150, 4, 3
    PW, PH = fw << Nv, fw << Nh
    cal_d = syn.synth_rig(W, H, PW, PH)
    if rig == "tangential":
        cal_d["dp"] = np.array([-0.05, 0.02, 0.001, -0.0015, 0.0])
    elif rig == "radial":
        cal_d["dp"] = np.array([-0.05, 0.02, 0.0, 0.0, 0.0])
    cal = syn.cal_tuple(cal_d)
    rng = np.random.default_rng(Nv * 10 + Nh)
    lassos = np.zeros((V + 1, H, W), np.uint8)
    for k in range(V + 1):
        lassos[k, 40 + 2 * k:100 + 2 * k, 90 + 3 * k:220 + 3 * k] = 1
        lassos[k][rng.random((H, W)) < 0.01] = 0
    refs = [_oracle_valid(m) for m in lassos]
    tail = f", {rig_class}, %d, true, false>"
    with S.Scanner(W, H, PW, PH, Nv, Nh, fw, fw, max_views=V) as sc, S.Scanner(W, H, PW, PH, Nv, Nh, fw, fw, max_views=V, eager_mask=True) as eager:
        for c in (sc, eager):
            c.set_calibration(*cal)
            for v in range(V):
                c.synth_view(v, plane=(1.5 * v, 0.05, 0.04), view_id=v, noise=2)
        for n, clouds in ((1, False), (V, False), (1, True), (V, True)):
            for c in (sc, eager):
                c.set_masks(lassos[V], 0, V)      # the history: every view sparsely selected, and known to be
                c.run(0, V)
                c.synchronize()
                c.set_masks(lassos[:n])
                got = c.fused_clouds(0, n) if clouds else c.run(0, n)
                if c is sc:
                    mine = got
            name = sc.last_fused_kernel_name()
            assert name.endswith(tail % (6 if clouds else 4)), name
            assert ("true" if Nv == Nh else "false") + f", {rig_class}, " in name, name
            assert eager.last_fused_kernel_name().endswith(tail % (2 if clouds else 0)), eager.last_fused_kernel_name()
            for v in range(n):
                a, b = sc.points(v), eager.points(v)
                # (stage 3's valid map is the oracle's literal scan of the selection; stage 5 then drops what the projector does not light)
                assert np.array_equal(a[1], b[1]) and not (a[1] & ~refs[v]).any() and a[1].sum() > 0.5 * refs[v].sum(), (n, clouds, v)
                if clouds:
                    assert np.array_equal(mine[v], got[v]) and len(mine[v]) == int(a[1].sum()), (n, v)
                else:
                    assert np.array_equal(a[0], b[0], equal_nan=True), (n, v)
                assert np.array_equal(_mask_plane(sc, v), _mask_plane(eager, v)), (n, clouds, v)


def test_fused_mask_route_several_views_per_lane():
    """4.2 Mpx: a launch of 3 views runs 2 views per lane (tools: views_per_lane) -- every view of an item evaluates its own selection."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, N, fw, V = 2560, 1640, 6, 8, 3
    PW, PH = fw << N, fw << N
    rng = np.random.default_rng(3)
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    masks = np.stack([_mask(rng, W, H, t) for t in (1, 2, 0)])
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V) as sc, S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V, eager_mask=True) as eager:
        for c in (sc, eager):
            c.set_calibration(*cal)
            for v in range(V):
                c.synth_view(v, plane=(1.5 * v, 0.05, 0.04), view_id=v, noise=2)
            c.set_masks(masks)
            c.run(0, V)
        assert sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
        for v in range(V):
            a, b = sc.points(v), eager.points(v)
            assert np.array_equal(a[1], b[1]), v
            assert np.array_equal(a[0], b[0], equal_nan=True), v
            assert np.array_equal(_mask_plane(sc, v), _mask_plane(eager, v)), v


@pytest.mark.parametrize("rig", ["reference", "radial"])
def test_small_and_large_launches_on_ill_conditioned_pixels(rig):
    """RANDOM frames make random correspondences, i.e. triangulations whose rays are nearly parallel -- where a camera table rounded to
    f32 showed up as 2.7e-6 (round 2; round 6 measured a 4-byte form again and dropped it: profiles/r06_camera_table_4_bytes_ab.txt).
    Against the oracle, small launches (the ordinary one and the MASKIN one) and large ones stay at the f32 rounding of the result: 2e-7
    of the point norm on every pixel."""
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    W, H, N, fw, V = 640, 360, 7, 4, 6
    PW, PH = fw << N, fw << N
    rng = np.random.default_rng(11)
    cal_d = syn.synth_rig(W, H, PW, PH)
    if rig == "radial":
        cal_d["dp"] = np.array([-0.05, 0.02, 0.0, 0.0, 0.0])
    cal = syn.cal_tuple(cal_d)
    planes = ([rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * N)], [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * N)])
    mask = syn.default_mask(W, H)
    o = Oracle(W, H, PW, PH, N, N, fw, fw)
    o.set_mask(mask)
    o.set_calibration(*cal)
    oxyz, ovalid, _ = o.run_scan_rowmajor(*planes)
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=V) as sc:
        sc.set_calibration(*cal)
        sc.set_masks(mask)
        for v in range(V):
            sc.set_frames(0, planes[0], view=v)
            sc.set_frames(1, planes[1], view=v)
        sc.run(0, V)
        big = sc.points(0)
        sc.run(0, 1)
        small = sc.points(0)
        sc.set_mask(mask, view=0)
        sc.run(0, 1)
        assert sc.last_fused_kernel_name().endswith(MASKIN_DENSE)
        fused = sc.points(0)
    v = ovalid == 1
    assert v.sum() > 0.5 * W * H
    ref = oxyz[v].astype(np.float64)
    nrm = np.linalg.norm(ref, axis=-1)
    for name, got in (("large launch", big), ("small launch", small), ("MASKIN launch", fused)):
        assert np.array_equal(got[1], ovalid), name
        e = np.linalg.norm(got[0][v] - ref, axis=-1) / nrm
        print(f"max relative point error vs the oracle, {name}: {e.max():.3e}")
        assert e.max() <= 2e-7, name
        assert_points_close(got[0], oxyz, ovalid == 1)
