"""GPU tests (-m gpu) added in round 4: the multi-GPU code of sl3d_group.cpp with every stripe on its OWN communication side
(SL3D_FLAG_GROUP_DISTINCT_SIDES) -- peer-copy transport in process, the N-rank RCCL exchange through a test double of librccl in
a fresh process -- and the reference's own distorted-projector calibrations through the table rig."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_points_close, pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="session")
def fake_rccl(tmp_path_factory):
    """tests/native/fake_rccl.cpp built next to the tests' temporary files (hipcc is part of the image, here and on the GPU box)."""
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "-O2", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"), "-o", out])
    return out


def _random_mask(rng, W, H, p=0.1):
    m = np.ones((H, W), np.uint8)
    m[0, :] = m[-1, :] = 0
    m[:, 0] = m[:, -1] = 0
    m[rng.random((H, W)) < p] = 0
    return m


# ---- every stripe its own side, (peer) copy transport, in process ---------------------------------------------------------------
@pytest.mark.parametrize("n_stripes", [4, 7, 8])
def test_group_distinct_sides_copy_transport(n_stripes):
    """All stripes on device 0, but each with its own GpuSide (communication stream + event): the `S.gpu != 0` branches of
    sl3d_group.cpp -- a stripe waits for the ROOT's communication stream as well as its own side's before it overwrites results a
    gather may still read (group_launch), the root's stream waits for the stripe's kernel, hipMemcpyPeerAsync (device 0 -> device 0)
    moves the slab -- run, and the result is the single-context result bit for bit: pipelined run(v + 1); gather(v), the same views
    re-run while their gather is in flight, compacted clouds, and gather -> process_views -> get_points."""
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw, NV = 320, 203, 512, 384, 7, 4, 4
    rng = np.random.default_rng(n_stripes)
    caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, view=v, noise=2, plane=(3.0 * v, 0.05, 0.02 * v)) for v in range(NV)]
    cal = syn.cal_tuple(caps[0]["cal"])
    masks = [caps[0]["mask"]] + [_random_mask(rng, W, H) for _ in range(NV - 1)]
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v, c in enumerate(caps):
            sc.set_mask(masks[v], view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, NV)
        ref = [sc.points(v) for v in range(NV)]
    flags = S.SL3D_FLAG_GROUP_DISTINCT_SIDES | S.SL3D_FLAG_GROUP_NO_RCCL
    with S.Group(W, H, PW, PH, N, N, fw, fw, devices=[0] * n_stripes, max_views=NV, flags=flags) as g:
        assert g.transport == "copy"
        g.set_calibration(*cal)
        for v, c in enumerate(caps):
            g.set_mask(masks[v], view=v)
            g.set_frames(0, c["planes_v"], view=v)
            g.set_frames(1, c["planes_h"], view=v)
        for rep in range(3):
            for v in range(NV):
                g.run(v, 1)
                g.gather(v, 1)
            for v in range(NV):
                xyz, val = g.points(v)
                assert np.array_equal(val, ref[v][1]), (rep, v)
                assert np.array_equal(xyz, ref[v][0], equal_nan=True), (rep, v)
        g.run_clouds(0, NV)
        counts = g.gather_clouds(0, NV)
        for v in range(NV):
            cl = g.cloud(v)
            assert counts[v] == len(cl) == int((ref[v][1] == 1).sum())
            assert np.array_equal(cl, ref[v][0][ref[v][1] == 1]), v
        # a gather in flight, then the host pipelines over the same result slots: the assembled planes stay the gathered ones, and
        # the pipelines deliver the views they were given
        g.run(0, NV)
        g.gather(0, NV)
        order = list(reversed(range(NV)))
        frames = np.stack([np.stack(caps[v]["planes_v"] + caps[v]["planes_h"]) for v in order])
        pxyz, pval = g.process_views(frames)
        for v in range(NV):
            xyz, val = g.points(v)
            assert np.array_equal(val, ref[v][1]) and np.array_equal(xyz, ref[v][0], equal_nan=True), v
        # (slot k processed view order[k] under slot k's mask)
        with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sc:
            sc.set_calibration(*cal)
            for k, v in enumerate(order):
                sc.set_mask(masks[k], view=k)
                sc.set_frames(0, caps[v]["planes_v"], view=k)
                sc.set_frames(1, caps[v]["planes_h"], view=k)
            sc.run(0, NV)
            for k in range(NV):
                xyz, val = sc.points(k)
                assert np.array_equal(pval[k], val) and np.array_equal(pxyz[k], xyz, equal_nan=True), k
        g.synchronize()


# ---- the N-rank RCCL exchange through the test double, in a fresh process ----------------------------------------------------------
@pytest.mark.parametrize("n_stripes", [4, 7, 8])
def test_group_n_rank_exchange_through_fake_rccl(fake_rccl, n_stripes):
    """tests/group_fake_rccl_driver.py in a fresh process with SL3D_RCCL_LIB = the double: every stripe is a rank of an N-rank
    communicator (ncclCommInitAll over N entries), sends on its own side's stream, the root receives -- pairing, the 256-message
    group split and the variable-size cloud gather execute with N > 1 ranks and reproduce one context bit for bit."""
    env = dict(os.environ)
    env["SL3D_RCCL_LIB"] = fake_rccl
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "group_fake_rccl_driver.py"), fake_rccl, str(n_stripes)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert st["many_messages"]["ranks"] == 2 * n_stripes          # two communicators of n ranks each were created
    assert st["many_messages"]["pairs"] > 0 and st["many_messages"]["max_pairs_in_group"] <= 256


def test_fake_rccl_fails_loudly(fake_rccl):
    """The double is only worth something if it refuses what real RCCL would hang on: an unmatched send, an unmatched recv, a size
    mismatch and a point-to-point call outside a group all fail, with a message."""
    L = ctypes.CDLL(fake_rccl)
    vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.ncclCommInitAll.argtypes = [ctypes.POINTER(vp), i, ctypes.POINTER(i)]
    L.ncclSend.argtypes = [vp, sz, i, i, vp, vp]
    L.ncclRecv.argtypes = [vp, sz, i, i, vp, vp]
    L.ncclCommDestroy.argtypes = [vp]
    L.ncclGetErrorString.restype = ctypes.c_char_p
    comms = (vp * 3)()
    assert L.ncclCommInitAll(comms, 3, (i * 3)(0, 0, 0)) == 0
    buf = vp(0x1000)   # never dereferenced: every case below fails before any copy
    ncclFloat = 7
    assert L.ncclSend(buf, 4, ncclFloat, 1, comms[0], None) != 0            # outside a group
    assert L.ncclGroupStart() == 0
    assert L.ncclSend(buf, 4, ncclFloat, 1, comms[0], None) == 0
    rc = L.ncclGroupEnd()
    assert rc != 0 and b"no matching recv" in L.ncclGetErrorString(rc)
    assert L.ncclGroupStart() == 0
    assert L.ncclRecv(buf, 4, ncclFloat, 2, comms[0], None) == 0
    rc = L.ncclGroupEnd()
    assert rc != 0 and b"no matching send" in L.ncclGetErrorString(rc)
    assert L.ncclGroupStart() == 0
    assert L.ncclSend(buf, 4, ncclFloat, 0, comms[1], None) == 0
    assert L.ncclRecv(buf, 8, ncclFloat, 1, comms[0], None) == 0
    rc = L.ncclGroupEnd()
    assert rc != 0 and b"expected" in L.ncclGetErrorString(rc)
    assert L.ncclSend(buf, 4, ncclFloat, 5, comms[0], None) != 0            # peer out of range
    for c in comms:
        L.ncclCommDestroy(c)


# ---- the reference's own distorted projectors --------------------------------------------------------------------------------
def _alt_projector_cal(name, W, H, PW, PH):
    """The synthetic rig (the reference's camera + extrinsics rescaled to the camera size) with the projector intrinsics and
    distortion of one of the reference's own OpenCV projector calibrations (tests/golden/calibration.json: _alt_projectors)."""
    import json
    from conftest import GOLDEN
    syn = pkg("synth")
    alt = json.load(open(os.path.join(GOLDEN, "calibration.json")))["_alt_projectors"][name]
    cal = syn.synth_rig(W, H, PW, PH)
    cal["Kp"] = np.array(alt["Kp"], dtype=np.float64)
    cal["dp"] = np.array(alt["dp"], dtype=np.float64)
    return cal


@pytest.mark.parametrize("name", ["Sharp", "Viewsonic"])
def test_reference_distorted_projector_calibrations(name):
    """Projector_calibration/Matrices/OPencv calib/{Sharp,Viewsonic}: k1 = -1.01 / -1.16, k2 = 8.28 / 2.60 -- an order of magnitude
    stronger than the synthetic distortions of the other tests, and radial only.  1080p camera, 1280x720 projector (the files'
    own size), two views (full mask; holes): the timed mode (camera-frame solve + the projector's radial table in LDS, the rig
    class 3 kernels; 7/triangulation.cpp:352-378) and the parity mode (the 5 iterations evaluated per pixel) against the oracle on the frames the
    context processed -- valid map and correspondences bit exact, points within 1e-5; the observed error of the table path is
    printed (pytest -s) and bounded well below the bar."""
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 1920, 1080, 1280, 720, 10, 2
    cal = syn.cal_tuple(_alt_projector_cal(name, W, H, PW, PH))
    rng = np.random.default_rng(7)
    masks = [syn.default_mask(W, H), _random_mask(rng, W, H, p=0.05)]
    worst = 0.0
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=2) as sc:
        sc.set_calibration(*cal)
        assert ", 3, 0, " in sc.fused_kernel_name(2)            # radial only: the LDS-table instantiation (rig class 3)
        for v, m in enumerate(masks):
            sc.set_mask(m, view=v)
            sc.synth_view(v, plane=(2.0 * v, 0.05, 0.04), view_id=v, noise=2)
        sc.run(0, 2)
        got = [sc.points(v) for v in range(2)]
        sc.run(1, 1)                                          # the small-launch instantiation too
        one = sc.points(1)
        assert np.array_equal(one[1], got[1][1]) and np.array_equal(one[0], got[1][0], equal_nan=True)
        frames = [(sc.frames(0, v), sc.frames(1, v)) for v in range(2)]
        clouds = sc.fused_clouds(0, 2)
    for v in range(2):
        o = Oracle(W, H, PW, PH, N, N, fw, fw)
        o.set_mask(masks[v])
        o.set_calibration(*cal)
        oxyz, ovalid, _ = o.run_scan_rowmajor(*frames[v])
        assert int(ovalid.sum()) > 100_000, "the scene must be seen by both devices"
        assert np.array_equal(got[v][1], ovalid), v
        worst = max(worst, assert_points_close(got[v][0], oxyz, ovalid == 1))
        assert np.array_equal(clouds[v], got[v][0][ovalid == 1])
    print(f"{name}: max relative point error of the timed (table) path against the oracle: {worst:.3e}")
    assert worst < 2e-6
    # parity mode (every stage-boundary plane; the 5-iteration projector undistortion evaluated per pixel) on a smaller frame of the
    # same rig: valid maps and correspondences bit exact, intersection_points within 1e-5
    Wc, Hc = 640, 360
    cal_d = _alt_projector_cal(name, Wc, Hc, PW, PH)
    cap = syn.make_capture(Wc, Hc, PW, PH, N, N, fw, fw, cal=cal_d, noise=2)
    calc = syn.cal_tuple(cal_d)
    with S.Scanner(Wc, Hc, PW, PH, N, N, fw, fw, keep_stages=True) as sc:
        sc.set_calibration(*calc)
        sc.set_mask(cap["mask"])
        sc.set_frames(0, cap["planes_v"])
        sc.set_frames(1, cap["planes_h"])
        sc.run()
        xyz, val = sc.points()
        cpm, ip = sc.c_p_map(), sc.intersection_points()
    o = Oracle(Wc, Hc, PW, PH, N, N, fw, fw)
    o.set_mask(cap["mask"])
    o.set_calibration(*calc)
    o.run_scan(cap["planes_v"], cap["planes_h"])
    ov = o.valid_map(2) == 1
    assert int(ov.sum()) > 10_000
    assert np.array_equal(val == 1, ov)
    assert np.array_equal(cpm[ov], o.c_p_map()[ov])
    assert_points_close(ip, o.intersection_points(), ov)
    assert_points_close(xyz, o.intersection_points(), ov)


# ---- a context larger than 4 GiB: results do not depend on where in the address space a view lies -------------------------------
def test_views_across_4gib_boundaries():
    """The kernels address a view's planes and results as (wave-uniform 64-bit base) + (32-bit lane offset).  400 views of 1 Mpx keep
    5 GB of points and 20 GB of frames in single allocations, so the bases of the views picked here differ in every one of the address
    bits 31..34 (a base whose low half has bit 31 set once came back sign-extended from a scalar read: intermittent faults, caught
    by the A/B runs, not by a test -- hence this one).  Dense results and ordered clouds of views spread over the allocation, each
    launched alone (small-launch kernels) and in a batch of 6 (the large-launch kernels), against the oracle."""
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 1024, 1024, 1024, 768, 9, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    mask = syn.default_mask(W, H)
    picks = [0, 97, 171, 172, 255, 342, 394]
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=400) as sc:
        sc.set_calibration(*cal)
        for v in picks + list(range(394, 400)):
            sc.set_mask(mask, view=v)
            sc.synth_view(v, plane=(0.01 * v, 0.04, 0.03), view_id=v, noise=2)
        got = {}
        for v in picks:
            sc.run(v, 1)
            got[v] = sc.points(v)
            cl = sc.fused_clouds(v, 1)[0]
            assert np.array_equal(cl, got[v][0][got[v][1] == 1]), v
        sc.run(394, 6)
        batch = [sc.points(v) for v in range(394, 400)]
        clouds = sc.fused_clouds(394, 6)
        assert np.array_equal(batch[0][1], got[394][1]) and np.array_equal(batch[0][0], got[394][0], equal_nan=True)
        for i in range(6):
            assert np.array_equal(clouds[i], batch[i][0][batch[i][1] == 1]), i
        frames = {v: (sc.frames(0, v), sc.frames(1, v)) for v in picks + [399]}
        got[399] = batch[5]
    for v in picks + [399]:
        o = Oracle(W, H, PW, PH, N, N, fw, fw)
        o.set_mask(mask)
        o.set_calibration(*cal)
        oxyz, ovalid, _ = o.run_scan_rowmajor(*frames[v])
        assert int(ovalid.sum()) > 100_000
        assert np.array_equal(got[v][1], ovalid), v
        assert_points_close(got[v][0], oxyz, ovalid == 1)


# ---- small launches of 1..4 views, every pipelined rig class, a window whose last tile is partial, masks with whole waves off ------
@pytest.mark.parametrize("n_views", [1, 2, 3, 4])
@pytest.mark.parametrize("rig", ["reference", "radial", "distorted"])
def test_small_launches_partial_tile_and_masked_waves(n_views, rig):
    """The small-launch instantiations (at most 4 views per launch: the reference's one scan per call) of the three pipelined rig
    classes.  1912 x 1083: 2,023 tiles, the last one owns only 183 of its 256 quads (lanes past the last row leave at once; the wave
    that straddles the end stores 16-byte pieces); masks with per-pixel holes AND whole rows / a block without a valid pixel (the
    plane requests of every view but a lane's first are masked: waves that skip them).  Dense results and ordered clouds against the oracle, view by view.
    (Written for a schedule with two tiles per block that was measured and rejected -- profiles/r04_two_tiles_per_block_ab.txt;
    the case stays.)"""
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    W, H, PW, PH, N, fw = 1912, 1083, 1920, 1080, 10, 2
    cal_d = syn.synth_rig(W, H, PW, PH)
    if rig == "radial":
        cal_d["dp"] = np.array([-0.05, 0.02, 0.0, 0.0, 0.0])
    if rig == "distorted":
        cal_d["dp"] = np.array([-0.05, 0.02, 0.001, -0.0005, 0.0])
        cal_d["dc"] = np.array(cal_d["dc"], dtype=np.float64) + np.array([0.0, 0.0, 0.0008, -0.0006, 0.0])
    cal = syn.cal_tuple(cal_d)
    rng = np.random.default_rng(100 + n_views)
    masks = [_random_mask(rng, W, H, p=0.3 if v % 2 else 0.02) for v in range(n_views)]
    for v, m in enumerate(masks):          # whole waves without a valid pixel: rows of the first tiles, a block in the middle, the last rows
        m[: 3 + 5 * v, :] = 0
        m[300 + 40 * v:520, 250:1500] = 0
        if v % 2 == 0:
            m[H - 4:, :] = 0
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=n_views) as sc:
        sc.set_calibration(*cal)
        want = {"reference": ", 1, 0, false, true>", "radial": ", 3, 0, false, true>", "distorted": ", 2, 0, false, true>"}[rig]   # small launch
        assert sc.fused_kernel_name(n_views).endswith(want), sc.fused_kernel_name(n_views)
        for v in range(n_views):
            sc.set_mask(masks[v], view=v)
            sc.synth_view(v, plane=(1.5 * v, 0.05, 0.04 - 0.01 * v), view_id=v, noise=2)
        sc.run(0, n_views)
        got = [sc.points(v) for v in range(n_views)]
        clouds = sc.fused_clouds(0, n_views)
        frames = [(sc.frames(0, v), sc.frames(1, v)) for v in range(n_views)]
    for v in range(n_views):
        o = Oracle(W, H, PW, PH, N, N, fw, fw)
        o.set_mask(masks[v])
        o.set_calibration(*cal)
        oxyz, ovalid, _ = o.run_scan_rowmajor(*frames[v])
        assert int(ovalid.sum()) > 100_000
        assert np.array_equal(got[v][1], ovalid), v
        assert_points_close(got[v][0], oxyz, ovalid == 1)
        assert np.isnan(got[v][0][ovalid == 0]).all()
        assert np.array_equal(clouds[v], got[v][0][ovalid == 1]), v


# ---- unequal Gray depths on the two axes: the straight-line kernels with the shorter axis padded in front --------------------------
@pytest.mark.parametrize("Nv,Nh,fwv,fwh,PW,PH", [(6, 5, 32, 32, 1280, 720), (12, 3, 1, 64, 600, 400), (7, 11, 4, 1, 500, 333), (10, 4, 2, 32, 1024, 500),
                                                (5, 5, 16, 16, 500, 400), (2, 12, 64, 1, 200, 1000), (9, 10, 2, 2, 1000, 1000), (1, 1, 64, 64, 100, 100)])
def test_padded_gray_axes_take_the_straight_line_kernels(Nv, Nh, fwv, fwh, PW, PH):
    """Any pair of axes with max(N_v, N_h) <= 12 that is not an exact instantiation (N_v = N_h in 6..12) takes the PADDED straight-line
    instantiation for NMAX = max(6, N_v, N_h): the shorter axis is padded IN FRONT with virtual planes that decode to G = 0
    (issue_gray / decode_gray in sl3d_fused.h) -- the reference's own capture set is 6 / 5.  Random frame bytes (every code value, ties, out-of-range correspondences), 6 views in one launch (the
    large-launch kernels), each view alone (the small-launch kernels), ordered clouds: valid maps and point counts bit exact against
    the oracle, points within 1e-5, and one launch equal to the other bit for bit."""
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    W, H, V = 168, 60, 6
    cal_d = syn.synth_rig(W, H, PW, PH)
    cal = syn.cal_tuple(cal_d)
    rng = np.random.default_rng(1000 * Nv + Nh)
    planes = [([rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * Nv)],
               [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * Nh)]) for _ in range(V)]
    masks = [(rng.random((H, W)) < 0.85).astype(np.uint8) for _ in range(V)]
    with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, max_views=V) as sc:
        sc.set_calibration(*cal)
        nmax = max(6, Nv, Nh)
        exact = "true" if Nv == Nh and Nv >= 6 else "false"   # (equal axes of 6..12 planes keep the exact form; "false" with nmax <= 12 = padded)
        for n in (V, 1):
            assert sc.fused_kernel_name(n).startswith(f"sl3d::k_fused<false, {nmax}, false, {exact}, "), sc.fused_kernel_name(n)
        for v in range(V):
            sc.set_mask(masks[v], view=v)
            sc.set_frames(0, planes[v][0], view=v)
            sc.set_frames(1, planes[v][1], view=v)
        sc.run(0, V)
        batch = [sc.points(v) for v in range(V)]
        clouds = sc.fused_clouds(0, V)
        for v in range(V):
            sc.run(v, 1)
            one = sc.points(v)
            assert np.array_equal(one[1], batch[v][1]) and np.array_equal(one[0], batch[v][0], equal_nan=True), v
            assert np.array_equal(sc.fused_clouds(v, 1)[0], clouds[v]), v
    n_valid = 0
    for v in range(V):
        o = Oracle(W, H, PW, PH, Nv, Nh, fwv, fwh)
        o.set_mask(masks[v])
        o.set_calibration(*cal)
        oxyz, ovalid, _ = o.run_scan_rowmajor(*planes[v])
        assert np.array_equal(batch[v][1], ovalid), v
        assert_points_close(batch[v][0], oxyz, ovalid == 1)
        assert np.array_equal(clouds[v], batch[v][0][ovalid == 1]), v
        n_valid += int(ovalid.sum())
    assert n_valid > 0 or min(Nv, Nh) <= 2


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
        # known is still the view's LAST selection: sparse -> k_mask_prepare + the gated kernel once more; that pass counts the new
        # selection, and from then on a new mask + one view is ONE launch of the MASKIN instantiation
        sc.set_mask(dense, view=0)
        assert sc.fused_kernel_name(1).endswith(", 1, 0, true, false>"), sc.fused_kernel_name(1)
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
