"""CPU tests: the oracle against the reference's own known answers (golden crops of the real
captures and of the KAT images the reference wrote), and the closed form of the boundary removal."""
import numpy as np
import pytest

from conftest import golden_calibration, load_golden
from oracle.oracle import Oracle

CROPS = ["real_inside", "real_edge"]


def _run_crop(g, with_cal=True):
    H, W = g["mask"].shape
    N_v, N_h, fw_v, fw_h, nc_v, nc_h = [int(v) for v in g["params"]]
    PW, PH = int(g["full"][2]), int(g["full"][3])
    o = Oracle(W, H, PW, PH, N_v, N_h, fw_v, fw_h, ncodes_v=nc_v, ncodes_h=nc_h,
               col0=int(g["origin"][0]), row0=int(g["origin"][1]))
    o.set_mask(g["mask"])
    o.compute_wrapped_phase(0, list(g["fringe_v"]))
    o.compute_wrapped_phase(1, list(g["fringe_h"]))
    o.unwrap_phase(0, list(g["gray_v"]), list(g["inv_v"]))
    o.unwrap_phase(1, list(g["gray_h"]), list(g["inv_h"]))
    if with_cal:
        cal, _ = golden_calibration()
        o.set_calibration(*cal)
        o.compute_c_p_map()
        o.triangulate()
    return o


# A crop is processed as its own small image, so the boundary removal differs from the full-frame
# run within 3 pixels of the crop border (the scan's `visited` logic looks 2 pixels up/left and the
# crop border is never scanned); everything is compared on the crop interior.
I = np.s_[3:-3, 3:-3]


@pytest.mark.parametrize("name", CROPS)
def test_stage3_kat(name):
    """Stage 3 is pinned by the reference's Wrapped_phase_image.bmp (both axes), bit exact."""
    g = load_golden(name)
    o = _run_crop(g, with_cal=False)
    for a, k in ((0, "kat_wrapped_v"), (1, "kat_wrapped_h")):
        assert np.array_equal(o.debug_image(3, a)[I], g[k][I])
        assert np.array_equal(o.valid_map(a)[I] == 1, g[k][I] != 0)


@pytest.mark.parametrize("name", CROPS)
def test_stage4_kat(name):
    """Stage 4 is pinned by the reference's Unwrapped_phase_{vertical,horizontal}.bmp, bit exact."""
    g = load_golden(name)
    o = _run_crop(g, with_cal=False)
    for a, k in ((0, "kat_unwrapped_v"), (1, "kat_unwrapped_h")):
        assert np.array_equal(o.debug_image(4, a)[I], g[k][I])


@pytest.mark.parametrize("name", CROPS)
def test_crop_matches_full_frame_run(name):
    """The crop run (with the crop origin for stage 7) reproduces the full-frame oracle run."""
    g = load_golden(name)
    o = _run_crop(g)
    v = g["valid"][I] == 1
    assert np.array_equal(o.valid_map(2)[I] == 1, v)
    assert np.array_equal(o.code(0)[I][v], g["code_v"][I][v])
    assert np.array_equal(o.code(1)[I][v], g["code_h"][I][v])
    assert np.array_equal(o.unwrapped_phi(0)[I][v], g["unwrapped_v"][I][v])
    assert np.array_equal(o.unwrapped_phi(1)[I][v], g["unwrapped_h"][I][v])
    assert np.array_equal(o.c_p_map()[I][v], g["c_p_map"][I][v])
    assert np.array_equal(o.intersection_points()[I][v], g["points"][I][v])
    A_cam, A_proj = o.projection_matrices()
    assert np.array_equal(A_cam, g["A_cam"]) and np.array_equal(A_proj, g["A_proj"])


def closed_form_valid(S):
    """Closed form of 3/wrapped_phase.cpp:253-279 used by the HIP kernels (see sl3d_kernels.hip)."""
    H, W = S.shape
    V = S == 1

    def at(A, dx, dy, fill):
        out = np.full_like(A, fill)
        out[max(-dy, 0):H + min(-dy, 0), max(-dx, 0):W + min(-dx, 0)] = A[max(dy, 0):H + min(dy, 0), max(dx, 0):W + min(dx, 0)]
        return out  # out[p] = A[p + d]

    interior = np.zeros((H, W), bool)
    interior[1:-1, 1:-1] = True
    later, earlier = [(1, 0), (-1, 1), (0, 1), (1, 1)], [(-1, -1), (0, -1), (1, -1), (-1, 0)]
    L = np.zeros((H, W), bool)
    for d in later:
        L |= ~at(V, *d, True)
    border_unsel = ~interior & ~V
    B = np.zeros((H, W), bool)
    for d in earlier:
        B |= at(border_unsel, *d, False)
    OK = V | (interior & (L | B))
    out = V & ~L
    for d in earlier:
        out &= at(OK, *d, True)
    return np.where(interior, out, V)


def test_boundary_removal_closed_form():
    """The scan-order dependent boundary removal has the closed form the kernels implement."""
    rng = np.random.default_rng(7)
    H, W = 37, 53
    fr = [np.zeros((H, W), np.uint8)] * 3
    o = Oracle(W, H, 64, 64, 3, 3, 8, 8)
    for trial in range(120):
        p = rng.choice([0.02, 0.1, 0.3, 0.5, 0.8, 0.95, 1.0])
        S = (rng.random((H, W)) < p).astype(np.uint8)
        if trial % 3 == 0:
            S[:] = 0
            for _ in range(4):
                y, x, h, w = rng.integers(0, H), rng.integers(0, W), rng.integers(1, 20), rng.integers(1, 20)
                S[y:y + h, x:x + w] = 1
            S ^= (rng.random((H, W)) < 0.02).astype(np.uint8)
        if trial % 5 == 0:
            S[rng.integers(0, H)] = 2  # only the value 1 selects a pixel
        o.set_mask(S)
        o.compute_wrapped_phase(0, fr)
        assert np.array_equal(o.valid_map(0).astype(bool), closed_form_valid(S)), trial


def test_boundary_removal_is_not_an_erosion():
    """Documents the finding: a 3x3 erosion would also remove the top row (it would keep [4:8, 4:8])."""
    S = np.zeros((12, 12), np.uint8)
    S[3:9, 3:9] = 1
    o = Oracle(12, 12, 64, 64, 3, 3, 8, 8)
    o.set_mask(S)
    o.compute_wrapped_phase(0, [np.zeros((12, 12), np.uint8)] * 3)
    v = o.valid_map(0)
    expect = np.zeros_like(S)
    expect[3:8, 4:8] = 1  # left/right columns and the bottom row go (E, SW, S, SE are 'later'), the top row stays
    assert np.array_equal(v, expect)


def test_float_row_index_matches_integer_below_2p24():
    """7/triangulation.cpp:265 derives the row with float division; exact for the sizes we test."""
    for W, H in ((640, 480), (1920, 1080), (4096, 3000)):
        f = np.arange(W * H, dtype=np.int64)
        lit = np.floor(f.astype(np.float32) / np.float32(W)).astype(np.int64)
        assert np.array_equal(lit, f // W)


def test_synthetic_capture_decodes(synth):
    """Oracle on a synthetic plane: decoded projector coordinates follow the analytic ones except where the
    22/7 pattern period drifts against the Gray code (a property of the reference's patterns)."""
    W, H, PW, PH, N, fw = 320, 240, 512, 384, 7, 4
    cap = synth.make_capture(W, H, PW, PH, N, N, fw, fw)
    o = Oracle(W, H, PW, PH, N, N, fw, fw)
    o.set_mask(cap["mask"])
    o.set_calibration(*synth.cal_tuple(cap["cal"]))
    o.run_scan(cap["planes_v"], cap["planes_h"])
    v = (o.valid_map(2) == 1) & cap["lit"]
    assert v.mean() > 0.7
    cp = o.c_p_map()
    ex = np.abs(cp[..., 0] - cap["xp"])[v]
    ey = np.abs(cp[..., 1] - cap["yp"])[v]
    # errors are either sub-pixel (8-bit quantisation) or exactly one fringe period
    assert np.mean(ex < 1.5) > 0.9 and np.mean(ey < 1.5) > 0.9
    assert ex.max() < fw + 1.5 and ey.max() < fw + 1.5
    good = v & (np.abs(cp[..., 0] - cap["xp"]) < 1.5) & (np.abs(cp[..., 1] - cap["yp"]) < 1.5)
    err = np.linalg.norm(o.intersection_points() - cap["world"], axis=-1)[good]
    assert np.median(err) < 0.5  # mm, 8-bit phase noise through a ~100 mm stand-off


def test_point_cloud_order():
    """O1: compaction in row-major scan order with the double -> float cast."""
    g = load_golden("real_edge")
    o = _run_crop(g)
    cloud = o.point_cloud()
    v = o.valid_map(2) == 1
    assert cloud.shape == (int(v.sum()), 3)
    assert np.array_equal(cloud, o.intersection_points()[v].astype(np.float32))


def test_registration_oracle_matches_numpy():
    """N3 oracle against a numpy restatement of the same float/double steps."""
    from oracle.oracle import register_point_clouds
    rng = np.random.default_rng(2)
    clouds = [rng.normal(0, 50, (n, 3)).astype(np.float32) for n in (17, 0, 301)]
    t = np.array([10.5, -2.0, 33.0], np.float32)
    step = np.float32(40.0)
    out = register_point_clouds(clouds, float(t[0]), float(t[1]), float(t[2]), float(step))
    theta, parts = np.float32(0.0), []
    for c in clouds:
        a = float(theta) * 22.0 / 7.0 / 180.0
        R = np.zeros((4, 4), np.float32)
        R[1, 1] = R[3, 3] = 1
        R[0, 0] = R[2, 2] = np.float32(np.cos(a)); R[0, 2] = np.float32(-1.0 * np.sin(a)); R[2, 0] = np.float32(np.sin(a))
        p = np.concatenate([c - t, np.ones((len(c), 1), np.float32)], 1).astype(np.float32)
        q = (R.astype(np.float64) @ p.astype(np.float64).T).T.astype(np.float32)[:, :3] + t
        parts.append(q.astype(np.float32))
        theta = np.float32(theta + step)
    assert np.allclose(out, np.concatenate(parts), rtol=1e-6, atol=1e-5)
    assert out.shape == (318, 3)


# ---- N1: projector pattern generator (1/pattern_generator.cpp), pinned by the reference's own pattern images -------
def _pattern_fixture():
    import os
    from conftest import ROOT
    return np.load(os.path.join(ROOT, "tests", "golden", "patterns_ref.npz"))


def test_pattern_generator_matches_reference_images():
    """The 1-D profiles of the 45 pattern images the reference generated (1280x720, F=3, fringe width 32; each image
    is that profile replicated along the other axis: make_golden.py checks it on every pixel)."""
    from oracle import oracle as O
    fx = _pattern_fixture()
    PW, PH, F, fwv, fwh = (int(v) for v in fx["config"])
    n = 0
    for axis, extent, fw in ((0, PW, fwv), (1, PH, fwh)):
        ncodes, nplanes = O.pattern_counts(extent, fw)
        assert (ncodes, nplanes) == ((40, 6) if axis == 0 else (23, 5))  # common_variables.h:6-9,23-24
        for kind, key, count in ((O.PATTERN_FRINGE, "fringe", F), (O.PATTERN_GRAY, "gray", nplanes + 1),
                                 (O.PATTERN_INVERSE_GRAY, "inverse", nplanes + 1), (O.PATTERN_BINARY, "binary", nplanes + 1)):
            for i in range(count):
                ref = fx[f"{key}_{'vh'[axis]}_{i}"]
                assert np.array_equal(O.pattern_profile(kind, i, extent, fw, nplanes, F), ref), (key, axis, i)
                n += 1
    assert n == 45
    img = O.pattern_image(O.PATTERN_GRAY, 1, 2, PW, PH, fwh, 5, F)
    assert np.array_equal(img, np.broadcast_to(fx["gray_h_2"][:, None], (PH, PW)))


def test_pattern_gray_code_properties():
    """Size-independent properties: consecutive codes differ in exactly one Gray bit, the Gray -> binary decode of
    stage 4 (running xor, MSB first) returns column // fringe_width, inverse = 255 - pattern."""
    from oracle import oracle as O
    for extent, fw in ((1920, 2), (1280, 32), (1000, 7), (4096, 4)):
        ncodes, N = O.pattern_counts(extent, fw)
        assert ncodes == -(-extent // fw) and (1 << N) >= ncodes
        planes = np.stack([O.pattern_profile(O.PATTERN_GRAY, i, extent, fw, N) for i in range(N)]) // 255
        inv = np.stack([O.pattern_profile(O.PATTERN_INVERSE_GRAY, i, extent, fw, N) for i in range(N)])
        assert np.array_equal(inv, 255 - planes * 255)
        b = np.zeros(extent, dtype=np.int64)
        code = np.zeros(extent, dtype=np.int64)
        for i in range(N):
            b ^= planes[i]
            code = code * 2 + b
        assert np.array_equal(code, np.arange(extent) // fw)
        per_code = planes[:, ::fw]
        assert np.all(np.abs(np.diff(per_code.astype(int), axis=1)).sum(axis=0) == 1)


def test_rowmajor_openmp_baseline_is_bit_identical():
    """CPU baseline (b): the fused row-major OpenMP restatement gives exactly the sequential oracle's valid map and
    float points (same per-pixel operations), for 1 and for all threads, with distortion on both devices, F = 3 and 4."""
    from conftest import pkg
    syn = pkg("synth")
    W, H, PW, PH, N, fw = 200, 120, 256, 192, 6, 8
    for F in (3, 4):
        cap = syn.make_capture(W, H, PW, PH, N, 5, fw, fw, noise=2, n_fringe=F)
        cal = {k: np.array(v, dtype=np.float64).copy() for k, v in cap["cal"].items()}
        cal["dp"] = np.array([0.05, -0.02, 0.001, -0.0005, 0.01])
        mask = cap["mask"].copy()
        mask[30:50, 40:90] = 0
        mask[np.random.default_rng(3).random((H, W)) < 0.02] = 0
        o = Oracle(W, H, PW, PH, N, 5, fw, fw, F=F)
        o.set_mask(mask)
        o.set_calibration(*syn.cal_tuple(cal))
        o.run_scan(cap["planes_v"], cap["planes_h"])
        v = o.valid_map(2) == 1
        ref = o.intersection_points().astype(np.float32)
        for threads in (1, 0):
            xyz, valid, n = o.run_scan_rowmajor(cap["planes_v"], cap["planes_h"], threads=threads)
            assert n >= 1
            assert np.array_equal(valid == 1, v)
            assert np.array_equal(xyz[v], ref[v])
            assert np.isnan(xyz[~v]).all()


def test_undistort_restatement_sanity():
    """N4 (parity unpinned): the restated cvUndistort2 is the identity without distortion, stays within one grey level of
    a float bilinear resampling of the same map on a smooth image, treats channels independently, and fills with 0 where
    the map leaves the image."""
    from oracle import oracle as O
    H, W = 120, 160
    yy, xx = np.mgrid[0:H, 0:W]
    img = (128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.uint8)
    K = [150.0, 0, 80.0, 0, 152.0, 60.0, 0, 0, 1]
    d = [0.1, -0.05, 0.001, 0.0005, 0.01]
    assert np.array_equal(O.undistort(img, K, [0, 0, 0, 0, 0]), img)
    out = O.undistort(img, K, d)
    fx, fy, cx, cy = K[0], K[4], K[2], K[5]
    x, y = (xx - cx) / fx, (yy - cy) / fy
    r2 = x * x + y * y
    kr = 1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2
    u = fx * (x * kr + d[2] * 2 * x * y + d[3] * (r2 + 2 * x * x)) + cx
    v = fy * (y * kr + d[2] * (r2 + 2 * y * y) + d[3] * 2 * x * y) + cy
    x0, y0 = np.floor(u).astype(int), np.floor(v).astype(int)
    a, b = u - x0, v - y0

    def g(r, c):
        ok = (c >= 0) & (c < W) & (r >= 0) & (r < H)
        return np.where(ok, img[np.clip(r, 0, H - 1), np.clip(c, 0, W - 1)], 0).astype(float)

    ref = (1 - b) * ((1 - a) * g(y0, x0) + a * g(y0, x0 + 1)) + b * ((1 - a) * g(y0 + 1, x0) + a * g(y0 + 1, x0 + 1))
    inside = (x0 >= 1) & (x0 + 2 < W) & (y0 >= 1) & (y0 + 2 < H)  # away from the hard 0 border, where 1/32 px moves a lot
    assert inside.mean() > 0.8 and np.abs(out.astype(float) - ref)[inside].max() <= 1.0
    c3 = np.stack([img, 255 - img, img // 2], -1)
    o3 = O.undistort(c3, K, d)
    assert np.array_equal(o3[..., 0], out) and np.array_equal(o3[..., 1], O.undistort(255 - img, K, d))
    far = O.undistort(np.full((H, W), 200, np.uint8), K, [-0.9, 0.0, 0, 0, 0])  # strong barrel term: corners sample outside
    assert far[H // 2, W // 2] == 200


def golden_relative_geometry():
    import json, os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "calibration.json")) as f:
        c = json.load(f)
    g = c["_relative_geometry"]
    return np.array(g["proj_cam_rot_mat"], dtype=np.float64).reshape(3, 3), np.array(g["proj_cam_trans_vect"], dtype=np.float64)


def test_T0_kat():
    """T0 is pinned by reference-held OpenCV 2.4 output: stage 6 ran cvRodrigues2 on the two rotation vectors stage 7 reads, then
    cvTranspose, cvMatMul, cvMatMul, cvSub, and saved Rc*Rp^T and tc - (Rc*Rp^T)*tp (6/system_calibration.cpp:1488-1516 ->
    Triangulation/Relative_geometry/proj_cam_rot_mat.xml, proj_cam_trans_vect.xml).  The oracle's restatements of those
    routines -- the same ones compute_A() uses for A = K[R|t] (7/triangulation.cpp:1069-1116) -- reproduce all 12 doubles BIT FOR BIT."""
    from oracle import oracle as O
    ct, _ = golden_calibration()
    cal = dict(zip(("Kc", "dc", "rc", "tc", "Kp", "dp", "rp", "tp"), ct))
    kat_R, kat_t = golden_relative_geometry()
    R, t = O.relative_geometry(cal["rc"], cal["tc"], cal["rp"], cal["tp"])
    assert np.array_equal(R.view(np.uint64), kat_R.view(np.uint64)), "Rc * Rp^T differs from the reference's proj_cam_rot_mat.xml"
    assert np.array_equal(t.view(np.uint64), kat_t.view(np.uint64)), "tc - (Rc Rp^T) tp differs from the reference's proj_cam_trans_vect.xml"
    # the rotation blocks the KAT pins are the ones inside the oracle's projection matrices: with K = I, A = [R|t]
    o = Oracle(16, 16, 1280, 720, 6, 5, 32, 32)
    I3, z5 = np.eye(3).ravel(), np.zeros(5)
    o.set_calibration(I3, z5, cal["rc"], cal["tc"], I3, z5, cal["rp"], cal["tp"])
    A_cam, A_proj = (np.array(a).reshape(3, 4) for a in o.projection_matrices())
    assert np.array_equal(A_cam[:, 3], cal["tc"]) and np.array_equal(A_proj[:, 3], cal["tp"])
    assert np.array_equal(relative_from_projection(A_cam, A_proj)[0].view(np.uint64), kat_R.view(np.uint64))
    assert np.array_equal(relative_from_projection(A_cam, A_proj)[1].view(np.uint64), kat_t.view(np.uint64))


def relative_from_projection(A_cam, A_proj):
    """Rc * Rp^T and tc - (Rc Rp^T) tp from two [R|t] matrices with cvGEMM's summation order (double accumulator from 0, k
    ascending), so that bit-identical rotation blocks give the bit-identical known answer."""
    Rc, Rp, tc, tp = A_cam[:, :3], A_proj[:, :3], A_cam[:, 3], A_proj[:, 3]
    R = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            acc = 0.0
            for k in range(3):
                acc += float(Rc[i, k]) * float(Rp[j, k])
            R[i, j] = acc
    t = np.zeros(3)
    for i in range(3):
        acc = 0.0
        for k in range(3):
            acc += float(R[i, k]) * float(tp[k])
        t[i] = float(tc[i]) - acc
    return R, t


def test_restated_opencv_routines_against_independent_implementations():
    """Stage 7 calls OpenCV routines that are not in the reference tree; the oracle restates them.  Independent checks:
    cvRodrigues2 + cvGEMM (T0: A = K [R|t]) against scipy's rotation-vector conversion, cvUndistortPoints against the
    forward Brown model it inverts (5 fixed-point iterations converge to ~1e-9 px for the reference's coefficients),
    cvInvert + the literal (P^T P)^-1 P^T F of T3 against numpy's least squares."""
    from scipy.spatial.transform import Rotation
    ct, _ = golden_calibration()
    cal = dict(zip(("Kc", "dc", "rc", "tc", "Kp", "dp", "rp", "tp"), ct))
    o = Oracle(64, 48, 1280, 720, 6, 5, 32, 32)
    o.set_calibration(*ct)
    A_cam, A_proj = (np.array(a).reshape(3, 4) for a in o.projection_matrices())
    for A, K, r, t in ((A_cam, cal["Kc"], cal["rc"], cal["tc"]), (A_proj, cal["Kp"], cal["rp"], cal["tp"])):
        Rm = Rotation.from_rotvec(np.array(r)).as_matrix()
        ref = np.array(K).reshape(3, 3) @ np.hstack([Rm, np.array(t).reshape(3, 1)])
        assert np.allclose(A, ref, rtol=1e-13, atol=1e-10)
    # undistort_point + re-projection (T1): distorting the result with the forward model returns the pixel
    K, d = np.array(cal["Kc"]).reshape(3, 3), np.array(cal["dc"])
    for col, row in ((0, 0), (63, 47), (10, 40), (33, 7)):
        u, v = o.undist_point(0, col, row)[:2]
        x, y = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
        r2 = x * x + y * y
        kr = 1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2
        xd = x * kr + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
        yd = y * kr + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
        assert abs(K[0, 0] * xd + K[0, 2] - col) < 1e-6 and abs(K[1, 1] * yd + K[1, 2] - row) < 1e-6
    # T2 + T3 on a synthetic scan: every valid point is numpy's least-squares solution of its 4x3 system
    from conftest import pkg
    syn = pkg("synth")
    W, H, PW, PH, N, fw = 96, 64, 128, 96, 5, 4
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw)
    o2 = Oracle(W, H, PW, PH, N, N, fw, fw)
    o2.set_mask(cap["mask"])
    ct = syn.cal_tuple(cap["cal"])
    o2.set_calibration(*ct)
    o2.run_scan(cap["planes_v"], cap["planes_h"])
    Ac, Ap = (np.array(a).reshape(3, 4) for a in o2.projection_matrices())
    v = o2.valid_map(2) == 1
    cp, pts = o2.c_p_map(), o2.intersection_points()
    rng = np.random.default_rng(0)
    ys, xs = np.nonzero(v)
    for k in rng.choice(len(ys), size=40, replace=False):
        r, c = ys[k], xs[k]
        cu, cv = o2.undist_point(0, c, r)[:2]
        pu, pv = o2.undist_point(1, int(cp[r, c, 0]), int(cp[r, c, 1]))[:2]
        P = np.array([Ac[0, :3] - cu * Ac[2, :3], Ac[1, :3] - cv * Ac[2, :3], Ap[0, :3] - pu * Ap[2, :3], Ap[1, :3] - pv * Ap[2, :3]])
        F = np.array([Ac[2, 3] * cu - Ac[0, 3], Ac[2, 3] * cv - Ac[1, 3], Ap[2, 3] * pu - Ap[0, 3], Ap[2, 3] * pv - Ap[1, 3]])
        X = np.linalg.lstsq(P, F, rcond=None)[0]
        assert np.linalg.norm(X - pts[r, c]) <= 1e-9 * max(1.0, np.linalg.norm(X))
