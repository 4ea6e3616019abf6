"""GPU tests (-m gpu): SURVEY 8f's "next" rows -- N1 pattern generator and synthetic captures, N3 turntable registration, N4 capture-side
cvUndistort2 -- against the oracle and the reference's own pattern images."""
import numpy as np
import pytest

from conftest import assert_points_close, golden_calibration, load_golden, pkg
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _scanner():
    return pkg("scanner")


def test_undistort_matches_restated_opencv_algorithm():
    """N4: sl3d_undistort == the oracle's restatement of OpenCV 2.4.0's cvUndistort2, byte for byte: 1 and 3 channels,
    widths that give stripes of 1, 2 and many rows, a skewed K, tangential terms, maps that leave the image."""
    from oracle import oracle as O
    S = _scanner()
    rng = np.random.default_rng(21)
    with S.Scanner(64, 48, 64, 48, 5, 5, 2, 2) as sc:
        for (H, W, cn), K, d in (
                ((120, 160, 1), [150.0, 0, 80.0, 0, 152.0, 60.0, 0, 0, 1], [0.1, -0.05, 0.001, 0.0005, 0.01]),
                ((75, 2049, 1), [1900.0, 0, 1020.3, 0, 1905.5, 36.2, 0, 0, 1], [-0.2, 0.07, 0, 0, 0]),        # stripe = 1 row... and 2
                ((600, 800, 3), [1411.4, 0, 396.9, 0, 1418.2, 295.8, 0, 0, 1], [0.0813, -0.1102, 0, 0, 0]),   # the reference's camera, halved
                ((97, 131, 3), [120.0, 0.7, 60.0, 0, 118.0, 50.0, 0, 0, 1], [0.3, -0.2, 0.004, -0.003, 0.05]),  # skew + tangential
                ((64, 64, 1), [40.0, 0, 32.0, 0, 40.0, 32.0, 0, 0, 1], [-0.6, 0.1, 0, 0, 0]),                  # leaves the image
        ):
            img = rng.integers(0, 256, size=(H, W) if cn == 1 else (H, W, cn), dtype=np.uint8)
            got = sc.undistort(img, K, d)
            assert np.array_equal(got, O.undistort(img, K, d)), (H, W, cn)
            img2 = rng.integers(0, 256, size=img.shape, dtype=np.uint8)   # same calibration and size: the cached map is reused
            assert np.array_equal(sc.undistort(img2, K, d), O.undistort(img2, K, d)), (H, W, cn, "cached map")
            d2 = list(d); d2[0] += 0.01                                    # another calibration: the map is rebuilt
            assert np.array_equal(sc.undistort(img2, K, d2), O.undistort(img2, K, d2)), (H, W, cn, "new map")
        with pytest.raises(S.Sl3dError):
            sc.undistort(np.zeros((8, 8, 2), np.uint8), [1.0, 0, 0, 0, 1, 0, 0, 0, 1], [0] * 5)


def test_raw_frames_path_equals_undistort_then_set_frames():
    """sl3d_set_frames_raw == cvUndistort2 (oracle restatement) of every plane with the camera calibration, then the
    normal path: the frame stack holds the same bytes and the scan gives the same points; re-calibration rebuilds the map."""
    from oracle import oracle as O
    syn = pkg("synth")
    S = _scanner()
    W, H, PW, PH, N, fw = 200, 120, 256, 192, 6, 8
    cap = syn.make_capture(W, H, PW, PH, N, 5, fw, fw, noise=2)
    rng = np.random.default_rng(9)
    raw_v = [rng.integers(0, 256, size=(H, W), dtype=np.uint8) for _ in cap["planes_v"]]  # any bytes will do for the byte check
    raw_h = [np.ascontiguousarray(p[:, ::-1]) for p in cap["planes_h"]]
    cal = {k: np.array(v, dtype=np.float64).copy() for k, v in cap["cal"].items()}
    with S.Scanner(W, H, PW, PH, N, 5, fw, fw) as sc, S.Scanner(W, H, PW, PH, N, 5, fw, fw) as ref:
        for dc in ([0.0813, -0.1102, 0.0, 0.0, 0.0], [0.2, -0.1, 0.002, -0.001, 0.03]):
            cal["dc"] = np.array(dc)
            ct = syn.cal_tuple(cal)
            for s_ in (sc, ref):
                s_.set_calibration(*ct)
                s_.set_mask(cap["mask"])
            sc.set_frames_raw(0, raw_v)
            sc.set_frames_raw(1, raw_h)
            und_v = [O.undistort(p, cal["Kc"], cal["dc"]) for p in raw_v]
            und_h = [O.undistort(p, cal["Kc"], cal["dc"]) for p in raw_h]
            for got, exp in zip(sc.frames(0, 0) + sc.frames(1, 0), und_v + und_h):
                assert np.array_equal(got, exp)
            ref.set_frames(0, und_v)
            ref.set_frames(1, und_h)
            sc.run()
            ref.run()
            a, b = sc.points(), ref.points()
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0], equal_nan=True)
    with S.Scanner(W, 60, PW, PH, N, 5, fw, fw, full_size=(W, H), origin=(0, 30)) as stripe:
        stripe.set_calibration(*syn.cal_tuple(cal))
        with pytest.raises(S.Sl3dError):
            stripe.set_frames_raw(0, [p[30:90] for p in raw_v])


def test_turntable_registration():
    """N3 (9/register_point_clouds.cpp): per-view rotation about Y with Pi = 22/7, float accumulation of theta,
    float GEMM with double accumulator -- bit exact against the oracle, clouds in the reference's scan order."""
    from oracle.oracle import register_point_clouds
    S, syn = _scanner(), pkg("synth")
    W, H, PW, PH, N, fw = 200, 96, 256, 256, 6, 4
    caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, view=v, noise=1, plane=(1.0 * v, 0.05, 0.03)) for v in range(3)]
    rng = np.random.default_rng(11)
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=3) as sc:
        sc.set_calibration(*syn.cal_tuple(caps[0]["cal"]))
        for v, c in enumerate(caps):
            m = c["mask"].copy()
            m[rng.random((H, W)) < 0.1 * v] = 0
            sc.set_mask(m, view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, 3)
        clouds = [sc.cloud(v) for v in range(3)]
        assert len({len(c) for c in clouds}) > 1
        for tx, ty, tz, step in ((50.0, 30.0, -5.0, 30.0), (0.0, 0.0, 0.0, 0.0), (12.5, -3.25, 7.0, 7.3)):
            got = sc.register_views(0, 3, tx, ty, tz, step)
            ref = register_point_clouds(clouds, tx, ty, tz, step)
            assert got.shape == ref.shape
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_device_synthetic_capture():
    """N1: the device generator writes the same captures as the host twin (3dscan_amd/synth.py) up to last-ulp
    differences of the trig functions (a grey level on a few bytes per million), its noise hash is bit-identical,
    and the pipeline run on the generated frames equals the oracle run on the downloaded bytes."""
    S, syn = _scanner(), pkg("synth")
    W, H, PW, PH, N, fw = 640, 240, 1024, 768, 8, 4
    host = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=2, view=3, plane=(1.0, 0.04, 0.06))
    cal = syn.cal_tuple(host["cal"])
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, keep_stages=True) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(host["mask"])
        sc.synth_view(0, plane=(1.0, 0.04, 0.06), view_id=3, noise=2)
        dv, dh = sc.frames(0), sc.frames(1)
        total = diff = 0
        for d, h in zip(dv + dh, host["planes_v"] + host["planes_h"]):
            delta = np.abs(d.astype(int) - h.astype(int))
            assert delta.max() <= 255
            diff += int((delta != 0).sum()); total += delta.size
        assert diff / total < 2e-4, diff / total
        # noise-only check (flat scene lit everywhere is hard to isolate): Gray planes are 0/255 patterns, so away from code
        # boundaries every byte must be identical -> covered by the mismatch bound above
        sc.run()
        o = Oracle(W, H, PW, PH, N, N, fw, fw)
        o.set_mask(host["mask"]); o.set_calibration(*cal); o.run_scan(dv, dh)
        v = o.valid_map(2) == 1
        assert np.array_equal(sc.valid_map(2) == 1, v)
        assert np.array_equal(sc.c_p_map()[v], o.c_p_map()[v])
        assert_points_close(sc.points()[0], o.intersection_points(), v)


def test_projector_patterns_match_oracle_and_reference():
    """N1: sl3d_generate_pattern against the oracle (pinned on the reference's pattern images) -- the reference's own
    configuration against the committed profiles of those images, then other sizes / fringe counts / widths, including
    widths that do not divide the extent and a projector width that is not a multiple of 16."""
    import os
    from conftest import ROOT
    from oracle import oracle as O
    S = _scanner()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "patterns_ref.npz"))
    PWr, PHr, F, fwv, fwh = (int(v) for v in fx["config"])
    with S.Scanner(64, 48, PWr, PHr, 6, 5, fwv, fwh, n_fringe=F) as sc:
        for axis in (0, 1):
            N = 6 if axis == 0 else 5
            for kind, key, count in ((S.PATTERN_FRINGE, "fringe", F), (S.PATTERN_GRAY, "gray", N + 1),
                                     (S.PATTERN_INVERSE_GRAY, "inverse", N + 1), (S.PATTERN_BINARY, "binary", N + 1)):
                for i in range(count):
                    prof = fx[f"{key}_{'vh'[axis]}_{i}"]
                    ref = np.broadcast_to(prof[None, :] if axis == 0 else prof[:, None], (PHr, PWr))
                    assert np.array_equal(sc.generate_pattern(kind, axis, i), ref), (key, axis, i)
        with pytest.raises(S.Sl3dError):
            sc.generate_pattern(S.PATTERN_GRAY, 0, 8)
        with pytest.raises(S.Sl3dError):
            sc.generate_pattern(S.PATTERN_FRINGE, 0, 3)
    for PW, PH, Fx, fw_v, fw_h in ((1920, 1080, 3, 2, 2), (1000, 700, 4, 7, 5), (1366, 768, 5, 3, 16)):
        (_, Nv), (_, Nh) = S.pattern_counts(PW, fw_v), S.pattern_counts(PH, fw_h)
        with S.Scanner(64, 48, PW, PH, Nv, Nh, fw_v, fw_h, n_fringe=Fx) as sc:
            for axis, N, fw in ((0, Nv, fw_v), (1, Nh, fw_h)):
                for kind, count in ((S.PATTERN_FRINGE, Fx), (S.PATTERN_GRAY, N + 1), (S.PATTERN_INVERSE_GRAY, N + 1), (S.PATTERN_BINARY, N + 1)):
                    for i in sorted({0, 1, count // 2, count - 2, count - 1}):
                        if 0 <= i < count:
                            assert np.array_equal(sc.generate_pattern(kind, axis, i), O.pattern_image(kind, axis, i, PW, PH, fw, N, Fx)), (PW, kind, axis, i)
