"""GPU parity tests (-m gpu): the HIP kernels, called through the C ABI, against the CPU oracle and the committed golden vectors of the
reference's real captures -- per stage and fused, small / ragged / degenerate shapes, every rig class, every Gray depth and fringe count,
small and large launches.  Bars: valid maps, Gray-code indices, correspondences, wrapped and absolute phase bit exact; 3-D points within
1e-5 relative (assert_points_close).  (The BASELINE configurations at full size: test_gpu_baseline_configs.py.)"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_points_close, golden_calibration, load_golden, pkg
from oracle.oracle import Oracle
from test_oracle import golden_relative_geometry, relative_from_projection

pytestmark = pytest.mark.gpu


def _scanner():
    return pkg("scanner")


# ---- golden vectors from the reference's real captures -------------------------------------------
@pytest.mark.parametrize("name", ["real_inside", "real_edge"])
@pytest.mark.parametrize("staged", [True, False])
def test_golden_real_captures(name, staged):
    """Crops of the real 1600x1200 captures, processed as a window of the full frame; expected values are the
    reference's own KAT images (stage 3/4, pinned) and the full-frame oracle run (stage 5/7)."""
    S = _scanner()
    g = load_golden(name)
    cal, dims = golden_calibration()
    H, W = g["mask"].shape
    x0, y0 = [int(v) for v in g["origin"]]
    N_v, N_h, fw_v, fw_h, nc_v, nc_h = [int(v) for v in g["params"]]
    full = np.zeros((dims["H"], dims["W"]), np.uint8)
    full[y0 - 2:y0 + H + 2, x0 - 2:x0 + W + 2] = g["mask_halo2"]
    with S.Scanner(W, H, dims["PW"], dims["PH"], N_v, N_h, fw_v, fw_h, n_codes_v=nc_v, n_codes_h=nc_h,
                   keep_stages=True, full_size=(dims["W"], dims["H"]), origin=(x0, y0)) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(full)
        sc.set_frames(0, list(g["fringe_v"]) + list(g["gray_v"]) + list(g["inv_v"]))
        sc.set_frames(1, list(g["fringe_h"]) + list(g["gray_h"]) + list(g["inv_h"]))
        if staged:
            sc.run_stages()
            # reference-provided known answers, bit exact, on every pixel of the crop
            assert np.array_equal(sc.debug_image(3, 0), g["kat_wrapped_v"])
            assert np.array_equal(sc.debug_image(3, 1), g["kat_wrapped_h"])
            assert np.array_equal(sc.debug_image(4, 0), g["kat_unwrapped_v"])
            assert np.array_equal(sc.debug_image(4, 1), g["kat_unwrapped_h"])
        else:
            sc.run()
        v = g["valid"] == 1
        assert np.array_equal(sc.valid_map(0) == 1, g["kat_wrapped_v"] != 0)
        assert np.array_equal(sc.valid_map(2) == 1, v)
        assert np.array_equal(sc.code(0)[v], g["code_v"][v]) and np.array_equal(sc.code(1)[v], g["code_h"][v])
        assert np.array_equal(sc.wrapped_phase(0)[v], g["wrapped_v"][v])
        assert np.array_equal(sc.wrapped_phase(1)[v], g["wrapped_h"][v])
        assert np.array_equal(sc.unwrapped_phase(0)[v], g["unwrapped_v"][v])
        assert np.array_equal(sc.unwrapped_phase(1)[v], g["unwrapped_h"][v])
        assert np.array_equal(sc.c_p_map()[v], g["c_p_map"][v])
        assert_points_close(sc.intersection_points(), g["points"], v)
        xyz, valid = sc.points()
        assert np.array_equal(valid == 1, v)
        assert_points_close(xyz, g["points"], v)


def _run_both(W, H, PW, PH, Nv, Nh, fwv, fwh, cap, mask, F=3):
    S = _scanner()
    o = _oracle_for(cap, W, H, PW, PH, Nv, Nh, fwv, fwh, mask, F=F)
    cal = pkg("synth").cal_tuple(cap["cal"])
    for staged in (True, False):
        with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, n_fringe=F, keep_stages=True) as sc:
            sc.set_calibration(*cal)
            sc.set_mask(mask)
            sc.set_frames(0, cap["planes_v"])
            sc.set_frames(1, cap["planes_h"])
            if staged:
                sc.run_stages()
            else:
                sc.run()
            _compare(sc, o, mask == 1, staged, "staged" if staged else "fused")
    # the timed mode (no stage planes) must give the same points as parity mode
    with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, n_fringe=F) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        sc.set_frames(0, cap["planes_v"])
        sc.set_frames(1, cap["planes_h"])
        sc.run()
        xyz, valid = sc.points()
        v = o.valid_map(2) == 1
        assert np.array_equal(valid == 1, v)
        assert_points_close(xyz, o.intersection_points(), v)
        # ... and the cloud compacted inside the kernel is exactly those points in scan order
        assert np.array_equal(sc.fused_clouds(0, 1)[0], xyz[v])
    return o


def _oracle_for(cap, W, H, PW, PH, Nv, Nh, fwv, fwh, mask, F=3):
    o = Oracle(W, H, PW, PH, Nv, Nh, fwv, fwh, F=F)
    o.set_mask(mask)
    o.set_calibration(*pkg("synth").cal_tuple(cap["cal"]))
    o.run_scan(cap["planes_v"], cap["planes_h"])
    return o


def _compare(sc, o, sel, staged, what):
    """sel = selected pixels (mask == 1).  staged=True: per-stage kernels (all planes defined as the reference
    leaves them); False: fused kernel in parity mode (planes defined on valid pixels)."""
    for a in (0, 1):
        assert np.array_equal(sc.valid_map(a), o.valid_map(a)), f"{what}: valid map axis {a}"
    vm = o.valid_map(2)
    assert np.array_equal(sc.valid_map(2), vm), f"{what}: merged valid map"
    v = vm == 1
    va = [o.valid_map(0) == 1, o.valid_map(1) == 1]
    for a in (0, 1):
        assert np.array_equal(sc.code(a)[va[a]], o.code(a)[va[a]]), f"{what}: code axis {a}"
        where = sel if staged else va[a]
        assert np.array_equal(sc.wrapped_phase(a)[where], o.wrapped_phi(a)[where]), f"{what}: wrapped phase axis {a}"
        assert np.array_equal(sc.unwrapped_phase(a)[va[a]], o.unwrapped_phi(a)[va[a]]), f"{what}: unwrapped phase axis {a}"
        if staged:
            assert np.array_equal(sc.code(a), o.code(a))
            assert np.array_equal(sc.debug_image(3, a), o.debug_image(3, a)), f"{what}: stage-3 debug image axis {a}"
            assert np.array_equal(sc.debug_image(4, a), o.debug_image(4, a)), f"{what}: stage-4 debug image axis {a}"
    assert np.array_equal(sc.c_p_map()[v], o.c_p_map()[v]), f"{what}: c_p_map"
    ref = o.intersection_points()
    assert_points_close(sc.intersection_points(), ref, v)
    xyz, valid = sc.points()
    assert np.array_equal(valid, vm)
    assert_points_close(xyz, ref, v, rel=1e-5)
    assert np.isnan(xyz[~v]).all()
    cloud = sc.cloud()
    assert cloud.shape == (int(v.sum()), 3)
    assert np.array_equal(cloud, xyz[v])


# ---- synthetic captures: BASELINE.json configs -----------------------------------------------------
def test_config1_640x480():
    """configs[0]: 640x480, 3 phase + 8-bit Gray code per axis."""
    syn = pkg("synth")
    W, H, PW, PH, N, fw = 640, 480, 1024, 768, 8, 4
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw)
    _run_both(W, H, PW, PH, N, N, fw, fw, cap, cap["mask"])


def test_config2_1920x1080_noise():
    """configs[1]: 1920x1080, 3 phase + 10 Gray per axis; camera noise +-2 so Gray thresholds see ties."""
    syn = pkg("synth")
    W, H, PW, PH, N, fw = 1920, 1080, 1920, 1080, 10, 2
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=2)
    o = _run_both(W, H, PW, PH, N, N, fw, fw, cap, cap["mask"])
    assert (o.valid_map(2) == 1).mean() > 0.5


# ---- edge cases -------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H", [(101, 37), (64, 5), (19, 64), (130, 3)])
def test_ragged_sizes_random_masks(W, H):
    """Widths that are not multiples of 4/16, tiny heights, random masks that touch the frame border
    (border pixels keep their selection, the unwrap skips the first/last column/row)."""
    syn = pkg("synth")
    rng = np.random.default_rng(W * 1000 + H)
    PW, PH, N, fw = 256, 256, 6, 4
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=3)
    for p in (0.0, 0.5, 0.9, 1.0):
        mask = (rng.random((H, W)) < p).astype(np.uint8)
        if p == 0.9:
            mask[rng.random((H, W)) < 0.05] = 2  # only the value 1 selects
        _run_both(W, H, PW, PH, N, N, fw, fw, cap, mask)


def test_saturated_and_flat_frames():
    """All-equal frames: atan2(0,0) = 0, Gray ties decode as 1 (THRESH 0, >=)."""
    W, H, PW, PH, N, fw = 96, 40, 128, 128, 5, 4
    syn = pkg("synth")
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw)
    for val in (0, 255, 17):
        cap["planes_v"] = [np.full((H, W), val, np.uint8) for _ in cap["planes_v"]]
        cap["planes_h"] = [np.full((H, W), val, np.uint8) for _ in cap["planes_h"]]
        _run_both(W, H, PW, PH, N, N, fw, fw, cap, cap["mask"])


def test_random_bytes():
    """Uniform random frame bytes: exercises the whole atan2 lattice, all codes and out-of-range rejections."""
    W, H, PW, PH, N, fw = 256, 128, 300, 200, 7, 3
    syn = pkg("synth")
    cap = syn.make_capture(W, H, 384, 384, N, N, fw, fw)
    rng = np.random.default_rng(3)
    cap["planes_v"] = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in cap["planes_v"]]
    cap["planes_h"] = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in cap["planes_h"]]
    o = _run_both(W, H, PW, PH, N, N, fw, fw, cap, cap["mask"])
    frac = (o.valid_map(2) == 1).mean()
    assert 0.0 < frac < 1.0  # some correspondences fall outside the projector and are rejected


def test_four_step_and_five_step():
    """F=4 uses the 4-step formula (3/wrapped_phase.cpp:188-204); F=5 yields no valid pixel, as in the reference."""
    W, H, PW, PH, N, fw = 128, 48, 256, 256, 6, 4
    syn = pkg("synth")
    for F in (4, 5):
        cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, n_fringe=F, noise=1)
        o = _run_both(W, H, PW, PH, N, N, fw, fw, cap, cap["mask"], F=F)
        if F == 5:
            assert (o.valid_map(2) == 1).sum() == 0


def test_projector_distortion_and_skew():
    """Non-zero projector distortion (5 iterations on the projector side too), tangential terms, a skewed K."""
    syn = pkg("synth")
    W, H, PW, PH, N, fw = 320, 200, 512, 384, 7, 4
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw)
    cal = {k: np.array(v, dtype=np.float64).copy() for k, v in cap["cal"].items()}
    cal["dp"] = np.array([0.05, -0.02, 0.001, -0.0005, 0.01])
    cal["dc"] = np.array([0.0813, -0.1102, 0.0007, -0.0003, 0.02])
    cal["Kc"][1] = 0.3  # skew
    cap["cal"] = cal
    _run_both(W, H, PW, PH, N, N, fw, fw, cap, cap["mask"])


def test_distorted_projector_table_path_and_recalibration():
    """Plain camera K + distorted projector: the timed kernel takes the projector's undistorted point from the
    per-calibration table (rig class 2).  Then the same context is re-calibrated to the reference's kind of rig (class 1)
    and back: the table is rebuilt, results follow the calibration."""
    syn = pkg("synth")
    S = _scanner()
    W, H, PW, PH, N, fw = 320, 200, 512, 384, 7, 4
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=1)
    cal_a = {k: np.array(v, dtype=np.float64).copy() for k, v in cap["cal"].items()}
    cal_a["dp"] = np.array([0.05, -0.02, 0.001, -0.0005, 0.01])
    cal_a["dc"] = np.array([0.0813, -0.1102, 0.0007, -0.0003, 0.02])
    cal_b = {k: np.array(v, dtype=np.float64).copy() for k, v in cap["cal"].items()}
    cal_c = {k: v.copy() for k, v in cal_a.items()}
    cal_c["dp"] = np.array([-0.08, 0.03, 0.0, 0.0, 0.0])
    cap_a = dict(cap, cal=cal_a)
    _run_both(W, H, PW, PH, N, N, fw, fw, cap_a, cap["mask"])
    with S.Scanner(W, H, PW, PH, N, N, fw, fw) as sc:
        sc.set_mask(cap["mask"])
        sc.set_frames(0, cap["planes_v"])
        sc.set_frames(1, cap["planes_h"])
        # (cal_b first: its radial-only camera table, one double per pixel, has to grow into cal_a's two-double one)
        for cal in (cal_b, cal_a, cal_b, cal_c, cal_a):
            o = _oracle_for(dict(cap, cal=cal), W, H, PW, PH, N, N, fw, fw, cap["mask"])
            sc.set_calibration(*syn.cal_tuple(cal))
            sc.run()
            xyz, valid = sc.points()
            v = o.valid_map(2) == 1
            assert np.array_equal(valid == 1, v)
            assert_points_close(xyz, o.intersection_points(), v)


# ---- size-independent properties at full size --------------------------------------------------------
def test_12mp_fused_equals_staged_and_row_shards():
    """configs[2] (4096x3000): the oracle would take minutes, so use properties: (a) the fused kernel equals the
    per-stage kernels bit for bit on codes / correspondences / valid, (b) processing the frame as two row
    stripes (the multi-GPU decomposition) reproduces the single-context result exactly, (c) a 64-row stripe
    equals the oracle run on that stripe (interior rows)."""
    S, syn = _scanner(), pkg("synth")
    W, H, PW, PH, N, fw = 4096, 3000, 2048, 2048, 10, 2
    R0, RH = 1400, 64  # oracle stripe
    cap = syn.make_capture(W, RH, PW, PH, N, N, fw, fw, row0=R0, full=(W, H), noise=1)
    full_mask = syn.default_mask(W, H)
    cal = syn.cal_tuple(cap["cal"])

    def stripe(rows0, rows, planes_v, planes_h, keep):
        sc = S.Scanner(W, rows, PW, PH, N, N, fw, fw, keep_stages=keep, full_size=(W, H), origin=(0, rows0))
        sc.set_calibration(*cal)
        sc.set_mask(full_mask)
        sc.set_frames(0, planes_v)
        sc.set_frames(1, planes_h)
        return sc

    with stripe(R0, RH, cap["planes_v"], cap["planes_h"], True) as a:
        a.run()
        fused = (a.valid_map(2), a.code(0), a.code(1), a.c_p_map(), a.points()[0])
        a.run_stages()
        v = a.valid_map(2) == 1
        assert np.array_equal(fused[0] == 1, v)
        assert np.array_equal(fused[1][v], a.code(0)[v]) and np.array_equal(fused[2][v], a.code(1)[v])
        assert np.array_equal(fused[3][v], a.c_p_map()[v])
        assert np.array_equal(fused[4][v], a.points()[0][v])
    # (b) two half stripes
    h2 = RH // 2
    parts = []
    for k in range(2):
        sl = slice(k * h2, (k + 1) * h2)
        with stripe(R0 + k * h2, h2, [p[sl] for p in cap["planes_v"]], [p[sl] for p in cap["planes_h"]], False) as s2:
            s2.run()
            parts.append(s2.points())
    xyz2 = np.concatenate([p[0] for p in parts]); val2 = np.concatenate([p[1] for p in parts])
    assert np.array_equal(val2, fused[0])
    # the timed mode (camera-frame solve, camera-side T1 from its per-calibration table) against the parity mode: the same
    # points to the last bit or two of the f32 output; two stripes against ONE timed-mode stripe: bit for bit
    assert_points_close(xyz2, fused[4], val2 == 1, rel=1e-6)
    with stripe(R0, RH, cap["planes_v"], cap["planes_h"], False) as s1:
        s1.run()
        xyz1, val1 = s1.points()
    assert np.array_equal(val1, val2) and np.array_equal(xyz1[val1 == 1], xyz2[val2 == 1])
    # (c) oracle on the stripe (its own small image with the stripe origin): compare away from the stripe's top/bottom rows
    o = Oracle(W, RH, PW, PH, N, N, fw, fw, row0=R0)
    o.set_mask(full_mask[R0:R0 + RH])
    o.set_calibration(*cal)
    o.run_scan(cap["planes_v"], cap["planes_h"])
    I = np.s_[3:-3, :]
    vo = o.valid_map(2)[I] == 1
    assert np.array_equal(fused[0][I] == 1, vo)
    assert np.array_equal(fused[3][I][vo], o.c_p_map()[I][vo])
    assert_points_close(fused[4][I], o.intersection_points()[I], vo)


def test_batch_of_views_matches_single_views():
    """A batch launch over several views (one kernel) equals running the views one by one."""
    S, syn = _scanner(), pkg("synth")
    W, H, PW, PH, N, fw = 640, 200, 1024, 768, 8, 4
    caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, view=v, noise=2, plane=(0.0 + 3 * v, 0.05, 0.02 * v)) for v in range(3)]
    cal = syn.cal_tuple(caps[0]["cal"])
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=3) as sc:
        sc.set_calibration(*cal)
        for v, c in enumerate(caps):
            sc.set_mask(c["mask"], view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, 3)
        batch = [sc.points(v) for v in range(3)]
        for v in range(3):
            sc.run(v, 1)
            xyz, val = sc.points(v)
            assert np.array_equal(val, batch[v][1])
            assert np.array_equal(xyz[val == 1], batch[v][0][val == 1])
    assert not np.array_equal(batch[0][0][batch[0][1] == 1][:100], batch[1][0][batch[1][1] == 1][:100])


def test_above_2p24_pixels_integer_pixel_indices():
    """BASELINE config 5's shape (8192x6144 camera and projector, N = 12): the reference indexes its stage-7 tables with
    floorf((float)f / (float)W) (7/triangulation.cpp:264-265), which is wrong above 2^24 pixels.  The product uses integer
    rows / columns: it equals the oracle run with exact indices (to the float rounding of the output).  The oracle that
    reproduces the reference's float index puts the first / last columns of projector rows beyond 2^24 / PW in the wrong
    row (float(f) is only exact to +-2 there), which is visible in its table and in any correspondence that lands there."""
    syn = pkg("synth")
    S = _scanner()
    W, H, N, fw, rows, row0 = 8192, 6144, 12, 2, 24, 3000   # a stripe whose projector rows lie beyond 2^24 / PW = 2048
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    mask = syn.default_mask(W, H)
    with S.Scanner(W, rows, W, H, N, N, fw, fw, full_size=(W, H), origin=(0, row0)) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        sc.synth_view(0, plane=(0.0, 0.05, 0.05), view_id=0, noise=2)
        sc.run()
        xyz, valid = sc.points()
        pv, ph = sc.frames(0), sc.frames(1)
    I = np.s_[3:rows - 3]  # the oracle treats the stripe as its own image: skip its first and last rows
    worst, edge = {}, {}
    for exact in (True, False):
        o = Oracle(W, rows, W, H, N, N, fw, fw, exact_index=exact, row0=row0)
        o.set_mask(mask[row0:row0 + rows])
        o.set_calibration(*cal)
        o.run_scan(pv, ph)
        v = o.valid_map(2) == 1
        assert np.array_equal(valid[I] == 1, v[I])
        ref, got = o.intersection_points()[I][v[I]], xyz[I][v[I]].astype(np.float64)
        worst[exact] = float(np.max(np.linalg.norm(got - ref, axis=-1) / np.linalg.norm(ref, axis=-1)))
        edge[exact] = o.undist_point(1, W - 1, 3001)  # projector pixel (8191, 3001): index 24,592,383 > 2^24
        del o
    assert worst[True] < 2e-7, worst
    assert abs(edge[True][1] - edge[False][1]) > 0.5, edge  # the reference's float index is a row off there


@pytest.mark.parametrize("Nv,Nh,fwv,fwh,PW,PH", [(6, 5, 32, 32, 1280, 720), (12, 12, 1, 1, 600, 400), (16, 9, 3, 5, 500, 333),
                                                (1, 2, 64, 64, 100, 200), (9, 8, 2, 4, 1024, 768)])
def test_gray_depths_and_fringe_widths(Nv, Nh, fwv, fwh, PW, PH):
    """Different bit depths per axis (the reference's own 6/5 with fw=32), the maximum of 16 planes, a single plane,
    fringe widths that are not powers of two, projector sizes that are not multiples of the fringe width; random
    frame bytes so every code value and many out-of-range correspondences occur."""
    syn = pkg("synth")
    W, H = 168, 60
    cap = syn.make_capture(W, H, 256, 256, 8, 8, 4, 4)
    rng = np.random.default_rng(Nv * 100 + Nh)
    cap["planes_v"] = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * Nv)]
    cap["planes_h"] = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3 + 2 * Nh)]
    mask = (rng.random((H, W)) < 0.85).astype(np.uint8)
    _run_both(W, H, PW, PH, Nv, Nh, fwv, fwh, cap, mask)


def test_rerun_is_idempotent_and_masks_can_change():
    """Running the same view twice gives identical bits; changing only the mask changes only validity."""
    S, syn = _scanner(), pkg("synth")
    W, H, PW, PH, N, fw = 256, 64, 512, 256, 7, 4
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=2)
    with S.Scanner(W, H, PW, PH, N, N, fw, fw) as sc:
        sc.set_calibration(*syn.cal_tuple(cap["cal"]))
        sc.set_mask(cap["mask"])
        sc.set_frames(0, cap["planes_v"])
        sc.set_frames(1, cap["planes_h"])
        sc.run()
        a = sc.points()
        sc.run()
        b = sc.points()
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        m2 = cap["mask"].copy()
        m2[:, W // 2:] = 0
        sc.set_mask(m2)
        sc.run()
        c = sc.points()
        assert c[1][:, W // 2:].sum() == 0
        keep = c[1] == 1
        assert np.array_equal(c[0][keep].view(np.uint32), a[0][keep].view(np.uint32))


def test_randomised_configurations():
    """tests/fuzz_parity.py: 120 random combinations of frame / window / projector sizes, Gray depths, fringe widths and
    counts, masks (including bytes other than 0/1), noise and rigs, timed and parity mode against the oracle
    (1200 cases over three other seeds were run clean when this was written; in round 2, with the fused compaction, the
    row-stripe groups on both transports and multi-view batches added to every case, 4750 more over six seeds, and 370 with frames up to 4200x2200 -- FUZZ_MAXW / FUZZ_MAXH)."""
    import importlib.util
    import os
    import sys
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tests", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["fuzz_parity.py", "120", "7"]
        assert mod.main() == 0
    finally:
        sys.argv = argv


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


# ---- segmented clouds: the wait-free ordered compaction and its consumers ------------------------------------------------
def _masks(rng, W, H):
    full = np.zeros((H, W), np.uint8)
    full[1:H - 1, 1:W - 1] = 1
    sparse = (rng.random((H, W)) < 0.07).astype(np.uint8) * full
    holes = full.copy()
    holes[H // 3:H // 2, W // 4:W // 2] = 0
    holes[rng.random((H, W)) < 0.02] = 0
    return [full, sparse, np.zeros((H, W), np.uint8), holes]


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


def _random_mask(rng, W, H, p=0.1):
    m = np.ones((H, W), np.uint8)
    m[0, :] = m[-1, :] = 0
    m[:, 0] = m[:, -1] = 0
    m[rng.random((H, W)) < p] = 0
    return m


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
