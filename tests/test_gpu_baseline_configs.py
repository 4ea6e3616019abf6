"""GPU tests (-m gpu): BASELINE.json's configurations at their FULL sizes against the oracle -- configs[2] (4096x3000), configs[3] (64 views x
8 row stripes), configs[4] (8192x6144, two axes, 12 Gray planes, as 8 stripes on one GPU) -- plus the other rig classes, Gray depths and
fringe counts at 1920x1080 and contexts whose views straddle 4-GiB boundaries."""
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


# ---- BASELINE configs at their full sizes ---------------------------------------------------------------------------------
def _oracle_stripe(W, RH, PW, PH, N, fw, cal, full_mask, R0, planes_v, planes_h, exact=False):
    o = Oracle(W, RH, PW, PH, N, N, fw, fw, row0=R0, exact_index=exact)
    o.set_mask(full_mask[R0:R0 + RH])
    o.set_calibration(*cal)
    o.run_scan([p[R0:R0 + RH] for p in planes_v], [p[R0:R0 + RH] for p in planes_h])
    return o


def test_config2_full_12mp_frame():
    """configs[2]: ONE 4096x3000 context.  (a) fused == per-stage kernels on valid / codes / c_p_map / points over the whole
    frame, (b) the oracle on 64-row stripes at the top border, mid frame and the bottom border (the oracle treats a stripe
    as its own image, so rows next to an artificial stripe edge are skipped, rows at the TRUE frame border are compared),
    (c) 8 row stripes (375 rows: configs[3]/[4]'s decomposition) reproduce the whole frame bit for bit, as a group with
    the gather, (d) the compacted cloud of the whole frame, (b') the whole frame against the oracle's row-major restatement."""
    S, syn = _S(), pkg("synth")
    W, H, PW, PH, N, fw, RH = 4096, 3000, 2048, 2048, 10, 2, 64
    cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH))
    mask = syn.default_mask(W, H)
    mask[2:40, 100:900] = 0          # structure near the top border
    mask[H - 30:H - 3, 2000:2600] = 0
    mask[1400:1500, 1:700] = 0
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, keep_stages=True) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        sc.synth_view(0, plane=(0.0, 0.05, 0.05), view_id=0, noise=2)
        pv, ph = sc.frames(0), sc.frames(1)
        sc.run()
        fused = dict(valid=sc.valid_map(2), code0=sc.code(0), code1=sc.code(1), cp=sc.c_p_map(), xyz=sc.points()[0])
        sc.run_stages()
        v = sc.valid_map(2) == 1
        assert np.array_equal(fused["valid"] == 1, v) and int(v.sum()) > 11_000_000
        assert np.array_equal(fused["code0"][v], sc.code(0)[v]) and np.array_equal(fused["code1"][v], sc.code(1)[v])
        assert np.array_equal(fused["cp"][v], sc.c_p_map()[v])
        # (the fused kernel takes an undistorted projector's point as the correspondence itself, the stage-7 kernel runs it
        # through K * ((x - c) / f) + c: equal to ~1e-13 px, so the f32 points agree to the last bit or two)
        assert_points_close(fused["xyz"], sc.points()[0], v, rel=1e-6)
    for R0, I in ((0, np.s_[0:RH - 3]), (1472, np.s_[3:RH - 3]), (H - RH, np.s_[3:RH])):
        o = _oracle_stripe(W, RH, PW, PH, N, fw, cal, mask, R0, pv, ph)
        vo = o.valid_map(2)[I] == 1
        sl = np.s_[R0:R0 + RH]
        assert np.array_equal(fused["valid"][sl][I] == 1, vo), R0
        assert np.array_equal(fused["cp"][sl][I][vo], o.c_p_map()[I][vo]), R0
        assert_points_close(fused["xyz"][sl][I], o.intersection_points()[I], vo)
        assert vo.sum() > 100_000
    v = fused["valid"] == 1
    # (b') ALL 12.3 Mpx against the oracle's fused row-major restatement (bit-identical to the stage functions, tests/test_oracle.py)
    o = Oracle(W, H, PW, PH, N, N, fw, fw)
    o.set_mask(mask)
    o.set_calibration(*cal)
    oxyz, ovalid, _ = o.run_scan_rowmajor(pv, ph)
    assert np.array_equal(ovalid == 1, v)
    assert_points_close(fused["xyz"], oxyz.astype(np.float64), v)
    del o, oxyz, ovalid
    # the timed mode on the whole frame (camera-frame solve: equal to the parity mode's general solve to ~1e-12, not bit for bit)
    with S.Scanner(W, H, PW, PH, N, N, fw, fw) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        sc.set_frames(0, pv)
        sc.set_frames(1, ph)
        sc.run()
        txyz, tval = sc.points()
        assert np.array_equal(tval, fused["valid"])
        assert_points_close(txyz, fused["xyz"], v, rel=1e-6)
        fused["xyz"] = txyz
    with S.Group(W, H, PW, PH, N, N, fw, fw, devices=[0] * 8, flags=S.SL3D_FLAG_GROUP_NO_RCCL) as g:
        assert [s[1] for s in g.stripes()] == [375] * 8
        g.set_calibration(*cal)
        g.set_mask(mask)
        g.set_frames(0, pv)
        g.set_frames(1, ph)
        g.run(0, 1)
        g.gather(0, 1)
        xyz, val = g.points(0)
        assert np.array_equal(val, fused["valid"])
        assert np.array_equal(xyz[v], fused["xyz"][v])
        g.run_clouds(0, 1)
        n = g.gather_clouds(0, 1)[0]
        assert n == int(v.sum())
        assert np.array_equal(g.cloud(0), fused["xyz"][v])


def test_config3_shape_8_stripes_of_135_rows():
    """configs[3]'s per-view decomposition: 1920x1080 views as 8 stripes of 135 rows (one per GPU of the node; here 8
    contexts on one GPU behind sl3d_group_*), 4 views: the gathered views equal the whole-frame run bit for bit and one
    view equals the oracle."""
    S, syn = _S(), pkg("synth")
    W, H, N, fw, NV = 1920, 1080, 10, 2, 4
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    mask = syn.default_mask(W, H)
    with S.Scanner(W, H, W, H, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        frames = []
        for v in range(NV):
            sc.set_mask(mask, view=v)
            sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
            frames.append((sc.frames(0, v), sc.frames(1, v)))
        sc.run(0, NV)
        whole = [sc.points(v) for v in range(NV)]
    with S.Group(W, H, W, H, N, N, fw, fw, devices=[0] * 8, max_views=NV, flags=S.SL3D_FLAG_GROUP_NO_RCCL) as g:
        assert [s[1] for s in g.stripes()] == [135] * 8
        g.set_calibration(*cal)
        for v in range(NV):
            g.set_mask(mask, view=v)
            g.set_frames(0, frames[v][0], view=v)
            g.set_frames(1, frames[v][1], view=v)
        for v in range(NV):      # per-view pipeline: view v+1 computes while view v is gathered
            g.run(v, 1)
            g.gather(v, 1)
        for v in range(NV):
            xyz, val = g.points(v)
            assert np.array_equal(val, whole[v][1]), v
            assert np.array_equal(xyz, whole[v][0], equal_nan=True), v
    o = Oracle(W, H, W, H, N, N, fw, fw)
    o.set_mask(mask)
    o.set_calibration(*cal)
    o.run_scan(*frames[2])
    vo = o.valid_map(2) == 1
    assert np.array_equal(whole[2][1] == 1, vo)
    assert_points_close(whole[2][0], o.intersection_points(), vo)


def test_config3_full_batch_64_views_8_stripes():
    """configs[3] at its full size: 64 views of 1920x1080, every view cut into 8 stripes of 135 rows (the 8 GPUs of the node;
    here 8 contexts on one GPU behind sl3d_group_*), inputs generated on the device inside every stripe, computed and
    gathered chunk by chunk (16 views per launch, the next chunk computing while the previous one is gathered): the 64
    assembled views equal a whole-frame context's results bit for bit, dense and compacted."""
    import ctypes as C
    S, syn = _S(), pkg("synth")
    W, H, N, fw, NV, CH = 1920, 1080, 10, 2, 64, 16
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    mask = syn.default_mask(W, H)
    L = S.load_library()

    def synth(handle, v):
        pl = np.asarray((0.75 * (v % 16), 0.05, 0.05 - 0.003 * (v % 16)), dtype=np.float64)
        assert L.sl3d_synth_view(handle, v, pl.ctypes.data, 0x3D5CA11, v, 2, C.c_float(0.8), C.c_float(10.0)) == 0

    with S.Scanner(W, H, W, H, N, N, fw, fw, max_views=NV) as whole, \
         S.Group(W, H, W, H, N, N, fw, fw, devices=[0] * 8, max_views=NV, flags=S.SL3D_FLAG_GROUP_NO_RCCL) as g:
        whole.set_calibration(*cal)
        g.set_calibration(*cal)
        stripes = g.stripes()
        assert [s[1] for s in stripes] == [135] * 8
        for v in range(NV):
            whole.set_mask(mask, view=v)
            g.set_mask(mask, view=v)
            synth(whole._h, v)
            for (_, _, _, h) in stripes:
                synth(h, v)
        for c in range(0, NV, CH):
            g.run(c, CH)
            g.gather(c, CH)
        whole.run(0, NV)
        for v in range(NV):
            xyz, val = g.points(v)
            wx, wv = whole.points(v)
            assert np.array_equal(val, wv), v
            assert np.array_equal(xyz, wx, equal_nan=True), v
        for c in range(0, NV, CH):
            g.run_clouds(c, CH)
            counts = g.gather_clouds(c, CH)
            for k in (0, CH - 1):
                wx, wv = whole.points(c + k)
                assert counts[k] == int((wv == 1).sum())
                assert np.array_equal(g.cloud(c + k), wx[wv == 1]), c + k


@pytest.mark.parametrize("rank", [0, 7])
def test_config4_stripes_at_frame_top_and_bottom(rank):
    """configs[4]: 8192x6144, N = 12, two axes; the stripes of rank 0 (frame top) and rank 7 (frame bottom), 768 rows each.
    Fused compaction == dense on the stripe; the oracle (exact pixel indices: the frame exceeds 2^24 pixels)
    on 24-row sub-stripes at the true frame border and inside the stripe."""
    S, syn = _S(), pkg("synth")
    W, H, N, fw, rows = 8192, 6144, 12, 2, 768
    row0 = rank * rows
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    mask = syn.default_mask(W, H)
    mask[0:6, 4000:5000] = 1 if rank == 0 else 0     # selected pixels ON the top border row
    mask[H - 5:H, 100:300] = 1                       # ... and on the bottom border rows
    with S.Scanner(W, rows, W, H, N, N, fw, fw, full_size=(W, H), origin=(0, row0)) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        sc.synth_view(0, plane=(0.0, 0.05, 0.05), view_id=0, noise=2)
        sc.run()
        xyz, valid = sc.points()
        pv, ph = sc.frames(0), sc.frames(1)
        cl = sc.fused_clouds(0, 1)[0]
        assert np.array_equal(cl, xyz[valid == 1])
    RH = 24
    subs = [(0, np.s_[0:RH - 3]), (400, np.s_[3:RH - 3])] if rank == 0 else [(rows - RH, np.s_[3:RH]), (400, np.s_[3:RH - 3])]
    for r, I in subs:
        o = Oracle(W, RH, W, H, N, N, fw, fw, row0=row0 + r, exact_index=True)
        o.set_mask(mask[row0 + r:row0 + r + RH])
        o.set_calibration(*cal)
        o.run_scan([p[r:r + RH] for p in pv], [p[r:r + RH] for p in ph])
        vo = o.valid_map(2)[I] == 1
        assert np.array_equal(valid[r:r + RH][I] == 1, vo), (rank, r)
        assert_points_close(xyz[r:r + RH][I], o.intersection_points()[I], vo, rel=1e-5)
        assert vo.sum() > 50_000


def test_config4_full_frame_8_stripes_vs_oracle():
    """configs[4] whole: ONE 8192x6144 view (50.3 Mpx, N = 12, two axes) as 8 row stripes of 768 rows behind sl3d_group_*
    (all on device 0, RCCL path forced), gathered dense and compacted == one 50-Mpx context bit for bit, and the whole frame
    against the oracle (fused row-major restatement on all host cores; exact pixel indices -- the frame exceeds 2^24 pixels)."""
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max" and int(lim) < (24 << 30):
            pytest.skip("needs ~6 GB of host memory beside the rest of the session")
    except OSError:
        pass
    S, syn = _S(), pkg("synth")
    W, H, N, fw = 8192, 6144, 12, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    mask = syn.default_mask(W, H)
    mask[0:6, 4000:5000] = 1          # selected pixels on the frame's top border row
    mask[H - 5:H, 100:300] = 1        # ... and on its bottom rows
    mask[767:770, 1000:3000] = 0      # a hole across the first stripe boundary (boundary removal needs the neighbour stripe's mask rows)
    with S.Scanner(W, H, W, H, N, N, fw, fw) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        sc.synth_view(0, plane=(0.0, 0.05, 0.05), view_id=0, noise=2)
        sc.run()
        xyz, valid = sc.points()
        pv, ph = sc.frames(0), sc.frames(1)
        cl = sc.fused_clouds(0, 1)[0]
    assert np.array_equal(cl, xyz[valid == 1])
    with S.Group(W, H, W, H, N, N, fw, fw, devices=[0] * 8, flags=S.SL3D_FLAG_GROUP_FORCE_RCCL) as g:
        assert [(r0, n) for r0, n, _, _ in g.stripes()] == [(768 * k, 768) for k in range(8)]
        g.set_calibration(*cal)
        g.set_mask(mask)
        g.set_frames(0, pv)
        g.set_frames(1, ph)
        g.run()
        g.gather()
        gx, gv = g.points()
        assert np.array_equal(gv, valid) and np.array_equal(gx, xyz, equal_nan=True)
        del gx, gv
        g.run_clouds()
        assert g.gather_clouds() == [int((valid == 1).sum())]
        assert np.array_equal(g.cloud(), cl)
    del cl
    o = Oracle(W, H, W, H, N, N, fw, fw, exact_index=True)
    o.set_mask(mask)
    o.set_calibration(*cal)
    oxyz, ovalid, _ = o.run_scan_rowmajor(pv, ph)
    vo = ovalid == 1
    assert np.array_equal(valid == 1, vo)
    assert vo.sum() > 49_000_000
    worst = 0.0
    for r in range(0, H, 512):   # (in slabs: fp64 copies of 50 M points at once would double the footprint of the test)
        m = vo[r:r + 512]
        a, b = xyz[r:r + 512][m].astype(np.float64), oxyz[r:r + 512][m].astype(np.float64)
        worst = max(worst, float((np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)).max()))
    assert worst <= 1e-5, worst


@pytest.mark.parametrize("rig", ["distorted", "general"])
def test_other_rigs_full_hd_vs_oracle(rig):
    """The instantiations bench.py reports as side figures -- distorted projector (RIG 2: projector table) and general rig (RIG 0:
    skewed camera matrix, tangential camera distortion) -- at 1920x1080, 3 views in one launch, timed mode and fused compaction,
    against the oracle on every pixel."""
    S, syn = _S(), pkg("synth")
    W, H, N, fw, NV = 1920, 1080, 10, 2, 3
    cal_d = syn.synth_rig(W, H, W, H)
    cal_d["dp"] = np.array([-0.05, 0.02, 0.001, -0.0005, 0.0])
    cal_d["dc"] = np.array(cal_d["dc"], dtype=np.float64) + np.array([0.0, 0.0, 0.0008, -0.0006, 0.0])
    if rig == "general":
        Kc = np.array(cal_d["Kc"], dtype=np.float64).reshape(3, 3).copy()
        Kc[0, 1] = 0.35
        cal_d["Kc"] = Kc.ravel()
    cal = syn.cal_tuple(cal_d)
    mask = syn.default_mask(W, H)
    mask[300:340, 500:900] = 0
    with S.Scanner(W, H, W, H, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v in range(NV):
            sc.set_mask(mask, view=v)
            sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
        sc.run(0, NV)
        dense = [sc.points(v) for v in range(NV)]
        clouds = sc.fused_clouds(0, NV)
        for v in range(NV):
            assert np.array_equal(clouds[v], dense[v][0][dense[v][1] == 1]), v
            o = Oracle(W, H, W, H, N, N, fw, fw)
            o.set_mask(mask)
            o.set_calibration(*cal)
            oxyz, ovalid, _ = o.run_scan_rowmajor(sc.frames(0, v), sc.frames(1, v))
            assert np.array_equal(dense[v][1], ovalid), v
            assert int((ovalid == 1).sum()) > 1_500_000
            assert_points_close(dense[v][0], oxyz, ovalid == 1)


@pytest.mark.parametrize("nv,nh,fwv,fwh", [(10, 9, 2, 4), (9, 9, 4, 4), (12, 11, 1, 1)])
def test_gray_depths_full_hd_vs_oracle(nv, nh, fwv, fwh):
    """Gray depths other than the benchmark's 10/10 (the kernel is instantiated per max(N_v, N_h); unequal depths take the
    shorter axis at run time) at 1920x1080, 2 views per launch, timed mode and fused compaction, every pixel against the oracle."""
    S, syn = _S(), pkg("synth")
    W, H, NV = 1920, 1080, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    mask = syn.default_mask(W, H)
    with S.Scanner(W, H, W, H, nv, nh, fwv, fwh, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v in range(NV):
            sc.set_mask(mask, view=v)
            sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
        sc.run(0, NV)
        dense = [sc.points(v) for v in range(NV)]
        clouds = sc.fused_clouds(0, NV)
        for v in range(NV):
            assert np.array_equal(clouds[v], dense[v][0][dense[v][1] == 1]), v
            o = Oracle(W, H, W, H, nv, nh, fwv, fwh)
            o.set_mask(mask)
            o.set_calibration(*cal)
            oxyz, ovalid, _ = o.run_scan_rowmajor(sc.frames(0, v), sc.frames(1, v))
            assert np.array_equal(dense[v][1], ovalid), v
            assert int((ovalid == 1).sum()) > 1_500_000
            assert_points_close(dense[v][0], oxyz, ovalid == 1)


@pytest.mark.parametrize("F", [4, 5])
def test_fringe_counts_full_hd_vs_oracle(F):
    """4- and 5-step fringe sets (3/wrapped_phase.cpp:188-229; with 5 the reference's validity assignment is commented out,
    :117-129: nothing is valid) at 1920x1080 in the timed mode, against the oracle on every pixel."""
    S, syn = _S(), pkg("synth")
    W, H, N, fw = 1920, 1080, 10, 2
    cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
    mask = syn.default_mask(W, H)
    with S.Scanner(W, H, W, H, N, N, fw, fw, n_fringe=F) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        sc.synth_view(0, plane=(0.0, 0.05, 0.05), view_id=0, noise=2)
        sc.run()
        xyz, valid = sc.points()
        cl = sc.fused_clouds(0, 1)[0]
        assert np.array_equal(cl, xyz[valid == 1])
        if F == 5:
            assert not valid.any() and len(cl) == 0
            return
        o = Oracle(W, H, W, H, N, N, fw, fw, F=F)
        o.set_mask(mask)
        o.set_calibration(*cal)
        oxyz, ovalid, _ = o.run_scan_rowmajor(sc.frames(0), sc.frames(1))
        assert np.array_equal(valid, ovalid)
        assert int((ovalid == 1).sum()) > 1_500_000
        assert_points_close(xyz, oxyz, ovalid == 1)


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
