"""GPU tests (-m gpu): row-stripe groups behind the one caller (sl3d_group_*: the multi-GPU path of the reference's single process) with several
stripes on ONE GPU -- device copies, RCCL forced, every stripe on its own communication side, the N-rank exchange through a test double of
librccl, per-stripe host assembly.  (Real multi-device runs: test_zz_multi_device.py, armed on >= 2 GPUs.)"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_points_close, golden_calibration, pkg
from oracle.oracle import Oracle
from test_oracle import golden_relative_geometry, relative_from_projection

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


# ---- row-stripe groups behind the C ABI -----------------------------------------------------------------------------------
def _single_context_reference(W, H, PW, PH, N, fw, cal, masks, caps):
    S = _S()
    out = []
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=len(caps)) as sc:
        sc.set_calibration(*cal)
        for v, c in enumerate(caps):
            sc.set_mask(masks[v], view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, len(caps))
        for v in range(len(caps)):
            out.append(sc.points(v))
    return out


@pytest.mark.parametrize("n_stripes,H", [(4, 200), (4, 203), (7, 64), (1, 50)])
@pytest.mark.parametrize("transport", ["copy", "rccl"])
def test_group_stripes_equal_single_context(n_stripes, H, transport):
    """sl3d_group_*: n row stripes (unequal heights included) on GPU 0, run + gather == the single-context result bit for
    bit (valid, xyz), dense and compacted; with device copies and with the RCCL send/recv path forced (self sends)."""
    S, syn = _S(), pkg("synth")
    W, PW, PH, N, fw, NV = 320, 512, 384, 7, 4, 3
    rng = np.random.default_rng(H + n_stripes)
    caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, view=v, noise=2, plane=(3.0 * v, 0.05, 0.02 * v)) for v in range(NV)]
    cal = syn.cal_tuple(caps[0]["cal"])
    masks = [caps[0]["mask"]] + [_random_mask(rng, W, H) for _ in range(NV - 1)]
    ref = _single_context_reference(W, H, PW, PH, N, fw, cal, masks, caps)
    flags = S.SL3D_FLAG_GROUP_FORCE_RCCL if transport == "rccl" else S.SL3D_FLAG_GROUP_NO_RCCL
    with S.Group(W, H, PW, PH, N, N, fw, fw, devices=[0] * n_stripes, max_views=NV, flags=flags) as g:
        assert g.transport == transport
        st = g.stripes()
        assert st[0][0] == 0 and sum(s[1] for s in st) == H and max(s[1] for s in st) - min(s[1] for s in st) <= 1
        g.set_calibration(*cal)
        for v, c in enumerate(caps):
            g.set_mask(masks[v], view=v)
            g.set_frames(0, c["planes_v"], view=v)
            g.set_frames(1, c["planes_h"], view=v)
        # pipelined the way a caller would: compute view v+1 while view v's stripes travel
        for v in range(NV):
            g.run(v, 1)
            g.gather(v, 1)
        for rep in range(2):
            for v in range(NV):
                xyz, val = g.points(v)
                assert np.array_equal(val, ref[v][1]), (rep, v)
                assert np.array_equal(xyz, ref[v][0], equal_nan=True), (rep, v)
            g.run(0, NV)      # the whole batch in one launch per stripe, one exchange
            g.gather(0, NV)
        g.run_clouds(0, NV)
        counts = g.gather_clouds(0, NV)
        for v in range(NV):
            cl = g.cloud(v)
            assert counts[v] == len(cl) == int((ref[v][1] == 1).sum())
            assert np.array_equal(cl, ref[v][0][ref[v][1] == 1]), v
        g.synchronize()


def test_group_many_messages_rccl_path():
    """A gather of 40 views x 7 stripes x (xyz + valid) = 560 messages through the forced RCCL path: the exchange is cut into
    several RCCL groups of at most 256 send/recv pairs, in the same order on both sides."""
    S, syn = _S(), pkg("synth")
    W, H, PW, PH, N, fw, NV, NS = 64, 28, 128, 96, 6, 4, 40, 7
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=1)
    cal = syn.cal_tuple(cap["cal"])
    rng = np.random.default_rng(3)
    masks = [_random_mask(rng, W, H, holes=3) for _ in range(NV)]
    ref = _single_context_reference(W, H, PW, PH, N, fw, cal, masks, [cap] * NV)
    with S.Group(W, H, PW, PH, N, N, fw, fw, devices=[0] * NS, max_views=NV, flags=S.SL3D_FLAG_GROUP_FORCE_RCCL) as g:
        g.set_calibration(*cal)
        for v in range(NV):
            g.set_mask(masks[v], view=v)
            g.set_frames(0, cap["planes_v"], view=v)
            g.set_frames(1, cap["planes_h"], view=v)
        g.run(0, NV)
        g.gather(0, NV)
        for v in range(NV):
            xyz, val = g.points(v)
            assert np.array_equal(val, ref[v][1]) and np.array_equal(xyz, ref[v][0], equal_nan=True), v
        g.run_clouds(0, NV)
        counts = g.gather_clouds(0, NV)
        for v in range(NV):
            assert np.array_equal(g.cloud(v), ref[v][0][ref[v][1] == 1]) and counts[v] == int((ref[v][1] == 1).sum())


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


def _random_mask_round4(rng, W, H, p=0.1):
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
    masks = [caps[0]["mask"]] + [_random_mask_round4(rng, W, H) for _ in range(NV - 1)]
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


@pytest.fixture(scope="session")
def fake_rccl(tmp_path_factory):
    """tests/native/fake_rccl.cpp built next to the tests' temporary files (hipcc is part of the image, here and on the GPU box)."""
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "-O2", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"), "-o", out])
    return out


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
