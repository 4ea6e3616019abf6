"""GPU tests (-m gpu) of the C boundary itself: error returns, two host threads on two contexts, host-buffer hand-over
(pipelined views, one copy per axis), the reference's own [col][row] layouts in and out -- and: no C++ exception crosses it.  An LD_PRELOADed `operator new`
(tests/native/failing_new.cpp) throws std::bad_alloc at the N-th allocation made on behalf of libsl3d / libsl3d_shim, for N = 0, 1, 2 ...
until a call gets through with the charge unused: every allocation site of every entry point exercised here has then thrown once.  The
call must return SL3D_E_NOMEM (or succeed, if the failing allocation was not needed), never unwind into the C caller; the context must
stay usable, and a final scan must still equal the oracle."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, assert_points_close, pkg
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

DRIVER = textwrap.dedent('''
    import ctypes as C, importlib, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["SL3D_ROOT"])
    fn = C.CDLL(os.environ["SL3D_FAILING_NEW"])           # already in the process (LD_PRELOAD): the handle to arm it
    fn.failing_new_arm.argtypes = [C.c_long]
    fn.failing_new_fired.restype = C.c_long
    fn.failing_new_left.restype = C.c_long
    sc = importlib.import_module("3dscan_amd.scanner")
    syn = importlib.import_module("3dscan_amd.synth")
    from oracle.oracle import Oracle
    L = sc.load_library()
    OK, NOMEM = 0, -6
    W, H, PW, PH, N, fw = 192, 64, 256, 192, 6, 8
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=1)
    cal = syn.cal_tuple(cap["cal"])
    report = {}

    def sweep(name, call, allowed=(OK, NOMEM), limit=200, clean=OK):
        """call() with the N-th allocation failing, N = 0 ... until the charge stays unused"""
        fired = 0
        for n in range(limit):
            fn.failing_new_arm(n)
            before = fn.failing_new_fired()
            rc = call()
            left = fn.failing_new_left()
            fn.failing_new_arm(-1)
            assert rc in allowed, (name, n, rc)
            if fn.failing_new_fired() == before:
                assert rc == clean, (name, n, rc, "unexpected status without an injected fault")
                break
            fired += 1
            assert rc == NOMEM or rc == clean, (name, n, rc)
        else:
            raise AssertionError(name + ": still allocating after %d injected faults" % limit)
        report[name] = fired

    # ---- sl3d_create / sl3d_destroy: a context whose construction is interrupted anywhere leaves nothing behind
    cfg = sc.Config(W, H, W, H, 0, 0, PW, PH, 3, N, N, fw, fw, 0, 0, 2, 0, 0, None)
    h = C.c_void_p()
    def create():
        rc = L.sl3d_create(C.byref(cfg), C.byref(h))
        if rc == OK:
            L.sl3d_destroy(h)
        else:
            assert not h.value, "a failed sl3d_create handed out a context"
        return rc
    sweep("sl3d_create", create)

    s = sc.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=2)
    a = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()) for v in cal]
    sweep("sl3d_set_calibration", lambda: L.sl3d_set_calibration(s._h, *[v.ctypes.data for v in a]))
    mask = np.ascontiguousarray(cap["mask"], dtype=np.uint8)
    sweep("sl3d_set_mask", lambda: L.sl3d_set_mask(s._h, 0, mask.ctypes.data, mask.strides[0]))
    for axis, planes in ((0, cap["planes_v"]), (1, cap["planes_h"])):
        arrs = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        ptrs = (C.c_void_p * len(arrs))(*[p.ctypes.data for p in arrs])
        sweep("sl3d_set_frames[%d]" % axis, lambda: L.sl3d_set_frames(s._h, 0, axis, ptrs, len(arrs), arrs[0].strides[0]))
    sweep("sl3d_run", lambda: L.sl3d_run(s._h, 0, 1))
    sweep("sl3d_run_clouds", lambda: L.sl3d_run_clouds(s._h, 0, 1))
    xyz = np.zeros((H, W, 3), np.float32); valid = np.zeros((H, W), np.uint8)
    sweep("sl3d_get_points", lambda: L.sl3d_get_points(s._h, 0, xyz.ctypes.data, valid.ctypes.data))
    cloud = np.zeros((W * H, 3), np.float32); cnt = (C.c_int64 * 1)()
    sweep("sl3d_download_clouds", lambda: L.sl3d_download_clouds(s._h, 0, 1, cloud.ctypes.data, W * H, cnt))
    sweep("sl3d_get_cloud", lambda: L.sl3d_get_cloud(s._h, 0, cloud.ctypes.data, W * H, cnt))
    total = (C.c_int64 * 1)()
    sweep("sl3d_register_clouds", lambda: L.sl3d_register_clouds(s._h, 0, 1, C.c_float(1), C.c_float(2), C.c_float(3), C.c_float(7.5), cloud.ctypes.data, W * H, total))
    name = C.create_string_buffer(256)
    sweep("sl3d_last_fused_kernel_name", lambda: L.sl3d_last_fused_kernel_name(s._h, name, 256))
    bad = C.c_void_p()
    sweep("sl3d_set_mask(view out of range)", lambda: L.sl3d_set_mask(s._h, 7, mask.ctypes.data, mask.strides[0]), allowed=(-1, NOMEM), clean=-1)   # the error text allocates too

    # ---- a group of two stripes on this GPU
    devs = (C.c_int * 2)(0, 0)
    g = C.c_void_p()
    gcfg = sc.Config(W, H, W, H, 0, 0, PW, PH, 3, N, N, fw, fw, 0, 0, 1, 0, 4, None)   # SL3D_FLAG_GROUP_NO_RCCL
    def gcreate():
        rc = L.sl3d_group_create(C.byref(gcfg), devs, 2, C.byref(g))
        if rc == OK:
            L.sl3d_group_destroy(g)
        else:
            assert not g.value
        return rc
    sweep("sl3d_group_create", gcreate)

    # ---- and after all that the context still computes the right thing
    s.set_calibration(*cal); s.set_mask(cap["mask"]); s.set_frames(0, cap["planes_v"]); s.set_frames(1, cap["planes_h"])
    s.run()
    xyz, valid = s.points()
    o = Oracle(W, H, PW, PH, N, N, fw, fw)
    o.set_mask(cap["mask"]); o.set_calibration(*cal); o.run_scan(cap["planes_v"], cap["planes_h"])
    v = o.valid_map(2) == 1
    assert np.array_equal(valid == 1, v)
    ref = o.intersection_points()[v]
    assert (np.linalg.norm(xyz[v] - ref, axis=-1) <= 1e-5 * np.linalg.norm(ref, axis=-1)).all()
    s.close()
    print("REPORT", report)
''')


def test_no_exception_crosses_the_c_abi(tmp_path):
    lib = str(tmp_path / "libfailing_new.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "failing_new.cpp"), "-o", lib, "-ldl"])
    script = tmp_path / "driver.py"
    script.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD=lib, SL3D_FAILING_NEW=lib, SL3D_ROOT=ROOT)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("REPORT")]
    assert line, out.stdout[-2000:]
    report = eval(line[0][len("REPORT "):])
    # the sweep did inject faults where the library allocates on the host
    assert report["sl3d_create"] >= 3 and report["sl3d_group_create"] >= 3, report
    assert sum(report.values()) >= 15, report


def _scanner():
    return pkg("scanner")


@pytest.mark.parametrize("W", [200, 224])  # 224: width == device pitch, a view goes up as one copy; 200: plane by plane
def test_host_buffer_pipeline_matches_resident_path(W):
    """sl3d_process_views: 7 host-resident views through 3 view slots (upload / kernel / download on three streams) give
    exactly what the resident path gives view by view, with pinned and with pageable host memory, and a 1-slot context
    degenerates to the serial order."""
    syn = pkg("synth")
    S = _scanner()
    H, PW, PH, N, fw, NV = 120, 256, 192, 6, 8, 7
    caps = [syn.make_capture(W, H, PW, PH, N, 5, fw, fw, plane=(2.0 * v, 0.05 - 0.004 * v, 0.04), view=v, noise=2) for v in range(NV)]
    stack = np.stack([np.stack(c["planes_v"] + c["planes_h"]) for c in caps])
    cal = syn.cal_tuple(caps[0]["cal"])
    mask = caps[0]["mask"].copy()
    mask[20:40, 50:80] = 0
    ref = []
    with S.Scanner(W, H, PW, PH, N, 5, fw, fw) as sc:
        sc.set_calibration(*cal)
        sc.set_mask(mask)
        for c in caps:
            sc.set_frames(0, c["planes_v"])
            sc.set_frames(1, c["planes_h"])
            sc.run()
            ref.append(sc.points())
    for slots, pin in ((3, True), (3, False), (1, True)):
        with S.Scanner(W, H, PW, PH, N, 5, fw, fw, max_views=slots) as sc:
            sc.set_calibration(*cal)
            for s_ in range(slots):
                sc.set_mask(mask, view=s_)
            frames = sc.pinned(stack.shape, np.uint8) if pin else stack.copy()
            frames[...] = stack
            xyz = sc.pinned((NV, H, W, 3), np.float32) if pin else None
            for _ in range(2):  # the second batch reuses slots that still hold the first one's results
                out_xyz, out_valid = sc.process_views(frames, xyz=xyz)
                for v in range(NV):
                    assert np.array_equal(out_valid[v], ref[v][1]), (slots, pin, v)
                    assert np.array_equal(out_xyz[v], ref[v][0], equal_nan=True), (slots, pin, v)


def _oracle_for(cap, W, H, PW, PH, Nv, Nh, fwv, fwh, mask, F=3):
    o = Oracle(W, H, PW, PH, Nv, Nh, fwv, fwh, F=F)
    o.set_mask(mask)
    o.set_calibration(*pkg("synth").cal_tuple(cap["cal"]))
    o.run_scan(cap["planes_v"], cap["planes_h"])
    return o


def test_two_contexts_are_independent():
    """SURVEY 8b threading row: contexts are independent of each other.  Two contexts of different shapes, rigs and
    modes on the same GPU with their calls interleaved give what each gives alone."""
    syn = pkg("synth")
    S = _scanner()
    A = dict(W=320, H=200, PW=512, PH=384, N=7, fw=4)
    B = dict(W=200, H=120, PW=256, PH=192, N=6, fw=8)
    capA = syn.make_capture(A["W"], A["H"], A["PW"], A["PH"], A["N"], A["N"], A["fw"], A["fw"], noise=1)
    capB = syn.make_capture(B["W"], B["H"], B["PW"], B["PH"], B["N"], 5, B["fw"], B["fw"], noise=2, plane=(5.0, 0.02, 0.07))
    calB = {k: np.array(v, dtype=np.float64).copy() for k, v in capB["cal"].items()}
    calB["dp"] = np.array([0.04, -0.01, 0.0, 0.0, 0.0])  # context B: distorted projector (rig class 2)
    oA = _oracle_for(capA, A["W"], A["H"], A["PW"], A["PH"], A["N"], A["N"], A["fw"], A["fw"], capA["mask"])
    oB = _oracle_for(dict(capB, cal=calB), B["W"], B["H"], B["PW"], B["PH"], B["N"], 5, B["fw"], B["fw"], capB["mask"])
    with S.Scanner(A["W"], A["H"], A["PW"], A["PH"], A["N"], A["N"], A["fw"], A["fw"]) as a, \
         S.Scanner(B["W"], B["H"], B["PW"], B["PH"], B["N"], 5, B["fw"], B["fw"], keep_stages=True) as b:
        a.set_calibration(*syn.cal_tuple(capA["cal"]))
        b.set_calibration(*syn.cal_tuple(calB))
        b.set_mask(capB["mask"])
        a.set_mask(capA["mask"])
        a.set_frames(0, capA["planes_v"])
        b.set_frames(0, capB["planes_v"])
        b.set_frames(1, capB["planes_h"])
        a.set_frames(1, capA["planes_h"])
        for _ in range(3):
            a.run()
            b.run_stages()
            b.run()
            a.run()
        xa, va = a.points()
        xb, vb = b.points()
        assert np.array_equal(va == 1, oA.valid_map(2) == 1) and np.array_equal(vb == 1, oB.valid_map(2) == 1)
        assert_points_close(xa, oA.intersection_points(), oA.valid_map(2) == 1)
        assert_points_close(xb, oB.intersection_points(), oB.valid_map(2) == 1)
        assert np.array_equal(b.code(0), oB.code(0)) and np.array_equal(b.c_p_map()[vb == 1], oB.c_p_map()[vb == 1])


def test_api_errors():
    """Error behaviour of the C ABI on a live context: bad arguments and call-order violations are reported, not executed."""
    S = _scanner()
    with S.Scanner(64, 32, 128, 128, 5, 5, 4, 4) as sc:  # no KEEP_STAGES, no calibration yet
        with pytest.raises(S.Sl3dError, match="call order"):
            sc.run()
        with pytest.raises(S.Sl3dError, match="call order"):
            sc.compute_wrapped_phase(0)
        with pytest.raises(S.Sl3dError, match="invalid argument"):
            sc.set_frames(0, [np.zeros((32, 64), np.uint8)] * 5)  # needs 3 + 2*5 planes
        with pytest.raises(S.Sl3dError, match="invalid argument"):
            sc.run(0, 2)  # max_views is 1
        with pytest.raises(S.Sl3dError):
            sc.wrapped_phase(0)  # stage planes were not requested
    with pytest.raises(S.Sl3dError, match="invalid argument"):
        S.Scanner(64, 32, 128, 128, 5, 5, 4, 4, full_size=(32, 32))  # window larger than the frame
    with pytest.raises(S.Sl3dError, match="unsupported"):
        S.Scanner(64, 32, 128, 128, 17, 5, 4, 4)


def _S():
    return pkg("scanner")


def test_set_frames_one_copy_per_axis_and_download():
    """Planes that follow each other in host memory go up as ONE 2-D copy per axis (pageable and pinned sources); the result
    equals plane-by-plane uploads.  sl3d_download returns device addresses the library hands out."""
    S, syn = _S(), pkg("synth")
    W, H, PW, PH, N, fw = 203, 77, 256, 192, 6, 8     # width != device pitch: the copy really is 2-D
    cap = syn.make_capture(W, H, PW, PH, N, 5, fw, fw, noise=2)
    cal = syn.cal_tuple(cap["cal"])
    with S.Scanner(W, H, PW, PH, N, 5, fw, fw, max_views=3) as sc:
        sc.set_calibration(*cal)
        stack_v, stack_h = np.stack(cap["planes_v"]), np.stack(cap["planes_h"])
        pin_v, pin_h = sc.pinned(stack_v.shape, np.uint8), sc.pinned(stack_h.shape, np.uint8)
        pin_v[:], pin_h[:] = stack_v, stack_h
        for v, (pv, ph) in enumerate(((cap["planes_v"], cap["planes_h"]), (list(stack_v), list(stack_h)), (list(pin_v), list(pin_h)))):
            sc.set_mask(cap["mask"], view=v)
            sc.set_frames(0, pv, view=v)
            sc.set_frames(1, ph, view=v)
        sc.run(0, 3)
        ref = sc.points(0)
        for v in (1, 2):
            got = sc.points(v)
            assert np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0], equal_nan=True), v
            for a in (0, 1):
                for x, y in zip(sc.frames(a, v), sc.frames(a, 0)):
                    assert np.array_equal(x, y)
        b = sc.device_buffers()
        row = np.empty(W, dtype=np.uint8)
        sc._d2h(row, b.valid + 5 * b.valid_pitch)
        assert np.array_equal(row, ref[1][5])


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
