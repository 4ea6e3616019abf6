"""GPU tests (-m gpu) of the C boundary itself: no C++ exception crosses it.  An LD_PRELOADed `operator new`
(tests/native/failing_new.cpp) throws std::bad_alloc at the N-th allocation made on behalf of libsl3d / libsl3d_shim, for N = 0, 1, 2 ...
until a call gets through with the charge unused: every allocation site of every entry point exercised here has then thrown once.  The
call must return SL3D_E_NOMEM (or succeed, if the failing allocation was not needed), never unwind into the C caller; the context must
stay usable, and a final scan must still equal the oracle."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

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

    def sweep(name, call, allowed=(OK, NOMEM), limit=200):
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
                assert rc == OK, (name, n, rc, "failed without an injected fault")
                break
            fired += 1
            assert rc == NOMEM or rc == OK, (name, n, rc)
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
    sweep("sl3d_set_mask(view out of range)", lambda: L.sl3d_set_mask(s._h, 7, mask.ctypes.data, mask.strides[0]), allowed=(-1, NOMEM, -7))

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
