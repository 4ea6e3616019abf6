"""ctypes front-end of the CPU oracle (oracle/sl3d_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under 3dscan_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsl3d_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "sl3d_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsl3d_oracle.so"])
    return _LIB


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("W", "H", "PW", "PH", "F", "N_v", "N_h", "fw_v", "fw_h", "ncodes_v", "ncodes_h", "exact_index", "col0", "row0", "pcol0", "prow0")]


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        # SL3D_ORACLE_LIB: another build of the same source, e.g. `make -C oracle libsl3d_oracle_asan.so` run as
        #   LD_PRELOAD=$(gcc -print-file-name=libasan.so) SL3D_ORACLE_LIB=oracle/libsl3d_oracle_asan.so pytest tests/test_oracle.py
        L = C.CDLL(os.environ.get("SL3D_ORACLE_LIB") or _LIB)
        vp = C.c_void_p
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(_Cfg)]
        L.orc_destroy.argtypes = [vp]
        L.orc_set_mask.argtypes = [vp, vp, C.c_size_t]
        L.orc_compute_wrapped_phase.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.orc_unwrap_phase.argtypes = [vp, C.c_int, vp, vp, C.c_size_t]
        L.orc_compute_c_p_map.argtypes = [vp]
        L.orc_set_calibration.argtypes = [vp] + [vp] * 8
        L.orc_prepare_triangulation.argtypes = [vp]
        L.orc_triangulate.argtypes = [vp]
        L.orc_save_point_cloud.restype = C.c_long
        L.orc_save_point_cloud.argtypes = [vp, vp, C.c_long]
        for n in ("orc_get_valid_map", "orc_get_wrapped_phi", "orc_get_unwrapped_phi", "orc_get_code"):
            getattr(L, n).argtypes = [vp, C.c_int, vp]
        L.orc_get_c_p_map.argtypes = [vp, vp]
        L.orc_get_intersection_points.argtypes = [vp, vp]
        L.orc_get_debug_image.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_get_projection_matrices.argtypes = [vp, vp, vp]
        L.orc_get_undist_point.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_relative_geometry.argtypes = [vp] * 6
        L.orc_run_scan.argtypes = [vp, vp, vp, C.c_size_t]
        L.orc_run_scan_rowmajor.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp, vp]
        L.orc_undistort.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t]
        L.orc_pattern_counts.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_pattern_profile.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.orc_pattern_image.argtypes = [C.c_int] * 8 + [vp]
        L.orc_register_point_clouds.argtypes = [C.c_uint, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]
        _lib = L
    return _lib


def _planes(arrs):
    """list of 2-D uint8 row-major arrays (same stride) -> (pointer array, stride, keepalive)"""
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in arrs]
    stride = arrs[0].strides[0] if arrs else 0
    ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    return ptrs, stride, arrs


class Oracle:
    """One scan's worth of reference state (the globals of PROJECT_GLOBAL/common_variables.h)."""

    def __init__(self, W, H, PW, PH, N_v, N_h, fw_v, fw_h, F=3, ncodes_v=None, ncodes_h=None, exact_index=False,
                 col0=0, row0=0):
        L = _load()
        self.cfg = _Cfg(W, H, PW, PH, F, N_v, N_h, fw_v, fw_h,
                        ncodes_v if ncodes_v is not None else -(-PW // fw_v),
                        ncodes_h if ncodes_h is not None else -(-PH // fw_h), int(exact_index),
                        col0, row0, 0, 0)
        self.W, self.H = W, H
        self._s = L.orc_create(C.byref(self.cfg))
        if not self._s:
            raise MemoryError("orc_create failed")

    def close(self):
        if self._s:
            _load().orc_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs -------------------------------------------------------------
    def set_mask(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape == (self.H, self.W)
        _load().orc_set_mask(self._s, m.ctypes.data, m.strides[0])

    def set_calibration(self, Kc, dc, rc, tc, Kp, dp, rp, tp):
        a = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel()) for x in (Kc, dc, rc, tc, Kp, dp, rp, tp)]
        assert [x.size for x in a] == [9, 5, 3, 3, 9, 5, 3, 3]
        self._cal = a
        _load().orc_set_calibration(self._s, *[x.ctypes.data for x in a])

    # -- the four reference entry points -------------------------------------
    def compute_wrapped_phase(self, pattern_type, fringe):
        p, st, keep = _planes(fringe)
        _load().orc_compute_wrapped_phase(self._s, pattern_type, p, st)

    def unwrap_phase(self, pattern_type, gray, inv):
        pg, st, k1 = _planes(gray)
        pi, st2, k2 = _planes(inv)
        assert st == st2 or not k2
        _load().orc_unwrap_phase(self._s, pattern_type, pg, pi, st)

    def compute_c_p_map(self):
        _load().orc_compute_c_p_map(self._s)

    def triangulate(self):
        _load().orc_triangulate(self._s)

    def run_scan(self, planes_v, planes_h):
        """main()'s stage order on F + N + N planes per axis (fringe, gray, inverse gray)."""
        pv, st, k1 = _planes(planes_v)
        ph, st2, k2 = _planes(planes_h)
        assert st == st2
        _load().orc_run_scan(self._s, pv, ph, st)

    def run_scan_rowmajor(self, planes_v, planes_h, threads=0, out=None):
        """CPU baseline (b) of SURVEY 8d: the same maths fused, row-major, OpenMP over rows.
        Returns (xyz float32 [H,W,3] with NaN where invalid, valid uint8 [H,W], threads used)."""
        pv, st, k1 = _planes(planes_v)
        ph, st2, k2 = _planes(planes_h)
        assert st == st2
        W, H = self.W, self.H
        # out = (xyz, valid) of an earlier call: reused, so that a timing loop does not measure first-touch page faults
        xyz, valid = out if out is not None else (np.empty((H, W, 3), dtype=np.float32), np.empty((H, W), dtype=np.uint8))
        n = _load().orc_run_scan_rowmajor(self._s, pv, ph, st, int(threads), xyz.ctypes.data, valid.ctypes.data)
        if n < 0:
            raise ValueError("orc_run_scan_rowmajor: 3 or 4 fringe patterns only")
        return xyz, valid, n

    def invalidate_tables(self):
        """Force stage 7 to redo assign_3d_coordinates, as the reference does on every triangulate()."""
        _load().orc_set_calibration(self._s, *[x.ctypes.data for x in self._cal])

    # -- outputs (row-major) ---------------------------------------------------
    def _get(self, fn, which, dtype):
        out = np.empty((self.H, self.W), dtype=dtype)
        getattr(_load(), fn)(self._s, which, out.ctypes.data)
        return out

    def valid_map(self, which):
        return self._get("orc_get_valid_map", which, np.uint8)

    def wrapped_phi(self, which):
        return self._get("orc_get_wrapped_phi", which, np.float32)

    def unwrapped_phi(self, which):
        return self._get("orc_get_unwrapped_phi", which, np.float32)

    def code(self, which):
        return self._get("orc_get_code", which, np.int32)

    def c_p_map(self):
        out = np.empty((self.H, self.W, 2), dtype=np.int64)
        _load().orc_get_c_p_map(self._s, out.ctypes.data)
        return out

    def intersection_points(self):
        out = np.empty((self.H, self.W, 3), dtype=np.float64)
        _load().orc_get_intersection_points(self._s, out.ctypes.data)
        return out

    def debug_image(self, stage, which):
        out = np.empty((self.H, self.W), dtype=np.uint8)
        _load().orc_get_debug_image(self._s, stage, which, out.ctypes.data)
        return out

    def projection_matrices(self):
        A = np.empty((3, 4)); B = np.empty((3, 4))
        _load().orc_get_projection_matrices(self._s, A.ctypes.data, B.ctypes.data)
        return A, B

    def undist_point(self, dev, col, row):
        uv = np.empty(2)
        _load().orc_get_undist_point(self._s, dev, col, row, uv.ctypes.data)
        return uv

    def point_cloud(self):
        L = _load()
        n = L.orc_save_point_cloud(self._s, None, 0)
        out = np.empty((n, 3), dtype=np.float32)
        L.orc_save_point_cloud(self._s, out.ctypes.data, n)
        return out


def relative_geometry(rc, tc, rp, tp):
    """Rc * Rp^T (3x3) and tc - (Rc * Rp^T) * tp (3,) through the oracle's cvRodrigues2 / cvTranspose / cvGEMM restatements, in
    the order 6/system_calibration.cpp:1488-1503 applies them (the reference saved OpenCV's own answer: the T0 known answer)."""
    a = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel()) for x in (rc, tc, rp, tp)]
    assert all(x.size == 3 for x in a)
    R, t = np.zeros(9), np.zeros(3)
    _load().orc_relative_geometry(*[x.ctypes.data for x in a], R.ctypes.data, t.ctypes.data)
    return R.reshape(3, 3), t


def pattern_counts(proj_extent, fringe_width):
    """1/pattern_generator.cpp:224-229 -> (number of codes, number of bit planes)."""
    L = _load()
    nc, npl = C.c_int(), C.c_int()
    L.orc_pattern_counts(int(proj_extent), int(fringe_width), C.byref(nc), C.byref(npl))
    return nc.value, npl.value


PATTERN_FRINGE, PATTERN_GRAY, PATTERN_INVERSE_GRAY, PATTERN_BINARY = 0, 1, 2, 3


def pattern_profile(kind, index, extent, fringe_width, nplanes, F=3):
    """The values of a projector pattern along its varying axis (1/pattern_generator.cpp; see orc_pattern_profile)."""
    L = _load()
    out = np.zeros(extent, dtype=np.uint8)
    L.orc_pattern_profile(int(kind), int(F), int(index), int(extent), int(fringe_width), int(nplanes), out.ctypes.data)
    return out


def pattern_image(kind, axis, index, PW, PH, fringe_width, nplanes, F=3):
    L = _load()
    out = np.zeros((PH, PW), dtype=np.uint8)
    L.orc_pattern_image(int(kind), int(axis), int(F), int(index), int(PW), int(PH), int(fringe_width), int(nplanes), out.ctypes.data)
    return out


def undistort(src, K, dist):
    """cvUndistort2(src, dst, K, dist) on an (H, W) or (H, W, 3) uint8 image (N4; parity unpinned, see orc_undistort)."""
    L = _load()
    a = np.ascontiguousarray(src, dtype=np.uint8)
    cn = 1 if a.ndim == 2 else a.shape[2]
    H, W = a.shape[:2]
    out = np.zeros_like(a)
    Kd = np.ascontiguousarray(np.asarray(K, dtype=np.float64).ravel())
    dd = np.ascontiguousarray(np.asarray(dist, dtype=np.float64).ravel())
    L.orc_undistort(a.ctypes.data, a.strides[0], W, H, cn, Kd.ctypes.data, dd.ctypes.data, out.ctypes.data, out.strides[0])
    return out


def register_point_clouds(clouds, tx, ty, tz, rot_step):
    """9/register_point_clouds.cpp on in-memory clouds: list of (n_i,3) float32 arrays -> concatenated (sum n_i,3)."""
    L = _load()
    arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    counts = (C.c_long * len(arrs))(*[len(a) for a in arrs])
    out = np.empty((sum(len(a) for a in arrs), 3), dtype=np.float32)
    L.orc_register_point_clouds(len(arrs), ptrs, counts, tx, ty, tz, rot_step, out.ctypes.data)
    return out
