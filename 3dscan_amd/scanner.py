"""ctypes binding of the C ABI (include/sl3d.h) -- the Python face of the product library.

There is no fallback: if libsl3d.so is missing or no HIP device is present, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SL3D_LIB: another build of the SAME library (kernel A/B experiments, tools/ab.sh); never a fallback
LIB_PATH = os.environ.get("SL3D_LIB") or os.path.join(_HERE, "libsl3d.so")

SL3D_FLAG_KEEP_STAGES = 1
SL3D_FLAG_GROUP_FORCE_RCCL, SL3D_FLAG_GROUP_NO_RCCL, SL3D_FLAG_GROUP_DISTINCT_SIDES = 2, 4, 16
SL3D_FLAG_EAGER_MASK = 32
SL3D_FLAG_SERIAL_LAUNCHES = 64
AXIS_VERTICAL, AXIS_HORIZONTAL = 0, 1
PATTERN_FRINGE, PATTERN_GRAY, PATTERN_INVERSE_GRAY, PATTERN_BINARY = 0, 1, 2, 3
VALID_VERTICAL, VALID_HORIZONTAL, VALID_MERGED = 0, 1, 2

# every symbol include/sl3d.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "sl3d_version", "sl3d_strerror", "sl3d_last_error", "sl3d_create", "sl3d_destroy",
    "sl3d_set_calibration", "sl3d_get_projection_matrices", "sl3d_set_mask", "sl3d_set_masks", "sl3d_set_mask_colrow", "sl3d_set_frames_range", "sl3d_get_global_colrow", "sl3d_set_frames", "sl3d_copy_view", "sl3d_synth_view", "sl3d_get_frames",
    "sl3d_compute_wrapped_phase", "sl3d_unwrap_phase", "sl3d_compute_c_p_map", "sl3d_triangulate",
    "sl3d_run", "sl3d_run_clouds", "sl3d_get_cloud_counts", "sl3d_get_cloud_segments", "sl3d_download_clouds", "sl3d_register_clouds", "sl3d_fused_kernel_name", "sl3d_last_fused_kernel_name", "sl3d_launch_counts", "sl3d_camera_table_bytes_per_pixel", "sl3d_run_timed", "sl3d_synchronize", "sl3d_timer_start", "sl3d_timer_stop",
    "sl3d_get_valid_map", "sl3d_get_wrapped_phase", "sl3d_get_unwrapped_phase", "sl3d_get_code",
    "sl3d_get_debug_image", "sl3d_get_c_p_map", "sl3d_get_intersection_points", "sl3d_get_points",
    "sl3d_get_cloud", "sl3d_set_texture", "sl3d_get_cloud_rgb", "sl3d_compact", "sl3d_compact_views", "sl3d_get_clouds", "sl3d_register_views", "sl3d_transform_cloud", "sl3d_host_alloc", "sl3d_host_free", "sl3d_process_views", "sl3d_undistort", "sl3d_set_frames_raw", "sl3d_pattern_counts", "sl3d_generate_pattern",
    "sl3d_get_device_buffers", "sl3d_download", "sl3d_download_2d",
    "sl3d_group_create", "sl3d_group_destroy", "sl3d_group_last_error", "sl3d_group_size", "sl3d_group_stripe", "sl3d_group_transport",
    "sl3d_group_set_calibration", "sl3d_group_set_mask", "sl3d_group_set_frames", "sl3d_group_run", "sl3d_group_gather",
    "sl3d_group_get_points", "sl3d_group_download_points", "sl3d_group_process_views", "sl3d_group_get_device_buffers", "sl3d_group_run_clouds", "sl3d_group_gather_clouds",
    "sl3d_group_get_cloud", "sl3d_group_synchronize",
)


class Sl3dError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "full_width", "full_height", "col0", "row0", "proj_width", "proj_height",
        "n_fringe", "n_gray_v", "n_gray_h", "fringe_width_v", "fringe_width_h", "n_codes_v", "n_codes_h",
        "max_views", "device")] + [("flags", C.c_uint32), ("stream", C.c_void_p)]


class DeviceBuffers(C.Structure):
    _fields_ = [
        ("frames", C.c_void_p), ("frame_pitch", C.c_size_t), ("plane_stride", C.c_size_t), ("view_stride", C.c_size_t),
        ("planes_per_view", C.c_int32),
        ("mask", C.c_void_p), ("mask_pitch", C.c_size_t), ("mask_view_stride", C.c_size_t),
        ("points", C.c_void_p), ("points_pitch", C.c_size_t), ("points_view_stride", C.c_size_t),
        ("valid", C.c_void_p), ("valid_pitch", C.c_size_t), ("valid_view_stride", C.c_size_t),
    ]


class CloudSegments(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("counts", C.c_void_p), ("offsets", C.c_void_p), ("n_segments", C.c_int32), ("segment_points", C.c_int32),
                ("view_stride_points", C.c_size_t), ("view_stride_segments", C.c_size_t)]


_lib = None


def load_library(path=None):
    """dlopen the product library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise Sl3dError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C 3dscan_amd/csrc` (there is no CPU fallback)")
    L = C.CDLL(p)
    vp, i = C.c_void_p, C.c_int
    L.sl3d_version.restype = C.c_char_p
    L.sl3d_strerror.restype = C.c_char_p
    L.sl3d_strerror.argtypes = [i]
    L.sl3d_last_error.restype = C.c_char_p
    L.sl3d_last_error.argtypes = [vp]
    L.sl3d_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.sl3d_destroy.argtypes = [vp]
    L.sl3d_destroy.restype = None
    L.sl3d_set_calibration.argtypes = [vp] + [vp] * 8
    L.sl3d_get_projection_matrices.argtypes = [vp, vp, vp]
    L.sl3d_set_mask.argtypes = [vp, i, vp, C.c_size_t]
    L.sl3d_set_frames.argtypes = [vp, i, i, vp, i, C.c_size_t]
    try:
        L.sl3d_set_masks.argtypes = [vp, i, i, vp, C.c_size_t, C.c_size_t]
        L.sl3d_last_fused_kernel_name.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.sl3d_camera_table_bytes_per_pixel.argtypes = [vp, i]
    except AttributeError:   # an older build of the library under SL3D_LIB (A/B runs against a previous round)
        if not os.environ.get("SL3D_LIB"):
            raise
    L.sl3d_set_mask_colrow.argtypes = [vp, i, vp]
    L.sl3d_set_frames_range.argtypes = [vp, i, i, i, vp, i, C.c_size_t]
    L.sl3d_get_global_colrow.argtypes = [vp, i, i, vp, i, i]
    for n in ("sl3d_compute_wrapped_phase", "sl3d_unwrap_phase", "sl3d_copy_view"):
        getattr(L, n).argtypes = [vp, i, i]
    for n in ("sl3d_compute_c_p_map", "sl3d_triangulate"):
        getattr(L, n).argtypes = [vp, i]
    L.sl3d_synth_view.argtypes = [vp, i, vp, C.c_uint64, i, i, C.c_float, C.c_float]
    L.sl3d_get_frames.argtypes = [vp, i, i, vp, i, C.c_size_t]
    L.sl3d_run.argtypes = [vp, i, i]
    L.sl3d_run_clouds.argtypes = [vp, i, i]
    L.sl3d_get_cloud_counts.argtypes = [vp, i, i, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    L.sl3d_get_cloud_segments.argtypes = [vp, i, i, C.POINTER(CloudSegments), C.POINTER(C.c_int64)]
    L.sl3d_download_clouds.argtypes = [vp, i, i, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sl3d_register_clouds.argtypes = [vp, i, i, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sl3d_run_timed.argtypes = [vp, i, i, C.POINTER(C.c_float)]
    try:
        L.sl3d_fused_kernel_name.argtypes = [vp, i, i, C.c_char_p, C.c_size_t]
    except AttributeError:   # an older build of the library under SL3D_LIB (A/B runs against a previous round): the name is cosmetic
        if not os.environ.get("SL3D_LIB"):
            raise
    L.sl3d_synchronize.argtypes = [vp]
    L.sl3d_timer_start.argtypes = [vp]
    L.sl3d_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.sl3d_get_valid_map.argtypes = [vp, i, i, vp, C.c_size_t]
    for n in ("sl3d_get_wrapped_phase", "sl3d_get_unwrapped_phase", "sl3d_get_code"):
        getattr(L, n).argtypes = [vp, i, i, vp, C.c_size_t]
    L.sl3d_get_debug_image.argtypes = [vp, i, i, i, vp, C.c_size_t]
    L.sl3d_get_c_p_map.argtypes = [vp, i, vp]
    L.sl3d_get_intersection_points.argtypes = [vp, i, vp]
    L.sl3d_get_points.argtypes = [vp, i, vp, vp]
    L.sl3d_get_cloud.argtypes = [vp, i, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sl3d_compact_views.argtypes = [vp, i, i, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    L.sl3d_get_clouds.argtypes = [vp, i, i, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sl3d_set_texture.argtypes = [vp, i, vp, C.c_size_t]
    L.sl3d_get_cloud_rgb.argtypes = [vp, i, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sl3d_compact.argtypes = [vp, i, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.sl3d_register_views.argtypes = [vp, i, i, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sl3d_host_alloc.restype = vp
    L.sl3d_host_alloc.argtypes = [C.c_size_t]
    L.sl3d_host_free.restype = None
    L.sl3d_host_free.argtypes = [vp]
    L.sl3d_process_views.argtypes = [vp, i, vp, C.c_size_t, vp, vp]
    L.sl3d_set_frames_raw.argtypes = [vp, i, i, vp, i, C.c_size_t]
    L.sl3d_undistort.argtypes = [vp, vp, C.c_size_t, i, i, i, vp, vp, vp, C.c_size_t]
    L.sl3d_pattern_counts.argtypes = [i, i, C.POINTER(i), C.POINTER(i)]
    L.sl3d_generate_pattern.argtypes = [vp, i, i, i, vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.sl3d_transform_cloud.argtypes = [vp, vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    L.sl3d_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.sl3d_download_2d.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]
    L.sl3d_get_device_buffers.argtypes = [vp, C.POINTER(DeviceBuffers)]
    L.sl3d_group_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), i, C.POINTER(vp)]
    L.sl3d_group_destroy.argtypes = [vp]
    L.sl3d_group_destroy.restype = None
    L.sl3d_group_last_error.restype = C.c_char_p
    L.sl3d_group_last_error.argtypes = [vp]
    L.sl3d_group_transport.restype = C.c_char_p
    L.sl3d_group_transport.argtypes = [vp]
    L.sl3d_group_size.argtypes = [vp]
    L.sl3d_group_stripe.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(vp)]
    L.sl3d_group_set_calibration.argtypes = [vp] + [vp] * 8
    L.sl3d_group_set_mask.argtypes = [vp, i, vp, C.c_size_t]
    L.sl3d_group_set_frames.argtypes = [vp, i, i, vp, i, C.c_size_t]
    for n in ("sl3d_group_run", "sl3d_group_gather", "sl3d_group_run_clouds"):
        getattr(L, n).argtypes = [vp, i, i]
    L.sl3d_group_gather_clouds.argtypes = [vp, i, i, C.POINTER(C.c_int64)]
    L.sl3d_group_get_points.argtypes = [vp, i, vp, vp]
    L.sl3d_group_download_points.argtypes = [vp, i, i, vp, vp]
    L.sl3d_group_process_views.argtypes = [vp, i, vp, C.c_size_t, vp, vp]
    L.sl3d_group_get_device_buffers.argtypes = [vp, C.POINTER(DeviceBuffers)]
    L.sl3d_group_get_cloud.argtypes = [vp, i, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sl3d_group_synchronize.argtypes = [vp]
    if path is None:
        _lib = L
    return L


def pattern_counts(proj_extent, fringe_width):
    """(number of codes, number of bit planes) of 1/pattern_generator.cpp:224-229."""
    L = load_library()
    nc, npl = C.c_int(), C.c_int()
    rc = L.sl3d_pattern_counts(proj_extent, fringe_width, C.byref(nc), C.byref(npl))
    if rc != 0:
        raise Sl3dError("sl3d_pattern_counts: " + L.sl3d_strerror(rc).decode())
    return nc.value, npl.value


class Scanner:
    """One context = one GPU + one stream + the HBM-resident frame stacks of `max_views` views."""

    def __init__(self, width, height, proj_width, proj_height, n_gray_v, n_gray_h, fringe_width_v, fringe_width_h,
                 n_fringe=3, n_codes_v=0, n_codes_h=0, max_views=1, device=0, keep_stages=False,
                 full_size=None, origin=(0, 0), stream=None, eager_mask=False, serial_launches=False):
        """eager_mask: SL3D_FLAG_EAGER_MASK -- every mask is prepared by k_mask_prepare when it is set; by default a timed context
        defers masks of up to 4 views to the next launch over them (the fused kernel evaluates the selection itself)."""
        self.L = load_library()
        fw, fh = full_size if full_size else (width, height)
        self.cfg = Config(width, height, fw, fh, origin[0], origin[1], proj_width, proj_height, n_fringe,
                          n_gray_v, n_gray_h, fringe_width_v, fringe_width_h, n_codes_v, n_codes_h,
                          max_views, device, (SL3D_FLAG_KEEP_STAGES if keep_stages else 0) | (SL3D_FLAG_EAGER_MASK if eager_mask else 0) |
                          (SL3D_FLAG_SERIAL_LAUNCHES if serial_launches else 0), stream)
        self.W, self.H = width, height
        self._h = C.c_void_p()
        rc = self.L.sl3d_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            raise Sl3dError(f"sl3d_create: {self.L.sl3d_strerror(rc).decode()}: {self.L.sl3d_last_error(None).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self.L.sl3d_destroy(self._h)
            self._h = None
        for p in getattr(self, "_pinned", []):
            self.L.sl3d_host_free(p)
        self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise Sl3dError(f"{what}: {self.L.sl3d_strerror(rc).decode()}: {self.L.sl3d_last_error(self._h).decode()}")

    # ---- inputs -------------------------------------------------------------------------------
    def set_calibration(self, Kc, dc, rc, tc, Kp, dp, rp, tp):
        a = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()) for v in (Kc, dc, rc, tc, Kp, dp, rp, tp)]
        assert [v.size for v in a] == [9, 5, 3, 3, 9, 5, 3, 3]
        self._chk(self.L.sl3d_set_calibration(self._h, *[v.ctypes.data for v in a]), "sl3d_set_calibration")

    def projection_matrices(self):
        """(A_cam, A_proj): the 3x4 matrices K [R|t] the library derived from the calibration (T0, compute_A)."""
        a, b = np.zeros(12), np.zeros(12)
        self._chk(self.L.sl3d_get_projection_matrices(self._h, a.ctypes.data, b.ctypes.data), "sl3d_get_projection_matrices")
        return a.reshape(3, 4), b.reshape(3, 4)

    def set_mask(self, full_frame_mask, view=0):
        m = np.ascontiguousarray(full_frame_mask, dtype=np.uint8)
        assert m.shape == (self.cfg.full_height, self.cfg.full_width), m.shape
        self._chk(self.L.sl3d_set_mask(self._h, view, m.ctypes.data, m.strides[0]), "sl3d_set_mask")

    def set_masks(self, masks, first_view=0, n_views=None):
        """One launch for several views.  masks: one full-frame mask (every view gets it) or an array [n][full_height][full_width]."""
        m = np.ascontiguousarray(masks, dtype=np.uint8)
        if os.environ.get("SL3D_LIB") and not hasattr(self.L, "sl3d_set_masks"):   # an older build under SL3D_LIB (A/B runs): view by view
            n = (self.cfg.max_views - first_view if n_views is None else n_views) if m.ndim == 2 else m.shape[0]
            for k in range(n):
                self.set_mask(m if m.ndim == 2 else m[k], view=first_view + k)
            return
        if m.ndim == 2:
            assert m.shape == (self.cfg.full_height, self.cfg.full_width), m.shape
            n, vs = (self.cfg.max_views - first_view if n_views is None else n_views), 0
        else:
            assert m.shape[1:] == (self.cfg.full_height, self.cfg.full_width), m.shape
            n, vs = m.shape[0], m.strides[0]
            assert n_views is None or n_views == n
        self._chk(self.L.sl3d_set_masks(self._h, first_view, n, m.ctypes.data, m.strides[-2], vs), "sl3d_set_masks")

    def set_masks_device(self, dev_ptr, stride, view_stride, first_view=0, n_views=1):
        """Masks that already live in device memory (a raw address, e.g. torch.Tensor.data_ptr())."""
        self._chk(self.L.sl3d_set_masks(self._h, first_view, n_views, C.c_void_p(dev_ptr), stride, view_stride), "sl3d_set_masks")

    def set_frames(self, axis, planes, view=0):
        arrs = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        for a in arrs:
            assert a.shape == (self.H, self.W), a.shape
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._chk(self.L.sl3d_set_frames(self._h, view, axis, ptrs, len(arrs), arrs[0].strides[0]), "sl3d_set_frames")

    def set_mask_colrow(self, selected_region, view=0):
        """selected_region as the reference holds it: int32 array [full_width][full_height] ([col][row]), selected iff == 1."""
        m = np.ascontiguousarray(selected_region, dtype=np.int32)
        assert m.shape == (self.cfg.full_width, self.cfg.full_height), m.shape
        self._chk(self.L.sl3d_set_mask_colrow(self._h, view, m.ctypes.data), "sl3d_set_mask_colrow")

    def set_frames_range(self, axis, first_plane, planes, view=0):
        arrs = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        for a in arrs:
            assert a.shape == (self.H, self.W), a.shape
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._chk(self.L.sl3d_set_frames_range(self._h, view, axis, first_plane, ptrs, len(arrs), arrs[0].strides[0]), "sl3d_set_frames_range")

    def global_colrow(self, which, view=0, out=None, row0=0):
        """One of the reference's globals in its own [col][row] layout and type (sl3d_get_global_colrow); `out`: a [W][H_total(,3)]
        array this window's rows are written into at row offset row0 (default: a fresh [W][H] array)."""
        dt, comps = (np.float64, 3) if which == 9 else (np.float32, 1) if 3 <= which <= 6 else (np.int32, 1)
        if out is None:
            out = np.empty((self.W, self.H) + ((3,) if comps == 3 else ()), dtype=dt)
        assert out.dtype == dt and out.flags["C_CONTIGUOUS"] and out.shape[0] == self.W
        self._chk(self.L.sl3d_get_global_colrow(self._h, view, which, out.ctypes.data, out.shape[1], row0), "sl3d_get_global_colrow")
        return out

    def set_frames_raw(self, axis, planes, view=0):
        """set_frames for raw captures: undistorted on the device with the camera calibration (whole frames only)."""
        arrs = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        for a in arrs:
            assert a.shape == (self.H, self.W), a.shape
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._chk(self.L.sl3d_set_frames_raw(self._h, view, axis, ptrs, len(arrs), arrs[0].strides[0]), "sl3d_set_frames_raw")

    def synth_view(self, view=0, plane=(0.0, 0.05, 0.05), seed=0x3D5CA11, view_id=None, noise=0, gain=0.8, offset=10.0):
        """Synthetic capture generated on the device into slot `view` (see 3dscan_amd/synth.py for the host twin)."""
        pl = np.asarray(plane, dtype=np.float64)
        self._chk(self.L.sl3d_synth_view(self._h, view, pl.ctypes.data, seed, view if view_id is None else view_id, noise, gain, offset),
                  "sl3d_synth_view")

    def frames(self, axis, view=0):
        n = self.cfg.n_fringe + 2 * (self.cfg.n_gray_v if axis == 0 else self.cfg.n_gray_h)
        arrs = [np.empty((self.H, self.W), dtype=np.uint8) for _ in range(n)]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        self._chk(self.L.sl3d_get_frames(self._h, view, axis, ptrs, n, self.W), "sl3d_get_frames")
        return arrs

    def copy_view(self, src_view, dst_view):
        self._chk(self.L.sl3d_copy_view(self._h, src_view, dst_view), "sl3d_copy_view")

    # ---- the reference's four stage entry points ----------------------------------------------
    def compute_wrapped_phase(self, pattern_type, view=0):
        self._chk(self.L.sl3d_compute_wrapped_phase(self._h, view, pattern_type), "sl3d_compute_wrapped_phase")

    def unwrap_phase(self, pattern_type, view=0):
        self._chk(self.L.sl3d_unwrap_phase(self._h, view, pattern_type), "sl3d_unwrap_phase")

    def compute_c_p_map(self, view=0):
        self._chk(self.L.sl3d_compute_c_p_map(self._h, view), "sl3d_compute_c_p_map")

    def triangulate(self, view=0):
        self._chk(self.L.sl3d_triangulate(self._h, view), "sl3d_triangulate")

    def run_stages(self, view=0):
        """main()'s order (m_tech_project_console.cpp:372-395) through the per-stage kernels."""
        self.compute_wrapped_phase(0, view); self.compute_wrapped_phase(1, view)
        self.unwrap_phase(0, view); self.unwrap_phase(1, view)
        self.compute_c_p_map(view)
        self.triangulate(view)

    # ---- fused hot path -----------------------------------------------------------------------
    def run(self, first_view=0, n_views=1):
        self._chk(self.L.sl3d_run(self._h, first_view, n_views), "sl3d_run")

    def run_clouds(self, first_view=0, n_views=1):
        """The fused pass with the ordered compaction inside the kernel (sl3d_run_clouds); asynchronous."""
        self._chk(self.L.sl3d_run_clouds(self._h, first_view, n_views), "sl3d_run_clouds")

    def cloud_counts(self, first_view=0, n_views=1, want_device_copy=True):
        """(device address of the first CONTIGUOUS cloud, points between clouds, [count per view]) after run_clouds; synchronises.
        want_device_copy=False: counts only (address None) -- segmented clouds then need no gap-closing launch."""
        counts = (C.c_int64 * n_views)()
        ptr, stride = C.c_void_p(), C.c_size_t()
        self._chk(self.L.sl3d_get_cloud_counts(self._h, first_view, n_views, C.byref(ptr) if want_device_copy else None, C.byref(stride), counts),
                  "sl3d_get_cloud_counts")
        return ptr.value, stride.value, [int(c) for c in counts]

    def cloud_segments(self, first_view=0, n_views=1):
        """(CloudSegments, [count per view]): the segmented clouds of the last run_clouds where they lie in HBM."""
        seg = CloudSegments()
        counts = (C.c_int64 * n_views)()
        self._chk(self.L.sl3d_get_cloud_segments(self._h, first_view, n_views, C.byref(seg), counts), "sl3d_get_cloud_segments")
        return seg, [int(c) for c in counts]

    def download_clouds(self, first_view=0, n_views=1, out=None):
        """Host copies of the clouds of the last run_clouds: list of (n_k, 3) float32 arrays (views of `out`, a flat float32 buffer --
        pinned for the zero-copy route -- or of a fresh array)."""
        counts = (C.c_int64 * n_views)()
        self._chk(self.L.sl3d_download_clouds(self._h, first_view, n_views, None, 0, counts), "sl3d_download_clouds")
        total = sum(counts)
        if out is None:
            out = np.empty(max(total, 1) * 3, dtype=np.float32)
        assert out.dtype == np.float32 and out.size >= 3 * total
        self._chk(self.L.sl3d_download_clouds(self._h, first_view, n_views, out.ctypes.data, out.size // 3, counts), "sl3d_download_clouds")
        res, off = [], 0
        for n in counts:
            res.append(out[3 * off:3 * (off + n)].reshape(n, 3))
            off += n
        return res

    def download_cloud_into(self, view, out):
        """ONE call of sl3d_download_clouds for one view into `out` (flat float32 buffer; capacity = out.size // 3 points): the number
        of valid points of the view (which may exceed the capacity: then the first `capacity` points were written).  With a pinned
        buffer behind a run_clouds of a few views this is the route without a scan launch (the gap-closing kernel scans on entry)."""
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
        counts = (C.c_int64 * 1)()
        self._chk(self.L.sl3d_download_clouds(self._h, view, 1, out.ctypes.data, out.size // 3, counts), "sl3d_download_clouds")
        return int(counts[0])

    def register_clouds(self, first_view, n_views, tx, ty, tz, rot_step):
        """register_views on the clouds of the last run_clouds (rotation applied while the segments are concatenated)."""
        n = C.c_int64(0)
        self._chk(self.L.sl3d_register_clouds(self._h, first_view, n_views, tx, ty, tz, rot_step, None, 0, C.byref(n)), "sl3d_register_clouds")
        out = np.empty((n.value, 3), dtype=np.float32)
        self._chk(self.L.sl3d_register_clouds(self._h, first_view, n_views, tx, ty, tz, rot_step, out.ctypes.data, n.value, C.byref(n)),
                  "sl3d_register_clouds")
        return out

    def fused_clouds(self, first_view=0, n_views=1):
        """run_clouds + host copies: list of (n_k, 3) float32 arrays in the reference's scan order."""
        self.run_clouds(first_view, n_views)
        ptr, stride, counts = self.cloud_counts(first_view, n_views)
        out = []
        for k, n in enumerate(counts):
            a = np.empty((n, 3), dtype=np.float32)
            if n:
                self._d2h(a, ptr + 12 * k * stride)
            out.append(a)
        return out

    def _d2h(self, arr, dev_ptr):
        """Device -> host copy of an address the library handed out (sl3d_download)."""
        self._chk(self.L.sl3d_download(self._h, arr.ctypes.data, dev_ptr, arr.nbytes), "sl3d_download")

    def fused_kernel_name(self, n_views=1, clouds=False):
        """The k_fused instantiation a launch over n_views views runs, as rocprofv3 prints it."""
        if os.environ.get("SL3D_LIB") and not hasattr(self.L, "sl3d_fused_kernel_name"):
            return "(a build without sl3d_fused_kernel_name)"
        buf = C.create_string_buffer(256)
        self._chk(self.L.sl3d_fused_kernel_name(self._h, n_views, 1 if clouds else 0, buf, len(buf)), "sl3d_fused_kernel_name")
        return buf.value.decode()

    def camera_table_bytes_per_pixel(self, n_views=1):
        """Bytes per pixel a launch of n_views views reads from the camera-side table (once per launch); None with a build that cannot say."""
        if not hasattr(self.L, "sl3d_camera_table_bytes_per_pixel"):
            return None
        n = self.L.sl3d_camera_table_bytes_per_pixel(self._h, n_views)
        if n < 0:
            self._chk(n, "sl3d_camera_table_bytes_per_pixel")
        return n

    def launch_counts(self):
        """(fused launches that went to the context's stream, ... to its launch lanes)"""
        a, b = C.c_int64(), C.c_int64()
        self.L.sl3d_launch_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self._chk(self.L.sl3d_launch_counts(self._h, C.byref(a), C.byref(b)), "sl3d_launch_counts")
        return a.value, b.value

    def last_fused_kernel_name(self):
        """The k_fused instantiation the last fused launch of this context ran."""
        if os.environ.get("SL3D_LIB") and not hasattr(self.L, "sl3d_last_fused_kernel_name"):
            return "(a build without sl3d_last_fused_kernel_name)"
        buf = C.create_string_buffer(256)
        self._chk(self.L.sl3d_last_fused_kernel_name(self._h, buf, len(buf)), "sl3d_last_fused_kernel_name")
        return buf.value.decode()

    def run_timed(self, first_view=0, n_views=1):
        ms = C.c_float(0)
        self._chk(self.L.sl3d_run_timed(self._h, first_view, n_views, C.byref(ms)), "sl3d_run_timed")
        return ms.value

    def timer_start(self):
        self._chk(self.L.sl3d_timer_start(self._h), "sl3d_timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        self._chk(self.L.sl3d_timer_stop(self._h, C.byref(ms)), "sl3d_timer_stop")
        return ms.value

    def synchronize(self):
        self._chk(self.L.sl3d_synchronize(self._h), "sl3d_synchronize")

    # ---- outputs ------------------------------------------------------------------------------
    def _plane(self, fn, dtype, view, *pre, comps=1):
        shape = (self.H, self.W) if comps == 1 else (self.H, self.W, comps)
        out = np.empty(shape, dtype=dtype)
        self._chk(getattr(self.L, fn)(self._h, view, *pre, out.ctypes.data, self.W * comps), fn)
        return out

    def valid_map(self, which=VALID_MERGED, view=0):
        return self._plane("sl3d_get_valid_map", np.uint8, view, which)

    def wrapped_phase(self, axis, view=0):
        return self._plane("sl3d_get_wrapped_phase", np.float32, view, axis)

    def unwrapped_phase(self, axis, view=0):
        return self._plane("sl3d_get_unwrapped_phase", np.float32, view, axis)

    def code(self, axis, view=0):
        return self._plane("sl3d_get_code", np.int32, view, axis)

    def debug_image(self, stage, axis, view=0):
        return self._plane("sl3d_get_debug_image", np.uint8, view, stage, axis)

    def c_p_map(self, view=0):
        out = np.empty((self.H, self.W, 2), dtype=np.int64)
        self._chk(self.L.sl3d_get_c_p_map(self._h, view, out.ctypes.data), "sl3d_get_c_p_map")
        return out

    def intersection_points(self, view=0):
        out = np.empty((self.H, self.W, 3), dtype=np.float64)
        self._chk(self.L.sl3d_get_intersection_points(self._h, view, out.ctypes.data), "sl3d_get_intersection_points")
        return out

    def points(self, view=0):
        xyz = np.empty((self.H, self.W, 3), dtype=np.float32)
        valid = np.empty((self.H, self.W), dtype=np.uint8)
        self._chk(self.L.sl3d_get_points(self._h, view, xyz.ctypes.data, valid.ctypes.data), "sl3d_get_points")
        return xyz, valid

    def download_views(self, first_view, n_views, xyz, valid):
        """Dense results of views [first_view, first_view + n_views) into caller arrays (n, H, W, 3) f32 / (n, H, W) u8 (pinned
        for full-rate DMA): one 2-D copy per view and plane, one wait."""
        assert xyz.shape == (n_views, self.H, self.W, 3) and valid.shape == (n_views, self.H, self.W)
        b = self.device_buffers()
        for k in range(n_views):
            self._chk(self.L.sl3d_download_2d(self._h, xyz[k].ctypes.data, self.W * 12, b.points + (first_view + k) * b.points_view_stride,
                                              b.points_pitch, self.W * 12, self.H), "sl3d_download_2d")
            self._chk(self.L.sl3d_download_2d(self._h, valid[k].ctypes.data, self.W, b.valid + (first_view + k) * b.valid_view_stride,
                                              b.valid_pitch, self.W, self.H), "sl3d_download_2d")
        self.synchronize()

    def cloud(self, view=0):
        n = C.c_int64(0)
        self._chk(self.L.sl3d_get_cloud(self._h, view, None, 0, C.byref(n)), "sl3d_get_cloud")
        out = np.empty((n.value, 3), dtype=np.float32)
        self._chk(self.L.sl3d_get_cloud(self._h, view, out.ctypes.data, n.value, C.byref(n)), "sl3d_get_cloud")
        return out

    def compact_views(self, first_view, n_views):
        """Batched device compaction (three launches for all views); returns the per-view counts, clouds stay in HBM."""
        counts = (C.c_int64 * n_views)()
        self._chk(self.L.sl3d_compact_views(self._h, first_view, n_views, None, None, counts), "sl3d_compact_views")
        return [int(c) for c in counts]

    def clouds(self, first_view, n_views):
        """The compacted clouds of a batch of views as a list of (n_k, 3) float32 arrays."""
        counts = (C.c_int64 * n_views)()
        self._chk(self.L.sl3d_get_clouds(self._h, first_view, n_views, None, 0, counts), "sl3d_get_clouds")
        total = sum(counts)
        flat = np.empty((total, 3), dtype=np.float32)
        self._chk(self.L.sl3d_get_clouds(self._h, first_view, n_views, flat.ctypes.data, total, counts), "sl3d_get_clouds")
        out, off = [], 0
        for n in counts:
            out.append(flat[off:off + n])
            off += n
        return out

    def set_texture(self, bgr, view=0):
        """The colour image save_point_cloud() takes r,g,b from: (H, W, 3) uint8, B,G,R order (cvLoadImage)."""
        t = np.ascontiguousarray(bgr, dtype=np.uint8)
        assert t.shape == (self.H, self.W, 3), t.shape
        self._chk(self.L.sl3d_set_texture(self._h, view, t.ctypes.data, t.strides[0]), "sl3d_set_texture")

    def cloud_rgb(self, view=0):
        """(n,3) float32 xyz and (n,3) uint8 r,g,b of the valid pixels in the reference's scan order."""
        n = C.c_int64(0)
        self._chk(self.L.sl3d_get_cloud_rgb(self._h, view, None, None, 0, C.byref(n)), "sl3d_get_cloud_rgb")
        xyz = np.empty((n.value, 3), dtype=np.float32)
        rgb = np.empty((n.value, 3), dtype=np.uint8)
        self._chk(self.L.sl3d_get_cloud_rgb(self._h, view, xyz.ctypes.data, rgb.ctypes.data, n.value, C.byref(n)), "sl3d_get_cloud_rgb")
        return xyz, rgb

    def register_views(self, first_view, n_views, tx, ty, tz, rot_step):
        """register_point_clouds(): rotate view k by k*rot_step degrees about Y through (tx,ty,tz), concatenate."""
        n = C.c_int64(0)
        self._chk(self.L.sl3d_register_views(self._h, first_view, n_views, tx, ty, tz, rot_step, None, 0, C.byref(n)), "sl3d_register_views")
        out = np.empty((n.value, 3), dtype=np.float32)
        self._chk(self.L.sl3d_register_views(self._h, first_view, n_views, tx, ty, tz, rot_step, out.ctypes.data, n.value, C.byref(n)),
                  "sl3d_register_views")
        return out

    def pinned(self, shape, dtype):
        """numpy array in pinned host memory (sl3d_host_alloc); freed when the Scanner is closed."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.L.sl3d_host_alloc(n)
        if not p:
            raise Sl3dError("sl3d_host_alloc failed")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return np.frombuffer((C.c_char * n).from_address(p), dtype=dtype).reshape(shape)

    def process_views(self, frames, xyz=None, valid=None):
        """frames: (n_views, planes_per_view, H, W) uint8 host array (pinned for full overlap); returns (xyz, valid)."""
        n, ppv, H, W = frames.shape
        assert (H, W) == (self.H, self.W) and frames.dtype == np.uint8 and frames.strides[3] == 1 and frames.strides[2] >= W
        if xyz is None:
            xyz = np.empty((n, H, W, 3), dtype=np.float32)
        if valid is None:
            valid = np.empty((n, H, W), dtype=np.uint8)
        ptrs = (C.c_void_p * (n * ppv))(*[frames[v, p].ctypes.data for v in range(n) for p in range(ppv)])
        self._chk(self.L.sl3d_process_views(self._h, n, ptrs, frames.strides[2], xyz.ctypes.data, valid.ctypes.data), "sl3d_process_views")
        return xyz, valid

    def undistort(self, image, K, dist):
        """cvUndistort2 on an (H, W) or (H, W, 3) uint8 image (any size), on the device."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        cn = 1 if a.ndim == 2 else a.shape[2]
        out = np.empty_like(a)
        Kd = np.ascontiguousarray(np.asarray(K, dtype=np.float64).ravel())
        dd = np.ascontiguousarray(np.asarray(dist, dtype=np.float64).ravel())
        self._chk(self.L.sl3d_undistort(self._h, a.ctypes.data, a.strides[0], a.shape[1], a.shape[0], cn, Kd.ctypes.data, dd.ctypes.data,
                                        out.ctypes.data, out.strides[0]), "sl3d_undistort")
        return out

    def generate_pattern(self, kind, axis, index):
        """One projector pattern of generate_pattern() (1/pattern_generator.cpp): (proj_height, proj_width) uint8."""
        out = np.empty((self.cfg.proj_height, self.cfg.proj_width), dtype=np.uint8)
        self._chk(self.L.sl3d_generate_pattern(self._h, kind, axis, index, out.ctypes.data, out.strides[0], None, None), "sl3d_generate_pattern")
        return out

    def transform_cloud(self, xyz, theta_deg, tx, ty, tz):
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.empty_like(a)
        self._chk(self.L.sl3d_transform_cloud(self._h, a.ctypes.data, len(a), theta_deg, tx, ty, tz, out.ctypes.data), "sl3d_transform_cloud")
        return out

    def device_buffers(self):
        b = DeviceBuffers()
        self._chk(self.L.sl3d_get_device_buffers(self._h, C.byref(b)), "sl3d_get_device_buffers")
        return b


class Group:
    """sl3d_group_*: n row stripes of one frame window, stripe i on HIP device devices[i], assembled on stripe 0's GPU
    (RCCL send/recv between GPUs, device copies inside one)."""

    def __init__(self, width, height, proj_width, proj_height, n_gray_v, n_gray_h, fringe_width_v, fringe_width_h, devices,
                 n_fringe=3, max_views=1, full_size=None, origin=(0, 0), flags=0):
        self.L = load_library()
        fw, fh = full_size if full_size else (width, height)
        self.cfg = Config(width, height, fw, fh, origin[0], origin[1], proj_width, proj_height, n_fringe,
                          n_gray_v, n_gray_h, fringe_width_v, fringe_width_h, 0, 0, max_views, 0, flags, None)
        self.W, self.H = width, height
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        rc = self.L.sl3d_group_create(C.byref(self.cfg), devs, len(devices), C.byref(self._h))
        if rc != 0:
            raise Sl3dError(f"sl3d_group_create: {self.L.sl3d_strerror(rc).decode()}: {self.L.sl3d_group_last_error(None).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self.L.sl3d_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise Sl3dError(f"{what}: {self.L.sl3d_strerror(rc).decode()}: {self.L.sl3d_group_last_error(self._h).decode()}")

    @property
    def transport(self):
        return self.L.sl3d_group_transport(self._h).decode()

    def stripes(self):
        """[(row0, rows, device, ctx handle)] of every stripe."""
        out = []
        for k in range(self.L.sl3d_group_size(self._h)):
            r0, n, d, h = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
            self._chk(self.L.sl3d_group_stripe(self._h, k, C.byref(r0), C.byref(n), C.byref(d), C.byref(h)), "sl3d_group_stripe")
            out.append((r0.value, n.value, d.value, h))
        return out

    def set_calibration(self, Kc, dc, rc, tc, Kp, dp, rp, tp):
        a = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()) for v in (Kc, dc, rc, tc, Kp, dp, rp, tp)]
        self._chk(self.L.sl3d_group_set_calibration(self._h, *[v.ctypes.data for v in a]), "sl3d_group_set_calibration")

    def set_mask(self, full_frame_mask, view=0):
        m = np.ascontiguousarray(full_frame_mask, dtype=np.uint8)
        assert m.shape == (self.cfg.full_height, self.cfg.full_width), m.shape
        self._chk(self.L.sl3d_group_set_mask(self._h, view, m.ctypes.data, m.strides[0]), "sl3d_group_set_mask")

    def set_frames(self, axis, planes, view=0):
        arrs = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
        for a in arrs:
            assert a.shape == (self.H, self.W), a.shape
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._chk(self.L.sl3d_group_set_frames(self._h, view, axis, ptrs, len(arrs), arrs[0].strides[0]), "sl3d_group_set_frames")

    def run(self, first_view=0, n_views=1):
        self._chk(self.L.sl3d_group_run(self._h, first_view, n_views), "sl3d_group_run")

    def gather(self, first_view=0, n_views=1):
        self._chk(self.L.sl3d_group_gather(self._h, first_view, n_views), "sl3d_group_gather")

    def points(self, view=0):
        xyz = np.empty((self.H, self.W, 3), dtype=np.float32)
        valid = np.empty((self.H, self.W), dtype=np.uint8)
        self._chk(self.L.sl3d_group_get_points(self._h, view, xyz.ctypes.data, valid.ctypes.data), "sl3d_group_get_points")
        return xyz, valid

    def download_points(self, first_view=0, n_views=1, xyz=None, valid=None):
        """Every stripe's rows straight into the caller's dense host images (no gather to the root): (n, H, W, 3) f32, (n, H, W) u8."""
        if xyz is None:
            xyz = np.empty((n_views, self.H, self.W, 3), dtype=np.float32)
        if valid is None:
            valid = np.empty((n_views, self.H, self.W), dtype=np.uint8)
        assert xyz.shape == (n_views, self.H, self.W, 3) and valid.shape == (n_views, self.H, self.W)
        self._chk(self.L.sl3d_group_download_points(self._h, first_view, n_views, xyz.ctypes.data, valid.ctypes.data), "sl3d_group_download_points")
        return xyz, valid

    def process_views(self, frames, xyz=None, valid=None):
        """frames: (n_views, planes_per_view, H, W) uint8 host array (pinned for concurrent DMA); returns (xyz, valid) dense images."""
        n, ppv, H, W = frames.shape
        assert (H, W) == (self.H, self.W) and frames.dtype == np.uint8 and frames.strides[3] == 1 and frames.strides[2] >= W
        if xyz is None:
            xyz = np.empty((n, H, W, 3), dtype=np.float32)
        if valid is None:
            valid = np.empty((n, H, W), dtype=np.uint8)
        ptrs = (C.c_void_p * (n * ppv))(*[frames[v, p].ctypes.data for v in range(n) for p in range(ppv)])
        self._chk(self.L.sl3d_group_process_views(self._h, n, ptrs, frames.strides[2], xyz.ctypes.data, valid.ctypes.data), "sl3d_group_process_views")
        return xyz, valid

    def run_clouds(self, first_view=0, n_views=1):
        self._chk(self.L.sl3d_group_run_clouds(self._h, first_view, n_views), "sl3d_group_run_clouds")

    def gather_clouds(self, first_view=0, n_views=1):
        counts = (C.c_int64 * n_views)()
        self._chk(self.L.sl3d_group_gather_clouds(self._h, first_view, n_views, counts), "sl3d_group_gather_clouds")
        return [int(c) for c in counts]

    def cloud(self, view=0):
        n = C.c_int64(0)
        self._chk(self.L.sl3d_group_get_cloud(self._h, view, None, 0, C.byref(n)), "sl3d_group_get_cloud")
        out = np.empty((n.value, 3), dtype=np.float32)
        self._chk(self.L.sl3d_group_get_cloud(self._h, view, out.ctypes.data, n.value, C.byref(n)), "sl3d_group_get_cloud")
        return out

    def synchronize(self):
        self._chk(self.L.sl3d_group_synchronize(self._h), "sl3d_group_synchronize")
