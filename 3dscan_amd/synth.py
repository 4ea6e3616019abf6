"""Deterministic synthetic captures for the decode->triangulate path.

A plane Z = z0 + a*X + b*Y in the world frame is viewed by a camera + projector pair.  For
every camera pixel the (distorted-model) ray is cast, intersected with the plane and projected
into the projector, giving continuous projector coordinates (xp, yp).  The projected patterns
follow the reference's pattern generator:

  fringe k   : 127 + 128*cos((p/fw)*2*Pi - Pi - Pi/2 + k*Pi/2), Pi = 22/7   (1/pattern_generator.cpp:302,313)
  Gray bit i : G_i = B_{i-1} xor B_i of code = floor(p/fw), MSB first, x255  (1/pattern_generator.cpp:80-105)
  inverse    : 255 - pattern                                                (1/pattern_generator.cpp:497)

then a simple camera model (gain, offset, counter-hash noise) is applied.  This is host-side
plumbing for tests and benchmark inputs; it is not part of the timed path.
"""
import numpy as np

from .calibration import CAL_KEYS, scaled_calibration

PI_REF = 22.0 / 7.0           # PROJECT_GLOBAL/global_cv.h:62
DEFAULT_SEED = 0x3D5CA11


def rodrigues(rvec):
    r = np.asarray(rvec, dtype=np.float64)
    th = float(np.sqrt((r * r).sum()))
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * K


def undistort_normalized(px, py, K, d, iters=5):
    """Iterative inverse of the 5-parameter Brown model, as stage 7 applies it to pixel coordinates."""
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)
    k1, k2, p1, p2, k3 = [float(v) for v in d]
    x0 = (px - K[0, 2]) / K[0, 0]
    y0 = (py - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return x, y


def projector_coordinates(W, H, cal, plane=(0.0, 0.05, 0.05), col0=0, row0=0):
    """Continuous projector coordinates (xp, yp) seen by every camera pixel of a W x H window whose
    top-left pixel is (col0,row0) of the camera frame, for the plane Z = z0 + a*X + b*Y."""
    z0, a, b = plane
    cols = np.arange(col0, col0 + W, dtype=np.float64)[None, :].repeat(H, 0)
    rows = np.arange(row0, row0 + H, dtype=np.float64)[:, None].repeat(W, 1)
    x, y = undistort_normalized(cols, rows, cal["Kc"], cal["dc"])
    Rc, Rp = rodrigues(cal["rc"]), rodrigues(cal["rp"])
    tc, tp = np.asarray(cal["tc"], float), np.asarray(cal["tp"], float)
    d = np.stack([x, y, np.ones_like(x)], -1)            # camera-frame ray
    dw = d @ Rc                                           # Rc^T d  (row-vector form)
    ow = -(Rc.T @ tc)                                     # camera centre in the world
    n = np.array([-a, -b, 1.0])
    lam = (z0 - ow @ n) / (dw @ n)
    Xw = ow[None, None, :] + lam[..., None] * dw
    Xp = Xw @ Rp.T + tp
    Kp = np.asarray(cal["Kp"], float).reshape(3, 3)
    xp = Kp[0, 0] * Xp[..., 0] / Xp[..., 2] + Kp[0, 2]
    yp = Kp[1, 1] * Xp[..., 1] / Xp[..., 2] + Kp[1, 2]
    return xp, yp, Xw


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _noise(shape, seed, view, frame, amp, col0=0, row0=0):
    if amp <= 0:
        return np.zeros(shape, dtype=np.int32)
    H, W = shape
    with np.errstate(over="ignore"):
        idx = (np.arange(row0, row0 + H, dtype=np.uint64)[:, None] << np.uint64(20)) + np.arange(col0, col0 + W, dtype=np.uint64)[None, :]
        key = np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(view) * np.uint64(1000003) + np.uint64(frame)
        h = _splitmix64(idx ^ _splitmix64(np.full((1, 1), key, dtype=np.uint64)))
    return (h % np.uint64(2 * amp + 1)).astype(np.int32) - amp


def _camera(I, lit, gain, offset, noise):
    v = np.where(lit, gain * I.astype(np.float32), 0.0) + offset + noise
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)


def _axis_frames(p, lit, fw, N, n_fringe, gain, offset, seed, view, axis, noise_amp, col0, row0):
    """fringe (n_fringe), gray (N), inverse-gray (N) frames for one axis; p = projector coordinate."""
    pf = p.astype(np.float32)
    a = (pf / np.float32(fw)).astype(np.float64)
    out = []
    fidx = axis * 1000
    for k in range(n_fringe):
        arg = a * 2.0 * 22.0 / 7.0 - 22.0 / 7.0 - ((22.0 / 7.0) / 2.0) + (22.0 / 7.0 / 2.0) * float(k)
        t = np.float32(127.0) + np.float32(128.0) * np.cos(arg.astype(np.float32))
        I = np.clip(t, 0, 255).astype(np.uint8)
        out.append(_camera(I, lit, gain, offset, _noise(p.shape, seed, view, fidx, noise_amp, col0, row0)))
        fidx += 1
    code = np.floor(pf / np.float32(fw)).astype(np.int64)
    code = np.clip(code, 0, (1 << N) - 1)
    gray_code = code ^ (code >> 1)
    pats = [(((gray_code >> (N - 1 - i)) & 1) * 255).astype(np.uint8) for i in range(N)]
    for I in pats:
        out.append(_camera(I, lit, gain, offset, _noise(p.shape, seed, view, fidx, noise_amp, col0, row0)))
        fidx += 1
    for I in pats:
        out.append(_camera(255 - I, lit, gain, offset, _noise(p.shape, seed, view, fidx, noise_amp, col0, row0)))
        fidx += 1
    return out


def default_mask(W, H):
    """H0's default selection: 1 on [1..W-2]x[1..H-2], 0 on the border."""
    m = np.zeros((H, W), dtype=np.uint8)
    m[1:-1, 1:-1] = 1
    return m


def synth_rig(W, H, PW, PH):
    """Calibration for synthetic runs: the reference rig rescaled to the requested resolutions, with the
    projector's focal length halved so its image covers the camera's field of view on the Z~0 plane."""
    cal = scaled_calibration(W, H, PW, PH)
    Kp = cal["Kp"].reshape(3, 3).copy()
    Kp[0, 0] *= 0.5
    Kp[1, 1] *= 0.5
    Kp[0, 2] = 0.5 * PW
    Kp[1, 2] = 0.5 * PH
    cal["Kp"] = Kp.ravel()
    return cal


def make_capture(W, H, PW, PH, N_v, N_h, fw_v, fw_h, cal=None, plane=(0.0, 0.05, 0.05), n_fringe=3,
                 seed=DEFAULT_SEED, view=0, noise=0, gain=0.8, offset=10.0, col0=0, row0=0, full=None):
    """One synthetic view.  Returns dict(planes_v, planes_h, mask, xp, yp, world, cal) where planes_*
    are lists of HxW uint8 arrays ordered fringe[0..F), gray[0..N), inverse[0..N)."""
    fullW, fullH = full if full is not None else (W, H)
    if cal is None:
        cal = synth_rig(fullW, fullH, PW, PH)
    assert PW <= fw_v * (1 << N_v) and PH <= fw_h * (1 << N_h), "Gray code too short for the projector"
    xp, yp, Xw = projector_coordinates(W, H, cal, plane, col0, row0)
    lit = (xp >= 0) & (xp < PW) & (yp >= 0) & (yp < PH)
    pv = _axis_frames(xp, lit, fw_v, N_v, n_fringe, gain, offset, seed, view, 0, noise, col0, row0)
    ph = _axis_frames(yp, lit, fw_h, N_h, n_fringe, gain, offset, seed, view, 1, noise, col0, row0)
    mask = default_mask(fullW, fullH)[row0:row0 + H, col0:col0 + W]
    return dict(planes_v=pv, planes_h=ph, mask=mask, xp=xp, yp=yp, world=Xw, lit=lit, cal=cal)


def cal_tuple(cal):
    return tuple(np.asarray(cal[k], dtype=np.float64).ravel() for k in CAL_KEYS)
