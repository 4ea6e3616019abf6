#!/usr/bin/env python3
"""Generates the committed golden fixtures from the reference's own data artefacts.

Run in the build container only (it reads /root/reference, which does not exist
on the GPU box):   python tests/golden/make_golden.py

What it does
 1. Loads the real 1600x1200 captures stage 3/4 compute on (the Gray_captured_image_*.bmp
    re-saves of 3/wrapped_phase.cpp:46-52 and 4/phase_unwrap.cpp:79-86,118-125) and the four
    known-answer images the reference wrote (Wrapped_phase_image.bmp, Unwrapped_phase_*.bmp).
 2. PINS THE ORACLE: replays oracle stages 3 and 4 on the full frames and requires the debug
    images to equal the reference's KAT images on every pixel (358,580 valid px per axis).
 2b. PINS T0: the oracle's cvRodrigues2 / cvTranspose / cvGEMM restatements must reproduce, bit for bit, the 12 doubles of
    Triangulation/Relative_geometry/proj_cam_rot_mat.xml + proj_cam_trans_vect.xml that the reference's stage 6 computed with
    OpenCV itself from the rotation / translation vectors stage 7 reads (6/system_calibration.cpp:1488-1516).
 3. Runs oracle stages 5 and 7 on the full frames with the reference's 8 calibration XMLs
    (unpinned stages: these outputs are regression goldens, not reference answers) and CORROBORATES them with an
    independent fp64 NumPy restatement of the same stages (independent_stage_5_7: no code shared with oracle/):
    identical 355,608 correspondences and 3-D points within 1e-12 of their norm are required on the full scan.
 4. Writes small crops (inputs + expected outputs) to tests/golden/*.npz and the calibration
    to tests/golden/calibration.json.  Only derived data is written: no reference source.
 5. PINS THE PATTERN GENERATOR (N1): the oracle's fringe / Gray / inverse-Gray / binary patterns must equal the 45
    pattern images the reference generated (Generated_patterns/**) on every pixel; their 1-D profiles go to
    tests/golden/patterns_ref.npz  (`python tests/golden/make_golden.py patterns` runs this step alone).
"""
import json
import os
import re
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.oracle import Oracle  # noqa: E402

REF = "/root/reference/M_tech_project_console/"
W, H, PW, PH = 1600, 1200, 1280, 720          # global_cv.h:49-53
N_V, N_H, FW, NCODES_V, NCODES_H = 6, 5, 32, 40, 23  # common_variables.h:6-9,23-24


def bmp(path):
    im = Image.open(REF + path)
    assert im.mode == "L" and im.size == (W, H), (path, im.mode, im.size)
    return np.array(im)


def xml_data(path):
    txt = open(REF + path).read()
    return [float(x) for x in re.search(r"<data>(.*?)</data>", txt, re.S).group(1).split()]


def dilate3(m):
    p = np.pad(m, 1)
    out = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            out |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
    return out


def erode3(m):
    p = np.pad(m, 1)
    out = np.ones_like(m)
    for dy in range(3):
        for dx in range(3):
            out &= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
    return out


def shift(m, dx, dy):
    out = np.zeros_like(m)
    h, w = m.shape
    out[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)] = m[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)]
    return out


def independent_stage_5_7(phi_v, phi_h, valid_v, valid_h, cal):
    """An INDEPENDENT fp64 NumPy restatement of stage 5 (C1, C2) and stage 7 (T0-T3), written from the reference's source
    (5/compute_correspondance.cpp:60-77,642-679; 7/triangulation.cpp:252-307,352-378,1061-1126,1134-1218) and the published
    algorithms of the five OpenCV 2.4 routines it calls -- it shares NO code with oracle/ (different language, vectorised,
    the 4x3 systems solved by LAPACK instead of the adjugate).  Nothing in the reference tree can pin these stages (the
    artefacts that would are missing blobs), so this is corroboration, not a pin: two restatements written separately agree.
    phi_*: unwrapped phase planes (float32, stage 4 -- pinned by the KATs); valid_*: per-axis valid maps.
    Returns (valid [H,W] bool, c_p_map [H,W,2] int64, points [H,W,3] float64)."""
    f64 = np.float64
    # C1: merge_valid_maps
    valid = (valid_v == 1) & (valid_h == 1)
    # C2: x = lrint(fw * (phi / (2.0*Pi))), Pi = 22.0/7.0 unparenthesised -> (2.0*22.0)/7.0; round-half-even
    two_pi = (2.0 * 22.0) / 7.0
    with np.errstate(invalid="ignore"):
        x = np.rint(f64(FW) * (phi_v.astype(f64) / two_pi))
        y = np.rint(f64(FW) * (phi_h.astype(f64) / two_pi))
    finite = np.isfinite(x) & np.isfinite(y)                       # lrint raising FE_INVALID clears the pixel
    in_range = finite & (x >= 0) & (y >= 0) & (x <= PW - 1) & (y <= PH - 1)
    valid = valid & in_range
    cp = np.zeros(phi_v.shape + (2,), dtype=np.int64)
    cp[valid, 0] = x[valid].astype(np.int64)
    cp[valid, 1] = y[valid].astype(np.int64)

    # T0: Rodrigues (cvRodrigues2, vector -> matrix) and A = K [R|t]
    def rodrigues(r):
        r = np.asarray(r, f64)
        th = np.sqrt(r @ r)
        if th < np.finfo(f64).eps:
            return np.eye(3)
        k = r / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx

    def proj_matrix(K, r, t):
        return np.asarray(K, f64).reshape(3, 3) @ np.hstack([rodrigues(r), np.asarray(t, f64).reshape(3, 1)])

    Ac, Ap = proj_matrix(cal["Kc"], cal["rc"], cal["tc"]), proj_matrix(cal["Kp"], cal["rp"], cal["tp"])

    # T1: cvUndistortPoints (5 fixed-point iterations, no R / P) then K (x, y, 1) and the division by w
    def undistort_reproject(u, v, K, d):
        K = np.asarray(K, f64).reshape(3, 3)
        k1, k2, p1, p2, k3 = [f64(c) for c in d]
        x0 = (u - K[0, 2]) / K[0, 0]
        y0 = (v - K[1, 2]) / K[1, 1]
        xx, yy = x0.copy(), y0.copy()
        for _ in range(5):
            r2 = xx * xx + yy * yy
            icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx = 2 * p1 * xx * yy + p2 * (r2 + 2 * xx * xx)
            dy = p1 * (r2 + 2 * yy * yy) + 2 * p2 * xx * yy
            xx = (x0 - dx) * icdist
            yy = (y0 - dy) * icdist
        h = np.stack([xx, yy, np.ones_like(xx)], -1) @ K.T
        return h[..., 0] / h[..., 2], h[..., 1] / h[..., 2]

    rows, cols = np.nonzero(valid)
    uc, vc = undistort_reproject(cols.astype(f64), rows.astype(f64), cal["Kc"], cal["dc"])
    up, vp = undistort_reproject(cp[rows, cols, 0].astype(f64), cp[rows, cols, 1].astype(f64), cal["Kp"], cal["dp"])
    # T2: P (4x3), F (4)
    P = np.stack([Ac[0, :3] - uc[:, None] * Ac[2, :3], Ac[1, :3] - vc[:, None] * Ac[2, :3],
                  Ap[0, :3] - up[:, None] * Ap[2, :3], Ap[1, :3] - vp[:, None] * Ap[2, :3]], axis=1)
    Fv = np.stack([Ac[2, 3] * uc - Ac[0, 3], Ac[2, 3] * vc - Ac[1, 3], Ap[2, 3] * up - Ap[0, 3], Ap[2, 3] * vp - Ap[1, 3]], axis=1)
    # T3: V = (P^T P)^-1 P^T F, here through LAPACK's solver on the normal equations
    PtP = np.einsum("nki,nkj->nij", P, P)
    PtF = np.einsum("nki,nk->ni", P, Fv)
    V = np.linalg.solve(PtP, PtF[..., None])[..., 0]
    pts = np.zeros(phi_v.shape + (3,), dtype=f64)
    pts[rows, cols] = V
    return valid, cp, pts


def main():
    ax = {0: "Vertical", 1: "Horizontal"}
    N = {0: N_V, 1: N_H}
    fringe = {a: [bmp(f"Captured_patterns/Fringe_patterns/{ax[a]}/Undistorted/Gray_captured_image_{i}.bmp")
                  for i in range(3)] for a in (0, 1)}
    gray = {a: [bmp(f"Captured_patterns/Coded_patterns/Gray_coded/{ax[a]}/Undistorted/Gray_captured_image_{i}.bmp")
                for i in range(N[a])] for a in (0, 1)}
    inv = {a: [bmp(f"Captured_patterns/Coded_patterns/Gray_coded/{ax[a]}/Undistorted/inverse_Gray_captured_image_{i}.bmp")
               for i in range(N[a])] for a in (0, 1)}
    kat3 = {a: bmp(f"Wrapped_phase_images/{ax[a]}/Wrapped_phase_image.bmp") for a in (0, 1)}
    kat4 = {0: bmp("Unwrapped_phase_images/Gray_coded/Vertical/Unwrapped_phase_vertical.bmp"),
            1: bmp("Unwrapped_phase_images/Gray_coded/Horizontal/Unwrapped_phase_horizontal.bmp")}

    # The lasso mask itself was not saved losslessly.  The final valid mask E is the non-zero set
    # of the stage-3 KAT (its formula never yields 0 on a valid pixel).  The reference's boundary
    # removal (3/wrapped_phase.cpp:266-279) is NOT a plain 3x3 erosion: invalid pixels that get
    # marked `visited` stop counting, which makes it scan-order dependent (E is not 3x3-open:
    # erode(dilate(E)) != E on 848 px, so no plain erosion could have produced it).  A selection
    # S with f(S) == E under the literal algorithm is built here: dilate E by the "later in scan
    # order" neighbours {E,SW,S,SE}, then add the earlier neighbours of any pixel still lost.
    E = (kat3[0] != 0)
    assert np.array_equal(E, kat3[1] != 0)
    print("valid pixels in KAT:", int(E.sum()))
    print("E is 3x3-open:", bool(np.array_equal(erode3(dilate3(E.astype(np.uint8))), E.astype(np.uint8))))
    orc = Oracle(W, H, PW, PH, N_V, N_H, FW, FW, ncodes_v=NCODES_V, ncodes_h=NCODES_H)
    S = E.astype(np.uint8)
    for dx, dy in ((1, 0), (-1, 1), (0, 1), (1, 1)):
        S |= shift(E.astype(np.uint8), dx, dy)
    for it in range(8):
        orc.set_mask(S)
        orc.compute_wrapped_phase(0, fringe[0])
        V = orc.valid_map(0).astype(bool)
        lost, extra = (~V & E), (V & ~E)
        print(f"mask pre-image iteration {it}: lost {int(lost.sum())} extra {int(extra.sum())}")
        if not lost.any() and not extra.any():
            break
        for y, x in zip(*np.nonzero(lost)):
            for dx, dy in ((-1, -1), (0, -1), (1, -1), (-1, 0)):
                S[y + dy, x + dx] = 1
    else:
        raise AssertionError("no selection mask reproduces the KAT's valid mask")
    assert S[0].sum() == 0 and S[-1].sum() == 0 and S[:, 0].sum() == 0 and S[:, -1].sum() == 0

    for a in (0, 1):
        orc.compute_wrapped_phase(a, fringe[a])
        d = orc.debug_image(3, a)
        bad = int((d != kat3[a]).sum())
        print(f"stage 3 axis {a}: {bad} mismatching pixels of {W*H}")
        assert bad == 0, "oracle stage 3 does not reproduce the reference KAT"
        assert np.array_equal(orc.valid_map(a).astype(bool), E)
    for a in (0, 1):
        orc.unwrap_phase(a, gray[a], inv[a])
        d = orc.debug_image(4, a)
        bad = int((d != kat4[a]).sum())
        print(f"stage 4 axis {a}: {bad} mismatching pixels of {W*H}")
        assert bad == 0, "oracle stage 4 does not reproduce the reference KAT"

    cal = {
        "Kc": xml_data("Camera_calibration/Matrices/cam_intrinsic_mat.xml"),
        "dc": xml_data("Camera_calibration/Matrices/cam_distortion_vect.xml"),
        "rc": xml_data("Triangulation/Camera_extrinsic_parametrs/world_to_cam_rot_vect.xml"),
        "tc": xml_data("Triangulation/Camera_extrinsic_parametrs/world_to_cam_trans_vect.xml"),
        "Kp": xml_data("Projector_calibration/Matrices/proj_intrinsic_mat.xml"),
        "dp": xml_data("Projector_calibration/Matrices/proj_distortion_vect.xml"),
        "rp": xml_data("Triangulation/Projector_extrinsic_parametrs/world_to_proj_rot_vect.xml"),
        "tp": xml_data("Triangulation/Projector_extrinsic_parametrs/world_to_proj_trans_vect.xml"),
    }
    cal["_source"] = "values of the 8 calibration XMLs read by 7/triangulation.cpp:152-168,1069-1083"
    cal["_dims"] = {"W": W, "H": H, "PW": PW, "PH": PH, "N_v": N_V, "N_h": N_H, "fw_v": FW, "fw_h": FW,
                    "ncodes_v": NCODES_V, "ncodes_h": NCODES_H}
    # PINS T0 (cvRodrigues2, cvTranspose, cvGEMM): stage 6 ran the same OpenCV routines on the same two rotation vectors
    # stage 7 reads and saved the result (6/system_calibration.cpp:1488-1516): Rc*Rp^T and tc - (Rc*Rp^T)*tp.  The oracle's
    # restatements must reproduce all 12 doubles of the two files BIT FOR BIT before any golden is written.
    from oracle import oracle as O
    kat_R = np.array(xml_data("Triangulation/Relative_geometry/proj_cam_rot_mat.xml"))
    kat_t = np.array(xml_data("Triangulation/Relative_geometry/proj_cam_trans_vect.xml"))
    got_R, got_t = O.relative_geometry(cal["rc"], cal["tc"], cal["rp"], cal["tp"])
    same = int((got_R.ravel().view(np.uint64) == kat_R.view(np.uint64)).sum() + (got_t.view(np.uint64) == kat_t.view(np.uint64)).sum())
    print(f"T0 known answer (Relative_geometry/*.xml, OpenCV 2.4's own output): {same} of 12 doubles bit-identical")
    assert same == 12, "oracle rodrigues / transpose / mat_mul do not reproduce the reference's saved relative geometry"
    cal["_relative_geometry"] = {
        "proj_cam_rot_mat": [float(x) for x in kat_R], "proj_cam_trans_vect": [float(x) for x in kat_t],
        "_source": "Triangulation/Relative_geometry/proj_cam_rot_mat.xml + proj_cam_trans_vect.xml, written by "
                   "6/system_calibration.cpp:1488-1516 (cvRodrigues2 x2, cvTranspose, cvMatMul x2, cvSub) from the same rc/rp/tc/tp "
                   "7/triangulation.cpp:1069-1083 reads: the reference-held known answer for T0"}
    # the reference's OTHER projector calibrations: two OpenCV calibrations of the projectors it was used with, both strongly
    # distorted (k1 = -1.01 / -1.16, k2 = 8.28 / 2.60) -- stage 7 reads whatever proj_intrinsic_mat.xml / proj_distortion_vect.xml
    # hold (7/triangulation.cpp:152-168), so these are inputs the path really meets (tests/test_gpu_round4.py: table rig)
    cal["_alt_projectors"] = {
        name: {"Kp": xml_data(f"Projector_calibration/Matrices/OPencv calib/{name}/Projector_intrinsic_mat.xml"),
               "dp": xml_data(f"Projector_calibration/Matrices/OPencv calib/{name}/Projector_dist_vect.xml")}
        for name in ("Sharp", "Viewsonic")}
    cal["_alt_projectors"]["_source"] = "Projector_calibration/Matrices/OPencv calib/{Sharp,Viewsonic}/Projector_{intrinsic_mat,dist_vect}.xml"
    with open(os.path.join(HERE, "calibration.json"), "w") as f:
        json.dump(cal, f, indent=1)

    orc.set_calibration(*[cal[k] for k in ("Kc", "dc", "rc", "tc", "Kp", "dp", "rp", "tp")])
    orc.compute_c_p_map()
    orc.triangulate()
    valid = orc.valid_map(2)
    cp = orc.c_p_map()
    pts = orc.intersection_points()
    print("in-range correspondences:", int(valid.sum()), "mean xyz:", pts[valid == 1].mean(axis=0))
    # corroboration of the unpinned stages: the independent NumPy restatement on the SAME full 1600x1200 scan must give the
    # identical correspondences and the same points before any golden is written
    iv, icp, ipts = independent_stage_5_7(orc.unwrapped_phi(0), orc.unwrapped_phi(1), orc.valid_map(0), orc.valid_map(1), cal)
    assert np.array_equal(iv, valid == 1), "independent restatement: valid map after stage 5 differs"
    assert np.array_equal(icp[iv], cp[iv]), "independent restatement: correspondences differ"
    rel = np.linalg.norm(ipts[iv] - pts[iv], axis=-1) / np.linalg.norm(pts[iv], axis=-1)
    print(f"independent NumPy restatement of stages 5 + 7: {int(iv.sum())} identical correspondences, "
          f"max relative point difference {rel.max():.3e} (median {np.median(rel):.1e})")
    assert int(iv.sum()) == 355608, "SURVEY's count of in-range correspondences on the real scan"
    assert rel.max() <= 1e-12, "independent restatement: 3-D points differ by more than 1e-12 of their norm"
    A_cam, A_proj = orc.projection_matrices()

    ys, xs = np.nonzero(E)
    print("valid bbox x:[%d,%d] y:[%d,%d]" % (xs.min(), xs.max(), ys.min(), ys.max()))
    # crop A: fully inside the valid region; crop B: straddles the lasso boundary (erosion edge,
    # invalid pixels); both 128x64.
    crops = {"real_inside": (800, 400), "real_edge": (int(xs.min()) - 40, int(ys[xs == xs.min()][0]) - 32)}
    CW, CH = 128, 64
    for name, (x0, y0) in crops.items():
        x0 -= x0 % 4
        sl = np.s_[y0:y0 + CH, x0:x0 + CW]
        frac = E[sl].mean()
        print(f"crop {name}: origin ({x0},{y0}) valid fraction {frac:.3f}")
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            origin=np.array([x0, y0]), full=np.array([W, H, PW, PH]),
            params=np.array([N_V, N_H, FW, FW, NCODES_V, NCODES_H]),
            mask=S[sl], mask_halo2=S[y0 - 2:y0 + CH + 2, x0 - 2:x0 + CW + 2],
            fringe_v=np.stack([f[sl] for f in fringe[0]]), fringe_h=np.stack([f[sl] for f in fringe[1]]),
            gray_v=np.stack([f[sl] for f in gray[0]]), inv_v=np.stack([f[sl] for f in inv[0]]),
            gray_h=np.stack([f[sl] for f in gray[1]]), inv_h=np.stack([f[sl] for f in inv[1]]),
            # reference-provided answers (PINNED): crops of the reference's own KAT images
            kat_wrapped_v=kat3[0][sl], kat_wrapped_h=kat3[1][sl],
            kat_unwrapped_v=kat4[0][sl], kat_unwrapped_h=kat4[1][sl],
            # oracle outputs of the full-frame run (regression goldens for the unpinned stages)
            valid=valid[sl], code_v=orc.code(0)[sl], code_h=orc.code(1)[sl],
            wrapped_v=orc.wrapped_phi(0)[sl], wrapped_h=orc.wrapped_phi(1)[sl],
            unwrapped_v=orc.unwrapped_phi(0)[sl], unwrapped_h=orc.unwrapped_phi(1)[sl],
            c_p_map=cp[sl], points=pts[sl], A_cam=A_cam, A_proj=A_proj,
        )
    print("golden fixtures written to", HERE)


def patterns():
    """N1: pins the oracle's pattern generator on the reference's own pattern images
    (M_tech_project_console/Generated_patterns, written by 1/pattern_generator.cpp:414-470 for 1280x720, 3 fringe
    patterns, fringe width 32 on both axes) and writes tests/golden/patterns_ref.npz: every pattern is constant along
    one axis (checked here), so its 1-D profile is the whole image; plus the 1078-byte BMP header + palette the
    reference's cvSaveImage wrote and the SHA-256 of every file (the shim's generate_pattern() reproduces the files)."""
    import hashlib
    from oracle import oracle as O
    root = REF + "Generated_patterns/"
    fw, F = FW, 3
    out = {"config": np.array([PW, PH, F, fw, fw])}
    names, hashes = [], []
    header = None
    n_files = 0
    for axis, ax_name, extent in ((0, "Vertical", PW), (1, "Horizontal", PH)):
        ncodes, nplanes = O.pattern_counts(extent, fw)
        assert (ncodes, nplanes) == ((NCODES_V, N_V) if axis == 0 else (NCODES_H, N_H))
        for kind, key, fmt, count in ((O.PATTERN_FRINGE, "fringe", "Fringe_patterns/%s/Pattern_%d.bmp", F),
                                      (O.PATTERN_GRAY, "gray", "Coded_patterns/Gray_coded/%s/Pattern_%d.bmp", nplanes + 1),
                                      (O.PATTERN_INVERSE_GRAY, "inverse", "Coded_patterns/Gray_coded/%s/inverse_Pattern_%d.bmp", nplanes + 1),
                                      (O.PATTERN_BINARY, "binary", "Coded_patterns/Binary_coded/%s/Pattern_%d.bmp", nplanes + 1)):
            for i in range(count):
                rel = fmt % (ax_name, i)
                raw = open(root + rel, "rb").read()
                im = Image.open(root + rel)
                assert im.mode == "L" and im.size == (PW, PH), (rel, im.mode, im.size)
                kat = np.array(im)
                prof = kat[0, :] if axis == 0 else kat[:, 0]
                assert np.array_equal(kat, np.broadcast_to(prof[None, :] if axis == 0 else prof[:, None], kat.shape)), rel
                got = O.pattern_image(kind, axis, i, PW, PH, fw, nplanes, F)
                assert np.array_equal(got, kat), f"oracle pattern differs from the reference's {rel}: {(got != kat).sum()} px"
                out[f"{key}_{'vh'[axis]}_{i}"] = prof.copy()
                names.append(rel)
                hashes.append(hashlib.sha256(raw).hexdigest())
                header = raw[:1078] if header is None else header
                assert raw[:1078] == header and len(raw) == 1078 + PW * PH
                n_files += 1
    out["bmp_header"] = np.frombuffer(header, dtype=np.uint8)
    out["file_names"] = np.array(names)
    out["file_sha256"] = np.array(hashes)
    np.savez_compressed(os.path.join(HERE, "patterns_ref.npz"), **out)
    print(f"pattern generator pinned on {n_files} reference images (every pixel equal); patterns_ref.npz written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "patterns":
        patterns()
    else:
        main()
        patterns()
