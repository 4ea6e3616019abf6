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
 3. Runs oracle stages 5 and 7 on the full frames with the reference's 8 calibration XMLs
    (unpinned stages: these outputs are regression goldens, not reference answers).
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
    with open(os.path.join(HERE, "calibration.json"), "w") as f:
        json.dump(cal, f, indent=1)

    orc.set_calibration(*[cal[k] for k in ("Kc", "dc", "rc", "tc", "Kp", "dp", "rp", "tp")])
    orc.compute_c_p_map()
    orc.triangulate()
    valid = orc.valid_map(2)
    cp = orc.c_p_map()
    pts = orc.intersection_points()
    print("in-range correspondences:", int(valid.sum()), "mean xyz:", pts[valid == 1].mean(axis=0))
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
