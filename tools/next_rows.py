#!/usr/bin/env python3
"""Exercises the kernels of the "next" rows at 1920x1080 so that `rocprofv3 --kernel-trace --stats -- python3 tools/next_rows.py`
gives their durations: N1 k_pattern / k_synth, N2 k_compact_*, N3 k_register, N4 k_undist_map / k_undist_remap, and the
projector undistortion table k_proj_table."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

syn = importlib.import_module("3dscan_amd.synth")
scm = importlib.import_module("3dscan_amd.scanner")
W, H, N, fw, V = 1920, 1080, 10, 2, 4
cal = syn.synth_rig(W, H, W, H)
cal["dp"] = np.array([-0.05, 0.02, 0.001, -0.0005, 0.0])  # distorted projector: k_proj_table runs
sc = scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=V)
sc.set_calibration(*syn.cal_tuple(cal))
mask = syn.default_mask(W, H)
for v in range(V):
    sc.set_mask(mask, view=v)
    sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05), view_id=v, noise=2)
rng = np.random.default_rng(1)
for rep in range(5):
    sc.run(0, V)
    sc.compact_views(0, V)
    sc.register_views(0, V, 60.0, 35.0, 0.0, 10.0)
    sc.generate_pattern(scm.PATTERN_FRINGE, 0, rep % 3)
    sc.generate_pattern(scm.PATTERN_GRAY, 1, rep % 5)
img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
for rep in range(5):
    sc.undistort(img, cal["Kc"], cal["dc"])
    sc.undistort(img[..., 0], cal["Kc"], cal["dc"])
sc.close()
print("done")
