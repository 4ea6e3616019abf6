#!/usr/bin/env python3
"""One view per launch from HBM (8 resident views) against the fraction of the frame the selection mask covers: a centred rectangle of
100 % / 50 % / 19 % (the reference's real captures select 358,580 of 1,920,000 pixels) / 5 % of the pixels.  Small launches request
their first view's planes before the mask is known (EARLY): with a sparse mask that is traffic for nothing.
    python3 tools/sparse_mask.py            (SL3D_NO_SMALL=1 with a -DSL3D_MEASURE build: the large-launch kernel for one view)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
scm = importlib.import_module("3dscan_amd.scanner")
syn = importlib.import_module("3dscan_amd.synth")

W, H, N, fw, R = 1920, 1080, 10, 2, 8
with scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=R) as sc:
    sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, W, H)))
    for v in range(R):
        sc.set_mask(syn.default_mask(W, H), view=v)
        sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
    for frac in (1.0, 0.5, 0.19, 0.05):
        m = np.zeros((H, W), np.uint8)
        s = frac ** 0.5
        h, w = int(H * s), int(W * s)
        y0, x0 = (H - h) // 2, (W - w) // 2
        m[y0:y0 + h, x0:x0 + w] = 1
        m[0, :] = m[-1, :] = 0
        m[:, 0] = m[:, -1] = 0
        for v in range(R):
            sc.set_mask(m, view=v)
        sc.synchronize()
        for n_views, clouds in ((1, False), (8, False), (1, True), (8, True)):
            if clouds:
                run = (lambda i: sc.run_clouds(i % R, 1)) if n_views == 1 else (lambda i: sc.run_clouds(0, 8))
            else:
                run = (lambda i: sc.run(i % R, 1)) if n_views == 1 else (lambda i: sc.run(0, 8))
            reps = 3000 if n_views == 1 else 600
            for i in range(reps // 3):
                run(i)
            sc.synchronize()
            sc.timer_start()
            for i in range(reps):
                run(i)
            ms = sc.timer_stop() / reps
            what = "ordered clouds (sl3d_run_clouds)" if clouds else "dense planes   (sl3d_run)       "
            print(f"mask covers {frac * 100:5.1f} %: {what} {n_views} view(s) per launch {ms * 1e3 / n_views:7.2f} us per view   {sc.fused_kernel_name(n_views, clouds=clouds)}")
