#!/usr/bin/env python3
"""Launch time of the fused kernel for pattern sets with DIFFERENT numbers of Gray planes on the two axes (the reference's own
capture set: 1600x1200 camera, 1280x720 projector, fringe width 32, N_v = 6, N_h = 5 -- global_cv.h:49-62) next to the exact
instantiations (N_v = N_h).  Prints the instantiation, microseconds per launch and the fraction of the 8 TB/s roofline on the
algorithmic bytes 2*3 + 2*N_v + 2*N_h + 14 per pixel.
    python3 tools/nvnh.py [views per launch] [fringe patterns per axis: 3 (default) or 4]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (its HIP runtime first)
scm = importlib.import_module("3dscan_amd.scanner")
syn = importlib.import_module("3dscan_amd.synth")

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H, PW, PH, fw = 1600, 1200, 1280, 720, 32
for Nv, Nh in ((6, 5), (6, 6), (7, 6), (7, 7), (10, 9), (10, 10), (12, 7), (12, 12)):
    with scm.Scanner(W, H, PW, PH, Nv, Nh, fw, fw, max_views=V, n_fringe=F) as sc:
        sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, PW, PH)))
        m = syn.default_mask(W, H)
        for v in range(V):
            sc.set_mask(m, view=v)
            sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
        sc.timer_start()
        sc.run(0, V)
        reps = 1000 if sc.timer_stop() < 2.0 else 20   # (a first launch of milliseconds: a slow corner -- do not spend a minute on it)
        for _ in range(reps // 2):
            sc.run(0, V)
        sc.synchronize()
        sc.timer_start()
        for _ in range(reps):
            sc.run(0, V)
        ms = sc.timer_stop() / reps
        alg = 2 * F + 2 * Nv + 2 * Nh + 14
        print(f"F={F} N_v={Nv:2d} N_h={Nh:2d} {V} views {W}x{H}: {ms * 1e3:7.1f} us per launch, {alg} B/px, frac of 8 TB/s {alg * V * W * H / (ms * 1e-3) / 8e12:.3f}   {sc.fused_kernel_name(V)}")
