#!/usr/bin/env python3
"""Launch times of the corners of the fused kernel that no benchmark line covers (each of them is parity-tested; this prints what they
COST): the general rig class 0 in the timed mode (a camera matrix with K[1][0] != 0: the un-pipelined kernel), a row stripe of
BASELINE config 4 (135 rows of 64 views), 13 Gray planes per axis (the kernels with per-plane tests), the parity mode (every
stage-boundary plane written).  1920x1080, N = 10 unless stated.
    python3 tools/corners.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (its HIP runtime first)
scm = importlib.import_module("3dscan_amd.scanner")
syn = importlib.import_module("3dscan_amd.synth")


def timed(sc, V, first=0):
    sc.timer_start()
    sc.run(first, V)
    reps = 600 if sc.timer_stop() < 3.0 else 10
    for _ in range(reps // 2):
        sc.run(first, V)
    sc.synchronize()
    sc.timer_start()
    for _ in range(reps):
        sc.run(first, V)
    return sc.timer_stop() / reps


def case(label, W, H, PW, PH, N, fw, V, cal_edit=None, keep=False, full=None, origin=(0, 0)):
    cal_d = syn.synth_rig(full[0] if full else W, full[1] if full else H, PW, PH)
    if cal_edit:
        cal_edit(cal_d)
    kw = dict(max_views=V, keep_stages=keep)
    if full:
        kw.update(full_size=full, origin=origin)
    with scm.Scanner(W, H, PW, PH, N, N, fw, fw, **kw) as sc:
        sc.set_calibration(*syn.cal_tuple(cal_d))
        m = syn.default_mask(full[0] if full else W, full[1] if full else H)
        for v in range(V):
            sc.set_mask(m, view=v)
            sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
        ms = timed(sc, V)
        alg = 2 * 3 + 4 * N + 14
        print(f"{label:58s} {ms * 1e3:9.1f} us per launch   frac of 8 TB/s on {alg} B/px {alg * V * W * H / (ms * 1e-3) / 8e12:.3f}   {sc.fused_kernel_name(V)}")


def perspective(c):
    K = np.array(c["Kc"], dtype=np.float64).reshape(3, 3).copy()
    K[1, 0] = 1e-3
    c["Kc"] = K.ravel()


W, H = 1920, 1080
case("reference rig, 16 views (the bench line)", W, H, W, H, 10, 2, 16)
case("rig class 0 (K[1][0] != 0), 16 views", W, H, W, H, 10, 2, 16, cal_edit=perspective)
case("rig class 0, 1 view", W, H, W, H, 10, 2, 1, cal_edit=perspective)
case("config 4 stripe: rows 405..539 of 64 views", W, 135, W, H, 10, 2, 64, full=(W, H), origin=(0, 405))
case("13 Gray planes per axis (per-plane tests), 16 views", W, H, W, H, 13, 1, 16)
case("parity mode (every stage plane), 1 view", W, H, W, H, 10, 2, 1, keep=True)
case("parity mode, 4 views", W, H, W, H, 10, 2, 4, keep=True)
