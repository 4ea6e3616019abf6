"""Probe: do one-view launches of two contexts on two HIP streams overlap their ramp / tail?  Per-launch average of alternating launches on
two streams against the same number of launches on one stream (8 resident views each, frames from HBM).   usage: two_stream_probe.py"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

scm = importlib.import_module("3dscan_amd.scanner")
syn = importlib.import_module("3dscan_amd.synth")
W, H, N, fw, V = 1920, 1080, 10, 2, 8
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs = []
for s in streams:
    sc = scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=V, stream=s.cuda_stream)
    sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, W, H)))
    sc.set_masks(syn.default_mask(W, H), 0, V)
    for v in range(V):
        sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
    sc.synchronize()
    ctxs.append(sc)


def loop(n, two):
    for i in range(n):
        ctxs[i & 1 if two else 0].run((i // 2) % V if two else i % V, 1)


for two in (False, True, False, True):
    loop(2000, two)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(4000, two)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 4000 * 1e6
    print(("two streams" if two else "one stream "), f"{us:.2f} us per one-view launch (wall clock over 4000 launches)")
