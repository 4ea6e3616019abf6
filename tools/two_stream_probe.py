"""Probe: do launches of two contexts on two HIP streams overlap their ramp / tail?  Per-launch average of alternating launches on two streams
against the same number of launches on one stream (resident views round robin, frames from HBM), for 1 / 2 / 4 / 16 views per launch.
usage: two_stream_probe.py"""
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
W, H, N, fw, V = 1920, 1080, 10, 2, 16
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


def loop(n, two, k):
    groups = V // k
    for i in range(n):
        c = ctxs[i & 1 if two else 0]
        c.run((((i // 2) if two else i) % groups) * k, k)


for k in (1, 2, 4, 16):
    n = 4000 // k
    for two in (False, True, False, True):
        loop(n // 2, two, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(n, two, k)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        print(f"{k:2d} view(s) per launch,", ("two streams" if two else "one stream "), f"{us:.2f} us per launch = {us / k:.2f} per view")
