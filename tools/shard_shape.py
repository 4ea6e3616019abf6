#!/usr/bin/env python3
"""Per-GPU rate for the stripe shapes bench.py --gpus N gives each rank (rows = 1080/N of every view, 16*N views per launch):
what weak scaling looks like from one GPU's point of view, measured on ONE GPU.   python tools/shard_shape.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (its HIP runtime first)

syn = importlib.import_module("3dscan_amd.synth")
scm = importlib.import_module("3dscan_amd.scanner")
W, H, N, fw = 1920, 1080, 10, 2
cal = syn.cal_tuple(syn.synth_rig(W, H, W, H))
mask = syn.default_mask(W, H)
for world in (1, 2, 4, 8):
    rows, nv = H // world, 16 * world
    row0 = rows * (world // 2)  # a middle stripe
    sc = scm.Scanner(W, rows, W, H, N, N, fw, fw, max_views=nv, full_size=(W, H), origin=(0, row0))
    sc.set_calibration(*cal)
    for v in range(nv):
        sc.set_mask(mask, view=v)
        sc.synth_view(v, plane=(0.75 * (v % 16), 0.05, 0.05 - 0.003 * (v % 16)), view_id=v, noise=2)
    sc.synchronize()
    for _ in range(300):
        sc.run(0, nv)
    sc.synchronize()
    t0 = time.perf_counter()
    steps = 1500
    for _ in range(steps):
        sc.run(0, nv)
    sc.synchronize()
    dt = time.perf_counter() - t0
    print(f"world {world}: {rows} rows x {nv} views per GPU: {nv * rows * W * steps / dt / 1e9:.1f} Gpx/s per GPU, {dt / steps * 1e3:.4f} ms per launch", flush=True)
    sc.close()
