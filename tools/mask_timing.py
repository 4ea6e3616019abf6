"""k_mask_prepare at one frame size: bench.py's side.per_scan_device (new device-resident mask + one-view launch, cold views) plus
the host-side hand-over (pinned / pageable, call and until-ready).  Under `rocprofv3 --kernel-trace --stats` the kernel table gives
the mask kernel's average per view (tools/recipes/mask.sh).   usage: mask_timing.py WIDTH HEIGHT"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench

W, H = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], "--width", str(W), "--height", str(H)] + (["--fringe-width", "4", "--cold-views", "3"] if W > 2048 else [])
args = bench.parse()
syn = importlib.import_module("3dscan_amd.synth")
scm = importlib.import_module("3dscan_amd.scanner")
LASSO = os.environ.get("LASSO", "0") == "1"   # the reference's 19 % selections: the gated kernels
CLOUDS = os.environ.get("PERSCAN_CLOUDS", "0") == "1"   # the ordered cloud instead of the dense planes
out = {"size": [W, H], "per_scan_device": bench.per_scan_device(args, scm, syn, np, torch, 0, lasso=LASSO, clouds=CLOUDS)}
full = syn.default_mask(W, H)
with scm.Scanner(W, H, W, H, args.ngray, args.ngray, args.fringe_width, args.fringe_width, max_views=16) as sc:
    pm = sc.pinned(full.shape, np.uint8)
    pm[:] = full
    for name, src in (("pinned", pm), ("pageable", full)):
        sc.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            sc.set_mask(src, view=0)
            tc = time.perf_counter() - t0
            sc.synchronize()
            ts.append((tc, time.perf_counter() - t0))
        out.setdefault("set_mask_us", {})[name] = {"call": round(sorted(x[0] for x in ts)[15] * 1e6, 1), "until_ready": round(sorted(x[1] for x in ts)[15] * 1e6, 1)}
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        sc.set_masks(pm)          # 16 views, one copy, one launch
        sc.synchronize()
        ts.append(time.perf_counter() - t0)
    out["set_masks_16_views_same_pinned_mask_until_ready_us"] = round(sorted(ts)[5] * 1e6, 1)
print(json.dumps(out))
