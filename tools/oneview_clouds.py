#!/usr/bin/env python3
"""One view per launch from HBM WITH the ordered cloud (sl3d_run_clouds: the fused kernel's segmented clouds + k_seg_scan) -- the
reference's real call pattern is one scan followed by save_point_cloud().  Prints the launch-to-launch time (HIP events); under
`rocprofv3 --kernel-trace --stats -- python3 tools/oneview_clouds.py` the per-kernel averages tell what the second launch costs."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
scm = importlib.import_module("3dscan_amd.scanner")
syn = importlib.import_module("3dscan_amd.synth")

W, H, N, fw, R = 1920, 1080, 10, 2, 8
with scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=R) as sc:
    sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, W, H)))
    m = syn.default_mask(W, H)
    for v in range(R):
        sc.set_mask(m, view=v)
        sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
    for what, run in (("dense", lambda v: sc.run(v, 1)), ("clouds", lambda v: sc.run_clouds(v, 1))):
        for i in range(1000):
            run(i % R)
        sc.synchronize()
        sc.timer_start()
        for i in range(3000):
            run(i % R)
        ms = sc.timer_stop() / 3000
        print(f"one view per launch from HBM, {what}: {ms * 1e3:.2f} us per launch")
    # the cloud where the reference's consumer wants it (a host cloud per scan): launch + download into pinned memory, wall clock
    import time
    import numpy as np
    pin = sc.pinned((W * H * 3,), np.float32)
    ts = []
    for i in range(40):
        t0 = time.perf_counter()
        sc.run_clouds(i % R, 1)
        n = sc.download_cloud_into(i % R, pin)
        ts.append(time.perf_counter() - t0)
    print(f"one view: sl3d_run_clouds + sl3d_download_clouds into pinned memory: {sorted(ts)[20] * 1e6:.1f} us per scan ({n} points)")
