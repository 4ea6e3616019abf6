import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
scm = importlib.import_module("3dscan_amd.scanner"); syn = importlib.import_module("3dscan_amd.synth")
W, H, N, fw, R = 1920, 1080, 10, 2, 8
with scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=R) as sc:
    sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, W, H)))
    for v in range(R):
        sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
    for frac in (0.5, 0.19, 0.05):
        m = np.zeros((H, W), np.uint8); s = frac ** 0.5; h, w = int(H * s), int(W * s); y0, x0 = (H - h) // 2, (W - w) // 2
        m[y0:y0 + h, x0:x0 + w] = 1
        sc.set_masks(m, 0, R); sc.synchronize()
        for nv in (2, 4):
            run = lambda i: sc.run((i * nv) % R, nv)
            for i in range(500): run(i)
            sc.synchronize(); sc.timer_start()
            for i in range(2000): run(i)
            ms = sc.timer_stop() / 2000
            print(f"mask {frac*100:4.0f} %: {nv} views per launch {ms*1e3/nv:6.2f} us per view  {sc.last_fused_kernel_name()}")
