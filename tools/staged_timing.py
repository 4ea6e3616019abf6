#!/usr/bin/env python3
"""Parity-mode timings at the reference's own size (1600x1200 camera, 1280x720 projector, N = 6/5, fringe width 32):
the four per-stage entry points in main()'s order (what the shim calls), the fused kernel in parity mode (all stage planes
stored), and the timed mode, one view."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
syn = importlib.import_module("3dscan_amd.synth"); scm = importlib.import_module("3dscan_amd.scanner")
W, H, PW, PH, NV, NH, fw = 1600, 1200, 1280, 720, 6, 5, 32
cal = syn.cal_tuple(syn.synth_rig(W, H, PW, PH)); mask = syn.default_mask(W, H)
def timed(f, n=200):
    for _ in range(20): f()
    sc.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    sc.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for keep in (True, False):
    sc = scm.Scanner(W, H, PW, PH, NV, NH, fw, fw, keep_stages=keep)
    sc.set_calibration(*cal); sc.set_mask(mask); sc.synth_view(0, noise=2)
    if keep:
        print(f"per-stage entry points (6 launches): {timed(sc.run_stages):8.1f} us per scan")
        print(f"fused kernel, parity mode:           {timed(sc.run):8.1f} us per scan")
    else:
        t = timed(sc.run)
        print(f"fused kernel, timed mode:            {t:8.1f} us per scan  ({W * H / t / 1e3:.1f} Gpx/s on ONE view)")
    sc.close()
