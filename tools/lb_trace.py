#!/usr/bin/env python3
"""Look-back trace of ONE sl3d_run_clouds launch (measurement build: tools/ab.sh build trace "-DSL3D_MEASURE -DSL3D_CX=128"):
per (view, tile) the clock at which the tile published its count, started and finished its look-back, and the XCD it ran on.
    SL3D_LIB=$PWD/ab/libsl3d_trace.so python3 tools/lb_trace.py  -> gpurun_out/lb_trace.npz + a summary"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (its HIP runtime first)
scm = importlib.import_module("3dscan_amd.scanner")
syn = importlib.import_module("3dscan_amd.synth")

W, H, N, fw, V = 1920, 1080, 10, 2, 16
sc = scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=V)
sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, W, H)))
m = syn.default_mask(W, H)
for v in range(V):
    sc.set_mask(m, view=v)
    sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
for _ in range(200):
    sc.run_clouds(0, V)
sc.cloud_counts(0, V)
sc.run_clouds(0, V)
sc.cloud_counts(0, V)
L = sc.L
dev, nbytes, nt = C.c_void_p(), C.c_size_t(), C.c_int()
L.sl3d_debug_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
assert L.sl3d_debug_buffer(sc._h, C.byref(dev), C.byref(nbytes), C.byref(nt)) == 0
a = np.empty(nbytes.value // 8, dtype=np.uint64)
sc._d2h(a, dev.value)
a = a.reshape(V, nt.value, 4)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "lb_trace.npz"), trace=a)
t0 = a[..., 0].astype(np.int64); ts = a[..., 1].astype(np.int64); te = a[..., 2].astype(np.int64)
base = t0.min()
us = lambda x: (x - base) / 100.0   # wall_clock64 ticks at 100 MHz
print("tiles/view", nt.value, "kernel span us", us(te.max()))
for v in (0, 3, 7, 8, 15):
    pub, st, en = us(t0[v]), us(ts[v]), us(te[v])
    d = np.diff(pub)
    print(f"view {v}: publish first/last {pub.min():.1f}/{pub.max():.1f} us; look-back duration mean {np.mean(en - st):.2f} p50 {np.median(en - st):.2f} p90 {np.percentile(en - st, 90):.2f} max {np.max(en - st):.2f};"
          f" slack (look-back start - own publish) mean {np.mean(st - pub):.2f}; predecessor lag (pub[t-1]-pub[t]) p50 {np.median(-d):.2f} p90 {np.percentile(-d, 90):.2f} p99 {np.percentile(-d, 99):.2f} max {np.max(-d):.2f}")
    # how late is the latest of the 32 nearest predecessors relative to my look-back start
    k = 32
    lat = np.array([pub[max(0, t - k):t].max() - st[t] for t in range(1, nt.value)])
    print(f"        latest publish among the {k} nearest predecessors minus my look-back start: p50 {np.median(lat):.2f} p90 {np.percentile(lat, 90):.2f} p99 {np.percentile(lat, 99):.2f} (positive = I have to wait)")
xcc = (a[..., 3] & np.uint64(0xF)).astype(int)
for x in range(8):
    sel = xcc[0] == x
    print("xcd", x, "tiles", int(sel.sum()), "mean publish us (view 0)", float(np.mean(us(t0[0])[sel])) if sel.any() else None)
