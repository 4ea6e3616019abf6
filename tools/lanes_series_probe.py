import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
scm = importlib.import_module("3dscan_amd.scanner"); syn = importlib.import_module("3dscan_amd.synth")
W, H, N, fw, V = 1920, 1080, 10, 2, 8
sc = scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=V)
sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, W, H)))
m = syn.default_mask(W, H)
for v in range(V):
    sc.set_mask(m, view=v)
    sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
sc.synchronize()
def series(n):
    sc.timer_start()
    for i in range(n): sc.run(i % V, 1)
    return sc.timer_stop() / n * 1e3
for rep in range(3):
    for n in (40, 40, 400, 2000, 2000):
        print(n, round(series(n), 2), end=" | ")
    print()
