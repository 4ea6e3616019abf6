#!/usr/bin/env python3
"""Phase trace of ONE dense launch of the fused kernel (measurement build: tools/ab.sh build trace "-DSL3D_MEASURE -DSL3D_TRACE"):
every wave stamps the 100 MHz wall clock at its phase boundaries (sl3d_kernels.hip, SL3D_STAMP).  Prints where the time of a
launch goes -- start-up, the rounds of blocks, how many waves sit in which phase at every microsecond -- i.e. what bounds the
one-view launch (the reference's real usage: one scan per call).
    SL3D_LIB=$PWD/ab/libsl3d_trace.so python3 tools/phase_trace.py [views] [cold] [maskin]   -> gpurun_out/phase_trace_<views>.npz + a summary
`cold` (with 1 view per launch): 8 views are resident and every launch takes the next one, so the traced launch reads its frames
from HBM, not from the Infinity Cache (side.one_view_cold of bench.py).  `maskin`: every launch is preceded by a new device-resident
selection (side.per_scan_device): the traced launch is a MASKIN launch; "item set-up" then contains the evaluation of the selection."""
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

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1
COLD = "cold" in sys.argv[2:]
MASKIN = "maskin" in sys.argv[2:]
R = 8 if COLD else 1            # resident batches the launches rotate over
W, H, N, fw = 1920, 1080, 10, 2
sc = scm.Scanner(W, H, W, H, N, N, fw, fw, max_views=V * R)
sc.set_calibration(*syn.cal_tuple(syn.synth_rig(W, H, W, H)))
m = syn.default_mask(W, H)
for v in range(V * R):
    sc.set_mask(m, view=v)
    sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=2)
d_m = torch.from_numpy(m).cuda() if MASKIN else None


def launch(i):
    if MASKIN:
        sc.set_masks_device(d_m.data_ptr(), W, 0, (i % R) * V, V)
    sc.run((i % R) * V, V)


for i in range(500):          # clocks up: the launch that is traced is one of a back-to-back series
    launch(i)
sc.synchronize()
sc.timer_start()
for i in range(200):
    launch(i)
ms = sc.timer_stop() / 200
print("kernel:", sc.last_fused_kernel_name())
sc.synchronize()
L = sc.L
dev, nbytes, nt = C.c_void_p(), C.c_size_t(), C.c_int()
L.sl3d_debug_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
assert L.sl3d_debug_buffer(sc._h, C.byref(dev), C.byref(nbytes), C.byref(nt)) == 0, "not a -DSL3D_TRACE build"
a = np.empty(nbytes.value // 8, dtype=np.uint64)
sc._d2h(a, dev.value)
a = a.reshape(-1, 8).astype(np.int64)
a = a[(a > 0).all(axis=1)]                      # waves that ran (padded blocks / lanes past the last row never stamp)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"phase_trace_{V}.npz"), stamps=a, ms_per_launch=ms)
t = (a - a[:, 0].min()) / 100.0                  # microseconds since the first wave entered the kernel
names = ["reciprocal table", "item set-up", "issue plane loads", "wait planes + decode", "phase A (stages 3-5)", "phase B (stage 7)", "issue stores"]
print(f"{V} view(s) per launch, back to back{' (8 resident batches round robin: frames from HBM)' if COLD else ''}: {ms * 1e3:.2f} us per launch (HIP events, 200 launches); traced launch: "
      f"{len(t)} waves, first entry -> last store issued {t[:, 7].max():.2f} us")
print("per-wave phase durations (us): median / p90")
for k, n in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print(f"  {n:26s} {np.median(d):6.2f} / {np.percentile(d, 90):6.2f}")
life = t[:, 7] - t[:, 0]
print(f"  {'wave life':26s} {np.median(life):6.2f} / {np.percentile(life, 90):6.2f}")
start = np.sort(t[:, 0])
print("waves entering the kernel per microsecond: " + " ".join(str(int(((start >= u) & (start < u + 1)).sum())) for u in range(int(start.max()) + 1)))
print("  t(us)  loading  decode+A  B  (waves in each state at t; 'loading' = plane loads issued, not landed)")
for u in np.arange(0.5, t[:, 7].max(), 1.0):
    loading = ((t[:, 2] <= u) & (t[:, 4] > u)).sum()
    a_ = ((t[:, 4] <= u) & (t[:, 5] > u)).sum()
    b_ = ((t[:, 5] <= u) & (t[:, 7] > u)).sum()
    print(f"  {u:5.1f}  {loading:6d}  {a_:6d}  {b_:6d}")
# bytes whose loads have LANDED by time u (46 dwords x 64 lanes per wave) -> achieved read bandwidth over the launch
landed = np.sort(t[:, 4])
tot = len(landed) * 46 * 256
for frac in (0.25, 0.5, 0.75, 1.0):
    k = int(frac * len(landed)) - 1
    print(f"  {int(frac * 100):3d}% of the frame bytes had landed after {landed[k]:6.2f} us  ({tot * frac / landed[k] / 1e6:6.2f} TB/s since entry)")
