#!/usr/bin/env python3
"""Randomised parity run on the GPU: random camera / projector sizes, windows (crops and row stripes of a larger frame),
Gray depths, fringe widths, fringe counts, masks, noise and rigs; every case compares the timed fused kernel (points, valid)
and the parity mode (codes, phases, correspondences) with the oracle.   python tests/fuzz_parity.py [cases] [seed]
(lives under tests/: it imports the oracle, which only tests, smoke() and the bench's cpu_baseline leg may do)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from oracle.oracle import Oracle
syn = importlib.import_module("3dscan_amd.synth"); S = importlib.import_module("3dscan_amd.scanner")

def close(xyz, ref, v):
    if not v.any():
        return 0.0
    a, b = xyz[v].astype(np.float64), ref[v]
    return float(np.max(np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)))

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    # FUZZ_MAXW / FUZZ_MAXH: larger frames (thousands of 1024-pixel tiles per view: long look-back chains of the fused compaction)
    MAXW, MAXH = int(os.environ.get("FUZZ_MAXW", 400)), int(os.environ.get("FUZZ_MAXH", 200))
    bad = 0
    maskin_cases = [0, 0, 0]
    for case in range(cases):
        fullW, fullH = int(rng.integers(5, MAXW)), int(rng.integers(5, MAXH))
        if rng.random() < 0.5:
            W, H, col0, row0 = fullW, fullH, 0, 0
        else:  # a window: column origin multiple of 4 (one lane = 4 pixels), any row origin
            col0 = int(rng.integers(0, max(1, fullW // 8))) * 4
            row0 = int(rng.integers(0, max(1, fullH // 2)))
            W, H = int(rng.integers(1, fullW - col0 + 1)), int(rng.integers(1, fullH - row0 + 1))
        Nv, Nh = int(rng.integers(0, 13)), int(rng.integers(0, 13))   # (0: an axis without Gray planes is a valid configuration)
        fwv, fwh = int(rng.choice([1, 2, 3, 4, 8, 32])), int(rng.choice([1, 2, 3, 5, 16]))
        PW = int(min(fwv * (1 << Nv), rng.integers(4, 3000)))
        PH = int(min(fwh * (1 << Nh), rng.integers(4, 2000)))
        F = int(rng.choice([3, 3, 3, 4, 5]))
        noise = int(rng.integers(0, 40))
        cap = syn.make_capture(W, H, PW, PH, Nv, Nh, fwv, fwh, n_fringe=F, noise=noise, col0=col0, row0=row0, full=(fullW, fullH),
                               plane=(float(rng.uniform(-20, 20)), float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2))))
        full_mask = (rng.random((fullH, fullW)) < rng.choice([0.5, 0.9, 0.99, 1.0])).astype(np.uint8)
        if rng.random() < 0.3:
            full_mask[:] = rng.integers(0, 3, size=full_mask.shape)  # values other than 0/1 count as unselected
        cal = {k: np.array(v, dtype=np.float64).copy() for k, v in cap["cal"].items()}
        rig = int(rng.integers(0, 5))
        if rig >= 1:
            cal["dp"] = np.array([0.05, -0.02, 0.001, -0.0005, 0.01]) * rng.uniform(0, 1)
        if rig == 2:
            cal["Kc"][1] = rng.uniform(-0.5, 0.5)  # skew (camera-frame solve with a skew term)
        if rig == 3:
            cal["dp"][2] = cal["dp"][3] = 0.0      # purely radial projector model: the LDS radial table (rig class 3)
        if rig == 4:
            cal["Kc"][3] = rng.uniform(-1e-3, 1e-3)  # K[1][0] != 0: the general, un-pipelined kernel (rig class 0)
        ct = syn.cal_tuple(cal)
        o = Oracle(W, H, PW, PH, Nv, Nh, fwv, fwh, F=F, col0=col0, row0=row0)
        # the oracle sees the window as its own image: give it the window of the mask, compare away from the window's edge
        o.set_mask(full_mask[row0:row0 + H, col0:col0 + W])
        o.set_calibration(*ct)
        o.run_scan(cap["planes_v"], cap["planes_h"])
        inner = np.zeros((H, W), bool)
        lo_r, hi_r = (0 if row0 == 0 else 3), (H if row0 + H == fullH else H - 3)
        lo_c, hi_c = (0 if col0 == 0 else 3), (W if col0 + W == fullW else W - 3)
        if hi_r > lo_r and hi_c > lo_c:
            inner[lo_r:hi_r, lo_c:hi_c] = True
        msg = []
        for keep in (False, True):
            with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, n_fringe=F, keep_stages=keep, full_size=(fullW, fullH), origin=(col0, row0)) as sc:
                sc.set_calibration(*ct)
                sc.set_mask(full_mask)
                sc.set_frames(0, cap["planes_v"])
                sc.set_frames(1, cap["planes_h"])
                sc.run()
                xyz, valid = sc.points()
                v = (o.valid_map(2) == 1) & inner
                if not np.array_equal((valid == 1) & inner, v):
                    msg.append(f"keep={keep}: valid map differs on {int(((valid == 1) & inner != v).sum())} px")
                    continue
                e = close(xyz, o.intersection_points(), v)
                if e > 1e-5:
                    msg.append(f"keep={keep}: point error {e:.3g}")
                if not keep:
                    # the cloud compacted inside the kernel is exactly xyz[valid], in scan order
                    cl = sc.fused_clouds(0, 1)[0]
                    if not np.array_equal(cl, xyz[valid == 1]):
                        msg.append("fused compaction != xyz[valid]")
                    if not np.array_equal(sc.valid_map(), valid):
                        msg.append("valid map changed by run_clouds")
                if keep:
                    for a in (0, 1):
                        va = (o.valid_map(a) == 1) & inner
                        if not np.array_equal(sc.code(a)[va], o.code(a)[va]): msg.append(f"code axis {a}")
                        if not np.array_equal(sc.unwrapped_phase(a)[va], o.unwrapped_phi(a)[va]): msg.append(f"unwrapped axis {a}")
                    if not np.array_equal(sc.c_p_map()[v], o.c_p_map()[v]): msg.append("c_p_map")
                    # the reference's [col][row] globals, transposed on the device, == the transposed row-major planes
                    rm = {0: sc.valid_map(0), 2: sc.valid_map(2), 4: sc.wrapped_phase(1), 5: sc.unwrapped_phase(0), 8: sc.code(1), 9: sc.intersection_points()}
                    for which, plane in rm.items():
                        want = plane.transpose(1, 0, 2) if which == 9 else plane.T
                        if not np.array_equal(sc.global_colrow(which), want.astype(np.float64 if which == 9 else np.float32 if 3 <= which <= 6 else np.int32), equal_nan=True):
                            msg.append(f"colrow global {which}")
        # the same window as row stripes behind sl3d_group_* (random stripe count, both transports) == the single context
        if H >= 2 and not msg:
            ns = int(rng.integers(1, min(H, 6) + 1))
            flags = S.SL3D_FLAG_GROUP_FORCE_RCCL if rng.random() < 0.5 else S.SL3D_FLAG_GROUP_NO_RCCL
            with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, n_fringe=F, full_size=(fullW, fullH), origin=(col0, row0)) as sc, \
                 S.Group(W, H, PW, PH, Nv, Nh, fwv, fwh, devices=[0] * ns, n_fringe=F, full_size=(fullW, fullH), origin=(col0, row0), flags=flags) as g:
                for x in (sc, g):
                    x.set_calibration(*ct)
                    x.set_mask(full_mask)
                    x.set_frames(0, cap["planes_v"])
                    x.set_frames(1, cap["planes_h"])
                sc.run()
                g.run(0, 1)
                g.gather(0, 1)
                a_, b_ = sc.points(), g.points(0)
                if not (np.array_equal(a_[1], b_[1]) and np.array_equal(a_[0], b_[0], equal_nan=True)):
                    msg.append(f"group of {ns} stripes ({g.transport}) != single context")
                g.run_clouds(0, 1)
                g.gather_clouds(0, 1)
                if not np.array_equal(g.cloud(0), a_[0][a_[1] == 1]):
                    msg.append(f"group cloud of {ns} stripes != xyz[valid]")
                g.run(0, 1)
                hx, hv = g.download_points(0, 1)   # every stripe straight into the host images
                if not (np.array_equal(hv[0], a_[1]) and np.array_equal(hx[0], a_[0], equal_nan=True)):
                    msg.append(f"group download_points of {ns} stripes != single context")
                fr = np.stack(cap["planes_v"] + cap["planes_h"])[None]
                hx, hv = g.process_views(fr)
                if not (np.array_equal(hv[0], a_[1]) and np.array_equal(hx[0], a_[0], equal_nan=True)):
                    msg.append(f"group process_views of {ns} stripes != single context")
        # a batch launch over several views with different masks (empty and full ones included) == the views one by one,
        # dense and compacted: exercises the view loop (views per lane, the pipelined plane loads, skipped views)
        if not msg:
            V = int(rng.integers(2, 10))
            with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, n_fringe=F, max_views=V, full_size=(fullW, fullH), origin=(col0, row0)) as sc:
                sc.set_calibration(*ct)
                for v in range(V):
                    m = full_mask.copy()
                    r = rng.random()
                    if r < 0.2:
                        m[:] = 0
                    elif r < 0.4:
                        m[:] = 1
                    elif r < 0.7:
                        m[rng.random(m.shape) < 0.5] = 0
                    sc.set_mask(m, view=v)
                    sc.set_frames(0, cap["planes_v"], view=v)
                    sc.set_frames(1, cap["planes_h"], view=v)
                sc.run(0, V)
                batch = [sc.points(v) for v in range(V)]
                bclouds = sc.fused_clouds(0, V)
                # the consumers of the segmented clouds: host copies (pageable, and pinned = written by the gap-closing kernel),
                # registration straight from the segments == registration of the dense results; the look-back context agrees
                got = sc.download_clouds(0, V)
                pin = sc.pinned((max(1, sum(len(c) for c in bclouds)) * 3,), np.float32)
                got_pin = sc.download_clouds(0, V, out=pin)
                for v in range(V):
                    if not (np.array_equal(got[v], bclouds[v]) and np.array_equal(got_pin[v], bclouds[v])):
                        msg.append(f"batch of {V}: download_clouds of view {v}")
                if not np.array_equal(sc.register_clouds(0, V, 3.5, -2.0, 250.0, 11.25), sc.register_views(0, V, 3.5, -2.0, 250.0, 11.25)):
                    msg.append(f"batch of {V}: register_clouds != register_views")
                for v in range(V):
                    sc.run(v, 1)
                    one = sc.points(v)
                    if not (np.array_equal(one[1], batch[v][1]) and np.array_equal(one[0], batch[v][0], equal_nan=True)):
                        msg.append(f"batch of {V}: view {v} differs from its single launch")
                    if not np.array_equal(bclouds[v], one[0][one[1] == 1]):
                        msg.append(f"batch of {V}: cloud of view {v} != xyz[valid]")
        # deferred masks (round 6): 1..4 views get NEW random selections and ONE launch -- a MASKIN launch wherever the configuration has
        # one (3-step fringes, a pipelined rig class, 1..12 Gray planes, no tangential camera terms), else k_mask_prepare + the ordinary
        # kernel -- must equal an SL3D_FLAG_EAGER_MASK context bit for bit: results, clouds, and the 0/1 plane it leaves behind
        if not msg:
            V = int(rng.integers(1, 5))
            kw = dict(n_fringe=F, max_views=V, full_size=(fullW, fullH), origin=(col0, row0))
            with S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, **kw) as sc, S.Scanner(W, H, PW, PH, Nv, Nh, fwv, fwh, eager_mask=True, serial_launches=True, **kw) as eg:
                masks = []
                for v in range(V):
                    m = (rng.random((fullH, fullW)) < rng.choice([0.7, 0.95, 1.0])).astype(np.uint8)
                    if rng.random() < 0.3:
                        m[rng.random(m.shape) < 0.3] = int(rng.integers(2, 256))
                    masks.append(m)
                clouds_route = rng.random() < 0.5
                # the history the launch is routed by: dense (the early-request MASKIN kernel) or sparse (the gated one)
                hist = np.ones((fullH, fullW), np.uint8) if rng.random() < 0.6 else (rng.random((fullH, fullW)) < 0.15).astype(np.uint8)
                for x in (sc, eg):
                    x.set_calibration(*ct)
                    for v in range(V):
                        x.set_frames(0, cap["planes_v"], view=v)
                        x.set_frames(1, cap["planes_h"], view=v)
                    x.set_masks(hist, 0, V)
                    x.run(0, V)
                    x.synchronize()
                    x.set_masks(np.stack(masks))
                    if not clouds_route:
                        x.run(0, V)
                    else:
                        x.run_clouds(0, V)
                names = sc.last_fused_kernel_name()
                for v in range(V):
                    a_, b_ = sc.points(v), eg.points(v)
                    if not (np.array_equal(a_[1], b_[1]) and np.array_equal(a_[0], b_[0], equal_nan=True)):
                        msg.append(f"deferred masks, {V} views ({names}): view {v} differs from the eager context")
                    ba, bb = sc.device_buffers(), eg.device_buffers()
                    pa, pb = np.empty((H + 4, ba.mask_pitch), np.uint8), np.empty((H + 4, bb.mask_pitch), np.uint8)
                    sc._d2h(pa, ba.mask + v * ba.mask_view_stride); eg._d2h(pb, bb.mask + v * bb.mask_view_stride)
                    if not np.array_equal(pa, pb):
                        msg.append(f"deferred masks, {V} views ({names}): the 0/1 plane of view {v} differs")
                # ... and a series of one-view launches over these views, nothing waited for in between: from the ninth launch on they run on
                # the launch lanes (two internal streams in turn); the eager context keeps every launch on its one stream
                if V > 1 and not msg:
                    for x in (sc, eg):
                        for i in range(20):
                            x.run(i % V, 1)
                    if sc.launch_counts()[1] > 0:
                        maskin_cases[2] += 1
                    for v in range(V):
                        a_, b_ = sc.points(v), eg.points(v)
                        if not (np.array_equal(a_[1], b_[1]) and np.array_equal(a_[0], b_[0], equal_nan=True)):
                            msg.append(f"a series of one-view launches on the lanes ({V} views): view {v} differs from the serial context")
                if any(t in names for t in (", 4, false, true>", ", 6, false, true>")):
                    maskin_cases[0] += 1
                if any(t in names for t in (", 4, true, false>", ", 6, true, false>")):
                    maskin_cases[1] += 1
        tag = f"case {case}: full {fullW}x{fullH} window {W}x{H}@({col0},{row0}) proj {PW}x{PH} N {Nv}/{Nh} fw {fwv}/{fwh} F {F} noise {noise} rig {rig} valid {int((o.valid_map(2) == 1).sum())}"
        if msg:
            bad += 1
            print("FAIL", tag, msg, flush=True)
        del o
    print(f"{cases} cases, {bad} failures; MASKIN launches: {maskin_cases[0]} early-request, {maskin_cases[1]} gated; series on the launch lanes: {maskin_cases[2]}")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
