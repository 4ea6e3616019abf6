#!/usr/bin/env python3
"""Runs in a FRESH process (started by tests/test_gpu_group.py) with SL3D_RCCL_LIB pointing at the test double of librccl
(tests/native/fake_rccl.cpp): the library binds its seven RCCL entry points once per process, so the double has to be in place
before the first group is created.

    python tests/group_fake_rccl_driver.py <libfake_rccl.so> <n_stripes>

Every stripe of the group is its own communication side (SL3D_FLAG_GROUP_DISTINCT_SIDES) = its own rank of an N-rank communicator,
all on device 0.  What is exercised and checked bit for bit against ONE context over the whole window:
  * sl3d_group_gather as an N-rank exchange: one ncclSend per (view, stripe) slab on the stripe's own side and stream, the matching
    ncclRecv on the root's -- pipelined `run(v + 1); gather(v)`, then re-run over the SAME views while their gather is in flight
    (the stripes must wait for the sides that still read them: sl3d_group.cpp group_launch);
  * a gather of more than 256 messages: several RCCL groups (the 256-pair split), same order on both sides;
  * sl3d_group_gather_clouds: variable message sizes from the counts (an empty stripe sends nothing), concatenation in stripe order;
  * gather -> sl3d_group_process_views -> get_points: the assembled planes of the gather survive the pipelines that overwrite the
    stripes' result slots right behind it.
Prints one JSON line with the double's counters; any mismatch raises."""
import ctypes
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    lib, ns = sys.argv[1], int(sys.argv[2])
    os.environ["SL3D_RCCL_LIB"] = lib
    import numpy as np
    try:
        import torch  # noqa: F401  (one ROCm stack per process: torch's first, as in tests/conftest.py)
    except Exception:
        pass
    S = importlib.import_module("3dscan_amd.scanner")
    syn = importlib.import_module("3dscan_amd.synth")
    fake = ctypes.CDLL(lib)   # the same handle the product dlopens: its counters are this process's

    def stats():
        v = [ctypes.c_int(0) for _ in range(5)]
        fake.fake_rccl_stats(*[ctypes.byref(x) for x in v])
        return dict(zip(("groups", "pairs", "max_pairs_in_group", "ranks", "self_pairs"), (x.value for x in v)))

    flags = S.SL3D_FLAG_GROUP_DISTINCT_SIDES
    out = {}

    # ---- 1. pipelined dense gather + re-run under a gather in flight + compacted gather -----------------------------------------
    W, H, PW, PH, N, fw, NV = 320, 203, 512, 384, 7, 4, 4
    rng = np.random.default_rng(ns)
    caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, view=v, noise=2, plane=(3.0 * v, 0.05, 0.02 * v)) for v in range(NV)]
    cal = syn.cal_tuple(caps[0]["cal"])
    masks = []
    for v in range(NV):
        m = caps[0]["mask"].copy()
        if v:
            m[rng.random((H, W)) < 0.1] = 0
        if v == 2:
            m[: H // ns + 1] = 0        # the root's own stripe (and a bit more) selects nothing: an empty message in the cloud gather
        masks.append(m)
    ref = []
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v, c in enumerate(caps):
            sc.set_mask(masks[v], view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, NV)
        ref = [sc.points(v) for v in range(NV)]
    with S.Group(W, H, PW, PH, N, N, fw, fw, devices=[0] * ns, max_views=NV, flags=flags) as g:
        assert g.transport == "rccl", g.transport
        g.set_calibration(*cal)
        for v, c in enumerate(caps):
            g.set_mask(masks[v], view=v)
            g.set_frames(0, c["planes_v"], view=v)
            g.set_frames(1, c["planes_h"], view=v)
        s0 = stats()
        for rep in range(3):
            for v in range(NV):      # run(v + 1) is enqueued while gather(v) travels; rep > 0: the same views again, gathers in flight
                g.run(v, 1)
                g.gather(v, 1)
            for v in range(NV):
                xyz, val = g.points(v)
                assert np.array_equal(val, ref[v][1]), (rep, v)
                assert np.array_equal(xyz, ref[v][0], equal_nan=True), (rep, v)
        s1 = stats()
        # every (view, stripe) slab of a stripe that is not the root's own went through the double as a pair of the stripe's rank
        # and rank 0: 2 messages (xyz, valid) per slab; the root's own stripe is a device copy
        assert s1["pairs"] - s0["pairs"] == 3 * NV * (ns - 1) * 2, (s0, s1)
        assert s1["self_pairs"] == 0
        assert s1["ranks"] >= ns
        g.run_clouds(0, NV)
        counts = g.gather_clouds(0, NV)
        for v in range(NV):
            cl = g.cloud(v)
            assert counts[v] == len(cl) == int((ref[v][1] == 1).sum()), v
            assert np.array_equal(cl, ref[v][0][ref[v][1] == 1]), v
        # gather, then the host pipelines over the same slots, then the assembled planes: still the gathered ones
        g.run(0, NV)
        g.gather(0, NV)
        frames = np.stack([np.stack(c["planes_v"] + c["planes_h"]) for c in reversed(caps)])   # other views than the slots hold
        pxyz, pval = g.process_views(frames)
        for v in range(NV):
            xyz, val = g.points(v)
            assert np.array_equal(val, ref[v][1]) and np.array_equal(xyz, ref[v][0], equal_nan=True), ("gather then process_views", v)
        g.synchronize()
    out["pipelined"] = stats()

    # ---- 2. more than 256 messages in one gather: several RCCL groups -----------------------------------------------------------
    W, H, PW, PH, N, fw, NV = 64, 28, 128, 96, 6, 4, 40
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=1)
    cal = syn.cal_tuple(cap["cal"])
    masks = []
    for v in range(NV):
        m = cap["mask"].copy()
        m[rng.random((H, W)) < 0.05 * (v % 5)] = 0
        masks.append(m)
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v in range(NV):
            sc.set_mask(masks[v], view=v)
            sc.set_frames(0, cap["planes_v"], view=v)
            sc.set_frames(1, cap["planes_h"], view=v)
        sc.run(0, NV)
        ref = [sc.points(v) for v in range(NV)]
    nsm = min(ns, H)
    with S.Group(W, H, PW, PH, N, N, fw, fw, devices=[0] * nsm, max_views=NV, flags=flags) as g:
        g.set_calibration(*cal)
        for v in range(NV):
            g.set_mask(masks[v], view=v)
            g.set_frames(0, cap["planes_v"], view=v)
            g.set_frames(1, cap["planes_h"], view=v)
        s0 = stats()
        g.run(0, NV)
        g.gather(0, NV)
        for v in range(NV):
            xyz, val = g.points(v)
            assert np.array_equal(val, ref[v][1]) and np.array_equal(xyz, ref[v][0], equal_nan=True), v
        s1 = stats()
        n_msgs = NV * (nsm - 1) * 2
        assert s1["pairs"] - s0["pairs"] == n_msgs, (s0, s1, n_msgs)
        assert s1["max_pairs_in_group"] <= 256
        assert s1["groups"] - s0["groups"] == (n_msgs + 255) // 256, (s0, s1)
        g.run_clouds(0, NV)
        counts = g.gather_clouds(0, NV)
        for v in range(NV):
            assert np.array_equal(g.cloud(v), ref[v][0][ref[v][1] == 1]), v
    out["many_messages"] = stats()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
