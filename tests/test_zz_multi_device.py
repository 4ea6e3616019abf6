"""GPU tests (-m gpu) that ARM THEMSELVES on a lease with two or more GPUs and skip on a one-GPU box: the code of sl3d_group.cpp,
of a context on a device other than 0 and of bench.py's RCCL branch that no one-GPU test can execute -- real ncclCommInitAll over
distinct devices, peer enabling and hipMemcpyPeerAsync between two GPUs, ncclSend/ncclRecv over xGMI, a context on `device != 0`.
Every case compares with what ONE context on GPU 0 produces, bit for bit (the stripes never exchange anything while they compute).
No scaling figure is claimed here or anywhere else until these have run.
(The file's name sorts it LAST: what has never run on hardware cannot keep `pytest -x` from reaching everything that has.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_points_close, pkg

pytestmark = pytest.mark.gpu


# SL3D_TEST_PRETEND_GPUS=n: run the test BODIES on a one-GPU box with every "device" mapped to GPU 0 (RCCL forced through the
# one-rank communicator, peer copies 0 -> 0) -- a rehearsal of this file itself, not of the multi-device code it exists for
PRETEND = int(os.environ.get("SL3D_TEST_PRETEND_GPUS", "0"))


def _n_gpus():
    if PRETEND:
        return PRETEND
    try:
        import torch
        return torch.cuda.device_count()   # (counting does not initialise the GPU)
    except Exception:
        return 0


def _dev(i):
    return 0 if PRETEND else i


needs_two = pytest.mark.skipif(_n_gpus() < 2, reason="needs a lease with at least 2 GPUs")


def _random_mask(rng, W, H, p=0.1):
    m = np.ones((H, W), np.uint8)
    m[0, :] = m[-1, :] = 0
    m[:, 0] = m[:, -1] = 0
    m[rng.random((H, W)) < p] = 0
    return m


@needs_two
@pytest.mark.parametrize("transport", ["rccl", "copy"])
def test_group_over_real_devices_equals_one_context(transport):
    """One row stripe per visible GPU (up to 8): the group's results -- dense planes through the pipelined run(v + 1); gather(v), the
    ordered clouds, the host-parallel download -- equal the single-context result bit for bit, over RCCL send/recv and over peer copies."""
    S, syn = pkg("scanner"), pkg("synth")
    devices = [_dev(i) for i in range(min(_n_gpus(), 8))]
    W, H, PW, PH, N, fw, NV = 640, 403, 512, 384, 8, 4, 4
    rng = np.random.default_rng(len(devices))
    caps = [syn.make_capture(W, H, PW, PH, N, N, fw, fw, view=v, noise=2, plane=(3.0 * v, 0.05, 0.02 * v)) for v in range(NV)]
    cal = syn.cal_tuple(caps[0]["cal"])
    masks = [caps[0]["mask"]] + [_random_mask(rng, W, H) for _ in range(NV - 1)]
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, max_views=NV) as sc:
        sc.set_calibration(*cal)
        for v, c in enumerate(caps):
            sc.set_mask(masks[v], view=v)
            sc.set_frames(0, c["planes_v"], view=v)
            sc.set_frames(1, c["planes_h"], view=v)
        sc.run(0, NV)
        ref = [sc.points(v) for v in range(NV)]
    flags = S.SL3D_FLAG_GROUP_NO_RCCL if transport == "copy" else (S.SL3D_FLAG_GROUP_FORCE_RCCL if PRETEND else 0)
    with S.Group(W, H, PW, PH, N, N, fw, fw, devices=devices, max_views=NV, flags=flags) as g:
        assert g.transport == transport, g.transport
        g.set_calibration(*cal)
        for v, c in enumerate(caps):
            g.set_mask(masks[v], view=v)
            g.set_frames(0, c["planes_v"], view=v)
            g.set_frames(1, c["planes_h"], view=v)
        for rep in range(3):
            g.run(0, 1)
            for v in range(NV):          # pipelined: the next view computes while this one's stripes travel
                if v + 1 < NV:
                    g.run(v + 1, 1)
                g.gather(v, 1)
            for v in range(NV):
                xyz, val = g.points(v)
                assert np.array_equal(val, ref[v][1]), (rep, v)
                assert np.array_equal(xyz, ref[v][0], equal_nan=True), (rep, v)
        g.run_clouds(0, NV)
        counts = g.gather_clouds(0, NV)
        for v in range(NV):
            cl = g.cloud(v)
            assert counts[v] == len(cl) == int((ref[v][1] == 1).sum())
            assert np.array_equal(cl, ref[v][0][ref[v][1] == 1]), v
        g.run(0, NV)
        if hasattr(g, "download_points"):   # every GPU's rows over its own PCIe link into one host array
            xyz, val = g.download_points(0, NV)
            for v in range(NV):
                assert np.array_equal(val[v], ref[v][1]) and np.array_equal(xyz[v], ref[v][0], equal_nan=True), v
        g.synchronize()


@needs_two
def test_context_on_the_last_device_matches_the_oracle():
    """A context on a device other than 0 (the current device of the calling thread stays 0): every entry point switches to the
    context's device and back.  Dense, clouds, device-resident masks living on THAT device, against the oracle."""
    import torch
    from oracle.oracle import Oracle
    S, syn = pkg("scanner"), pkg("synth")
    dev = _dev(_n_gpus() - 1)
    W, H, PW, PH, N, fw = 320, 203, 512, 384, 7, 4
    cap = syn.make_capture(W, H, PW, PH, N, N, fw, fw, noise=2)
    cal = syn.cal_tuple(cap["cal"])
    mask = _random_mask(np.random.default_rng(3), W, H)
    o = Oracle(W, H, PW, PH, N, N, fw, fw)
    o.set_mask(mask)
    o.set_calibration(*cal)
    oxyz, ovalid, _ = o.run_scan_rowmajor(cap["planes_v"], cap["planes_h"])
    with S.Scanner(W, H, PW, PH, N, N, fw, fw, device=dev) as sc:
        sc.set_calibration(*cal)
        sc.set_frames(0, cap["planes_v"])
        sc.set_frames(1, cap["planes_h"])
        for how in ("host", "device"):
            sc.set_mask(np.zeros_like(mask))
            if how == "host":
                sc.set_mask(mask)
            else:
                dm = torch.from_numpy(mask).to(f"cuda:{dev}")
                torch.cuda.synchronize(dev)
                sc.set_masks_device(dm.data_ptr(), W, 0, 0, 1)
            sc.run()
            xyz, val = sc.points(0)
            assert np.array_equal(val, ovalid), how
            assert_points_close(xyz, oxyz, ovalid == 1)
            assert np.array_equal(sc.fused_clouds(0, 1)[0], xyz[ovalid == 1]), how
    assert torch.cuda.current_device() == 0


def _bench(*argv):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@needs_two
@pytest.mark.skipif(PRETEND > 0, reason="RCCL needs one physical GPU per rank")
def test_bench_two_ranks_over_rccl_assemble_the_single_rank_result():
    """bench.py --gpus 2 with the RCCL backend (one rank per GPU, xGMI between them): the assembled dense planes and clouds on rank 0
    have the digests of a one-rank run over the same global batch; the headline of an N > 1 line is an ASSEMBLED figure, the
    compute-only rate sits beside it."""
    common = ["--steps", "3", "--warmup", "1", "--precondition-ms", "0", "--no-cpu-baseline", "--no-side", "--check", "--height", "1080"]
    one = _bench("--gpus", "1", "--views", "4", *common)
    two = _bench("--gpus", "2", "--backend", "nccl", "--views", "2", "--chunks", "2", *common)
    assert two["n_gpus"] == 2 and [r["device"] for r in two["ranks"]] == [0, 1]
    wa = two["with_assembly"]
    assert "error" not in wa, wa
    assert two["check"]["dense_sha256"] == one["check"]["dense_sha256"]
    assert two["check"]["compact_sha256"] == one["check"]["compact_sha256"]
    assert two["value_is"] in wa and two["value"] == wa[two["value_is"]]["headline_value"]
    assert two["compute_only"]["value"] >= two["value"]


@needs_two
@pytest.mark.skipif(PRETEND > 0, reason="RCCL needs one physical GPU per rank")
def test_bench_two_ranks_strong_scaling_and_rank_identity():
    """--scaling strong over RCCL: BASELINE configs[3] as stated (a FIXED batch of views, every view row-sharded over the ranks) -- the
    assembled digests equal the one-rank run over the same batch; the line names the RCCL version, the communicator size and one
    distinct PCI bus id per rank (what lets the driver confirm N ranks on N GPUs); the 19 % selection variant of the single-root
    compact gather is there and never the headline."""
    common = ["--steps", "3", "--warmup", "1", "--precondition-ms", "0", "--no-cpu-baseline", "--no-side", "--check", "--height", "1080",
              "--scaling", "strong", "--total-views", "4"]
    one = _bench("--gpus", "1", *common)
    two = _bench("--gpus", "2", "--backend", "nccl", "--chunks", "2", *common)
    assert two["scaling"] == "strong" and two["config"]["views_per_step"] == 4 and two["config"]["rows_per_gpu"] == 540
    assert two["check"]["dense_sha256"] == one["check"]["dense_sha256"] and two["check"]["compact_sha256"] == one["check"]["compact_sha256"]
    ids = [r["pci_bus_id"] for r in two["ranks"]]
    assert len(set(ids)) == 2 and all(ids), ids
    assert all(r["comm_size"] == 2 and r["rccl_version"] for r in two["ranks"]), two["ranks"]
    wa = two["with_assembly"]
    assert two["value_is"].startswith("dense_") and "compact_root_gather_19pct_selection" in wa and "error" not in wa["compact_root_gather_19pct_selection"]


@needs_two
@pytest.mark.parametrize("mode", ["memory", "memory+deferred_final"])
def test_shim_over_two_devices(tmp_path, mode):
    """The drop-in shim with SL3D_DEVICES=0,1: the scan as two row stripes on two GPUs, stage by stage and deferred -- every global
    and both cloud files as on one context (the same checks tests/test_gpu_shim.py makes with all stripes on GPU 0)."""
    import test_gpu_shim
    test_gpu_shim.test_shim_matches_oracle(tmp_path, "0,0" if PRETEND else "0,1", mode)
