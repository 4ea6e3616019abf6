#!/usr/bin/env python3
"""Benchmark of the fused decode -> unwrap -> correspond -> triangulate kernel (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RANK/LOCAL_RANK/WORLD_SIZE from env)

Workload (configs[1] of BASELINE.json): 1920x1080 camera, 3 phase-shift + 10 Gray-code bit planes per
axis, reference-faithful two-axis mode (every Gray frame is thresholded against its inverse frame, so a
view is 2*(3+10+10) = 46 frames), synthetic captures resident in HBM.  A step is ONE launch of the fused
kernel over a batch of `--views` views per GPU (default 16, i.e. 2 GB of frames + results: larger than
the 256 MiB Infinity Cache, so the kernel really streams from HBM).  With N GPUs every view is sharded by
image rows (1080/N rows per GPU) and the batch grows to N*views views, so per-GPU work is fixed (weak
scaling); no collective is needed by the per-pixel map itself (the mask halo comes from the input mask).
The optional assembly of the dense clouds over RCCL is measured separately and reported in `assemble`.

The defaults (300 warm-up + 2000 timed launches, ~1 s of GPU time) let the clocks settle: the kernel runs the package
into its power limit (~1.4 kW), and a 25 ms run from idle measures the ramp, not the steady state.  For the same reason
the set-up ends with --precondition-ms (0.4 s) of launches before the W warm-up steps, whatever W and K are.

Prints ONE JSON line (rank 0).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def measured_traffic(px_per_launch):
    """HBM bytes per launch of the fused kernel from the latest committed PMC run (tools/profile.sh +
    tools/summarize_profile.py: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, scaled by the factors
    calibrated on tools/membench in the same session).  The profile is taken on this same command; bytes scale
    with the pixels per launch.  None if no profile has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    t = json.load(open(files[-1]))
    scale = px_per_launch / (t["algorithmic_bytes_per_launch"] / 60.0)
    return round(t["hbm_bytes_per_launch"] * scale)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--views", type=int, default=16, help="views per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--ngray", type=int, default=10)
    ap.add_argument("--fringe-width", type=int, default=2)
    ap.add_argument("--noise", type=int, default=2)
    ap.add_argument("--precondition-ms", type=float, default=400.0,
                    help="part of the SETUP, before the W warm-up steps: run the kernel for this long so that the clocks have "
                         "left the idle state whatever W is (the package is power-managed; see profiles/README.md). 0 disables")
    ap.add_argument("--rig", default="reference", choices=["reference", "distorted"],
                    help="reference = the reference's calibration rescaled (BASELINE workload); distorted = the same rig with "
                         "projector distortion and camera tangential terms, i.e. the general stage-7 path (sweeps only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-assemble", action="store_true", help="skip the separate RCCL assembly measurement (N>1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of one view timed on the CPU (0 = whole view)")
    return ap.parse_args()


class _DevMem:
    """Expose a raw device pointer to torch (zero copy) through the CUDA array interface."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max = "quota period")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(args, cap, cal, gpu_valid, gpu_xyz, gpu_cp_note):
    """The oracle (CPU restatement of the reference loop: single thread, [col][row] arrays, pow() per bit,
    fenv per pixel, stage-7 tables rebuilt per scan as triangulate() does) timed on one view of the
    same workload; its results also check the GPU output of that view."""
    import numpy as np
    from oracle.oracle import Oracle

    W, H = args.width, args.height
    rows = args.cpu_sample_rows or H
    pv = [p[:rows] for p in cap["planes_v"]]
    ph = [p[:rows] for p in cap["planes_h"]]
    syn = importlib.import_module("3dscan_amd.synth")
    mask = syn.default_mask(W, H)[:rows]
    # above 2^24 pixels the reference's float pixel index (7/triangulation.cpp:264-265) goes wrong; the product uses integer
    # rows / columns, so the oracle is asked for the same there
    big = W * H > (1 << 24)
    o = Oracle(W, rows, W, H, args.ngray, args.ngray, args.fringe_width, args.fringe_width, exact_index=big)
    o.set_mask(mask)
    o.set_calibration(*cal)
    times = []
    for _ in range(5):
        o.invalidate_tables()
        t0 = time.perf_counter()
        o.run_scan(pv, ph)
        times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    v = o.valid_map(2) == 1
    I = np.s_[0:rows - 3]  # the oracle treats the sample as its own image: skip its last rows
    ok = bool(np.array_equal(gpu_valid[:rows][I] == 1, v[I]))
    ref = o.intersection_points()[I][v[I]]
    got = gpu_xyz[:rows][I][v[I]].astype(np.float64) if ok else None
    rel = float(np.max(np.linalg.norm(got - ref, axis=-1) / np.linalg.norm(ref, axis=-1))) if ok and len(ref) else None
    # (b) of SURVEY 8d: the same maths fused, row-major, OpenMP over rows on all host cores (bit-identical results)
    par = None
    try:
        tp = []
        ncores = usable_cores()
        pxyz, pvalid, nthreads = o.run_scan_rowmajor(pv, ph, threads=ncores)   # warm-up: thread pool, first touch of the outputs
        for _ in range(5):
            t0 = time.perf_counter()
            o.run_scan_rowmajor(pv, ph, threads=ncores, out=(pxyz, pvalid))
            tp.append(time.perf_counter() - t0)
        tpar = sorted(tp)[len(tp) // 2]
        same = bool(np.array_equal(pvalid == 1, v) and np.array_equal(pxyz[v], o.intersection_points().astype(np.float32)[v]))
        par = {"value": round(W * rows / tpar / 1e6, 3), "unit": "Mpixels/s", "cores": int(nthreads),
               "sample": f"same view, fused row-major restatement, OpenMP over rows on the {ncores} cores the cgroup quota / affinity grant "
                         f"(of {os.cpu_count()} visible), median of 5 runs ({tpar:.3f} s each)",
               "bit_identical_to_single_thread": same}
    except Exception as e:
        par = {"error": repr(e)}
    return {
        "value": round(W * rows / t / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port", "all_cores": par,
        "sample": f"1 view {W}x{rows} of the same workload (N={args.ngray}, two axes), single thread, reference loop order and "
                  f"[col][row] layout, median of 5 runs ({t:.2f} s each)",
        "gpu_matches_oracle": {"valid_map_bit_exact": ok, "max_rel_point_error": rel},
    }


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    # torch first: its bundled HIP runtime must be the one libsl3d.so binds to (one runtime per process)
    import torch
    import torch.distributed as dist

    dmod = importlib.import_module("3dscan_amd.distributed")
    rank, local_rank, world = dmod.env_ranks()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)  # one GPU per rank on a real node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    red_dev = dev if args.backend == "nccl" else None  # where the tiny timing reductions live

    syn = importlib.import_module("3dscan_amd.synth")
    scm = importlib.import_module("3dscan_amd.scanner")
    W, H, N, fw, V = args.width, args.height, args.ngray, args.fringe_width, args.views
    PW, PH = W, H
    row0, rows = dmod.shard_rows(H, world, rank)
    n_views = V * world  # batch grows with the GPU count; each GPU holds `rows` rows of every view

    # ---- synthetic inputs, generated on the device, resident in HBM before the timed region ----
    # every view is a different plane seen by the same rig, with its own noise stream (sl3d_synth_view / k_synth)
    cal_d = syn.synth_rig(W, H, PW, PH)
    if args.rig == "distorted":
        cal_d["dp"] = np.array([-0.05, 0.02, 0.001, -0.0005, 0.0])
        cal_d["dc"] = np.array(cal_d["dc"], dtype=np.float64) + np.array([0.0, 0.0, 0.0008, -0.0006, 0.0])
    cal = syn.cal_tuple(cal_d)
    sc = scm.Scanner(W, rows, PW, PH, N, N, fw, fw, max_views=n_views, device=dev_index, full_size=(W, H), origin=(0, row0))
    sc.set_calibration(*cal)
    full_mask = syn.default_mask(W, H)
    for v in range(n_views):
        sc.set_mask(full_mask, view=v)
        sc.synth_view(v, plane=(0.75 * v, 0.05, 0.05 - 0.003 * v), view_id=v, noise=args.noise)
    sc.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup: bring the device out of its idle power state (reported in the JSON; not part of the W + K steps)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.precondition_ms:
        for _ in range(50):
            sc.run(0, n_views)
        sc.synchronize()
    for _ in range(args.warmup):
        sc.run(0, n_views)
    barrier()
    t0 = time.perf_counter()
    sc.timer_start()                       # HIP event on the stream the kernel is launched on
    for _ in range(args.steps):
        sc.run(0, n_views)
    ev_ms = sc.timer_stop()                # second event, waited for
    barrier()
    dt = time.perf_counter() - t0
    dt = dmod.max_over_ranks(dt, red_dev)
    ev_ms = dmod.max_over_ranks(ev_ms, red_dev)

    px_per_launch = n_views * rows * W                     # pixels one launch processes on one GPU
    total_px = args.steps * px_per_launch * world
    value = total_px / dt / 1e6
    alg_bytes_px = 2 * 3 + 4 * N + 1 + 13                  # read 2F+2Nv+2Nh frame bytes + 1 mask, write xyz f32 + valid
    launch_s = ev_ms / 1e3 / args.steps
    achieved = alg_bytes_px * px_per_launch / launch_s / 1e9

    nmax = next(m for m in (6, 8, 10, 12, 16) if m >= N)   # the instantiation launch_fused picks (sl3d_kernels.hip)
    kernel_name = f"sl3d::k_fused<false, {nmax}, false, {'true' if nmax == N else 'false'}, {1 if args.rig == 'reference' else 2}>"

    out = {
        "metric": "Mpixels/s decode+unwrap+triangulate @1920\u00d71080",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic (planes through the reference rig, generated on the device, a different plane and noise stream per view)",
        "config": {"workload": f"configs[1]: {W}x{H} camera, 3 phase-shift + {N} Gray-code bit planes per axis, two axes, "
                               f"Gray frames thresholded against inverse frames (46 frames/view); one step = one fused-kernel "
                               f"launch over {V} views per GPU, frames resident in HBM",
                   "views_per_gpu_per_step": V, "rows_per_gpu": rows, "frames_per_view": 2 * (3 + 2 * N),
                   "projector": f"{PW}x{PH}", "fringe_width": fw, "sharding": "image rows" if world > 1 else "none",
                   "rig": args.rig, "setup_preconditioning_ms": args.precondition_ms},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": measured_traffic(px_per_launch) if alg_bytes_px == 60 else None,
                     "kernel": kernel_name, "algorithmic_bytes_per_pixel": alg_bytes_px,
                     "pixels_per_launch": px_per_launch, "avg_launch_ms": round(launch_s * 1e3, 4)},
    }

    # end to end from device-resident frames to compacted clouds: the fused kernel + the batched compaction of
    # every view (three more launches and one read-back of the counts per step); a side figure, never `value`
    try:
        for _ in range(20):
            sc.run(0, n_views)
            sc.compact_views(0, n_views)
        barrier()
        t0 = time.perf_counter()
        reps = max(20, args.steps // 10)
        for _ in range(reps):
            sc.run(0, n_views)
            counts = sc.compact_views(0, n_views)
        barrier()
        te = dmod.max_over_ranks((time.perf_counter() - t0) / reps, red_dev)
        out["to_compacted_clouds"] = {"value": round(px_per_launch * world / te / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(te * 1e3, 4),
                                      "valid_points_per_step_rank0": int(sum(counts))}
    except Exception as e:
        out["to_compacted_clouds"] = {"error": repr(e)}

    if world > 1 and not args.no_assemble:
        out["assemble"] = measure_assemble(torch, dist, dmod, sc, n_views, V, rows, rank, world)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        xyz, valid = sc.points(0)
        try:
            cap0 = {"planes_v": sc.frames(0, 0), "planes_h": sc.frames(1, 0)}  # the very bytes the GPU processed
            out["cpu_baseline"] = cpu_baseline(args, cap0, cal, valid, xyz, None)
            # what the C ABI delivers when the boundary hands over HOST buffers (never `value`):
            # (a) serial: upload of the 46 frames of one view, one launch, download of xyz + valid, pageable numpy memory
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                sc.set_frames(0, cap0["planes_v"], view=0)
                sc.set_frames(1, cap0["planes_h"], view=0)
                sc.run(0, 1)
                sc.points(0)
                ts.append(time.perf_counter() - t0)
            t = sorted(ts)[2]
            out["host_buffers_one_view"] = {"value": round(W * rows / t / 1e6, 1), "unit": "Mpixels/s", "ms": round(t * 1e3, 2),
                                            "note": "H2D of 46 frames + launch + D2H of xyz and valid for ONE view, pageable host memory"}
            # (b) pipelined: sl3d_process_views, 12 host-resident views through the view slots on three HIP streams, pinned memory
            nv = 12
            fr = sc.pinned((nv, 2 * (3 + 2 * N), rows, W), np.uint8)
            fr[:] = np.stack(cap0["planes_v"] + cap0["planes_h"])[None]
            px = sc.pinned((nv, rows, W, 3), np.float32)
            pv = sc.pinned((nv, rows, W), np.uint8)
            sc.process_views(fr, xyz=px, valid=pv)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                sc.process_views(fr, xyz=px, valid=pv)
                ts.append(time.perf_counter() - t0)
            t = sorted(ts)[1]
            same = bool(np.array_equal(pv[nv - 1], valid) and np.array_equal(px[nv - 1], xyz, equal_nan=True))
            out["host_buffers_pipelined"] = {"value": round(nv * W * rows / t / 1e6, 1), "unit": "Mpixels/s", "ms_per_view": round(t / nv * 1e3, 2),
                                             "note": f"{nv} host-resident views, upload / kernel / download overlapped on 3 streams, pinned memory",
                                             "equals_resident_result": same}
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"error": repr(e)}
    sc.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_assemble(torch, dist, dmod, sc, n_views, V, rows, rank, world):
    """Assembly of the dense clouds over RCCL, outside the timed region: (a) gather of every stripe to rank 0,
    (b) rotating roots (one all_to_all).  Reported as time per batch and the rate it would sustain."""
    res = {}
    try:
        b = sc.device_buffers()
        pitch = b.frame_pitch
        pts = torch.as_tensor(_DevMem(b.points, n_views * b.points_view_stride), device="cuda").view(torch.float32)
        pts = pts.view(n_views, rows, pitch * 3)
        val = torch.as_tensor(_DevMem(b.valid, n_views * b.valid_view_stride), device="cuda").view(n_views, rows, pitch)
        nbytes = pts.numel() * 4 + val.numel()
        for name, fn in (("root_gather", lambda: (dmod.assemble_root(pts, rows), dmod.assemble_root(val, rows))),
                         ("rotating_all_to_all", lambda: (dmod.assemble_rotating(pts, V), dmod.assemble_rotating(val, V)))):
            fn()
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(); dist.barrier()
            t = dmod.max_over_ranks((time.perf_counter() - t0) / reps, pts.device)
            res[name] = {"ms_per_batch": round(t * 1e3, 3), "GB_per_rank": round(nbytes / 1e9, 4),
                         "job_Mpixels_per_s_if_serialised": round(n_views * rows * world * (pitch) / t / 1e6, 1)}
    except Exception as e:
        res["error"] = repr(e)
    return res


if __name__ == "__main__":
    main()
