#!/usr/bin/env python3
"""Turns gpurun_out/profile_<tag>/ (written by tools/profile.sh on the GPU box) into the committed summaries
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and profiles/<tag>_traffic.json (read by bench.py)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/profile_{tag}"
os.makedirs("profiles", exist_ok=True)

# 1. kernel stats
stats = glob.glob(f"{src}/stats/*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    for r in rows[:4]:
        print(r)


def counters(prefix, kernel_substr):
    vals = {}
    for fn in glob.glob(f"{src}/pmc/{prefix}_*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if kernel_substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
    return vals


# the bench command launches two instantiations of k_fused: the dense kernel the bench line's `value` / `roofline` are about
# (last template argument 0) and the compacting one of its `to_compacted_clouds` leg (2 = segmented clouds; 1 = the look-back
# variant of the side figure): kept apart
def dense_name():
    try:
        return json.loads(open(f"{src}/stats_bench.json").read())["roofline"]["kernel"].replace("sl3d::", "")
    except Exception:
        return "k_fused<false, 10, false, true, 1, 0, true, true>"


dense = dense_name()
bench = counters("bench", dense)


def clouds_name():
    try:
        return json.loads(open(f"{src}/stats_bench.json").read())["to_compacted_clouds"]["roofline"]["kernel"].replace("sl3d::", "")
    except Exception:
        return dense.rsplit(", 0, ", 1)[0] + ", 2, true, true>"


compact_kernel = clouds_name()
compact = counters("bench", compact_kernel)
out = {"kernel": "sl3d::" + dense, "per_dispatch_mean": bench, "compacting_kernel": "sl3d::" + compact_kernel,
       "compacting_kernel_per_dispatch_mean": compact}
# 2. calibration of FETCH_SIZE / WRITE_SIZE on tools/membench mode 0 (one dword per lane per plane, 47 planes;
#    three 16-B stores + one dword per lane): the same access widths as the fused kernel, with KNOWN byte counts.
mem = counters("membench", "k_dword")
npx = 33177600  # tools/membench 33.1776 rounds to a multiple of 4096
npx = (33177600 + 4095) // 4096 * 4096
cal = {}
if "FETCH_SIZE" in mem and "WRITE_SIZE" in mem:
    cal = {"membench_pixels": npx, "known_read_bytes": 47 * npx, "known_write_bytes": 13 * npx,
           "FETCH_SIZE_KiB": mem["FETCH_SIZE"], "WRITE_SIZE_KiB": mem["WRITE_SIZE"],
           "read_bytes_per_FETCH_KiB": 47 * npx / mem["FETCH_SIZE"], "write_bytes_per_WRITE_KiB": 13 * npx / mem["WRITE_SIZE"]}
    out["calibration"] = cal
    if "FETCH_SIZE" in bench and "WRITE_SIZE" in bench:
        rd = bench["FETCH_SIZE"] * cal["read_bytes_per_FETCH_KiB"]
        wr = bench["WRITE_SIZE"] * cal["write_bytes_per_WRITE_KiB"]
        # pixels and algorithmic bytes per launch of the profiled command: from the bench line saved next to the stats
        px_launch, bpp = 16 * 1920 * 1080, 60
        try:
            bl = json.loads(open(f"{src}/stats_bench.json").read())
            px_launch, bpp = bl["roofline"]["pixels_per_launch"], bl["roofline"]["algorithmic_bytes_per_pixel"]
        except Exception:
            pass
        traffic = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                   "algorithmic_bytes_per_launch": bpp * px_launch, "ratio_to_algorithmic": (rd + wr) / (bpp * px_launch),
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB units); gfx950 counts a wide "
                             "coalesced read at half its bytes and leaves other widths uncalibrated, so both counters are scaled by "
                             "the factors measured on tools/membench mode 0 (same access widths, known byte counts) in the same session",
                   "raw_FETCH_SIZE_KiB": bench["FETCH_SIZE"], "raw_WRITE_SIZE_KiB": bench["WRITE_SIZE"]}
        out["traffic"] = traffic
        json.dump(traffic, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
        # the compacting (segmented) kernel of the same command: its algorithmic bytes depend on the valid fraction of the run
        if "FETCH_SIZE" in compact and "WRITE_SIZE" in compact:
            rdc, wrc = compact["FETCH_SIZE"] * cal["read_bytes_per_FETCH_KiB"], compact["WRITE_SIZE"] * cal["write_bytes_per_WRITE_KiB"]
            cbpp = None
            try:
                cbpp = bl["to_compacted_clouds"]["algorithmic_bytes_per_pixel"]
            except Exception:
                pass
            tc = {"kernel": "sl3d::" + compact_kernel, "pixels_per_launch": px_launch, "hbm_read_bytes_per_launch": rdc,
                  "hbm_write_bytes_per_launch": wrc, "hbm_bytes_per_launch": rdc + wrc,
                  "algorithmic_bytes_per_pixel": cbpp, "algorithmic_bytes_per_launch": cbpp * px_launch if cbpp else None,
                  "ratio_to_algorithmic": (rdc + wrc) / (cbpp * px_launch) if cbpp else None, "method": traffic["method"],
                  "raw_FETCH_SIZE_KiB": compact["FETCH_SIZE"], "raw_WRITE_SIZE_KiB": compact["WRITE_SIZE"]}
            out["traffic_clouds"] = tc
            json.dump(tc, open(f"profiles/{tag}_traffic_clouds.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
