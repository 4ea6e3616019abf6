"""CPU check of the benchmark contract on the committed line of the round's last build (profiles/r05_bench.json, written by
`python bench.py` on the GPU box): the keys the driver and the judge read are there, with the meaning the task gives them."""
import glob
import json
import os

from conftest import ROOT


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return json.load(open(files[-1]))


def test_committed_bench_line_keeps_the_contract():
    d = _latest("r[0-9][0-9]_bench.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "Mpixels/s" and d["metric"].split(";")[0].strip() in base["metric"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"].startswith("synthetic")
    assert abs(d["value"] - d["config"]["views_per_gpu_per_step"] * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    assert "workload" in d["config"] and "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - 60 * r["pixels_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["achieved"] < r["peak"] and r["traffic"] and r["traffic"] >= 0.99 * 60 * r["pixels_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "Mpixels/s" and c["value"] > 0 and c["sample"]
    assert c["gpu_matches_oracle"]["valid_map_bit_exact"] is True and c["gpu_matches_oracle"]["max_rel_point_error"] <= 1e-5
    s = d["side"]
    # the per-scan figures the round-4 review asked for sit beside the headline
    assert s["per_scan_device"]["scan_us"] <= 34.0 and s["per_scan_device"]["mask_us"] <= 8.0
    assert s["one_view_cold_clouds"]["frac"] >= 0.57
    assert d["set_mask_us"]["pinned"]["until_ready"] <= 70.0
    assert {"resident", "with_upload", "with_upload_prewarm"} <= set(s["one_scan_from_idle"])


def test_committed_kernel_stats_agree_with_the_bench_line():
    """The rocprofv3 average of the dense kernel (profiles/rNN_kernel_stats.csv) and the HIP-event figure of the bench line under the
    profiler belong to the same session: within 2 %."""
    import csv
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats.csv")))
    rows = list(csv.DictReader(open(files[-1])))
    under = json.load(open(files[-1].replace("_kernel_stats.csv", "_bench_under_rocprof.json")))
    name = under["roofline"]["kernel"].replace("sl3d::", "")
    k = [r for r in rows if name in r["Name"]]
    assert len(k) == 1, name
    avg_ms = float(k[0]["AverageNs"]) / 1e6
    assert abs(avg_ms - under["roofline"]["avg_launch_ms"]) < 0.02 * avg_ms
