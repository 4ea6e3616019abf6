"""GPU test (-m gpu) of bench.py's N > 1 branch end to end on ONE GPU: `python bench.py --gpus N` spawns its own ranks, every
rank computes its row stripe on GPU 0 (gloo backend, host-staged assembly: RCCL needs one GPU per rank), rank 0 assembles the
dense planes and the compacted clouds through the pipelined gather, and their digests equal those of a 1-rank run over the
same global batch of views."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*argv):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


COMMON = ["--steps", "3", "--warmup", "1", "--precondition-ms", "0", "--no-cpu-baseline", "--no-side", "--check"]


@pytest.mark.parametrize("world,height", [(2, 1080), (3, 1000)])
def test_bench_spawns_ranks_and_assembly_equals_single_rank(world, height):
    views = 2
    one = _bench("--gpus", "1", "--views", str(views * world), "--height", str(height), *COMMON)
    many = _bench("--gpus", str(world), "--backend", "gloo", "--devices", ",".join(["0"] * world), "--views", str(views),
                  "--height", str(height), "--chunks", "2", *COMMON)
    assert one["n_gpus"] == 1 and many["n_gpus"] == world and many["scaling"] == "weak"
    assert many["config"]["rows_per_gpu"] == -(-height // world)
    wa = many["with_assembly"]
    assert "error" not in wa, wa
    assert wa["dense_root_gather"]["value"] > 0 and wa["compact_root_gather"]["value"] > 0
    # the headline of an N > 1 line is an assembled figure over exactly --steps steps; the compute-only rate sits beside it
    assert many["value_is"] == wa["headline"] and many["value"] == wa[wa["headline"]]["headline_value"] > 0
    # ... of a variant that leaves COMPLETE DENSE views on a rank; the host-parallel and compacted variants sit beside it
    assert wa["headline"].startswith("dense_") and wa["host_parallel"]["value"] > 0
    assert wa["compact_root_gather_19pct_selection"]["value"] > 0 and abs(wa["compact_root_gather_19pct_selection"]["selected_fraction"] - 0.187) < 0.01
    assert 0 < wa["compact_root_gather_19pct_selection"]["points_per_step"] < 0.25 * wa["compact_root_gather"]["points_per_step"]
    assert [r["rank"] for r in many["ranks"]] == list(range(world)) and all(r["comm_size"] == world and r["pci_bus_id"] and r["rccl_version"] for r in many["ranks"])
    assert many["compute_only"]["value"] > 0
    assert wa["whole_views_no_exchange"]["value"] > 0 and many["value_is"] != "whole_views_no_exchange"   # the other sharding: reported, never the headline
    assert many["check"]["dense_sha256"] == one["check"]["dense_sha256"]
    assert many["check"]["compact_sha256"] == one["check"]["compact_sha256"]
    assert wa["compact_root_gather"]["points_per_step"] == one["to_compacted_clouds"]["valid_points_per_step_rank0"]


def test_bench_refuses_wrong_world_and_stacked_rccl():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
    # RCCL with two ranks on one GPU is refused loudly instead of stacking them
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--steps", "1"],
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "one GPU per rank" in (p.stderr + p.stdout)


def test_bench_strong_scaling_is_configs3_as_stated():
    """--scaling strong: BASELINE configs[3] itself -- a FIXED batch of views, each row-sharded over the N ranks; the assembled digests
    equal a 1-rank run over the same batch."""
    one = _bench("--gpus", "1", "--scaling", "strong", "--total-views", "6", "--height", "1000", *COMMON)
    many = _bench("--gpus", "3", "--backend", "gloo", "--devices", "0,0,0", "--scaling", "strong", "--total-views", "6", "--height", "1000", "--chunks", "2", *COMMON)
    assert one["scaling"] == many["scaling"] == "strong" and one["config"]["views_per_step"] == many["config"]["views_per_step"] == 6
    assert many["config"]["rows_per_gpu"] == 334 and one["config"]["rows_per_gpu"] == 1000
    assert many["check"]["dense_sha256"] == one["check"]["dense_sha256"] and many["check"]["compact_sha256"] == one["check"]["compact_sha256"]
