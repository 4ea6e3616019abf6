"""CPU tests of the multi-GPU decomposition (world_size 2, gloo): row sharding covers the frame exactly,
max-over-ranks timing, and both cloud-assembly schemes rebuild full views in the reference's row order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def test_shard_rows_partition():
    d = pkg("distributed")
    for H in (1080, 3000, 7, 135):
        for world in (1, 2, 3, 4, 8):
            rows = [d.shard_rows(H, world, r) for r in range(world)]
            assert rows[0][0] == 0
            assert sum(n for _, n in rows) == H
            for (a0, an), (b0, _) in zip(rows, rows[1:]):
                assert a0 + an == b0
            assert max(n for _, n in rows) - min(n for _, n in rows) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import importlib
    d = importlib.import_module("3dscan_amd.distributed")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert d.env_ranks() == (rank, rank, world)
        # a fake batch: V views per rank, full frames of H rows; every rank holds its row stripe of every view
        V, H, W = 3, 8, 5
        full = torch.arange(world * V * H * W * 3, dtype=torch.float32).reshape(world * V, H, W, 3)
        r0, rows = d.shard_rows(H, world, rank)
        stripe = full[:, r0:r0 + rows].contiguous()
        t = d.max_over_ranks(1.0 + rank)
        assert t == float(world)
        root = d.assemble_root(stripe, rows)
        if rank == 0:
            assert torch.equal(root, full)
        else:
            assert root is None
        mine = d.assemble_rotating(stripe, V)          # views [rank*V, (rank+1)*V) assembled here
        assert torch.equal(mine, full[rank * V:(rank + 1) * V])
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_assembly_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
