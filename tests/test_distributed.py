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
        # the chunked form bench.py pipelines: view i of a chunk is assembled on rank i % world
        chunk = stripe[:2 * world]
        mine = d.assemble_rotating_interleaved(chunk)
        assert torch.equal(mine, full[:2 * world][rank::world])
        mine = d.assemble_rotating_interleaved(chunk, [d.shard_rows(H, world, r)[1] for r in range(world)])
        assert torch.equal(mine, full[:2 * world][rank::world])
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker_pipeline(rank, world, port, q):
    """RootAssembler on CPU tensors (gloo): dense stripes of UNEQUAL heights land in place; compacted clouds are concatenated
    in rank order; the equalising collectives (assemble_root / assemble_rotating) accept unequal stripes too."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import importlib
    d = importlib.import_module("3dscan_amd.distributed")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, H, P = 5, 11, 16            # 11 rows on 2 ranks: 6 + 5
        g = torch.Generator().manual_seed(7)
        full = torch.rand((V, H, P * 3), generator=g)
        fval = (torch.rand((V, H, P), generator=g) > 0.4).to(torch.uint8)
        rows_by_rank = [d.shard_rows(H, world, r)[1] for r in range(world)]
        r0, rows = d.shard_rows(H, world, rank)
        pts, val = full[:, r0:r0 + rows].contiguous(), fval[:, r0:r0 + rows].contiguous()
        asm = d.RootAssembler(rows_by_rank)
        out_p = torch.zeros_like(full) if rank == 0 else None
        out_v = torch.zeros_like(fval) if rank == 0 else None
        for f, n in ((0, 2), (2, 3)):          # two chunks, as the benchmark's pipeline issues them
            asm.gather_dense(range(f, f + n), pts, val, out_p, out_v)
        if rank == 0:
            assert torch.equal(out_p, full) and torch.equal(out_v, fval)
        # compacted: every rank holds, per view, `count` points at the start of a fixed-stride region
        stride = rows * P
        clouds = torch.zeros(V * stride * 3)
        counts = []
        for v in range(V):
            sel = val[v].reshape(-1) == 1
            p3 = pts[v].reshape(-1, 3)[sel]
            clouds[3 * v * stride:3 * v * stride + p3.numel()] = p3.reshape(-1)
            counts.append(int(sel.sum()))
        allc = asm.gather_counts(counts)
        assert allc[rank] == counts and len(allc) == world
        tot = [sum(allc[r][i] for r in range(world)) for i in range(V)]
        offs = [3 * sum(tot[:i]) for i in range(V)]
        out_c = torch.zeros(3 * sum(tot)) if rank == 0 else None
        asm.gather_compact(range(V), clouds, stride, allc, out_c, offs)
        if rank == 0:
            ref = torch.cat([full[v].reshape(-1, 3)[fval[v].reshape(-1) == 1].reshape(-1) for v in range(V)])
            assert torch.equal(out_c, ref)
        root = d.assemble_root(pts)
        if rank == 0:
            assert torch.equal(root, full)
        full2 = full[:world * 2]
        mine = d.assemble_rotating(full2[:, r0:r0 + rows].contiguous(), 2)
        assert torch.equal(mine, full2[rank * 2:(rank + 1) * 2])
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_pipelined_assembly_unequal_stripes_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_spawn_ranks_environment(tmp_path):
    """bench.py's self-launch helper: fresh children with the torch.distributed.run environment, rank 0's stdout returned,
    a failing rank raises."""
    d = pkg("distributed")
    script = tmp_path / "child.py"
    script.write_text("import os, sys\nprint(os.environ['RANK'], os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], sys.argv[1])\n"
                      "sys.exit(3 if os.environ['RANK'] == sys.argv[2] else 0)\n")
    out = d.spawn_ranks(str(script), ["hello", "-1"], 3)
    assert out.split() == ["0", "0", "3", "127.0.0.1", "hello"]
    with pytest.raises(SystemExit):
        d.spawn_ranks(str(script), ["x", "2"], 3)


def test_spawn_ranks_supervises_its_children(tmp_path):
    """A rank that dies (before a rendezvous, inside a collective ...) must not leave the others behind: the survivors --
    here asleep for a minute, like a rank stuck in init_process_group -- are terminated and the failure is raised at once;
    the same for an overall timeout."""
    import os
    import time
    d = pkg("distributed")
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "open(os.path.join(sys.argv[1], 'pid' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                      "if os.environ['RANK'] == sys.argv[2]:\n    sys.exit(7)\n"
                      "time.sleep(60)\n")

    def alive(pid):
        try:
            os.kill(pid, 0)
            return open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (OSError, IOError):
            return False

    for args, kw in ((["1"], {}), (["-1"], {"timeout": 1.5})):
        for f in tmp_path.glob("pid*"):
            f.unlink()
        t0 = time.monotonic()
        with pytest.raises(SystemExit) as e:
            d.spawn_ranks(str(script), [str(tmp_path)] + args, 3, **kw)
        assert time.monotonic() - t0 < 45   # (the children would sleep for 60 s)
        assert ("failed" in str(e.value)) if not kw else ("did not finish" in str(e.value))
        pids = [int(f.read_text()) for f in tmp_path.glob("pid*")]
        assert len(pids) == 3 and not any(alive(p) for p in pids), "orphaned ranks"


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
