"""Multi-GPU decomposition of the hot path: one process per GPU, frames sharded by image rows.

Every stage is a per-pixel map whose only neighbourhood input is the selection MASK (an input, not
a computed quantity), so a row stripe needs no data from other GPUs: each rank gets the full-frame
mask and keeps 2 halo rows.  The only exchange is the ASSEMBLY of the point cloud of each view on one
GPU (north star: "a single RCCL gather over xGMI"), through torch.distributed (backend "nccl" = RCCL
over xGMI on MI355X; "gloo" with host staging in the plumbing tests):

  RootAssembler     : the gather, pipelined per chunk of views: while the fused kernel works on chunk
                      k+1 on the compute stream, the stripes of chunk k leave on a communication stream
                      as ONE grouped batch of point-to-point sends that land in place in the root's
                      dense [view][row] buffers (rank order = row order, so the reference's row-major
                      scan order of 8/save_point_cloud.cpp:85 is preserved; stripes may have unequal
                      heights).  dense = xyz f32 + valid u8 (13 B/px); compact = the valid points only,
                      compacted in scan order by the fused kernel itself (counts exchanged first).
  assemble_root     : the same gather as one blocking collective over a whole batch (equalised shapes).
  assemble_rotating : view v is assembled on rank v // views_per_rank, one all_to_all for the whole
                      batch, so all point-to-point xGMI links carry traffic at once instead of only
                      the 7 links into one root.

torch is plumbing here (process group, streams, device buffers); the compute never goes through it.
"""
import os
import socket
import subprocess
import sys


def env_ranks():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def shard_rows(height, world, rank):
    """Contiguous row block [row0, row0+rows) of rank `rank`; remainder rows go to the first ranks."""
    base, rem = divmod(height, world)
    rows = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, rows


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(script, argv, world, extra_env=None, timeout=None, poll_s=0.2):
    """Start `world` FRESH python processes of `script argv...`, one per rank, with the environment torch.distributed.run
    would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  The caller must not have touched the GPU
    (and never exec's): the children are ordinary subprocesses.  Rank 0's stdout is returned; every rank's stderr is
    inherited (ranks > 0 send their stdout there too, so nothing a rank says is lost).

    The ranks are SUPERVISED: all of them are polled; the first non-zero exit, `timeout` seconds without completion, or an
    interrupt of the parent terminates (then kills) the survivors -- a rank that died before or inside a collective never
    leaves the others holding their GPUs in init_process_group or an RCCL call.  Raises SystemExit naming the failed ranks."""
    import threading
    import time
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else 2))   # fd 2: the parent's stderr
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # the pipe never fills up
    reader.start()

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    why = None
    t0 = time.monotonic()
    try:
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                why = f"rank(s) failed: {bad}"
                break
            if all(rc == 0 for rc in rcs):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                why = f"ranks did not finish within {timeout} s (still running: {[r for r, rc in enumerate(rcs) if rc is None]})"
                break
            time.sleep(poll_s)
    except BaseException:   # KeyboardInterrupt included: no orphaned ranks
        stop_all()
        raise
    if why:
        stop_all()
    reader.join(timeout=5.0)
    out = b"".join(chunks).decode(errors="replace")
    if why:
        sys.stdout.write(out)
        raise SystemExit(why)
    return out


def max_over_ranks(seconds, device=None):
    """MAX of a python float over all ranks (1 rank: identity)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _equalise_rows(stripe, dim=1):
    """Pad `stripe` along `dim` to the largest row count of any rank (collectives need identical shapes); returns
    (padded stripe, list of every rank's true row count)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([stripe.shape[dim]], dtype=torch.int64, device=stripe.device if dist.get_backend() == "nccl" else "cpu")
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    rows = [int(r.item()) for r in rows]
    mx = max(rows)
    if stripe.shape[dim] < mx:
        pad = list(stripe.shape)
        pad[dim] = mx - stripe.shape[dim]
        stripe = torch.cat([stripe, stripe.new_zeros(pad)], dim=dim)
    return stripe.contiguous(), rows


def assemble_root(stripe, rows_per_rank=None, root=0):
    """stripe: [views, rows, ...] tensor of this rank's rows of every view (row counts may differ by rank: the stripes
    are padded to a common height for the collective and trimmed afterwards).
    Returns on `root` the assembled [views, sum(rows), ...] tensor (rank order = row order, so the
    reference's row-major scan order of 8/save_point_cloud.cpp:85 is preserved), None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if rows_per_rank is not None:
        assert stripe.shape[1] == rows_per_rank
    stripe, rows = _equalise_rows(stripe)
    if rank == root:
        parts = [torch.empty_like(stripe) for _ in range(world)]
        dist.gather(stripe, gather_list=parts, dst=root)
        return torch.cat([p[:, :n] for p, n in zip(parts, rows)], dim=1)
    dist.gather(stripe, gather_list=None, dst=root)
    return None


def assemble_rotating(stripe, views_per_rank):
    """stripe: [world*views_per_rank, rows, ...].  View v is assembled on rank v // views_per_rank.
    Returns this rank's [views_per_rank, sum(rows), ...] assembled views (one all_to_all_single; unequal stripe
    heights are padded for the collective and trimmed afterwards)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    assert stripe.shape[0] == world * views_per_rank
    stripe, rows = _equalise_rows(stripe)
    out = torch.empty_like(stripe)  # [src_rank * views_per_rank + j, rows, ...]
    dist.all_to_all_single(out, stripe)
    mx = stripe.shape[1]
    out = out.view(world, views_per_rank, mx, *stripe.shape[2:])
    return torch.cat([out[r, :, :rows[r]] for r in range(world)], dim=1)


def assemble_rotating_interleaved(stripe, rows_by_rank=None):
    """stripe: [n, rows, ...] with n a multiple of the world size -- a CHUNK of consecutive views.  View i of the chunk is assembled on
    rank i % world (so every chunk keeps every link busy, and a pipeline can exchange chunk k while chunk k + 1 computes).
    Returns this rank's [n // world, sum(rows), ...] assembled views: chunk views rank, rank + world, ...  One all_to_all_single.
    rows_by_rank: every rank's stripe height, if the caller knows it (no collective, no host wait to find out)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n = stripe.shape[0]
    assert n % world == 0, (n, world)
    m = n // world
    if rows_by_rank is None:
        stripe, rows = _equalise_rows(stripe)       # (one small all_gather and a host wait: a pipeline passes the row counts instead)
    else:
        rows, mx = list(rows_by_rank), max(rows_by_rank)
        if stripe.shape[1] < mx:
            pad = list(stripe.shape)
            pad[1] = mx - stripe.shape[1]
            stripe = torch.cat([stripe, stripe.new_zeros(pad)], dim=1)
    mx = stripe.shape[1]
    send = stripe.view(m, world, mx, *stripe.shape[2:]).transpose(0, 1).contiguous()   # [dst, j, rows, ...] = view j * world + dst
    out = torch.empty_like(send)                                                       # [src, j, rows, ...]
    dist.all_to_all_single(out, send)
    return torch.cat([out[r, :, :rows[r]] for r in range(world)], dim=1)


class RootAssembler:
    """Pipelined gather of row stripes to one root, chunk of views by chunk of views.

    dense mode   : every rank sends, per view, its [rows, pitch*3] f32 xyz slab and [rows, pitch] u8 valid slab; they land
                   in place in the root's [views, H, ...] buffers (a stripe of a view is a contiguous slab there).
    compact mode : every rank sends, per view, the `count` valid points of its stripe (compacted in scan order by the fused
                   kernel); the root concatenates them in rank order = the reference's scan order.  Counts travel first
                   (one small all_gather per chunk), the payload sizes follow from them.

    nccl: tensors are device tensors and the batch of sends/recvs of a chunk is ONE RCCL group on `comm_stream`, which
    waits for the compute event of that chunk -- so the compute stream never waits for communication.
    gloo (plumbing tests only): the same schedule, staged through host memory."""

    def __init__(self, rows_by_rank, root=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.root = dist.get_world_size(), dist.get_rank(), root
        self.rows = list(rows_by_rank)
        self.row0 = [sum(self.rows[:r]) for r in range(self.world)]
        self.nccl = dist.get_backend() == "nccl"
        self.comm_stream = torch.cuda.Stream() if torch.cuda.is_available() else None

    # -- helpers ---------------------------------------------------------------------------------------------------
    def _exchange(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(tensor, src)] -- one grouped batch.  Returns when the batch has been ENQUEUED
        (nccl: on the current stream) or has completed (gloo)."""
        dist = self.dist
        ops = [dist.P2POp(dist.isend, t, d) for t, d in sends] + [dist.P2POp(dist.irecv, t, s) for t, s in recvs]
        if not ops:
            return
        for w in dist.batch_isend_irecv(ops):
            w.wait()  # nccl: makes the current stream wait for the group, the host does not block

    def gather_dense(self, views, pts, val, out_pts, out_val):
        """pts [n, rows, pitch*3] f32 / val [n, rows, pitch] u8: this rank's stripes (device).  out_* (root only): dense
        [n, H, ...] device buffers.  Sends views `views` (an iterable of indices).  Enqueues on the current stream."""
        torch = self.torch
        r0, n = self.row0[self.rank], self.rows[self.rank]
        if self.rank == self.root:
            recvs = []
            stage = []
            for v in views:
                out_pts[v, r0:r0 + n].copy_(pts[v], non_blocking=True)
                out_val[v, r0:r0 + n].copy_(val[v], non_blocking=True)
                for src in range(self.world):
                    if src == self.root or self.rows[src] == 0:
                        continue
                    a, b = self.row0[src], self.row0[src] + self.rows[src]
                    if self.nccl:
                        recvs += [(out_pts[v, a:b], src), (out_val[v, a:b], src)]
                    else:
                        hp = torch.empty(out_pts[v, a:b].shape, dtype=out_pts.dtype)
                        hv = torch.empty(out_val[v, a:b].shape, dtype=out_val.dtype)
                        recvs += [(hp, src), (hv, src)]
                        stage += [(out_pts[v, a:b], hp), (out_val[v, a:b], hv)]
            self._exchange([], recvs)
            for dst, h in stage:
                dst.copy_(h)
        elif n > 0:
            sends = []
            for v in views:
                if self.nccl:
                    sends += [(pts[v], self.root), (val[v], self.root)]
                else:
                    sends += [(pts[v].cpu(), self.root), (val[v].cpu(), self.root)]
            self._exchange(sends, [])

    def gather_counts(self, counts):
        """counts: python list (one per view of the chunk) of this rank's valid points -> [world][n] list on every rank."""
        torch, dist = self.torch, self.dist
        dev = "cuda" if self.nccl else "cpu"
        mine = torch.tensor(counts, dtype=torch.int64, device=dev)
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine)
        return [[int(x) for x in c.tolist()] for c in allc]

    def gather_compact(self, views, clouds, view_stride_pts, all_counts, out_cloud, out_offsets):
        """clouds: flat f32 device tensor, view v's compacted points at [3*v*view_stride_pts, +3*count).  all_counts
        [world][len(views)] from gather_counts.  out_cloud (root): flat f32 device buffer; out_offsets (root): per view,
        the float offset at which that view's assembled cloud starts.  Enqueues on the current stream."""
        torch = self.torch
        if self.rank == self.root:
            recvs, stage = [], []
            for i, v in enumerate(views):
                off = out_offsets[i]
                for src in range(self.world):
                    c = all_counts[src][i]
                    if c:
                        dst = out_cloud[off:off + 3 * c]
                        if src == self.root:
                            dst.copy_(clouds[3 * v * view_stride_pts:3 * v * view_stride_pts + 3 * c], non_blocking=True)
                        elif self.nccl:
                            recvs.append((dst, src))
                        else:
                            h = torch.empty(3 * c, dtype=out_cloud.dtype)
                            recvs.append((h, src))
                            stage.append((dst, h))
                    off += 3 * c
            self._exchange([], recvs)
            for dst, h in stage:
                dst.copy_(h)
        else:
            sends = []
            for i, v in enumerate(views):
                c = all_counts[self.rank][i]
                if c:
                    t = clouds[3 * v * view_stride_pts:3 * v * view_stride_pts + 3 * c]
                    sends.append((t if self.nccl else t.cpu(), self.root))
            self._exchange(sends, [])
