"""Multi-GPU decomposition of the hot path: one process per GPU, frames sharded by image rows.

Every stage is a per-pixel map whose only neighbourhood input is the selection MASK (an input, not
a computed quantity), so a row stripe needs no data from other GPUs: each rank gets the full-frame
mask and keeps 2 halo rows.  The only exchange is the optional ASSEMBLY of the dense point cloud
(xyz f32 + valid u8) of each view on one GPU, through torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" in the CPU tests):

  assemble_root     : every stripe of every view goes to rank 0 (the literal "single gather").
  assemble_rotating : view v is assembled on rank v // views_per_rank, one all_to_all for the whole
                      batch, so all point-to-point xGMI links carry traffic at once instead of only
                      the 7 links into one root.

torch is plumbing here (process group + device buffers); the compute never goes through it.
"""
import os


def env_ranks():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def shard_rows(height, world, rank):
    """Contiguous row block [row0, row0+rows) of rank `rank`; remainder rows go to the first ranks."""
    base, rem = divmod(height, world)
    rows = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, rows


def max_over_ranks(seconds, device=None):
    """MAX of a python float over all ranks (1 rank: identity)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def assemble_root(stripe, rows_per_rank, root=0):
    """stripe: [views, rows, ...] tensor of this rank's rows of every view (equal rows on every rank).
    Returns on `root` the assembled [views, world*rows, ...] tensor (rank order = row order, so the
    reference's row-major scan order of 8/save_point_cloud.cpp:85 is preserved), None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    assert stripe.shape[1] == rows_per_rank
    stripe = stripe.contiguous()
    if rank == root:
        parts = [torch.empty_like(stripe) for _ in range(world)]
        dist.gather(stripe, gather_list=parts, dst=root)
        return torch.cat(parts, dim=1)
    dist.gather(stripe, gather_list=None, dst=root)
    return None


def assemble_rotating(stripe, views_per_rank):
    """stripe: [world*views_per_rank, rows, ...].  View v is assembled on rank v // views_per_rank.
    Returns this rank's [views_per_rank, world*rows, ...] assembled views (one all_to_all_single)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    assert stripe.shape[0] == world * views_per_rank
    stripe = stripe.contiguous()
    out = torch.empty_like(stripe)  # [src_rank * views_per_rank + j, rows, ...]
    dist.all_to_all_single(out, stripe)
    rows = stripe.shape[1]
    out = out.view(world, views_per_rank, rows, *stripe.shape[2:])
    return out.transpose(0, 1).reshape(views_per_rank, world * rows, *stripe.shape[2:])
