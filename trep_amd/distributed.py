"""Batch sharding across the GPUs of one node.

Trajectories are independent, so the batch is cut into contiguous per-rank slices and every rank
runs its slice on its own GPU with no data-path communication.  The only collective is an
all-gather of per-trajectory results (terminal states / costs) after a rollout: what a discopt
line-search needs to see on every rank.  One process per GPU; ``torch.distributed`` provides the
process group (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
torch is plumbing here, imported lazily; the integrator itself never needs it.
"""


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of `total` items owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(array, rank, world):
    lo, hi = shard_bounds(len(array), rank, world)
    return array[lo:hi]


def all_gather_rows(local, group=None):
    """All-gather row blocks of possibly different lengths; returns the concatenation in rank order.

    `local` is a torch tensor [n_local, width] (CUDA for nccl, CPU for gloo).  Shards produced by
    shard_bounds differ by at most one row, so shorter shards are padded by one row for the
    fixed-size collective and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    padded = local
    if local.shape[0] < n_max:
        pad = torch.zeros((n_max - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    out = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    parts = [out[r * n_max:r * n_max + counts[r]] for r in range(world)]
    return torch.cat(parts, 0)


def max_over_ranks(value, device="cpu", group=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
