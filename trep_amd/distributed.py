"""Batch sharding across the GPUs of one node.

Trajectories are independent, so the batch is cut into contiguous per-rank slices and every rank
runs its slice on its own GPU with no data-path communication.  The only collective is an
all-gather of per-trajectory results (terminal states / costs) after a rollout: what a discopt
line-search needs to see on every rank (``trep_amd/rccl.py``: RCCL behind the C ABI, no torch).
This module is the shard arithmetic only.
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


def padded_rows(total, world):
    """Rows of the largest shard: what every rank's block is padded to for the fixed-size all-gather."""
    return shard_bounds(total, 0, world)[1]
