"""Sharding of a batch of independent planning problems over the GPUs of one node.

The path has no exchange step: world `w` of a batch belongs to exactly one rank, ranks never communicate on
the data path, and the only collectives are the barrier and the MAX-reduction of the timing in bench.py.
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block partition of `total` problems: rank r owns [lo, hi); sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seeds(first_seed, total, rank, world):
    """World seeds owned by `rank` (armour_amd.worlds.random_problem(seed, O))."""
    lo, hi = shard_range(total, rank, world)
    return list(range(first_seed + lo, first_seed + hi))


def reduce_max_elapsed(elapsed_seconds, device=None):
    """max over ranks of a wall-clock interval (what bench.py divides the total work by)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_counts(count, device=None):
    """all-gather of one integer per rank (problems processed), for the aggregate throughput."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([count], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [int(x[0]) for x in out]
    return [int(count)]
