"""Multi-GPU inference = sharding independent clips / batches over ranks; there is no exchange step in the
data path, so no collective is used.  Only the tiny per-batch results (a few floats or
[L,2] frames) are gathered on the host at the end, through whatever torch.distributed backend is up
(RCCL on the GPU box, gloo in the CPU tests)."""
from __future__ import annotations

from typing import Any, Callable, List, Sequence, Tuple


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment: item i belongs to rank i % world.  Keeps every rank's work within one item of
    the others even when item cost varies slowly along the dataset (songs sorted by length)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return list(range(rank, n_items, world))


def gather_indexed(local: Sequence[Tuple[int, Any]], world: int) -> List[Tuple[int, Any]]:
    """All ranks -> every rank holds all (index, value) pairs, sorted by index."""
    if world == 1:
        return sorted(local, key=lambda p: p[0])
    import torch.distributed as dist
    bucket: List[Any] = [None] * world
    dist.all_gather_object(bucket, list(local))
    merged = [p for part in bucket for p in part]
    return sorted(merged, key=lambda p: p[0])


def map_sharded(fn: Callable[[int], Any], n_items: int, rank: int, world: int) -> List[Any]:
    """Apply fn to this rank's items, gather, return the values in item order (identical on every rank)."""
    local = [(i, fn(i)) for i in shard_indices(n_items, rank, world)]
    return [v for _, v in gather_indexed(local, world)]
