"""Data-parallel sharding of the chunk stream over the GPUs of one node (no counterpart in the reference,
which is single-process: SURVEY.md §8e).

Chunks are independent (reference: birdnet_stm32/evaluation/metrics.py:128-141), so rank ``r`` of ``R`` takes
the contiguous block ``[r*N/R, (r+1)*N/R)`` of the global chunk index, runs it in batches, and the scores meet
in ONE all-gather at the end (RCCL over xGMI on MI355X — ``torch.distributed`` backend ``nccl``; ``gloo`` on
CPU in the tests).  Ragged shards are padded to the largest shard for the collective and trimmed afterwards.
"""

from __future__ import annotations


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Half-open range of the global index owned by ``rank``; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_scores(local_scores, n_items: int, group=None):
    """All-gather per-rank ``[n_local, C]`` score tensors into the global ``[n_items, C]`` tensor (same on every rank)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        if local_scores.shape[0] != n_items:
            raise ValueError("single-process call must hold all items")
        return local_scores
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n_items, rank, world)
    if local_scores.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local_scores.shape[0]} rows, its shard has {hi - lo}")
    longest = -(-n_items // world)
    padded = local_scores.new_zeros((longest, local_scores.shape[1]))
    padded[: hi - lo] = local_scores
    gathered = local_scores.new_empty((world * longest, local_scores.shape[1]))
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_items, r, world)
        parts.append(gathered[r * longest : r * longest + (b - a)])
    return torch.cat(parts, dim=0)


def run_sharded(score_fn, n_items: int, batch_size: int, group=None):
    """Score this rank's shard in batches with ``score_fn(start, stop) -> [stop-start, C]`` and all-gather the result."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(n_items, rank, world)
    outs = [score_fn(s, min(s + batch_size, hi)) for s in range(lo, hi, batch_size)]
    local = torch.cat(outs, dim=0) if outs else None
    if local is None:
        raise ValueError("empty shard: fewer items than ranks is not supported")
    return all_gather_scores(local, n_items, group=group)
