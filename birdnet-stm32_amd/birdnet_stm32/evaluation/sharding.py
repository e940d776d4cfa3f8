"""Data-parallel sharding of the chunk stream over the GPUs of one node (no counterpart in the reference,
which is single-process: SURVEY.md §8e).

Chunks are independent (reference: birdnet_stm32/evaluation/metrics.py:128-141), so rank ``r`` of ``R`` takes
the contiguous block ``[r*N/R, (r+1)*N/R)`` of the global chunk index, runs it in batches, and the scores meet
in ONE all-gather at the end (RCCL over xGMI on MI355X — ``torch.distributed`` backend ``nccl``; ``gloo`` on
CPU in the tests).  Ragged shards are padded to the largest shard for the collective and trimmed afterwards;
equal shards (the benchmark's case: N a multiple of R) are gathered straight into the result.

Callers: ``bench.py`` (BASELINE configs[3]: every rank scores its block of the synthetic chunk stream through
:func:`run_sharded`) and ``evaluation.metrics.evaluate`` (:func:`score_files_sharded`: files are dealt to the ranks
in contiguous blocks, the chunk scores of all files meet in one all-gather, pooling happens after it).
"""

from __future__ import annotations

# With a process group of ONE rank the gathers below return the local tensor.  Set to True (bench.py --collective, the GPU test of the
# RCCL path) to run the collective anyway: the same calls as on 8 GPUs, executed on hardware that has a single one.
COLLECTIVE_AT_WORLD_1 = False


def _local_only(world: int) -> bool:
    import torch.distributed as dist

    return world == 1 and not (COLLECTIVE_AT_WORLD_1 and dist.is_available() and dist.is_initialized())


def world_info(group=None) -> tuple[int, int]:
    """(rank, world size) of the initialised process group, (0, 1) without one."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


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

    rank, world = world_info(group)
    if world == 1 and local_scores.shape[0] != n_items:
        raise ValueError("single-process call must hold all items")
    if _local_only(world):
        return local_scores
    lo, hi = shard_bounds(n_items, rank, world)
    if local_scores.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local_scores.shape[0]} rows, its shard has {hi - lo}")
    if n_items % world == 0:  # equal shards: the collective writes the result itself
        gathered = local_scores.new_empty((n_items, local_scores.shape[1]))
        dist.all_gather_into_tensor(gathered, local_scores.contiguous(), group=group)
        return gathered
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


def run_sharded(score_fn, n_items: int, batch_size: int, group=None, into=None):
    """Score this rank's shard in batches and all-gather the result: ``[n_items, C]`` on every rank.

    ``score_fn(start, stop) -> [stop-start, C]`` scores the global items ``[start, stop)``.  With ``into`` (a
    preallocated ``[shard size, C]`` tensor) the call is ``score_fn(start, stop, out_rows)`` and must fill ``out_rows``
    (no concatenation afterwards).  Fewer items than ranks is refused on EVERY rank before any work (the test depends on
    ``n_items`` and the world size only), so no rank is left waiting in the collective.
    """
    import torch

    rank, world = world_info(group)
    if n_items < world:
        raise ValueError(f"{n_items} items cannot be sharded over {world} ranks: every rank needs at least one")
    lo, hi = shard_bounds(n_items, rank, world)
    if into is not None:
        if into.shape[0] != hi - lo:
            raise ValueError(f"`into` has {into.shape[0]} rows, the shard of rank {rank} has {hi - lo}")
        for s in range(lo, hi, batch_size):
            e = min(s + batch_size, hi)
            score_fn(s, e, into[s - lo : e - lo])
        local = into
    else:
        outs = [score_fn(s, min(s + batch_size, hi)) for s in range(lo, hi, batch_size)]
        local = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    return all_gather_scores(local, n_items, group=group)


def all_gather_ragged(local, group=None):
    """All-gather ``[n_r, C]`` tensors whose row counts differ per rank: ``(global [sum n_r, C], counts per rank)``.

    Two collectives: the row counts (one int64 per rank), then ONE all-gather of the rows padded to the longest shard.
    """
    import torch
    import torch.distributed as dist

    rank, world = world_info(group)
    if _local_only(world):
        return local, [int(local.shape[0])]
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = [int(c) for c in counts.cpu()]
    longest = max(counts)
    if longest == 0:
        return local, counts
    padded = local.new_zeros((longest, local.shape[1]))
    padded[: local.shape[0]] = local
    gathered = local.new_empty((world * longest, local.shape[1]))
    dist.all_gather_into_tensor(gathered, padded, group=group)
    return torch.cat([gathered[r * longest : r * longest + counts[r]] for r in range(world)], dim=0), counts


def score_files_sharded(n_files: int, score_files_fn, width: int, device=None, group=None, bounds=None):
    """File-level sharding for ``evaluate``: rank r scores the files of its contiguous block, the chunk scores of all
    files meet in one (ragged) all-gather.

    ``score_files_fn(lo, hi) -> (scores [n_chunks, width], chunks per file [hi - lo])`` scores files ``[lo, hi)``.
    Returns ``(scores of all chunks in file order [N, width], chunks per file for all n_files)`` on every rank; pooling
    happens on the gathered tensor (the file -> chunk map stays on the host).

    ``bounds`` (``world + 1`` non-decreasing file indices from 0 to ``n_files``, the same list on every rank) replaces the
    equal-count blocks: ``evaluate`` passes blocks of equal CHUNK count (``audio.pipeline.balanced_bounds`` over the probed
    headers), so that a dataset of mixed file lengths does not leave ranks idle.
    """
    import torch
    import torch.distributed as dist

    rank, world = world_info(group)
    if bounds is not None:
        bounds = [int(b) for b in bounds]
        if len(bounds) != world + 1 or bounds[0] != 0 or bounds[-1] != n_files or any(a > b for a, b in zip(bounds, bounds[1:])):
            raise ValueError(f"bounds {bounds} do not partition {n_files} files over {world} ranks")
        block = lambda r: (bounds[r], bounds[r + 1])  # noqa: E731
    else:
        block = lambda r: shard_bounds(n_files, r, world)  # noqa: E731
    lo, hi = block(rank)
    if hi > lo:
        scores, per_file = score_files_fn(lo, hi)
    else:
        scores, per_file = torch.empty((0, width), dtype=torch.float32, device=device), []
    if len(per_file) != hi - lo or scores.shape[0] != sum(per_file):
        raise ValueError("score_files_fn must return one chunk count per file and as many rows as chunks")
    if _local_only(world):
        return scores, list(per_file)
    all_scores, _ = all_gather_ragged(scores, group=group)
    longest = max(1, max(block(r)[1] - block(r)[0] for r in range(world)))
    mine = torch.zeros(longest, dtype=torch.int64, device=scores.device)
    if per_file:
        mine[: hi - lo] = torch.tensor(per_file, dtype=torch.int64, device=scores.device)
    every = torch.empty(world * longest, dtype=torch.int64, device=scores.device)
    dist.all_gather_into_tensor(every, mine, group=group)
    every = every.cpu().tolist()
    counts: list[int] = []
    for r in range(world):
        a, b = block(r)
        counts += every[r * longest : r * longest + (b - a)]
    return all_scores, counts
