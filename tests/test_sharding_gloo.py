"""The N > 1 path on CPU: gloo processes (world size 2, 4 and 8) shard a chunk stream and meet in one all-gather."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, REPO

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from birdnet_stm32.evaluation.sharding import run_sharded, shard_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
N, C = int(sys.argv[1]), 5
calls = []
def score(a, b):   # a deterministic stand-in for the GPU path: row g of the result identifies chunk g
    calls.append((a, b))
    g = torch.arange(a, b, dtype=torch.float32)
    return torch.stack([g * (c + 1) + 0.25 * rank * 0 for c in range(C)], dim=1)
out = run_sharded(score, N, batch_size=4)
lo, hi = shard_bounds(N, rank, world)
assert calls[0][0] == lo and calls[-1][1] == hi and all(b - a <= 4 for a, b in calls)
expect = torch.arange(N, dtype=torch.float32)[:, None] * torch.arange(1, C + 1, dtype=torch.float32)[None, :]
assert out.shape == (N, C) and torch.equal(out, expect), (rank, out[:3])
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", lo, hi)
"""


WORKER2 = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from birdnet_stm32.evaluation.sharding import all_gather_ragged, run_sharded, score_files_sharded, shard_bounds
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
C = 7
# 1) bench.py's timed region with a fake scorer: K batches per rank into one buffer, ONE all-gather, equal shards
K, B = 3, 4
seen = []
def score_batch(k, out_rows):   # row j of the rank's batch k is global chunk (rank*K + k)*B + j
    seen.append(k)
    g = torch.arange(B, dtype=torch.float32) + (rank * K + k) * B
    out_rows.copy_(g[:, None] * torch.arange(1, C + 1, dtype=torch.float32)[None, :])
calls = {"barrier": 0, "sync": 0}
def barrier():
    calls["barrier"] += 1
    dist.barrier()
def sync():
    calls["sync"] += 1
dt, gathered = bench.timed_job(score_batch, K, B, C, torch.device("cpu"), barrier, sync)
assert seen == [0, 1, 2] and calls == {"barrier": 2, "sync": 2} and dt > 0
expect = torch.arange(world * K * B, dtype=torch.float32)[:, None] * torch.arange(1, C + 1, dtype=torch.float32)[None, :]
assert torch.equal(gathered, expect)
# 2) fewer items than ranks: every rank raises before any work (no rank is left in the collective)
try:
    run_sharded(lambda a, b: torch.zeros((b - a, C)), 1, 4)
    raise SystemExit("expected ValueError")
except ValueError as e:
    assert "cannot be sharded" in str(e)
# 3) ragged all-gather
local = torch.full((3 + 2 * rank, C), float(rank))
allrows, counts = all_gather_ragged(local)
assert counts == [3, 5] and allrows.shape == (8, C) and torch.equal(allrows[:3], torch.zeros(3, C)) and torch.equal(allrows[3:], torch.ones(5, C))
# 4) evaluate's file-level sharding: 5 files with 1..5 chunks; file f's chunk rows carry f
per = [1, 2, 3, 4, 5]
def score_files(lo, hi):
    rows = [torch.full((per[f], C), float(f)) for f in range(lo, hi)]
    return torch.cat(rows), per[lo:hi]
scores, counts = score_files_sharded(5, score_files, C)
assert counts == per and scores.shape == (15, C)
assert torch.equal(scores[:, 0], torch.repeat_interleave(torch.arange(5, dtype=torch.float32), torch.tensor(per)))
# fewer files than ranks: the rank without files still joins both collectives
scores, counts = score_files_sharded(1, lambda lo, hi: (torch.ones(2, C), [2]), C)
assert counts == [2] and scores.shape == (2, C)
# blocks of equal CHUNK count instead of equal file count (evaluate's dealing): 2 long files + 8 short ones
from birdnet_stm32.audio.pipeline import balanced_bounds
per = [20, 20, 1, 1, 1, 1, 1, 1, 1, 1]
bounds = balanced_bounds(per, world)
assert bounds == [0, 1, 10] if world == 2 else True   # 20 | 28 chunks (the nearer cut), not 40 | 8
seen_blocks = []
def score_files2(lo, hi):
    seen_blocks.append((lo, hi))
    rows = [torch.full((per[f], C), float(f)) for f in range(lo, hi)]
    return torch.cat(rows), per[lo:hi]
scores, counts = score_files_sharded(10, score_files2, C, bounds=bounds)
assert seen_blocks == [(bounds[rank], bounds[rank + 1])]
assert counts == per and torch.equal(scores[:, 0], torch.repeat_interleave(torch.arange(10, dtype=torch.float32), torch.tensor(per)))
try:
    score_files_sharded(10, score_files2, C, bounds=[0, 11, 10])
    raise SystemExit("expected ValueError")
except ValueError as e:
    assert "do not partition" in str(e)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(script, world, *argv):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), *map(str, argv)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    for rank in range(world):
        assert f"rank {rank} ok" in outs[rank]
    return outs


@pytest.mark.parametrize("world,n_items", [(2, 16), (2, 13), (4, 37), (8, 64), (8, 61)])
def test_sharding_and_all_gather(world, n_items, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % PKG)
    _run_world(script, world, n_items)


WORKER3 = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from birdnet_stm32.audio.pipeline import balanced_bounds
from birdnet_stm32.audio import _pcmio
from birdnet_stm32.evaluation.sharding import all_gather_ragged, score_files_sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
assert _pcmio.local_world_size() == world and _pcmio.default_threads() <= max(2, len(os.sched_getaffinity(0)) // world)
F, C = int(sys.argv[1]), 3
per = np.random.default_rng(11).integers(0, 13, F)           # chunks per file (some files have none), the same on every rank
per[:4] = [60, 1, 0, 45]                                      # long files at the head of the list
bounds = balanced_bounds(per, world)
assert bounds[0] == 0 and bounds[-1] == F and all(a <= b for a, b in zip(bounds, bounds[1:]))
loads = np.array([int(per[bounds[r]:bounds[r + 1]].sum()) for r in range(world)])
assert np.abs(loads - per.sum() / world).max() <= 60 + 1, loads     # nobody more than one file's chunks from the mean
blocks = []
def score_files(lo, hi):
    blocks.append((lo, hi))
    n = int(per[lo:hi].sum())
    owner = np.repeat(np.arange(lo, hi), per[lo:hi])
    rows = torch.from_numpy(np.stack([owner, owner * 2 + 1, np.full(n, rank)], axis=1).astype(np.float32)) if n else torch.zeros((0, C))
    return rows, per[lo:hi].tolist()
scores, counts = score_files_sharded(F, score_files, C, bounds=bounds)
assert blocks == [(bounds[rank], bounds[rank + 1])] and counts == per.tolist() and scores.shape == (int(per.sum()), C)
owner = np.repeat(np.arange(F), per)
assert np.array_equal(scores[:, 0].numpy(), owner.astype(np.float32)) and np.array_equal(scores[:, 1].numpy(), (owner * 2 + 1).astype(np.float32))
who = np.repeat(np.searchsorted(np.asarray(bounds[1:]), np.arange(F), side="right"), per)   # the rank whose block holds the file
assert np.array_equal(scores[:, 2].numpy(), who.astype(np.float32))                         # every row came from the rank that owns it
# ragged all-gather with an empty rank in the middle
local = torch.full((0 if rank == 1 else 2 + rank, C), float(rank))
allrows, cnt = all_gather_ragged(local)
assert cnt == [0 if r == 1 else 2 + r for r in range(world)] and allrows.shape[0] == sum(cnt)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,files", [(4, 1000), (8, 4096)])
def test_file_sharding_by_chunk_count_on_four_and_eight_ranks(world, files, tmp_path):
    """``evaluate``'s dealing on 4 and 8 ranks of one host: blocks of equal CHUNK count (disjoint, covering, nobody more than one file's
    chunks from the mean), every rank scores exactly its block, the ragged all-gather returns the rows in file order on every rank, and the
    reader-thread share follows LOCAL_WORLD_SIZE."""
    script = tmp_path / "worker3.py"
    script.write_text(WORKER3 % PKG)
    _run_world(script, world, files)


def test_shard_bounds_cover_the_range():
    from birdnet_stm32.evaluation.sharding import shard_bounds

    for n in (0, 1, 7, 8, 262144, 262145):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(262144, 3, 8) == (98304, 131072)  # BASELINE configs[3]: 32768 chunks per rank
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_bench_job_ragged_gather_and_file_sharding(tmp_path):
    """bench.py's timed region (run_sharded + one all-gather) with a fake scorer, the all-ranks refusal of an unshardable job,
    the ragged all-gather and evaluate's file-level sharding — world size 2 on gloo."""
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2 % (PKG, REPO))
    _run_world(script, 2)


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset starts torch.distributed.run as a child (never an exec of a process
    that touched the GPU) and refuses when fewer GPUs are visible."""
    import argparse

    import bench

    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    import torch

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    assert bench.self_launch(argparse.Namespace(gpus=4), ["--gpus", "4", "--steps", "8"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "8"]
    assert cmd[cmd.index("--master-port") + 2] == os.path.join(REPO, "bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.self_launch(argparse.Namespace(gpus=8), ["--gpus", "8"]) == 2
