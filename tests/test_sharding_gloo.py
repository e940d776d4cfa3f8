"""The N > 1 path on CPU: world_size-2 gloo processes shard a chunk stream and meet in one all-gather."""

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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_items", [16, 13])
def test_two_rank_sharding_and_all_gather(n_items, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % PKG)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(n_items)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]


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
