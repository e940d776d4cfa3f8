"""The RCCL path on hardware.  The test box has ONE GPU, so the collectives run at world size 1 — the same calls, the
same code path (``torch.distributed`` backend ``nccl`` = RCCL, ``init_process_group(device_id=...)``, ``all_gather_into_tensor`` on device
tensors) as the driver's 8-GPU run.  Every case starts a FRESH child with ``python -m torch.distributed.run`` (no exec from this process)."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_and_args, timeout=600):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; the product has no CPU path to fall back to")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           *script_and_args]
    p = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(lines[-1])


def test_bench_collective_runs_rccl_inside_the_timed_region():
    """``bench.py --gpus 1 --collective``: RCCL initialised with ``device_id``, warm-up + timed ``all_gather_into_tensor`` of the score
    buffer inside every repeat, the JSON line says what RCCL saw; the rate stays that of the plain one-GPU line (the gather of
    1.6 MB is microseconds)."""
    out = _torchrun([os.path.join(REPO, "bench.py"), "--gpus", "1", "--collective", "--steps", "4", "--repeats", "5", "--warmup", "2", "--batch", "1024",
                     "--no-cpu-baseline"])
    c = out["collective"]
    assert c["backend"] == "nccl" and c["ranks_seen"] == 1 and c["in_timed_region"] is True
    assert c["bytes_per_rank"] == 4 * 1024 * 100 * 4 and c["all_gather_ms"] > 0 and c["nccl_version"]
    assert out["n_gpus"] == 1 and out["repeats"] == 5 and out["scores_finite"] and out["value"] > 1e5
    assert "all-gather" in out["config"]["path"]
    assert out["value_min"] <= out["value"] <= out["value_max"]


def test_sharded_scoring_over_rccl_equals_unsharded():
    """``run_sharded`` (even / ragged / into a buffer), ``all_gather_ragged`` and ``score_files_sharded`` with backend ``nccl`` on device
    tensors: identical to scoring everything in one process."""
    out = _torchrun([os.path.join(REPO, "tests", "_rccl_child.py")])
    assert out["ok"] and out["backend"] == "nccl" and out["world"] == 1
