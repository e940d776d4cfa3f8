"""Child process of tests/test_gpu_rccl.py: the sharding module's collectives over RCCL (backend ``nccl``) on device tensors, at the
world size torchrun started (1 on the test box: ``COLLECTIVE_AT_WORLD_1`` makes the calls run anyway), against the unsharded results."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
from conftest import TFLITE_PATH, synth_chunks  # noqa: E402

from birdnet_stm32.evaluation import sharding  # noqa: E402
from birdnet_stm32.models.runners import load_model_runner  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
sharding.COLLECTIVE_AT_WORLD_1 = True
runner = load_model_runner(TFLITE_PATH, device=local, max_batch=64)
n = 150 * world + 3  # ragged over the ranks, several batches per rank
audio = torch.from_numpy(synth_chunks(n, seed=9)).to(dev)
direct = torch.cat([runner.infer_audio_device(audio[i : i + 64]).clone() for i in range(0, n, 64)])

calls = []


def score(lo, hi):
    calls.append((lo, hi))
    return runner.infer_audio_device(audio[lo:hi]).clone()


got = sharding.run_sharded(score, n, 64)
assert got.is_cuda and got.shape == (n, 100) and torch.equal(got, direct), "run_sharded over RCCL differs from the unsharded scores"
lo, hi = sharding.shard_bounds(n, rank, world)
assert calls[0][0] == lo and calls[-1][1] == hi
# equal shards into a preallocated buffer (bench.py's form)
m = 128 * world
buf = torch.empty((128, 100), dtype=torch.float32, device=dev)
got2 = sharding.run_sharded(lambda a, b, out: out.copy_(runner.infer_audio_device(audio[a:b])), m, 64, into=buf)
assert torch.equal(got2, direct[:m])
# ragged gather + file-level sharding
rag, counts = sharding.all_gather_ragged(direct[lo:hi])
assert counts == [sharding.shard_bounds(n, r, world)[1] - sharding.shard_bounds(n, r, world)[0] for r in range(world)] and torch.equal(rag, direct)
per_file = [1 + (f % 3) for f in range(40)]
starts = np.concatenate([[0], np.cumsum(per_file)])


def score_files(a, b):
    return direct[starts[a] : starts[b]].clone(), per_file[a:b]


sc, cnt = sharding.score_files_sharded(40, score_files, 100, device=dev)
assert cnt == per_file and torch.equal(sc, direct[: starts[-1]])
from birdnet_stm32.audio.pipeline import balanced_bounds  # noqa: E402

sc, cnt = sharding.score_files_sharded(40, score_files, 100, device=dev, bounds=balanced_bounds(per_file, world))
assert cnt == per_file and torch.equal(sc, direct[: starts[-1]])
# evaluate() itself through its sharded branch (chunk-balanced file blocks, pipeline per rank, ragged all-gather over RCCL, pooling of all files on every
# rank) against the same call without a process group in the picture
import tempfile  # noqa: E402

from birdnet_stm32.audio.io import save_wav  # noqa: E402
from birdnet_stm32.evaluation.metrics import evaluate  # noqa: E402
from birdnet_stm32.training.config import ModelConfig  # noqa: E402
from conftest import CONFIG_PATH  # noqa: E402

cfg = ModelConfig.load(CONFIG_PATH).to_dict()
cfg.update(sample_rate=24000, hop_length=281)
classes = cfg["class_names"]
root = tempfile.mkdtemp() if rank == 0 else None
box = [root]
dist.broadcast_object_list(box, src=0)
root = box[0]
wav = synth_chunks(12, seed=3)
files = []
for i in range(12):
    d = os.path.join(root, classes[i % 3])
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, f"f{i}.wav")
    if rank == 0:
        save_wav(np.concatenate([wav[i]] * (1 + i % 4))[: 72000 * (1 + i % 4) - 5000 * (i % 2)], p, 24000)
    files.append(p)
dist.barrier()
sharding.COLLECTIVE_AT_WORLD_1 = False
_, pf_a, _, ys_a = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=8) if world == 1 else (None, None, None, None)
sharding.COLLECTIVE_AT_WORLD_1 = True
_, pf_b, _, ys_b = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=8)
assert len(pf_b) == 12
if world == 1:
    assert [p["file"] for p in pf_a] == [p["file"] for p in pf_b] and np.array_equal(ys_a, ys_b), "sharded evaluate differs from the single-process one"
torch.cuda.synchronize()
if rank == 0:
    ver = torch.cuda.nccl.version()
    print(json.dumps({"ok": True, "backend": dist.get_backend(), "world": dist.get_world_size(), "nccl_version": list(ver) if isinstance(ver, tuple) else ver,
                      "items": n}))
runner.close()
dist.destroy_process_group()
