"""Soak of the INT8 exactness pass: the quantised input bytes of the guarded fast path (stft_exact = 2: float32 STFT + float64 pass over the
elements in doubt) against the all-float64 STFT (stft_exact = 1, itself checked against the oracle by tests/test_gpu_sweeps.py) on many batches of
random signals from a dozen families with random parameters — generated on the device, so the CPU oracle's speed does not limit the count.

    python tools/exact_soak.py [batches] [chunks per batch] [seed]      # prints one summary line; exit code 1 on any differing byte or score
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
from conftest import TFLITE_PATH  # noqa: E402

from birdnet_stm32 import _hip  # noqa: E402
from birdnet_stm32.models.runners import load_model_runner  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T, sr = 72000, 24000
dev = torch.device("cuda")
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1234
g = torch.Generator(device=dev).manual_seed(seed)
t = torch.arange(T, device=dev, dtype=torch.float64) / sr


def rnd(*shape, lo=0.0, hi=1.0):
    return lo + (hi - lo) * torch.rand(shape, generator=g, device=dev, dtype=torch.float64)


def batch(kind: int):
    f = rnd(B, 1, lo=60.0, hi=11500.0)
    tone = torch.sin(2 * np.pi * f * t[None, :] + rnd(B, 1, hi=6.28))
    noise = torch.randn((B, T), generator=g, device=dev, dtype=torch.float64)
    if kind == 0:
        x = rnd(B, 1, hi=1.0) * noise + tone
    elif kind == 1:
        x = tone + rnd(B, 1) * torch.sin(2 * np.pi * rnd(B, 1, lo=60.0, hi=11500.0) * t[None, :])
    elif kind == 2:
        x = noise
    elif kind == 3:
        x = torch.sin(2 * np.pi * (f * t[None, :] + rnd(B, 1, lo=-1500.0, hi=1500.0) * t[None, :] ** 2))
    elif kind == 4:
        x = (1 + 0.9 * torch.sin(2 * np.pi * rnd(B, 1, lo=1.0, hi=40.0) * t[None, :])) * tone + 0.01 * noise
    elif kind == 5:
        x = torch.clamp(3 * (0.3 * noise + tone), -1, 1)
    elif kind == 6:
        x = sum(torch.sin(2 * np.pi * (rnd(B, 1, lo=80.0, hi=400.0)) * h * t[None, :]) / h for h in range(1, 12)) + 0.02 * noise
    elif kind == 7:
        x = torch.where(t[None, :] > rnd(B, 1, hi=2.5), 0.2 * noise + tone, torch.zeros_like(tone))  # onset behind digital silence
    elif kind == 8:
        x = (torch.rand((B, T), generator=g, device=dev) < 2e-3).double() * noise + 1e-3 * noise
    elif kind == 9:
        x = 1e-4 * noise + tone  # almost noise-free
    elif kind == 10:
        x = torch.cumsum(noise, dim=1) / 50.0  # brown noise: strong low frequencies
    else:
        x = 0.05 * noise + torch.sign(tone)
    x = x * 10.0 ** rnd(B, 1, lo=-4.0, hi=0.0)  # any level: the normalisation is scale-free
    return x.to(torch.float32).contiguous()


runner = load_model_runner(TFLITE_PATH, max_batch=B)
bad_bytes = bad_scores = total = listed = whole = 0
for i in range(n_batches):
    x = batch(i % 12)
    with _hip.options(stft_exact=1):
        s1 = runner.infer_audio_device(x).clone()
        q1 = torch.from_numpy(runner.input_bytes(B))
    s2 = runner.infer_audio_device(x)
    q2 = torch.from_numpy(runner.input_bytes(B))
    st = runner.guard_stats(B)
    d = int((q1 != q2).sum())
    bad_bytes += d
    bad_scores += int((s1 != s2).any(dim=1).sum())
    total += B
    listed += st["listed"]
    whole += st["whole_minmax"] + st["whole_fix"]
    if d:
        print(f"batch {i} (family {i % 12}): {d} bytes differ", flush=True)
print(f"exactness soak (seed {seed}): {total} chunks of 12 signal families, {bad_bytes} differing input bytes, {bad_scores} chunks with differing scores; "
      f"{listed / (total * 257 * 256):.2e} of the elements re-evaluated in float64, {whole} chunks as whole float64 spectrograms")
sys.exit(1 if bad_bytes or bad_scores else 0)
