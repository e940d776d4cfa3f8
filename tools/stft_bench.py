#!/usr/bin/env python3
"""Runs only the batched STFT (bn_stft_mag) on synthetic chunks; used under rocprofv3 to study that kernel alone."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import torch
from birdnet_stm32 import _hip
from birdnet_stm32.models.runners import stft_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _hip.Context(0, B)
x = torch.randn(B, 72000, device="cuda").clamp_(-1, 1)
for _ in range(3):
    stft_device(ctx, x, normalize=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    stft_device(ctx, x, normalize=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"stft B={B}: {dt*1e3:.3f} ms/launch-group, {B*551168/dt/1e9:.1f} GB/s algorithmic")
