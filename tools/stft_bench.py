#!/usr/bin/env python3
"""Runs only the batched STFT on synthetic chunks — the spectrogram-writing variant (bn_stft_mag) and the mel-writing one
(bn_mel_spectrogram); used under rocprofv3 --kernel-trace --stats to study stft512_mag_kernel<false|true> at one batch size.

    python tools/stft_bench.py [B] [iters]
"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import torch
from birdnet_stm32 import _hip
from birdnet_stm32.audio.spectrogram import mel_spectrograms_device
from birdnet_stm32.models.runners import stft_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _hip.Context(0, B)
x = torch.randn(B, 72000, device="cuda").clamp_(-1, 1)


def timed(fn, label, bytes_per_chunk):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{label} B={B}: {dt*1e3:.3f} ms/call (host clock, all kernels of the call), {B*bytes_per_chunk/dt/1e9:.1f} GB/s algorithmic")


timed(lambda: stft_device(ctx, x, normalize=False), "stft -> spectrogram", 551168)
timed(lambda: mel_spectrograms_device(ctx, x), "stft -> mel (+ finishing pass)", 353536 + 2 * 65536)
