"""|S' - S| of the float32 STFT kernel against the float64 oracle, in units of the bound the exactness pass assumes
(csrc/bn_quant_in.h: kStftGuard * ||frame||_2), for several signal families; and bn_stft_mag_exact == oracle bit for bit.

    python tools/stft_error_stats.py [chunks per family]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
from oracle import stft  # noqa: E402

from birdnet_stm32 import _hip  # noqa: E402
from birdnet_stm32.models.runners import stft_device  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, W, sr = 72000, 256, 24000
rng = np.random.default_rng(5)
t = np.arange(T) / sr
fam = {}
fam["tone+noise"] = np.stack([0.3 * rng.standard_normal(T) + np.sin(2 * np.pi * (500 + 37 * b) * t) for b in range(N)])
fam["pure tone"] = np.stack([np.sin(2 * np.pi * (440.0 + 13.7 * b) * t) for b in range(N)])
fam["noise"] = rng.standard_normal((N, T))
fam["dc+step"] = np.concatenate([np.ones((N // 2, T)), np.repeat((np.arange(T) > T // 3)[None, :], N - N // 2, 0)]).astype(np.float64)
fam["impulses"] = (rng.random((N, T)) < 1e-3).astype(np.float64)
fam["zero tail"] = np.stack([np.where(np.arange(T) < 5000 + 997 * b, rng.standard_normal(T), 0.0) for b in range(N)])
fam["square"] = np.sign(np.sin(2 * np.pi * 1000.0 * t))[None, :].repeat(N, 0) + 0.0
fam["quiet"] = 1e-4 * fam["tone+noise"]
ctx = _hip.Context(0, N)
guard = 2.0**-15 * 1.01
for name, x in fam.items():
    x = (x / max(np.abs(x).max(), 1e-30) if name != "quiet" else x).astype(np.float32)
    d = torch.from_numpy(x).cuda()
    S_ref = np.stack([stft.stft_magnitude(a, 512, T // W)[:, :W] for a in x])
    S32 = stft_device(ctx, d, normalize=False).cpu().numpy()
    S64 = stft_device(ctx, d, normalize=False, exact=True).cpu().numpy()
    N64 = stft_device(ctx, d, normalize=True, exact=True).cpu().numpy()
    N_ref = np.stack([stft.minmax_normalize(s) for s in S_ref])
    # per-frame bound
    xp = np.pad(x, ((0, 0), (256, 256)))
    idx = np.arange(512)[None, :] + (T // W) * np.arange(W)[:, None]
    l2 = np.sqrt((xp[:, idx].astype(np.float64) ** 2).sum(-1))  # [N, W]
    bound = guard * l2[:, None, :]
    err = np.abs(S32.astype(np.float64) - S_ref)
    ratio = np.where(bound > 0, err / np.maximum(bound, 1e-300), np.where(err > 0, np.inf, 0.0))
    print(f"{name:12s} max |S'-S| / bound = {ratio.max():.4f}   rms ratio {np.sqrt((ratio**2).mean()):.5f}   exact kernel: "
          f"{int((S64 != S_ref).sum())} of {S_ref.size} magnitudes differ, {int((N64 != N_ref).sum())} normalised values differ", flush=True)
