#!/usr/bin/env python3
"""Random precomputed-frontend spectrogram cases (sample rate, chunk length, mel bins, width, mode, mag_scale, n_mfcc) and random pooling
layouts on the GPU against oracle/melspec.py / numpy.  A one-off fuzzing aid (noisy signals: PCEN and dB amplify round-off on pure tones).

    python tools/fuzz/melspec_fuzz.py [n_cases] [seed]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
import numpy as np
import torch
from birdnet_stm32 import _hip
from birdnet_stm32.audio.spectrogram import mel_spectrograms_device
from oracle import melspec

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = _hip.Context(0, 64)
bad = 0
for i in range(n):
    sr = int(rng.choice([16000, 22050, 24000, 32000, 44100, 48000]))
    secs = float(rng.choice([1.0, 2.0, 3.0, 5.0]))
    mels = int(rng.choice([16, 32, 40, 64, 96, 128]))
    W = int(rng.choice([64, 128, 192, 256, 384]))
    mode = str(rng.choice(["mel", "mel", "log_mel", "mfcc"]))
    mag = str(rng.choice(["none", "pwl", "db", "pcen"])) if mode == "mel" else "none"
    n_mfcc = int(rng.integers(8, min(mels, 40) + 1))
    T = int(sr * secs)
    t = np.arange(T) / sr
    x = np.stack([(0.4 * np.sin(2 * np.pi * rng.uniform(200, 0.4 * sr) * t) + 0.25 * rng.standard_normal(T)) for _ in range(3)]).astype(np.float32)
    x /= np.abs(x).max(axis=1, keepdims=True)
    try:
        got = mel_spectrograms_device(ctx, torch.from_numpy(x).cuda(), sr, 512, mels, W, mag, mode, n_mfcc).cpu().numpy()
        bar = {"none": 5e-5, "pwl": 5e-5, "db": 5e-4, "pcen": 5e-3}[mag] if mode == "mel" else (5e-5 if mode == "log_mel" else 1e-3)
        err = 0.0
        for b in range(3):
            want, _ = melspec.get_spectrogram(x[b], sr, 512, mels, W, mag, mode, n_mfcc)
            if got[b].shape != want.shape:
                err = float("inf")
                break
            err = max(err, float(np.abs(got[b] - want).max()))
        ok = err <= bar
    except Exception as e:  # noqa: BLE001
        print(i, "EXCEPTION", type(e).__name__, str(e)[:160], dict(sr=sr, secs=secs, mels=mels, W=W, mode=mode, mag=mag, n_mfcc=n_mfcc))
        bad += 1
        continue
    print(i, "ok" if ok else "MISMATCH", f"{err:.2e}", dict(sr=sr, secs=secs, mels=mels, W=W, mode=mode, mag=mag, n_mfcc=n_mfcc), flush=True)
    bad += not ok
# pooling: random segment layouts (empty files included)
from birdnet_stm32.audio.ingest import pool_scores_device
from birdnet_stm32.evaluation.pooling import pool_scores
for i in range(n):
    F, C = int(rng.integers(1, 40)), int(rng.integers(1, 130))
    counts = [int(v) for v in rng.integers(0, 12, F)]
    if sum(counts) == 0:
        counts[0] = 1
    scores = (rng.random((sum(counts), C)).astype(np.float32)) ** 3
    method = str(rng.choice(["avg", "max", "lme"]))
    beta = float(rng.choice([1.0, 5.0, 10.0, 20.0]))
    dev = pool_scores_device(ctx, torch.from_numpy(scores).cuda(), counts, method, beta=beta).cpu().numpy()
    at, ok = 0, True
    for f in range(F):
        want = np.asarray(pool_scores(scores[at : at + counts[f]], method, beta=beta), np.float32)
        at += counts[f]
        ok = ok and (np.abs(dev[f] - want).max() <= 2e-6 if method == "lme" else np.array_equal(dev[f].view(np.uint32), want.view(np.uint32)))
    if not ok:
        print("pool", i, "MISMATCH", F, C, method, beta, counts)
        bad += 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
