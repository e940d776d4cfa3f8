#!/usr/bin/env python3
"""Throughput of the device ingest and pooling kernels with inputs resident in HBM.

    python tools/ingest_bench.py [--sr_in 48000] [--channels 2] [--windows 128] [--seconds 30] [--steps 20]

One step = ``bn_ingest_resample`` + ``bn_ingest_chunks`` over all windows (PCM16 already uploaded), timed
with HIP events on the launch stream.  Prints one JSON line per kernel: chunks/s and the achieved fraction of
the HBM roofline on ALGORITHMIC bytes (PCM bytes in + mono float32 out; mono in + chunk matrix out;
score matrix in + pooled rows out).
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
from math import gcd

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "birdnet-stm32_amd"))

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sr_in", type=int, default=48000)
    ap.add_argument("--sr_out", type=int, default=24000)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--windows", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spec_batch", type=int, default=1024)
    args = ap.parse_args()

    import torch

    from birdnet_stm32 import _hip
    from birdnet_stm32.audio import ingest

    ctx = _hip.Context(0, 1)
    lib = ctx.lib
    dev = torch.device("cuda", 0)
    n_in = int(args.sr_in * args.seconds)
    g = gcd(args.sr_in, args.sr_out)
    up, down = args.sr_out // g, args.sr_in // g
    if up == down:
        taps, per_phase, pre = None, 0, 0
        n_out = n_in
    else:
        taps, per_phase, pre = ingest.polyphase_filter(up, down)
        n_out = ingest.resampled_length(n_in, up, down)
    W = args.windows
    gen = torch.Generator(device=dev).manual_seed(42)
    pcm = (torch.randn((W * n_in, args.channels), device=dev, generator=gen) * 6000.0).clamp_(-32768, 32767).to(torch.int16)
    in_off = torch.arange(W + 1, dtype=torch.int64, device=dev) * n_in
    out_off = torch.arange(W + 1, dtype=torch.int64, device=dev) * n_out
    d_taps = torch.from_numpy(taps).to(dev) if taps is not None else None
    mono = torch.empty(W * n_out, dtype=torch.float32, device=dev)
    peak = torch.empty(W, dtype=torch.float32, device=dev)
    starts, valid, owner, counts, T = ingest.chunk_table([n_out] * W, args.sr_out, 3.0, 0.0)
    N = starts.shape[0]
    d_src = torch.from_numpy(starts + owner.astype(np.int64) * n_out).to(dev)
    d_valid = torch.from_numpy(valid).to(dev)
    d_owner = torch.from_numpy(owner).to(dev)
    chunks = torch.empty((N, T), dtype=torch.float32, device=dev)
    scores = torch.rand((N, 100), dtype=torch.float32, device=dev, generator=gen)
    file_off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(dev)
    pooled = torch.empty((W, 100), dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def resample():
        _hip.check(lib.bn_ingest_resample(ctx.handle, pcm.data_ptr(), ingest.PCM_S16, args.channels, in_off.data_ptr(),
                                          out_off.data_ptr(), W, n_in, n_out, d_taps.data_ptr() if d_taps is not None else None,
                                          up, down, per_phase, pre, mono.data_ptr(), peak.data_ptr(), stream))

    def gather():
        _hip.check(lib.bn_ingest_chunks(ctx.handle, mono.data_ptr(), peak.data_ptr(), d_src.data_ptr(), d_valid.data_ptr(),
                                        d_owner.data_ptr(), N, T, chunks.data_ptr(), stream))

    def pool():
        _hip.check(lib.bn_pool_scores(ctx.handle, scores.data_ptr(), file_off.data_ptr(), W, 100, 2, 10.0, pooled.data_ptr(), stream))

    legs = [
        ("ingest_resample_kernel", resample, W * n_in * args.channels * 2 + W * n_out * 4),
        ("ingest_chunks_kernel", gather, N * T * 4 * 2),
        ("pool_scores_kernel", pool, N * 100 * 4 + W * 100 * 4),
    ]
    for _ in range(args.warmup):
        for _, fn, _ in legs:
            fn()
    torch.cuda.synchronize()
    total_ms = 0.0
    for name, fn, nbytes in legs:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        total_ms += ms if name != "pool_scores_kernel" else 0.0
        gbs = nbytes / ms / 1e6
        print(json.dumps({
            "kernel": name, "ms": round(ms, 4), "chunks": int(N), "chunks_per_s": round(N / ms * 1e3, 1),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "algorithmic_bytes_per_chunk": round(nbytes / N, 1),
            "config": {"workload": f"{W} windows x {args.seconds:g} s, {args.channels} ch PCM16 @ {args.sr_in} Hz -> {args.sr_out} Hz, 3 s chunks",
                       "up": up, "down": down, "taps_per_phase": per_phase},
        }))
    print(json.dumps({"metric": "ingested audio chunks/s (resample + chunk gather)", "value": round(N / total_ms * 1e3, 1), "ms_per_step": round(total_ms, 4)}))

    # precomputed-frontend spectrograms: [B, 72000] float32 -> [B, 64 | 20, 256]
    from birdnet_stm32.audio.spectrogram import mel_spectrograms_device

    B = args.spec_batch
    audio = chunks[:B].contiguous() if N >= B else torch.rand((B, T), dtype=torch.float32, device=dev, generator=gen) * 2 - 1
    for mode, mag in (("mel", "none"), ("mel", "pwl"), ("mel", "db"), ("mel", "pcen"), ("log_mel", "none"), ("mfcc", "none")):
        for _ in range(args.warmup):
            mel_spectrograms_device(ctx, audio, args.sr_out, 512, 64, 256, mag, mode, 20)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in ev:
            a.record()
            mel_spectrograms_device(ctx, audio, args.sr_out, 512, 64, 256, mag, mode, 20)
            b.record()
        torch.cuda.synchronize()
        ms = float(np.min([a.elapsed_time(b) for a, b in ev]))  # host-side allocation jitter is not the kernels' time
        rows = 20 if mode == "mfcc" else 64
        nbytes = B * (T * 4 + rows * 256 * 4)
        gbs = nbytes / ms / 1e6
        print(json.dumps({"kernel": f"stft512_mag_kernel<mel> + melspec_finish_kernel [{mode}/{mag}]", "ms": round(ms, 4), "chunks": B,
                          "chunks_per_s": round(B / ms * 1e3, 1),
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
                          "algorithmic_bytes_per_chunk": nbytes // B}))


if __name__ == "__main__":
    main()
