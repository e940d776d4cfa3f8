"""TEST INFRASTRUCTURE — CPU restatement of the audio-ingest step in front of the hot path.

Reference: birdnet_stm32/audio/io.py
  * :14-30   fast_resample      -> scipy.signal.resample_poly(y, sr_out/g, sr_in/g) in float32
  * :63-130  load_audio_window  -> frames.mean(axis=1), resample, y / max(abs(y))
  * :133-174 split_audio_into_chunks -> chunk start positions, single right zero pad for short files

The arithmetic lives in a third-party dependency that IS installed here and on the GPU box
(scipy 1.15.3 ``resample_poly`` -> ``firwin`` Kaiser(5.0) low-pass, ``upfirdn`` polyphase loop in the
input dtype).  This file restates that loop with explicit float32 rounding after every multiply and
every add, oldest input sample first — the order of scipy's ``_apply_impl`` — so the GPU kernel can be
held to bit equality.  Pinning: tests/test_oracle_pinning.py compares every function below with scipy /
numpy themselves (bit-exact) on the reference's fixture signals and on random multi-channel PCM.

Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

from math import gcd

import numpy as np


def design_filter(up: int, down: int):
    """(h_padded float32, n_pre_remove): scipy.signal.resample_poly's default filter and centring pads.

    ``h = firwin(2*half_len+1, 1/max(up,down), window=('kaiser', 5.0)).astype(float32) * up`` with
    ``half_len = 10*max(up,down)``; ``n_pre_pad = down - half_len % down`` zeros in front so that output
    sample 0 sits on input sample 0; ``n_pre_remove = (half_len + n_pre_pad) // down`` leading outputs dropped.
    """
    from scipy.signal import firwin

    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float32)
    h *= up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    return np.concatenate([np.zeros(n_pre_pad, np.float32), h]), n_pre_remove


def resampled_length(n_in: int, up: int, down: int) -> int:
    n = n_in * up
    return n // down + (1 if n % down else 0)


def rates_to_ratio(sr_in: int, sr_out: int):
    g = gcd(sr_in, sr_out)
    return sr_out // g, sr_in // g


def mono_mean(frames: np.ndarray) -> np.ndarray:
    """``frames.mean(axis=1)`` of a C-contiguous float32 ``[n, ch]`` array, in numpy's summation order.

    Up to 7 channels numpy adds left to right; from 8 channels on it runs its unrolled pairwise sum: eight
    running sums ``r[j] += a[8 i + j]``, combined as ``((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))``, then the tail
    left to right (valid below the 128-element block size).  The sum is divided by ch in float32.
    """
    f = np.asarray(frames, np.float32)
    n, ch = f.shape
    if ch < 8:
        acc = f[:, 0].copy()
        for c in range(1, ch):
            acc = (acc + f[:, c]).astype(np.float32)
    else:
        if ch >= 128:
            raise ValueError("more than 127 channels")
        r = [f[:, j].copy() for j in range(8)]
        i = 8
        while i + 8 <= ch:
            for j in range(8):
                r[j] = (r[j] + f[:, i + j]).astype(np.float32)
            i += 8
        acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for c in range(i, ch):
            acc = (acc + f[:, c]).astype(np.float32)
    return (acc / np.float32(ch)).astype(np.float32)


def resample_poly_f32(x: np.ndarray, up: int, down: int) -> np.ndarray:
    """scipy.signal.resample_poly(x float32, up, down) restated: float32 multiply, float32 add, oldest sample first."""
    x = np.asarray(x, np.float32)
    g = gcd(up, down)
    up, down = up // g, down // g
    if up == down == 1:
        return x.copy()
    h, n_pre_remove = design_filter(up, down)
    n_in = x.shape[0]
    n_out = resampled_length(n_in, up, down)
    n = np.arange(n_out, dtype=np.int64)
    t = (n + n_pre_remove) * down
    phase = t % up
    kmax = t // up
    taps_per_phase = -(-h.shape[0] // up)
    acc = np.zeros(n_out, np.float32)
    for i in range(taps_per_phase - 1, -1, -1):
        tap = phase + i * up
        k = kmax - i
        ok = (tap < h.shape[0]) & (k >= 0) & (k < n_in)
        prod = (x[np.clip(k, 0, n_in - 1)] * h[np.minimum(tap, h.shape[0] - 1)]).astype(np.float32)
        acc = np.where(ok, (acc + prod).astype(np.float32), acc)
    return acc


def ingest_window(frames: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """Decoded frames ``[n, ch]`` float32 -> mono, resampled, peak-normalised window (load_audio_window's tail)."""
    y = mono_mean(frames)
    if sr_in != sr_out:
        up, down = rates_to_ratio(sr_in, sr_out)
        y = resample_poly_f32(y, up, down)
    peak = float(np.abs(y).max()) if y.size else 0.0
    if peak > 0.0:
        y = (y / np.float32(peak)).astype(np.float32)
    return y


def chunk_starts(n: int, sample_rate: int, chunk_duration: float, chunk_overlap: float):
    """(starts, chunk_size) of split_audio_into_chunks; a window not longer than one chunk has the single start 0."""
    size = int(sample_rate * chunk_duration)
    if n <= 0 or size <= 0:
        return [], size
    if n <= size:
        return [0], size
    overlap = max(0.0, min(chunk_overlap, chunk_duration - 0.1))
    step = max(1, int(sample_rate * (chunk_duration - overlap)))
    starts = list(range(0, n - size + 1, step))
    if not starts or starts[-1] + size < n:
        starts.append(n - size)
    return starts, size


def split_chunks(y: np.ndarray, sample_rate: int, chunk_duration: float, chunk_overlap: float) -> np.ndarray:
    starts, size = chunk_starts(y.shape[0], sample_rate, chunk_duration, chunk_overlap)
    out = np.zeros((len(starts), size), np.float32)
    for i, s in enumerate(starts):
        seg = y[s : s + size]
        out[i, : seg.shape[0]] = seg
    return out
