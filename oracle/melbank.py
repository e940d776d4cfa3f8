"""Slaney mel filterbank, restating ``librosa.filters.mel(htk=False, norm='slaney')``.

ORACLE — test infrastructure only (see oracle/__init__.py).

Reference call sites: birdnet_stm32/models/frontend.py:257-276 (mel_mixer seeding,
fmin=150, fmax=sr//2, transposed and zero-padded 257->264 rows) and
birdnet_stm32/audio/spectrogram.py:64-77.  librosa 0.11.0 (requirements.txt:1) is not
vendored; the algorithm below is the published Slaney/Auditory-Toolbox construction.
The firmware restates the same formulas in C (reference: firmware/Src/audio_mel.c:24-96).
"""

from __future__ import annotations

import numpy as np

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    """Slaney mel scale: linear below 1 kHz, logarithmic above."""
    f = np.asarray(f, dtype=np.float64)
    lin = f / _F_SP
    log = _MIN_LOG_MEL + np.log(np.maximum(f, 1e-30) / _MIN_LOG_HZ) / _LOGSTEP
    return np.where(f >= _MIN_LOG_HZ, log, lin)


def mel_to_hz(m):
    """Inverse of :func:`hz_to_mel`."""
    m = np.asarray(m, dtype=np.float64)
    lin = _F_SP * m
    log = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL))
    return np.where(m >= _MIN_LOG_MEL, log, lin)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """Return the ``[n_mels, 1 + n_fft//2]`` float32 triangular filterbank with Slaney area norm."""
    n_bins = 1 + n_fft // 2
    fft_freqs = np.linspace(0.0, float(sr) / 2.0, n_bins)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    widths = np.diff(edges)
    ramps = edges[:, None] - fft_freqs[None, :]
    bank = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        rising = -ramps[i] / widths[i]
        falling = ramps[i + 2] / widths[i + 1]
        bank[i] = np.maximum(0.0, np.minimum(rising, falling))
    bank *= (2.0 / (edges[2 : n_mels + 2] - edges[:n_mels]))[:, None]
    return bank.astype(np.float32)


def hybrid_mel_mixer(sr: int, n_fft: int, n_mels: int, fmin: float = 150.0, fmax: float | None = None) -> np.ndarray:
    """Mel-mixer kernel ``[F_padded, n_mels]`` as the reference seeds it (frontend.py:257-276)."""
    upper = float(fmax) if fmax is not None else float(sr // 2)
    w = mel_filterbank(sr, n_fft, n_mels, fmin, upper).T.astype(np.float32)
    cin = w.shape[0]
    pad = (8 - cin % 8) % 8
    if pad:
        w = np.pad(w, ((0, pad), (0, 0)))
    return w
