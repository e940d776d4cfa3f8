"""Slaney-scale triangular mel filterbank (area-normalised), built without librosa.

The reference seeds the hybrid frontend's frozen 1x1 mel mixer from
``librosa.filters.mel(sr, n_fft, n_mels, fmin=150, fmax=sr//2, htk=False, norm='slaney')``
transposed and zero-padded on the bin axis to a multiple of 8
(reference: birdnet_stm32/models/frontend.py:257-276).  This is the same construction written
as one broadcast: piecewise mel scale (linear below 1 kHz, log above), triangle = min of the two
edge ramps, each filter scaled by 2 / (its Hz width).
"""

from __future__ import annotations

import numpy as np

_LIN_HZ_PER_MEL = 200.0 / 3.0
_BREAK_HZ = 1000.0
_BREAK_MEL = _BREAK_HZ / _LIN_HZ_PER_MEL
_LOG_STEP = np.log(6.4) / 27.0


def _to_mel(hz: np.ndarray) -> np.ndarray:
    hz = np.asarray(hz, np.float64)
    upper = _BREAK_MEL + np.log(np.maximum(hz, _BREAK_HZ) / _BREAK_HZ) / _LOG_STEP
    return np.where(hz < _BREAK_HZ, hz / _LIN_HZ_PER_MEL, upper)


def _to_hz(mel: np.ndarray) -> np.ndarray:
    mel = np.asarray(mel, np.float64)
    upper = _BREAK_HZ * np.exp(_LOG_STEP * (np.maximum(mel, _BREAK_MEL) - _BREAK_MEL))
    return np.where(mel < _BREAK_MEL, mel * _LIN_HZ_PER_MEL, upper)


def mel_filterbank(sample_rate: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """``[n_mels, n_fft//2 + 1]`` float32 filterbank."""
    bins = np.linspace(0.0, sample_rate / 2.0, n_fft // 2 + 1)[None, :]
    knots = _to_hz(np.linspace(_to_mel(fmin), _to_mel(fmax), n_mels + 2))
    left, centre, right = knots[:-2, None], knots[1:-1, None], knots[2:, None]
    up = (bins - left) / (centre - left)
    down = (right - bins) / (right - centre)
    tri = np.clip(np.minimum(up, down), 0.0, None)
    return (tri * (2.0 / (right - left))).astype(np.float32)


def hybrid_mel_mixer(sample_rate: int, n_fft: int, n_mels: int, fmin: float = 150.0, fmax: float | None = None) -> np.ndarray:
    """``[F_padded, n_mels]`` mixer kernel of the hybrid frontend (bins padded to a multiple of 8)."""
    top = float(fmax) if fmax is not None else float(sample_rate // 2)
    mixer = mel_filterbank(int(sample_rate), int(n_fft), int(n_mels), float(fmin), top).T
    pad = -mixer.shape[0] % 8
    return np.pad(mixer, ((0, pad), (0, 0))).astype(np.float32)
