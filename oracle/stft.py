"""Linear-magnitude STFT of the evaluate path, restated in numpy.

ORACLE — test infrastructure only (see oracle/__init__.py).  **Parity unpinned** for the
framing: librosa cannot be imported here, so the framing below is librosa 0.11's
documented ``stft`` default behaviour, anchored on the reference's call site.

Reference: birdnet_stm32/audio/spectrogram.py:12-21 (``normalize``), :61 (hop =
len(audio) // spec_width), :106-115 (``np.abs(librosa.stft(y, n_fft, hop_length,
win_length=n_fft, window='hann'))``), :133 (keep the first ``spec_width`` columns), :149
(min-max normalise); caller birdnet_stm32/evaluation/metrics.py:55-61.

librosa.stft defaults restated: ``center=True`` with ``pad_mode='constant'`` (n_fft//2
zeros each side), periodic Hann (``scipy.signal.get_window('hann', n, fftbins=True)`` =
``0.5 - 0.5 cos(2 pi k / n)``), frame *t* covers padded samples ``[t*hop, t*hop + n_fft)``,
``1 + len(y)//hop`` frames, window*frame and the real FFT evaluated in float64, the
result stored as complex64, and ``np.abs`` of complex64 returning float32.

``np.abs`` of a complex64 array is not the correctly rounded magnitude: numpy >= 1.25 on x86 with
FMA3 evaluates ``larger * sqrt(fma(ratio, ratio, 1))`` with ``ratio = smaller / larger`` of
(|re|, |im|), every step rounded to float32 (``simd_cabsf`` in numpy's
``loops_unary_complex.dispatch.c.src``).  ``cabs_numpy_simd`` restates that formula; it is **pinned**
to the installed numpy (tests/test_oracle_pinning.py: identical on 5e6 random complex64 values) and
is what the GPU's float64 pass reproduces (csrc/bn_quant_in.h: ``numpy_cabsf``).  ``stft_magnitude``
keeps calling ``np.abs`` itself — on a CPU without FMA3 numpy's result (and the reference's) would
differ from the restatement in the last bit of ~1e-2 of the values.
"""

from __future__ import annotations

import numpy as np


def hann_periodic(n: int) -> np.ndarray:
    """Periodic Hann window in float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def cabs_numpy_simd(z: np.ndarray) -> np.ndarray:
    """``np.abs`` of a complex64 array as numpy's SIMD loop evaluates it (x86 + FMA3), restated step by step in float32."""
    z = np.asarray(z, dtype=np.complex64)
    re, im = np.abs(z.real), np.abs(z.imag)
    larger, smaller = np.maximum(re, im), np.minimum(im, re)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(larger == 0, np.float32(0), smaller / larger).astype(np.float32)
    # fma(ratio, ratio, 1) rounded once to float32: ratio^2 is exact in float64 (48 bits); the sum may round in float64 first, which
    # could only matter on an exact tie of the second rounding (never seen: the pin test compares with numpy itself)
    t = (ratio.astype(np.float64) ** 2 + 1.0).astype(np.float32)
    return (np.sqrt(t) * larger).astype(np.float32)


def hann_symmetric(n: int) -> np.ndarray:
    """Symmetric Hann window ``0.5 (1 - cos(2 pi k / (n - 1)))`` in float64 (the reference FIRMWARE's window, firmware/Src/audio_stft.c)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * k / (n - 1)))


def stft_magnitude(y: np.ndarray, n_fft: int, hop: int, center: bool = True, window: str = "hann_periodic", n_frames: int | None = None) -> np.ndarray:
    """``abs(librosa.stft(y, n_fft, hop, win_length=n_fft, window='hann'))`` -> float32 ``[1+n_fft//2, 1+len(y)//hop]``.

    The defaults are librosa's (and the evaluate path's).  ``center=False`` / ``window='hann_symmetric'`` / ``n_frames`` select the framing
    of the reference's firmware STFT (frame t = samples ``[t*hop, t*hop + n_fft)`` of the signal, zero-extended at the end, symmetric Hann):
    the SAME function evaluated with that framing is pinned against the reference's own ``audio_stft.c`` compiled in place
    (tests/test_oracle_pinning.py), so everything in it but the two framing switches is reference-checked code."""
    y = np.asarray(y, dtype=np.float32)
    if hop <= 0:
        raise ValueError("hop must be positive")
    win = {"hann_periodic": hann_periodic, "hann_symmetric": hann_symmetric}[window](n_fft)
    if center:
        pad = n_fft // 2
        yp = np.concatenate([np.zeros(pad, np.float32), y, np.zeros(pad, np.float32)])
        if n_frames is None:
            n_frames = 1 + len(y) // hop
    else:
        if n_frames is None:
            n_frames = 1 + max(len(y) - n_fft, 0) // hop
        yp = np.concatenate([y, np.zeros(max(0, hop * (n_frames - 1) + n_fft - len(y)), np.float32)])
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = yp[idx].astype(np.float64) * win[None, :]
    spec = np.fft.rfft(frames, axis=1).astype(np.complex64)  # stored as complex64 like librosa
    return np.abs(spec).T.astype(np.float32, copy=False)


def minmax_normalize(S: np.ndarray) -> np.ndarray:
    """``(S - S.min()) / (S.max() - S.min() + 1e-10)`` (spectrogram.py:12-21), float32 result."""
    S = np.asarray(S, dtype=np.float32)
    lo = S.min()
    rng = np.float32(np.float64(S.max() - lo) + 1e-10)
    return ((S - lo) / rng).astype(np.float32)


def hybrid_spectrogram(audio: np.ndarray, n_fft: int = 512, spec_width: int = 256) -> np.ndarray:
    """``get_spectrogram_from_audio(audio, n_fft=n_fft, mel_bins=-1, spec_width=spec_width)``.

    Returns the normalised ``[n_fft//2+1, spec_width]`` float32 model input of the hybrid
    frontend.  If the chunk yields fewer than ``spec_width`` frames the result is narrower,
    exactly as the reference's slicing leaves it.
    """
    audio = np.asarray(audio, dtype=np.float32)
    hop = (len(audio) // spec_width) if spec_width > 0 else n_fft // 2
    S = stft_magnitude(audio, n_fft, hop)
    S = S[:, :spec_width]
    return minmax_normalize(S)
