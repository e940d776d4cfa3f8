"""TEST INFRASTRUCTURE — CPU restatement of the host-side spectrogram modes of the precomputed frontends.

Reference: birdnet_stm32/audio/spectrogram.py:24-149 ``get_spectrogram_from_audio`` for ``mel_bins > 0``:

  * :63-84    mode='mfcc'    melspectrogram(power=2) -> power_to_db(ref=max) -> mfcc(n_mfcc, norm='ortho') -> [:, :W] -> normalize
  * :86-104   mode='log_mel' melspectrogram(power=1) -> [:, :W] -> log1p -> normalize
  * :116-147  mode='mel'     melspectrogram(power=1) -> [:, :W] -> mag_scale {none, pcen, pwl, db} -> normalize

The arithmetic is librosa 0.11.0's (``requirements.txt:1``), which is NOT installed here and not vendored under
/root/reference, so — like oracle/stft.py — this file restates its published algorithm:

  * ``feature.melspectrogram``: ``abs(stft(y, n_fft, hop, window='hann', center=True)) ** power`` (float32), mixed by
    ``filters.mel(sr, n_fft, n_mels, fmin=150, fmax=sr//2, htk=False, norm='slaney')`` with a float32 ``einsum``;
  * ``power_to_db(S, ref, amin=1e-10, top_db=80)``: ``10 log10(max(amin, S)) - 10 log10(max(amin, ref))``, floored at
    ``max - top_db``; ``amplitude_to_db(S, ref=max)`` = ``power_to_db(S**2, ref=max**2)``;
  * ``pcen(S, sr, hop_length, gain=0.98, bias=2, power=0.5, time_constant=0.4, eps=1e-6)``: ``b = (sqrt(1 + 4 T^2) - 1) /
    (2 T^2)``, ``T = time_constant sr / hop``; ``M = scipy.signal.lfilter([b], [1, b - 1], S, zi=lfilter_zi)`` along time
    (float64), ``smooth = exp(-gain (log eps + log1p(M / eps)))``, ``bias**power * expm1(power log1p(S smooth / bias))``;
  * ``feature.mfcc(S=S_db, n_mfcc, norm='ortho')``: ``scipy.fftpack.dct(S_db, axis=-2, type=2, norm='ortho')[:n_mfcc]``.

**Parity unpinned** against librosa itself (it cannot be imported here and the reference's tests hold only shape/range
checks for these modes: tests/test_spectrogram.py).  Pinned pieces: the mel basis (equals the shipped checkpoint's mixer,
tests/test_oracle_pinning.py), the DCT and the IIR smoother (scipy's own ``dct`` / ``lfilter`` are called here, as librosa
calls them), ``normalize`` (reference spectrogram.py:12-21, restated in oracle/stft.py).

Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

import numpy as np

from oracle import melbank, stft


def mel_spectrogram(audio: np.ndarray, sample_rate: int, n_fft: int, n_mels: int, spec_width: int, power: float):
    """``(librosa.feature.melspectrogram(..., power) with all 1 + len // hop frames, hop)``, ``hop = len(audio) // spec_width``."""
    hop = len(audio) // spec_width if spec_width > 0 else n_fft // 2
    S = stft.stft_magnitude(np.asarray(audio, np.float32), n_fft, hop)  # [F, frames] float32
    if power != 1.0:
        S = S**power
    basis = melbank.mel_filterbank(sample_rate, n_fft, n_mels, 150.0, float(sample_rate // 2))  # [M, F] float32
    mel = np.einsum("ft,mf->mt", S, basis, optimize=True).astype(np.float32)
    return mel, hop


def power_to_db(S: np.ndarray, ref: float, amin: float = 1e-10, top_db: float = 80.0) -> np.ndarray:
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec = log_spec - 10.0 * np.log10(np.maximum(amin, ref))
    return np.maximum(log_spec, log_spec.max() - top_db)


def pcen_coefficient(sample_rate: int, hop: int, time_constant: float = 0.4) -> float:
    t_frames = time_constant * sample_rate / float(hop)
    return float((np.sqrt(1 + 4 * t_frames**2) - 1) / (2 * t_frames**2))


def pcen(S: np.ndarray, sample_rate: int, hop: int, gain=0.98, bias=2.0, power=0.5, eps=1e-6) -> np.ndarray:
    from scipy.signal import lfilter, lfilter_zi

    b = pcen_coefficient(sample_rate, hop)
    zi = np.empty((1, 1))
    zi[:] = lfilter_zi([b], [1, b - 1])[:]
    smooth_in, _ = lfilter([b], [1, b - 1], S, zi=zi, axis=-1)
    smooth = np.exp(-gain * (np.log(eps) + np.log1p(smooth_in / eps)))
    return (bias**power) * np.expm1(power * np.log1p(S * smooth / bias))


def get_spectrogram(audio: np.ndarray, sample_rate: int = 24000, n_fft: int = 512, mel_bins: int = 64, spec_width: int = 256,
                    mag_scale: str = "none", mode: str = "mel", n_mfcc: int = 20) -> np.ndarray:
    """The reference function for ``mel_bins > 0`` (the linear branch is oracle/stft.py's hybrid_spectrogram)."""
    if mode == "mfcc":
        from scipy.fftpack import dct

        S, hop = mel_spectrogram(audio, sample_rate, n_fft, mel_bins, spec_width, 2.0)
        S_log = power_to_db(S, ref=S.max())  # over ALL frames: the reference cuts to spec_width after the DCT (:80-83)
        out = dct(S_log, axis=-2, type=2, norm="ortho")[:n_mfcc]
        return stft.minmax_normalize(out[:, :spec_width]), hop
    S, hop = mel_spectrogram(audio, sample_rate, n_fft, mel_bins, spec_width, 1.0)
    S = S[:, :spec_width]  # "ensure fixed width" comes before the scaling in these modes (:104, :133)
    if mode == "log_mel":
        return stft.minmax_normalize(np.log1p(S)), hop
    if mag_scale == "pcen":
        S = pcen(S * (2.0**31), sample_rate, hop)
    elif mag_scale == "pwl":
        lo, hi = S.min(), S.max()
        x = (S - lo) / (hi - lo + 1e-10)
        relu = lambda z: np.maximum(z, 0.0)  # noqa: E731
        S = 0.40 * x + 0.25 * relu(x - 0.10) + 0.15 * relu(x - 0.35) + 0.08 * relu(x - 0.65)
    elif mag_scale == "db":
        S = power_to_db(np.square(S), ref=S.max() ** 2)
    return stft.minmax_normalize(S), hop
