"""Host-callable spectrogram entry points, computed on the MI355X.

``get_spectrogram_from_audio(audio, sample_rate, n_fft, mel_bins, spec_width, mag_scale, mode, n_mfcc)`` keeps the
reference signature and every mode of it (reference: birdnet_stm32/audio/spectrogram.py:24-149):

* ``mel_bins <= 0`` / ``mode='linear'`` — the hybrid frontend's input: ``normalize(abs(stft(y, n_fft, hop = len(y) //
  spec_width))[:, :spec_width])`` (reference :61,106-115,133,149) through ``bn_stft_mag``;
* ``mode='mel'`` with ``mag_scale`` none / pwl / pcen / db, ``mode='log_mel'``, ``mode='mfcc'`` — the precomputed
  frontends ('librosa', 'log_mel', 'mfcc'; reference :63-104,116-147) through ``bn_mel_spectrogram``: the same STFT
  kernel with the Slaney mel basis applied in LDS, then one finishing pass per chunk.

The reference applies ``mag_scale`` to linear spectrograms too (:135-147); no caller on the path does (the hybrid
model scales inside the graph), and that combination raises ``NotImplementedError`` here.

``spectrograms_from_chunks`` / ``mel_spectrograms_from_chunks`` are the batched forms the evaluator uses: one launch
group for all chunks of a file.  There is no CPU fallback: without a GPU these functions raise.
"""

from __future__ import annotations

import ctypes
import math

import numpy as np

_ctx = None
_basis_cache: dict = {}

_MODES = {"mel": 0, "log_mel": 1, "mfcc": 2}
_MAGS = {"none": 0, "pwl": 1, "pcen": 2, "db": 3}


def _context():
    global _ctx
    if _ctx is None:
        from birdnet_stm32 import _hip

        _ctx = _hip.Context(0, 1)
    return _ctx


def normalize(S: np.ndarray) -> np.ndarray:
    """Per-sample min-max normalisation to [0, 1] (reference :12-21)."""
    lo = S.min()
    return (S - lo) / (S.max() - lo + 1e-10)


def spectrograms_from_chunks(chunks: np.ndarray, n_fft: int = 512, spec_width: int = 256, normalize_out: bool = True,
                             exact: bool = False) -> np.ndarray:
    """``[B, T]`` float32 chunks -> ``[B, n_fft//2+1, spec_width]`` float32, one GPU launch group.

    Default: the float32 FFT (within 2e-6 of the peak of the reference's values) — what float32 models, PTQ calibration and
    ``get_spectrogram_from_audio`` need.  ``exact=True``: ``bn_stft_mag_exact`` — librosa's float64 arithmetic value for value (a float64
    FFT, about five times slower), so that an INT8 runner behind this host-side call quantises the bytes the reference quantises;
    ``evaluate``'s per-file loop passes it for INT8 runners only."""
    import torch

    from birdnet_stm32.models.runners import stft_device

    x = np.ascontiguousarray(np.asarray(chunks, np.float32))
    if x.ndim != 2:
        raise ValueError("chunks must be [B, T]")
    ctx = _context()  # raises when no MI355X is present: there is no CPU fallback
    d = torch.from_numpy(x).cuda()
    out = stft_device(ctx, d, n_fft=n_fft, spec_width=spec_width, normalize=normalize_out, exact=exact)
    return out.cpu().numpy()


def pcen_coefficient(sample_rate: int, hop: int, time_constant: float = 0.4) -> float:
    """Smoothing coefficient ``b`` librosa.pcen derives from ``time_constant * sr / hop_length`` frames."""
    t = time_constant * sample_rate / float(hop)
    return (math.sqrt(1.0 + 4.0 * t * t) - 1.0) / (2.0 * t * t)


def dct_ortho_rows(n_mfcc: int, n_mels: int) -> np.ndarray:
    """First ``n_mfcc`` rows of the orthonormal DCT-II matrix (``scipy.fftpack.dct(type=2, norm='ortho')`` along the mel axis)."""
    m = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    mat = 2.0 * np.cos(np.pi * k * (2.0 * m + 1.0) / (2.0 * n_mels))
    mat *= np.sqrt(1.0 / (2.0 * n_mels))
    mat[0] *= np.sqrt(0.5)
    return mat.astype(np.float32)


def _device_basis(ctx, dev, sample_rate: int, n_fft: int, mel_bins: int):
    """Band-sparse Slaney mel basis (fmin 150 Hz, fmax sr // 2: reference :70-75) resident on the device."""
    import torch

    key = (ctx.device, int(sample_rate), int(n_fft), int(mel_bins))
    if key not in _basis_cache:
        from birdnet_stm32.audio.melbank import mel_filterbank
        from birdnet_stm32.models._lower_f32 import mel_bands

        basis = mel_filterbank(int(sample_rate), int(n_fft), int(mel_bins), 150.0, float(sample_rate // 2))  # [M, F]
        vals, bands = mel_bands(np.ascontiguousarray(basis.T), n_fft // 2 + 1)
        _basis_cache[key] = (torch.from_numpy(vals).to(dev), torch.from_numpy(np.ascontiguousarray(bands)).to(dev))
    return _basis_cache[key]


def mel_spectrograms_device(ctx, audio, sample_rate: int = 24000, n_fft: int = 512, mel_bins: int = 64, spec_width: int = 256,
                            mag_scale: str = "none", mode: str = "mel", n_mfcc: int = 20, return_energies: bool = False):
    """``bn_mel_spectrogram`` on a CUDA tensor: float32 ``[B, T]`` -> float32 ``[B, mel_bins | n_mfcc, spec_width]``.

    ``return_energies`` also returns the un-normalised mel energies ``[B, mel_bins, frames]`` the finishing pass started from
    (tests use them to check that pass apart from the STFT's float32 noise floor)."""
    import torch

    from birdnet_stm32 import _hip

    if mode not in _MODES:
        raise ValueError(f"unknown spectrogram mode {mode!r}")
    if mag_scale not in _MAGS:
        raise ValueError(f"unknown mag_scale {mag_scale!r}")
    if not (audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous() and audio.dim() == 2):
        raise ValueError("audio must be a contiguous float32 CUDA tensor [B, T]")
    B, T = audio.shape
    hop = T // spec_width if spec_width > 0 else n_fft // 2
    dev = audio.device
    d_w, d_bands = _device_basis(ctx, dev, sample_rate, n_fft, mel_bins)
    rows = n_mfcc if mode == "mfcc" else mel_bins
    out = torch.empty((B, rows, spec_width), dtype=torch.float32, device=dev)
    frames = 1 + T // max(1, hop) if mode == "mfcc" else spec_width
    if B == 0:
        return (out, torch.empty((0, mel_bins, frames), dtype=torch.float32, device=dev)) if return_energies else out
    work = torch.empty(max(1, B) * (mel_bins * (1 + T // max(1, hop)) + 2), dtype=torch.float32, device=dev)
    d_dct = torch.from_numpy(dct_ortho_rows(n_mfcc, mel_bins)).to(dev) if mode == "mfcc" else None
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _hip.check(ctx.lib.bn_mel_spectrogram(
            ctx.handle, audio.data_ptr(), B, T, n_fft, hop, spec_width, d_w.data_ptr(), d_bands.data_ptr(), mel_bins, _MODES[mode],
            _MAGS[mag_scale], pcen_coefficient(sample_rate, hop) if hop > 0 else 0.0, d_dct.data_ptr() if d_dct is not None else None,
            n_mfcc, work.data_ptr(), out.data_ptr(), stream))
        torch.cuda.current_stream(dev).synchronize()  # `work` and the DCT rows are released on return
    if return_energies:
        return out, work[: B * mel_bins * frames].view(B, mel_bins, frames).clone()
    return out


def mel_spectrograms_from_chunks(chunks: np.ndarray, sample_rate: int = 24000, n_fft: int = 512, mel_bins: int = 64, spec_width: int = 256,
                                 mag_scale: str = "none", mode: str = "mel", n_mfcc: int = 20) -> np.ndarray:
    """Batched precomputed-frontend spectrograms: ``[B, T]`` float32 -> ``[B, mel_bins | n_mfcc, spec_width]`` float32."""
    import torch

    x = np.ascontiguousarray(np.asarray(chunks, np.float32))
    if x.ndim != 2:
        raise ValueError("chunks must be [B, T]")
    ctx = _context()
    d = torch.from_numpy(x).cuda()
    return mel_spectrograms_device(ctx, d, sample_rate, n_fft, mel_bins, spec_width, mag_scale, mode, n_mfcc).cpu().numpy()


def get_spectrogram_from_audio(audio: np.ndarray, sample_rate: int = 24000, n_fft: int = 512, mel_bins: int = 64,
                               spec_width: int = 256, mag_scale: str = "none", mode: str = "mel", n_mfcc: int = 20) -> np.ndarray:
    """Reference-compatible single-chunk entry point (all modes of the reference function)."""
    y = np.asarray(audio, np.float32)[None, :]
    if mode in ("mfcc", "log_mel"):
        return mel_spectrograms_from_chunks(y, sample_rate, n_fft, mel_bins, spec_width, "none", mode, n_mfcc)[0]
    if mel_bins <= 0 or mode == "linear":
        if mag_scale != "none":
            raise NotImplementedError("host-side magnitude scaling of linear spectrograms is not part of any frontend's path")
        return spectrograms_from_chunks(y, n_fft, spec_width)[0]
    return mel_spectrograms_from_chunks(y, sample_rate, n_fft, mel_bins, spec_width, mag_scale, "mel", n_mfcc)[0]
