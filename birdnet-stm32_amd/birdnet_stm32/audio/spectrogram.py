"""Host-callable spectrogram entry point of the hybrid frontend, computed on the MI355X.

``get_spectrogram_from_audio(audio, sample_rate, n_fft, mel_bins, spec_width, mag_scale, mode, n_mfcc)``
keeps the reference signature (reference: birdnet_stm32/audio/spectrogram.py:24-33).  The branch the hot
path uses — ``mel_bins <= 0`` / ``mode='linear'``: ``normalize(abs(stft(y, n_fft, hop = len(y) //
spec_width))[:, :spec_width])`` (reference :61,106-115,133,149) — runs through ``bn_stft_mag``.  The
host-side mel / MFCC / log-mel / PCEN branches (:63-104,116-147) belong to the precomputed frontends,
which this build does not accelerate; they raise ``NotImplementedError``.

``spectrograms_from_chunks`` is the batched form the evaluator uses: one launch for all chunks of a file.
"""

from __future__ import annotations

import numpy as np

_ctx = None


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


def spectrograms_from_chunks(chunks: np.ndarray, n_fft: int = 512, spec_width: int = 256, normalize_out: bool = True) -> np.ndarray:
    """``[B, T]`` float32 chunks -> ``[B, n_fft//2+1, spec_width]`` float32, one GPU launch group."""
    import torch

    from birdnet_stm32.models.runners import stft_device

    x = np.ascontiguousarray(np.asarray(chunks, np.float32))
    if x.ndim != 2:
        raise ValueError("chunks must be [B, T]")
    ctx = _context()  # raises when no MI355X is present: there is no CPU fallback
    d = torch.from_numpy(x).cuda()
    out = stft_device(ctx, d, n_fft=n_fft, spec_width=spec_width, normalize=normalize_out)
    return out.cpu().numpy()


def get_spectrogram_from_audio(audio: np.ndarray, sample_rate: int = 24000, n_fft: int = 512, mel_bins: int = 64,
                               spec_width: int = 256, mag_scale: str = "none", mode: str = "mel", n_mfcc: int = 20) -> np.ndarray:
    """Reference-compatible single-chunk entry point; only the linear-magnitude branch is accelerated."""
    if mel_bins <= 0 or mode == "linear":
        if mag_scale != "none":
            raise NotImplementedError("host-side magnitude scaling of linear spectrograms is not part of the hybrid path")
        return spectrograms_from_chunks(np.asarray(audio, np.float32)[None, :], n_fft, spec_width)[0]
    raise NotImplementedError(
        f"mode={mode!r} with mel_bins={mel_bins}: the precomputed frontends (librosa / mfcc / log_mel) have no MI355X path in "
        "this build; only the hybrid frontend's linear STFT magnitude (mel_bins=-1) is implemented"
    )
