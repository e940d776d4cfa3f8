"""Audio ingest on the MI355X: PCM frames -> mono -> resampled -> peak-normalised -> fixed-length chunks.

Device form of the reference's ``load_audio_window`` + ``split_audio_into_chunks``
(reference: birdnet_stm32/audio/io.py:63-130, :133-174), for many files per launch.  The host only parses
the RIFF header and hands the interleaved PCM payload over as it lies in the file (int16 / packed int24 /
int32 / float32); everything the reference computes on decoded samples —

* ``y.mean(axis=1)``                                     (io.py:119)
* ``resample_poly(y, sr_out // g, sr_in // g)``          (io.py:26-30, scipy's Kaiser(5.0) low-pass)
* ``y / max(abs(y))`` when the peak is positive          (io.py:123-125)
* chunk gathering with a single right zero pad            (io.py:155-174)

— runs in ``bn_ingest_resample`` + ``bn_ingest_chunks`` with numpy's and scipy's operation order, so the
chunks are bit-identical to the host functions in ``birdnet_stm32.audio.io``.  The filter coefficients come
from ``scipy.signal.firwin`` exactly as ``resample_poly`` designs them (a few thousand floats, cached per rate
pair); the chunk start positions depend only on lengths and are computed on the host.

There is no CPU fallback: without the HIP library or a GPU every function here raises.
"""

from __future__ import annotations

import ctypes
import warnings
from dataclasses import dataclass
from math import gcd

import numpy as np

from birdnet_stm32 import _hip
from birdnet_stm32.audio import io as _io

PCM_S16, PCM_S24, PCM_S32, PCM_F32 = 0, 1, 2, 3
_BYTES = {PCM_S16: 2, PCM_S24: 3, PCM_S32: 4, PCM_F32: 4}


@dataclass
class PcmWindow:
    """One file's read window as it lies in the container: ``payload`` holds ``frames * channels`` samples."""

    payload: np.ndarray  # uint8 view of the interleaved samples
    fmt: int             # PCM_*
    channels: int
    sample_rate: int

    @property
    def frames(self) -> int:
        return self.payload.shape[0] // (_BYTES[self.fmt] * self.channels)


def window_from_frames(frames: np.ndarray, sample_rate: int) -> PcmWindow:
    """Wrap already decoded float32 frames ``[n, ch]`` (or ``[n]``) as a float32 PCM window."""
    f = np.ascontiguousarray(np.asarray(frames, np.float32))
    if f.ndim == 1:
        f = f[:, None]
    return PcmWindow(f.reshape(-1).view(np.uint8), PCM_F32, f.shape[1], int(sample_rate))


def window_from_int16(frames: np.ndarray, sample_rate: int) -> PcmWindow:
    """Wrap int16 PCM frames ``[n, ch]`` (or ``[n]``)."""
    f = np.ascontiguousarray(np.asarray(frames, np.int16))
    if f.ndim == 1:
        f = f[:, None]
    return PcmWindow(f.reshape(-1).view(np.uint8), PCM_S16, f.shape[1], int(sample_rate))


def read_pcm_window(path: str, max_duration: float | None = 30, chunk_duration: float = 3.0,
                    random_offset: bool = False) -> PcmWindow | None:
    """The read window of ``load_audio_window`` (io.py:89-117) without decoding; ``None`` for an empty/unreadable file.

    PCM 16/24/32-bit and float32 WAV payloads are passed through untouched; any other encoding the host
    reader understands (8-bit, float64, non-WAV through ``soundfile`` when present) is decoded to float32
    frames first — format conversion only, the arithmetic still happens on the device.
    """
    try:
        if path.lower().endswith(".wav"):
            with open(path, "rb") as fh:
                raw = fh.read()
            code, ch, sr0, bits, off, nbytes = _io._wav_layout(raw)
            fmt = {(1, 16): PCM_S16, (1, 24): PCM_S24, (1, 32): PCM_S32, (3, 32): PCM_F32}.get((code, bits))
            if fmt is not None and ch >= 1 and sr0 > 0:
                frame = (bits // 8) * ch
                total = nbytes // frame
                first, count = _window_frames(total, sr0, max_duration, chunk_duration, random_offset)
                if count <= 0:
                    return None
                view = np.frombuffer(raw, np.uint8, count * frame, off + first * frame)
                return PcmWindow(view, fmt, ch, sr0)
        if path.lower().endswith(".flac"):  # decoded on the host (entropy coding is serial); mixing / resampling / scaling stay on the device
            from birdnet_stm32.audio import _flac

            with open(path, "rb") as fh:
                raw = fh.read()
            sr0, ch, bps, total = _flac.flac_info(raw)
            if total == 0:
                total = int(_flac.decode_flac(raw)[0].shape[0])  # (the whole stream: checked against its MD5)
            first, count = _window_frames(total, sr0, max_duration, chunk_duration, random_offset)
            if count <= 0:
                return None
            ints = _flac.decode_flac(raw, first, count, verify_md5=False)[0]
            if bps == 16:
                return PcmWindow(np.ascontiguousarray(ints.astype(np.int16)).reshape(-1).view(np.uint8), PCM_S16, ch, sr0)
            if bps <= 32 and bps != 16:  # libsndfile's scaling x / 2^(bps-1) == (x << (32 - bps)) / 2^31: the device's 32-bit PCM path
                return PcmWindow(np.ascontiguousarray(ints << (32 - bps) if bps < 32 else ints).astype(np.int32).reshape(-1).view(np.uint8), PCM_S32, ch, sr0)
        frames, sr0 = _io._read_window(path, max_duration, chunk_duration, random_offset)
        if frames.size == 0:
            return None
        return window_from_frames(frames, sr0)
    except Exception:
        return None


def _window_frames(total: int, sr0: int, max_duration, chunk_duration: float, random_offset: bool):
    """(first frame, frame count) of the read window (io.py:95-111)."""
    if total <= 0 or sr0 <= 0:
        return 0, 0
    duration = total / float(sr0)
    want = min(float(max_duration), duration) if max_duration and max_duration > 0 else duration
    offset_s = 0.0
    if random_offset:
        latest = max(0.0, duration - max(chunk_duration, want))
        offset_s = float(np.random.uniform(0.0, latest)) if latest > 0 else 0.0
    first = min(int(offset_s * sr0), total)
    return first, int(min(total - first, want * sr0))


_filters: dict[tuple[int, int], tuple[np.ndarray, int, int]] = {}


def polyphase_filter(up: int, down: int):
    """``(taps [up, taps_per_phase] float32, taps_per_phase, n_pre_remove)`` of ``resample_poly(x, up, down)``.

    scipy designs ``firwin(2 * half_len + 1, 1 / max(up, down), window=('kaiser', 5.0))`` with ``half_len =
    10 * max(up, down)``, casts it to the signal's float32 and scales it by ``up``; it then prepends
    ``down - half_len % down`` zeros so that output 0 is centred on input 0 and drops the first
    ``(half_len + n_pre_pad) // down`` outputs.  ``upfirdn`` stores the filter phase-major with each phase
    reversed (the coefficient of the oldest input sample first) — the layout ``bn_ingest_resample`` takes.
    """
    key = (int(up), int(down))
    if key not in _filters:
        from scipy.signal import firwin

        max_rate = max(up, down)
        half_len = 10 * max_rate
        h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float32)
        h *= up
        n_pre_pad = down - half_len % down
        n_pre_remove = (half_len + n_pre_pad) // down
        length = n_pre_pad + h.shape[0]
        per_phase = -(-length // up)
        full = np.zeros(per_phase * up, np.float32)
        full[n_pre_pad:length] = h
        taps = np.ascontiguousarray(full.reshape(per_phase, up).T[:, ::-1])
        _filters[key] = (taps, per_phase, n_pre_remove)
    return _filters[key]


def resampled_length(n_in: int, up: int, down: int) -> int:
    n = n_in * up
    return n // down + (1 if n % down else 0)


def chunk_table(lengths, sample_rate: int, chunk_duration: float, chunk_overlap: float):
    """Start offset (relative to its window), valid length and window index of every chunk, in window order."""
    size = int(sample_rate * chunk_duration)
    step = _io._step(sample_rate, chunk_duration, chunk_overlap)
    starts, valid, owner, counts = [], [], [], []
    for w, n in enumerate(lengths):
        if n <= 0 or size <= 0:
            counts.append(0)
            continue
        if n <= size:
            s = np.zeros(1, np.int64)
            v = np.full(1, n, np.int32)
        else:
            s = np.arange(0, n - size + 1, step, dtype=np.int64)
            if s.size == 0 or s[-1] + size < n:
                s = np.append(s, n - size)
            v = np.full(s.shape[0], size, np.int32)
        starts.append(s)
        valid.append(v)
        owner.append(np.full(s.shape[0], w, np.int32))
        counts.append(int(s.shape[0]))
    cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)  # noqa: E731
    return cat(starts, np.int64), cat(valid, np.int32), cat(owner, np.int32), counts, size


def ingest_windows_device(ctx: _hip.Context, windows: list[PcmWindow], sample_rate: int = 24000, chunk_duration: float = 3.0,
                          chunk_overlap: float = 0.0, return_windows: bool = False):
    """Chunks of all ``windows`` as one float32 CUDA tensor ``[N, T]`` plus the chunk count of every window.

    Windows sharing (format, channels, native rate) go through one ``bn_ingest_resample`` launch; a single
    ``bn_ingest_chunks`` launch gathers the peak-normalised chunks of all windows.  With ``return_windows`` the
    un-normalised resampled windows, their offsets and the peaks are returned as well (tests).
    """
    import torch

    lib = ctx.lib
    dev = torch.device("cuda", ctx.device)
    n_out = []
    for w in windows:
        g = gcd(w.sample_rate, sample_rate)
        n_out.append(resampled_length(w.frames, sample_rate // g, w.sample_rate // g) if w.sample_rate != sample_rate else w.frames)
    out_off = np.zeros(len(windows) + 1, np.int64)
    np.cumsum(n_out, out=out_off[1:])
    total_out = int(out_off[-1])
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        mono = torch.empty(max(total_out, 1), dtype=torch.float32, device=dev)
        peak = torch.zeros(max(len(windows), 1), dtype=torch.float32, device=dev)
        groups: dict[tuple[int, int, int], list[int]] = {}
        for i, w in enumerate(windows):
            groups.setdefault((w.fmt, w.channels, w.sample_rate), []).append(i)
        keep = []  # device buffers must outlive the asynchronous launches
        for (fmt, ch, sr0), members in groups.items():
            members = [i for i in members if n_out[i] > 0]
            for lo in range(0, len(members), 65535):
                part = members[lo : lo + 65535]
                payload = np.concatenate([windows[i].payload for i in part]) if len(part) > 1 else windows[part[0]].payload
                frames = np.array([windows[i].frames for i in part], np.int64)
                in_off = np.zeros(len(part) + 1, np.int64)
                np.cumsum(frames, out=in_off[1:])
                # each window of the group writes its own slice of `mono`; windows of a group need not be adjacent
                with warnings.catch_warnings():  # the payload is a read-only view of the file's bytes; it is only read
                    warnings.simplefilter("ignore", UserWarning)
                    d_pcm = torch.from_numpy(np.ascontiguousarray(payload)).to(dev)
                d_in = torch.from_numpy(in_off).to(dev)
                if sr0 == sample_rate:
                    up = down = 1
                    taps, per_phase, pre = None, 0, 0
                else:
                    g = gcd(sr0, sample_rate)
                    up, down = sample_rate // g, sr0 // g
                    taps, per_phase, pre = polyphase_filter(up, down)
                d_taps = torch.from_numpy(taps).to(dev) if taps is not None else None
                keep += [d_pcm, d_in, d_taps]
                # contiguous runs of windows share one launch (offset tables must be monotone per launch)
                run_start = 0
                while run_start < len(part):
                    run_end = run_start + 1
                    while run_end < len(part) and part[run_end] == part[run_end - 1] + 1:
                        run_end += 1
                    idx = part[run_start:run_end]
                    d_out = torch.from_numpy(out_off[idx[0] : idx[-1] + 2].copy()).to(dev)
                    keep.append(d_out)
                    _hip.check(lib.bn_ingest_resample(
                        ctx.handle, d_pcm.data_ptr(), fmt, ch, d_in.data_ptr() + 8 * run_start, d_out.data_ptr(), len(idx),
                        int(frames[run_start:run_end].max()), int(max(n_out[i] for i in idx)),
                        d_taps.data_ptr() if d_taps is not None else None, up, down, per_phase, pre,
                        mono.data_ptr(), peak.data_ptr() + 4 * idx[0], stream))
                    run_start = run_end
        starts, valid, owner, counts, size = chunk_table(n_out, sample_rate, chunk_duration, chunk_overlap)
        n_chunks = int(starts.shape[0])
        chunks = torch.empty((n_chunks, size), dtype=torch.float32, device=dev)
        if n_chunks:
            d_src = torch.from_numpy(starts + out_off[owner]).to(dev)
            d_valid = torch.from_numpy(valid).to(dev)
            d_owner = torch.from_numpy(owner).to(dev)
            keep += [d_src, d_valid, d_owner]
            _hip.check(lib.bn_ingest_chunks(ctx.handle, mono.data_ptr(), peak.data_ptr(), d_src.data_ptr(), d_valid.data_ptr(),
                                            d_owner.data_ptr(), n_chunks, size, chunks.data_ptr(), stream))
        torch.cuda.current_stream(dev).synchronize()
    if return_windows:
        return chunks, counts, mono[:total_out], out_off, peak[: len(windows)]
    return chunks, counts


def load_audio_files_device(ctx: _hip.Context, paths: list[str], sample_rate: int = 24000, max_duration: float | None = 30,
                            chunk_duration: float = 3.0, chunk_overlap: float = 0.0, random_offset: bool = False):
    """``load_audio_file`` for a list of files at once: ``(chunks CUDA [N, T], chunk count per path)``.

    An unreadable or empty file contributes zero chunks (the reference returns an empty list for it, io.py:206-207).
    """
    windows, slots = [], []
    for i, p in enumerate(paths):
        w = read_pcm_window(p, max_duration, chunk_duration, random_offset)
        if w is not None and w.frames > 0:
            windows.append(w)
            slots.append(i)
    counts = [0] * len(paths)
    chunks, per_window = ingest_windows_device(ctx, windows, sample_rate, chunk_duration, chunk_overlap)
    for i, c in zip(slots, per_window):
        counts[i] = c
    return chunks, counts


def pool_scores_device(ctx: _hip.Context, scores, counts, method: str = "average", beta: float = 10.0):
    """``pool_scores`` (reference: evaluation/pooling.py:25-47) for every file of a batch: ``[N, C]`` CUDA scores whose
    rows are grouped per file (``counts[i]`` rows each) -> ``[len(counts), C]`` CUDA tensor."""
    import torch

    key = method.lower()
    code = 0 if key in ("avg", "mean", "average") else 1 if key == "max" else 2 if key in ("lme", "log_mean_exp", "log_mean_exponential") else None
    if code is None:
        raise ValueError(f"Unsupported pooling method: {method}")
    if scores.dim() != 2:
        raise ValueError("chunk_scores must be [N_chunks, C]")
    if not (scores.is_cuda and scores.dtype == torch.float32 and scores.is_contiguous()):
        raise ValueError("scores must be a contiguous float32 CUDA tensor")
    off = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(np.asarray(counts, np.int64), out=off[1:])
    if int(off[-1]) != scores.shape[0]:
        raise ValueError(f"counts add up to {int(off[-1])} rows, scores has {scores.shape[0]}")
    out = torch.empty((len(counts), scores.shape[1]), dtype=torch.float32, device=scores.device)
    with torch.cuda.device(scores.device):
        d_off = torch.from_numpy(off).to(scores.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(scores.device).cuda_stream)
        _hip.check(ctx.lib.bn_pool_scores(ctx.handle, scores.data_ptr(), d_off.data_ptr(), len(counts), scores.shape[1], code,
                                          float(beta), out.data_ptr(), stream))
        torch.cuda.current_stream(scores.device).synchronize()
    return out
