"""Audio ingest: WAV decode, mono mix, polyphase resampling, peak normalisation, fixed-length chunking.

Behaviour of the reference's birdnet_stm32/audio/io.py:14-224, which the evaluate path calls once
per file (reference: birdnet_stm32/evaluation/metrics.py:44-46):

* ``load_audio_window``: read at most ``max_duration`` seconds from the start of the file (or a
  random offset), average the channels, resample with ``scipy.signal.resample_poly(up, down)``
  (``up = sr_out/g``, ``down = sr_in/g``), divide by the window's absolute peak; any failure yields
  an empty array (the caller skips the file).
* ``split_audio_into_chunks``: a waveform not longer than one chunk is right-padded with zeros once;
  otherwise chunk starts are ``arange(0, len - chunk + 1, step)`` plus a tail chunk at ``len - chunk``
  when the last start does not reach the end; ``step = int(sr * (chunk_duration - min(overlap,
  chunk_duration - 0.1)))``.

The reference decodes through libsndfile (``soundfile``); that package is not on the MI355X image, so
RIFF/WAVE files (PCM 8/16/24/32-bit, IEEE float 32/64, plain or WAVE_FORMAT_EXTENSIBLE) are parsed
here with libsndfile's scaling (int16 / 32768 etc.) and native FLAC streams are decoded by the plain-C
decoder of this build (``audio/_flac.py`` -> ``csrc/host/bn_flac.c``).  Other containers (Ogg, MP3, M4A)
are handed to ``soundfile`` when it is importable and otherwise count as unreadable; ``evaluate`` says
how many files it skipped for that reason.
"""

from __future__ import annotations

import struct
from math import gcd

import numpy as np
from scipy.signal import resample_poly


def fast_resample(y: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """Polyphase resampling to ``sr_out``; identity (as float32) when the rates agree."""
    if sr_in == sr_out:
        return y.astype(np.float32, copy=False)
    g = gcd(sr_in, sr_out)
    return resample_poly(y, sr_out // g, sr_in // g).astype(np.float32, copy=False)


def estimate_num_chunks(num_samples: int, sample_rate: int, chunk_duration: float, chunk_overlap: float = 0.0) -> int:
    """Number of chunks :func:`split_audio_into_chunks` emits for ``num_samples`` samples."""
    size = int(sample_rate * chunk_duration)
    if num_samples <= 0 or size <= 0:
        return 0
    if num_samples <= size:
        return 1
    step = _step(sample_rate, chunk_duration, chunk_overlap)
    rest = num_samples - size
    return 1 + rest // step + (1 if rest % step else 0)


def _step(sample_rate: int, chunk_duration: float, chunk_overlap: float) -> int:
    overlap = max(0.0, min(chunk_overlap, chunk_duration - 0.1))
    return max(1, int(sample_rate * (chunk_duration - overlap)))


def have_soundfile() -> bool:
    """Is libsndfile's Python binding importable (needed for containers other than RIFF/WAVE)?"""
    try:
        import soundfile  # noqa: F401
    except Exception:
        return False
    return True


# ------------------------------------------------------------------------------------------ WAV
def _wav_layout(raw: bytes):
    """Return (format_tag, channels, sample_rate, bits, data_offset, data_bytes) of a RIFF/WAVE file."""
    if len(raw) < 12 or raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        tag, size = raw[pos : pos + 4], struct.unpack_from("<I", raw, pos + 4)[0]
        body = pos + 8
        if tag == b"fmt ":
            code, ch, sr, _br, _align, bits = struct.unpack_from("<HHIIHH", raw, body)
            if code == 0xFFFE and size >= 26:  # WAVE_FORMAT_EXTENSIBLE: real code is the sub-format GUID's first word
                code = struct.unpack_from("<H", raw, body + 24)[0]
            fmt = (code, ch, sr, bits)
        elif tag == b"data":
            data = (body, min(size, len(raw) - body))
            break
        pos = body + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("WAVE file lacks a fmt or data chunk")
    return (*fmt, *data)


def _g711_tables() -> tuple[np.ndarray, np.ndarray]:
    """(A-law, mu-law) byte -> 16-bit sample tables of ITU-T G.711, the values libsndfile's decoders hold (it reads such WAVE files — format
    tags 6 and 7 — as those int16 samples, scaled by 1 / 32768 like 16-bit PCM)."""
    b = np.arange(256, dtype=np.int32)
    a = b ^ 0x55
    exp, man = (a >> 4) & 7, a & 0x0F
    mag = np.where(exp == 0, (man << 4) + 8, ((man << 4) + 0x108) << np.maximum(exp - 1, 0))
    alaw = np.where(a & 0x80, mag, -mag)              # (bit 7 set = positive, after the 0x55 toggle)
    u = ~b & 0xFF
    exp, man = (u >> 4) & 7, u & 0x0F
    mag = (((man << 3) + 0x84) << exp) - 0x84
    ulaw = np.where(u & 0x80, -mag, mag)
    return alaw.astype(np.int16), ulaw.astype(np.int16)


_G711 = None


def _decode_frames(raw: bytes, code: int, ch: int, bits: int, offset: int, nbytes: int, first: int, count: int) -> np.ndarray:
    """Frames [first, first+count) as float32 ``[count, ch]`` with libsndfile's integer scaling."""
    width = bits // 8
    frame = width * ch
    start = offset + first * frame
    count = max(0, min(count, (nbytes - first * frame) // frame))
    buf = raw[start : start + count * frame]
    if code == 3:  # IEEE float
        x = np.frombuffer(buf, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif code == 1:  # PCM
        if bits == 8:
            x = (np.frombuffer(buf, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(buf, "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(buf, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (np.frombuffer(buf, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif code in (6, 7) and bits == 8:  # G.711 A-law / mu-law: one byte per sample, companded 13- / 14-bit PCM
        global _G711
        if _G711 is None:
            _G711 = _g711_tables()
        x = _G711[0 if code == 6 else 1][np.frombuffer(buf, np.uint8)].astype(np.float32) / 32768.0
    else:
        raise ValueError(f"unsupported WAVE format tag {code}")
    return x.reshape(-1, ch)


def _read_window(path: str, max_duration, chunk_duration: float, random_offset: bool):
    """(frames float32 [n, ch], native sample rate) of the requested window."""
    if path.lower().endswith(".wav"):
        with open(path, "rb") as fh:
            raw = fh.read()
        code, ch, sr0, bits, off, nbytes = _wav_layout(raw)
        total = nbytes // max(1, (bits // 8) * ch)
        reader = lambda first, count: _decode_frames(raw, code, ch, bits, off, nbytes, first, count)  # noqa: E731
    elif path.lower().endswith(".flac"):
        from birdnet_stm32.audio import _flac

        with open(path, "rb") as fh:
            raw = fh.read()
        sr0, _ch, _bps, total = _flac.flac_info(raw)
        if total == 0:  # length not recorded in STREAMINFO: decode once to learn it
            total = int(_flac.decode_flac(raw)[0].shape[0])  # (the whole stream: checked against its MD5)
        reader = lambda first, count: _flac.read_flac_window(raw, first, count)[0]  # noqa: E731
    else:
        import soundfile as sf  # not installed on the MI355X image: such files count as unreadable

        info = sf.info(path)
        sr0, total = int(info.samplerate), int(info.frames)

        def reader(first, count):
            with sf.SoundFile(path, mode="r") as f:
                f.seek(first)
                return f.read(count, dtype="float32", always_2d=True)

    if total <= 0 or sr0 <= 0:
        return np.empty((0, 1), np.float32), sr0
    duration = total / float(sr0)
    want = min(float(max_duration), duration) if max_duration and max_duration > 0 else duration
    offset_s = 0.0
    if random_offset:
        latest = max(0.0, duration - max(chunk_duration, want))
        offset_s = float(np.random.uniform(0.0, latest)) if latest > 0 else 0.0
    first = min(int(offset_s * sr0), total)
    count = int(min(total - first, want * sr0))
    if count <= 0:
        return np.empty((0, 1), np.float32), sr0
    return reader(first, count), sr0


def load_audio_window(path: str, sample_rate: int = 24000, max_duration: float | None = 30, chunk_duration: float = 3.0,
                      random_offset: bool = False) -> np.ndarray:
    """One contiguous mono window, resampled to ``sample_rate`` and peak-normalised; empty on any error."""
    try:
        frames, sr0 = _read_window(path, max_duration, chunk_duration, random_offset)
        if frames.size == 0:
            return np.empty((0,), np.float32)
        y = frames.mean(axis=1).astype(np.float32, copy=False)
        if sr0 != sample_rate:
            y = fast_resample(y, sr0, sample_rate)
        peak = float(np.abs(y).max()) if y.size else 0.0
        if peak > 0.0:
            y = y / peak
        return y.astype(np.float32, copy=False)
    except Exception:
        return np.empty((0,), np.float32)


def split_audio_into_chunks(audio: np.ndarray, sample_rate: int = 24000, chunk_duration: float = 3.0,
                            chunk_overlap: float = 0.0) -> np.ndarray:
    """``[num_chunks, chunk_size]`` float32 chunks (see module docstring for the start positions)."""
    size = int(sample_rate * chunk_duration)
    if audio.size == 0 or size <= 0:
        return np.empty((0, max(size, 0)), np.float32)
    y = np.asarray(audio, np.float32).reshape(-1)
    n = y.shape[0]
    if n <= size:
        out = np.zeros((1, size), np.float32)
        out[0, :n] = y
        return out
    starts = list(range(0, n - size + 1, _step(sample_rate, chunk_duration, chunk_overlap)))
    if not starts or starts[-1] + size < n:
        starts.append(n - size)
    return np.stack([y[s : s + size] for s in starts]).astype(np.float32, copy=False)


def load_audio_file(path: str, sample_rate: int = 24000, max_duration: int = 30, chunk_duration: float = 3.0,
                    chunk_overlap: float = 0.0, random_offset: bool = False):
    """Load, resample, normalise and chunk a file; an unreadable file gives an empty list."""
    audio = load_audio_window(path, sample_rate=sample_rate, max_duration=max_duration, chunk_duration=chunk_duration,
                              random_offset=random_offset)
    if audio.size == 0:
        return []
    return split_audio_into_chunks(audio, sample_rate=sample_rate, chunk_duration=chunk_duration, chunk_overlap=chunk_overlap)


def save_wav(audio: np.ndarray, path: str, sample_rate: int = 24000, subtype: str = "PCM_16") -> None:
    """Write a mono WAV file (PCM_16 like soundfile's default for .wav, or FLOAT)."""
    x = np.asarray(audio, np.float32).reshape(-1)
    if subtype == "FLOAT":
        code, bits, payload = 3, 32, x.astype("<f4").tobytes()
    else:
        code, bits = 1, 16
        # libsndfile's float -> int16 write path scales by 0x7FFF and rounds to nearest (lrintf); samples inside [-1, 1] never clip
        payload = np.clip(np.rint(x * 32767.0), -32768, 32767).astype("<i2").tobytes()
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, code, 1, sample_rate,
                      sample_rate * bits // 8, bits // 8, bits, b"data", len(payload))
    with open(path, "wb") as fh:
        fh.write(hdr + payload)
