"""FLAC decoding for the ingest side, through the plain-C decoder ``csrc/host/bn_flac.c`` (``lib/libbn_host.so``).

The reference reads FLAC (like every other container) through libsndfile (reference: birdnet_stm32/audio/io.py:90,114-116);
``soundfile`` is not on the MI355X image.  ``read_flac_window`` returns the samples the way ``soundfile.read(dtype='float32',
always_2d=True)`` does — integers scaled by ``2^-(bits-1)`` — plus the raw integers for the device ingest, which uploads PCM as
int16 / int32 and does the arithmetic on the GPU.  Whenever a WHOLE stream is decoded (``decode_flac`` from frame 0 to the end: what the
length pass of ``load_audio_window`` / ``read_pcm_window`` does once per file that does not state its length, and any caller that asks for
everything) the audio is checked against the stream's MD5 when the file carries one; a window read of a longer file is covered by the
per-frame CRC-16 only, like libsndfile's.
"""

from __future__ import annotations

import ctypes
import hashlib
import os

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "lib", "libbn_host.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: run `make -C birdnet-stm32_amd/csrc` (or __graft_entry__.build())")
        lib = ctypes.CDLL(_LIB_PATH)
        lib.bn_flac_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)]
        lib.bn_flac_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        lib.bn_flac_decode.restype = ctypes.c_int64
        lib.bn_flac_md5.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        _lib = lib
    return _lib


_ERRORS = {-1: "malformed FLAC stream", -2: "FLAC frame checksum mismatch", -3: "FLAC feature not supported (negative LPC shift / 33-bit side channel)",
           -4: "out of memory"}


def flac_info(raw: bytes) -> tuple[int, int, int, int]:
    """(sample rate, channels, bits per sample, total inter-channel frames; 0 = unknown) of a FLAC byte string."""
    sr, ch, bps, total = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    rc = _load().bn_flac_info(raw, len(raw), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bps), ctypes.byref(total))
    if rc:
        raise ValueError(_ERRORS.get(rc, f"FLAC error {rc}"))
    return sr.value, ch.value, bps.value, int(total.value)


def decode_flac(raw: bytes, first: int = 0, count: int | None = None, verify_md5: bool = True) -> tuple[np.ndarray, int, int]:
    """Decode frames ``[first, first + count)``: ``(int32 [n, channels], sample rate, bits per sample)``."""
    sr, ch, bps, total = flac_info(raw)
    # no more frames than the bytes can hold: a frame is at least 11 bytes and carries at most 65 535 samples per channel (a forged
    # STREAMINFO total or an absurd `count` must not size the output buffer)
    ceiling = (len(raw) // 11 + 1) * 65535
    if count is not None:
        want = min(int(count), ceiling)
    elif total:
        want = min(max(total - first, 0), ceiling)
    else:  # the stream does not state its length (total = 0): count by decoding once without storing — no bound follows from the file size
        want = _load().bn_flac_decode(raw, len(raw), int(first), (1 << 62), None)  # (a constant subframe holds 65 535 samples in a few bytes)
        if want < 0:
            raise ValueError(_ERRORS.get(int(want), f"FLAC error {want}"))
    out = np.empty((max(int(want), 0), ch), np.int32)
    n = _load().bn_flac_decode(raw, len(raw), int(first), int(out.shape[0]), out.ctypes.data_as(ctypes.c_void_p))
    if n < 0:
        raise ValueError(_ERRORS.get(int(n), f"FLAC error {n}"))
    out = out[:n]
    whole = first == 0 and (n == total if total else count is None)
    if verify_md5 and whole:
        md5 = _stream_md5(raw)
        if md5 != b"\x00" * 16:
            width = (bps + 7) // 8
            le = out.astype("<i4").view(np.uint8).reshape(n, ch, 4)[:, :, :width]
            if hashlib.md5(np.ascontiguousarray(le).tobytes()).digest() != md5:
                raise ValueError("FLAC: decoded audio does not match the stream's MD5")
    return out, sr, bps


def _stream_md5(raw: bytes) -> bytes:
    """STREAMINFO's MD5 as the C parser locates it (behind an ID3v2 tag, at the marker it accepted — not the first ``fLaC`` byte pattern)."""
    out = ctypes.create_string_buffer(16)
    rc = _load().bn_flac_md5(raw, len(raw), out)
    if rc:
        raise ValueError(_ERRORS.get(rc, f"FLAC error {rc}"))
    return out.raw


def read_flac_window(raw: bytes, first: int, count: int) -> tuple[np.ndarray, np.ndarray, int, int]:
    """(float32 frames [n, ch] with libsndfile's integer scaling, the raw int32 frames, sample rate, bits per sample)."""
    ints, sr, bps = decode_flac(raw, first, count, verify_md5=False)
    return (ints.astype(np.float64) / float(1 << (bps - 1))).astype(np.float32), ints, sr, bps
