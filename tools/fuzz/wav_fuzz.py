#!/usr/bin/env python3
"""Random PCM WAV files written by Python's ``wave`` module (an independent RIFF writer), optionally with extra chunks spliced in
front of ``data``, through this build's WAV reader (birdnet_stm32/audio/io.py): window reads against the samples.  CPU only.

    python tools/fuzz/wav_fuzz.py [n_files] [seed]
"""
import io, os, struct, sys, tempfile, wave
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import numpy as np
from birdnet_stm32.audio import io as aio

n_files, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
bad = 0
tmp = tempfile.mkdtemp()
for i in range(n_files):
    width = int(rng.choice([1, 2, 3, 4]))
    ch = int(rng.choice([1, 2, 3, 6]))
    sr = int(rng.choice([8000, 16000, 22050, 24000, 44100, 48000]))
    n = int(rng.integers(0, 3 * sr))
    if width == 1:
        ints = rng.integers(0, 256, (n, ch))
        raw = ints.astype(np.uint8).tobytes()
        ref = (ints.astype(np.float64) - 128.0) / 128.0
    else:
        lim = 1 << (8 * width - 1)
        ints = rng.integers(-lim, lim, (n, ch))
        raw = ints.astype("<i4").view(np.uint8).reshape(n, ch, 4)[:, :, :width].tobytes() if width == 3 else ints.astype({2: "<i2", 4: "<i4"}[width]).tobytes()
        ref = ints.astype(np.float64) / float(lim)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(ch)
        w.setsampwidth(width)
        w.setframerate(sr)
        w.writeframes(raw)
    data = bytearray(buf.getvalue())
    if rng.integers(2):  # splice a LIST chunk (odd length, padded) between fmt and data
        payload = b"INFO" + bytes(rng.integers(0, 255, int(rng.integers(1, 40))).astype(np.uint8))
        chunk = b"LIST" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
        at = data.index(b"data")
        data[at:at] = chunk
        struct.pack_into("<I", data, 4, len(data) - 8)
    path = os.path.join(tmp, f"f{i}.wav")
    open(path, "wb").write(bytes(data))
    try:
        code, ch2, sr2, bits, offset, nbytes = aio._wav_layout(bytes(data))
        first = int(rng.integers(0, n + 1))
        count = int(rng.integers(0, n - first + 1))
        got = aio._decode_frames(bytes(data), code, ch2, bits, offset, nbytes, first, count)
        want = ref[first : first + count].astype(np.float32)
        ok = sr2 == sr and ch2 == ch and got.shape == want.shape and np.array_equal(got, want)
    except Exception as e:  # noqa: BLE001
        print(i, "EXCEPTION", type(e).__name__, str(e)[:160], dict(width=width, ch=ch, sr=sr, n=n))
        bad += 1
        continue
    if not ok:
        print(i, "MISMATCH", dict(width=width, ch=ch, sr=sr, n=n, first=first, count=count), float(np.abs(got - want).max()) if got.shape == want.shape and got.size else None)
        bad += 1
print("files:", n_files, "mismatches:", bad)
sys.exit(1 if bad else 0)
