#!/usr/bin/env python3
"""Random FLAC streams from the test encoder (tests/flac_writer.py) through the decoder (csrc/host/bn_flac.c): samples, window reads,
MD5 / CRC checks.  CPU only.

    python tools/fuzz/flac_fuzz.py [n_streams] [seed]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
import numpy as np
import flac_writer as fw
from birdnet_stm32.audio import _flac

n_streams, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
bad = 0
for i in range(n_streams):
    ch = int(rng.choice([1, 2]))
    bps = int(rng.choice([8, 12, 16, 20, 24]))
    sr = int(rng.choice([8000, 16000, 22050, 24000, 44100, 48000, 96000]))
    n_frames = int(rng.integers(1, 7))
    sizes = [int(rng.choice([16, 192, 200, 256, 576, 1000, 1152, 2048, 4096, 4608])) for _ in range(n_frames)]
    n = sum(sizes)
    amp = (1 << (bps - 1)) - 1
    t = np.arange(n)
    x = np.stack([np.clip(0.3 * amp * np.sin(2 * np.pi * rng.uniform(50, 3000) * t / sr + c) + rng.normal(0, 0.02 * amp, n), -amp - 1, amp)
                  for c in range(ch)], axis=1).astype(np.int64)
    frames = []
    pos = 0
    for bs in sizes:
        mode = str(rng.choice(["indep", "ls", "sr", "ms"])) if ch == 2 else "indep"
        wasted = int(rng.choice([0, 0, 0, 2, 4])) if bps >= 12 and mode == "indep" else 0
        if wasted:
            x[pos : pos + bs] = (x[pos : pos + bs] >> wasted) << wasted
        subs = []
        for c in range(ch):
            kind = str(rng.choice(["fixed", "fixed", "lpc", "verbatim", "constant"]))
            if kind == "constant":
                if mode != "indep":
                    kind = "fixed"
                else:
                    x[pos : pos + bs, c] = x[pos, c]
            pos_orders = [po for po in range(0, 5) if bs % (1 << po) == 0 and (bs >> po) > 8]
            po = int(rng.choice(pos_orders)) if pos_orders else 0
            if kind == "fixed":
                subs.append(dict(kind="fixed", order=int(rng.integers(0, 5)), po=po, rice2=bool(rng.integers(2)), wasted=wasted))
            elif kind == "lpc":
                order = int(rng.integers(1, 9))
                prec = int(rng.integers(4, 13))
                shift = int(rng.integers(1, prec))
                coefs = [int(v) for v in rng.integers(-(1 << (prec - 2)), 1 << (prec - 2), order)]
                subs.append(dict(kind="lpc", lpc=(coefs, prec, shift), po=po, rice2=bool(rng.integers(2)), wasted=wasted))
            elif kind == "verbatim":
                subs.append(dict(kind="verbatim", wasted=wasted))
            else:
                subs.append(dict(kind="constant", wasted=wasted))
        frames.append({"n": bs, "mode": mode, "sub": subs})
        pos += bs
    try:
        raw = fw.encode(x, sr, bps, frames, with_md5=bool(rng.integers(2)), id3=bool(rng.integers(2)), total_known=bool(rng.integers(2)))
    except Exception as e:  # the test encoder could not express the case (e.g. residual overflow): not a decoder finding
        continue
    try:
        ints, sr2, bps2 = _flac.decode_flac(raw)
        ok = sr2 == sr and bps2 == bps and np.array_equal(np.asarray(ints).reshape(n, ch), x)
        a = int(rng.integers(0, n))
        cnt = int(rng.integers(0, n - a + 1))
        part = np.asarray(_flac.decode_flac(raw, a, cnt, verify_md5=False)[0]).reshape(-1, ch)
        ok = ok and np.array_equal(part, x[a : a + cnt])
    except Exception as e:  # noqa: BLE001
        print(i, "EXCEPTION", type(e).__name__, str(e)[:200], dict(ch=ch, bps=bps, sr=sr, sizes=sizes, frames=frames))
        bad += 1
        continue
    if not ok:
        print(i, "MISMATCH", dict(ch=ch, bps=bps, sr=sr, sizes=sizes, frames=frames))
        bad += 1
print("streams:", n_streams, "mismatches:", bad)
sys.exit(1 if bad else 0)
