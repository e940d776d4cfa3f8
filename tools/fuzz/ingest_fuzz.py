#!/usr/bin/env python3
"""Random ingest cases (sample rate, channels, length, target rate, chunk length / overlap) on the GPU against oracle/ingest.py
(which is pinned to numpy + scipy): chunks bit for bit.  A one-off fuzzing aid.

    python tools/fuzz/ingest_fuzz.py [n_cases] [seed]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
import numpy as np
from birdnet_stm32 import _hip
from birdnet_stm32.audio import ingest
from oracle import ingest as oi

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = _hip.Context(0, 64)
bad = 0
for i in range(n):
    sr_in = int(rng.choice([8000, 11025, 12000, 16000, 22050, 24000, 32000, 37800, 44100, 48000, 88200, 96000]))
    sr_out = int(rng.choice([24000, 22050, 16000, 32000]))
    ch = int(rng.choice([1, 1, 2, 2, 3, 5]))
    cd = float(rng.choice([1.0, 2.0, 3.0]))
    overlap = float(rng.choice([0.0, 0.5, 1.0])) if cd > 1.0 else 0.0
    lengths = [int(sr_in * rng.uniform(0.05, 7.0)) for _ in range(int(rng.integers(1, 4)))]
    pcm = [np.clip(np.rint(9000.0 * rng.standard_normal((m, ch))), -32768, 32767).astype(np.int16) for m in lengths]
    try:
        wins = [ingest.window_from_int16(p, sr_in) for p in pcm]
        chunks, counts = ingest.ingest_windows_device(ctx, wins, sr_out, cd, overlap)[:2]
        chunks = chunks.cpu().numpy()
        at, ok = 0, True
        for k, p in enumerate(pcm):
            y = oi.ingest_window(p.astype(np.float32) / 32768.0, sr_in, sr_out)
            want = oi.split_chunks(y, sr_out, cd, overlap)
            got = chunks[at : at + counts[k]]
            at += counts[k]
            ok = ok and counts[k] == want.shape[0] and got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        ok = ok and at == chunks.shape[0]
    except Exception as e:  # noqa: BLE001
        print(i, "EXCEPTION", type(e).__name__, str(e)[:160], (sr_in, sr_out, ch, cd, overlap, lengths))
        bad += 1
        continue
    print(i, "ok" if ok else "MISMATCH", (sr_in, sr_out, ch, cd, overlap, lengths), flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
