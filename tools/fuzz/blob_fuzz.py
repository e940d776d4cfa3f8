#!/usr/bin/env python3
"""bn_blob_check (host only, no GPU) on mutated plans: random byte / word flips in the header, slot, tensor and operator records and
truncations must be answered with accept / BN_ERR_FORMAT, never with a crash.  CPU only; prints how many mutations were refused.

    python tools/fuzz/blob_fuzz.py [n_mutations] [seed]
"""
import os, struct, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import numpy as np
from birdnet_stm32 import _hip
from birdnet_stm32.models import _pack as pk
from birdnet_stm32.models.runners import lower_model_file

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ck = os.path.join(REPO, "birdnet-stm32_amd", "checkpoints")
blobs = [pk.pack_plan(lower_model_file(os.path.join(ck, f), keep_all=ka)) for f in ("birdnet_stm32n6_100.tflite", "birdnet_stm32n6_100.keras") for ka in (False, True)]
refused = accepted = 0
for i in range(n):
    base = blobs[i % len(blobs)]
    hdr = struct.unpack_from("<8s14I", base, 0)
    ops_off, n_ops = hdr[13], hdr[10]
    meta_end = ops_off + n_ops * (16 + 4 * (pk.OP_NP + pk.OP_NT + pk.OP_NF))
    b = bytearray(base)
    kind = int(rng.integers(0, 5))
    if kind == 0:    # one random byte in the records
        b[int(rng.integers(8, meta_end))] = int(rng.integers(0, 256))
    elif kind == 1:  # one random 32-bit word replaced by an extreme value
        at = int(rng.integers(2, meta_end // 4)) * 4
        struct.pack_into("<I", b, at, int(rng.choice([0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 65536, 1 << 20])))
    elif kind == 2:  # several random words
        for _ in range(int(rng.integers(2, 9))):
            at = int(rng.integers(2, meta_end // 4)) * 4
            struct.pack_into("<I", b, at, int(rng.integers(0, 1 << 32)))
    elif kind == 3:  # truncation
        b = b[: int(rng.integers(0, len(b)))]
    else:            # a word incremented / decremented
        at = int(rng.integers(2, meta_end // 4)) * 4
        v = struct.unpack_from("<i", b, at)[0] + int(rng.choice([-1, 1, -4, 4, 16]))
        struct.pack_into("<i", b, at, max(-(1 << 31), min((1 << 31) - 1, v)))
    try:
        _hip.blob_check(bytes(b))
        accepted += 1
    except _hip.HipError:
        refused += 1
print("mutations:", n, "refused:", refused, "accepted:", accepted)
