#!/usr/bin/env python3
"""Random float32 topologies on the GPU against the float oracle (production plan, scores): a one-off fuzzing aid.

    python tools/fuzz/f32_fuzz.py [n_configs] [seed]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
import numpy as np
from conftest import synth_chunks
from birdnet_stm32.models import build_model
from birdnet_stm32.models._lower_f32 import lower_f32
from birdnet_stm32.models.runners import HipRunner
from oracle import float_graph, stft

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
bad = 0
for i in range(n):
    kw = dict(num_mels=int(rng.choice([16, 32, 40, 48, 64])), spec_width=int(rng.choice([64, 128, 192, 256, 320])),
              alpha=float(rng.choice([0.5, 0.75, 1.0, 1.25, 1.5])), use_se=bool(rng.integers(2)), use_inverted_residual=bool(rng.integers(2)),
              use_attention_pooling=bool(rng.integers(2)), mag_scale=str(rng.choice(["pwl", "pcen", "none", "db"])),
              num_classes=int(rng.integers(3, 60)), class_activation=str(rng.choice(["sigmoid", "softmax"])), chunk_duration=int(rng.choice([2, 3])),
              depth_multiplier=int(rng.choice([1, 1, 2])), embeddings_size=int(rng.choice([128, 256])))
    norm = bool(rng.integers(2))
    try:
        spec = build_model("dscnn", sample_rate=24000, randomize_bn=True, seed=100 + i, **kw)
        spec.frontend.attrs["norm"] = norm
        chunks = synth_chunks(4, sr=24000, seconds=kw["chunk_duration"], seed=i)
        x = np.stack([stft.hybrid_spectrogram(a, spec_width=kw["spec_width"]) for a in chunks])[..., None].astype(np.float32)
        ref = float_graph.forward(spec, x, np.float64)
        r = HipRunner(lower_f32(spec), max_batch=4)
        got = r.predict(x)
        r.close()
        err = float(np.abs(got - ref).max())
        ok = err < 2e-4
    except NotImplementedError as e:
        print(i, "not lowered:", str(e)[:100], kw)
        continue
    print(i, "ok" if ok else "MISMATCH", f"{err:.2e}", "norm" if norm else "", kw, flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
