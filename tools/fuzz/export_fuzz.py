#!/usr/bin/env python3
"""Random exported INT8 topologies on the GPU against the INT8 oracle (production plan, scores): a one-off fuzzing aid.

    python tools/fuzz/export_fuzz.py [n_configs] [seed]
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
import numpy as np
from test_conversion import _export
from birdnet_stm32.models._lower_i8 import lower_i8
from birdnet_stm32.models.runners import HipRunner
from oracle.int8_graph import Int8Interpreter

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
bad = 0
for i in range(n):
    kw = dict(num_mels=int(rng.choice([16, 32, 48, 64])), spec_width=int(rng.choice([64, 128, 192, 256])), alpha=float(rng.choice([0.5, 0.75, 1.0, 1.25, 1.5])),
              use_se=bool(rng.integers(2)), use_inverted_residual=bool(rng.integers(2)), mag_scale=str(rng.choice(["pwl", "pcen", "none"])),
              frontend_norm=bool(rng.integers(2)), num_classes=int(rng.integers(3, 60)), class_activation=str(rng.choice(["sigmoid", "softmax"])),
              chunk_duration=int(rng.choice([2, 3])), depth_multiplier=int(rng.choice([1, 1, 2])))
    try:
        spec, model, _, x = _export(kw, n_cal=2, seed=i)
        ref = Int8Interpreter(model).invoke(x)
        r = HipRunner(lower_i8(model), max_batch=x.shape[0])
        got = r.predict(x)
        r.close()
        ok = np.allclose(got, ref, atol=1e-6) if kw["class_activation"] == "softmax" else np.array_equal(got, ref)
    except NotImplementedError as e:
        print(i, "not lowered:", str(e)[:100], kw)
        continue
    print(i, "ok" if ok else "MISMATCH", kw, flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
