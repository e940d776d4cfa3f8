#!/usr/bin/env python3
"""Throughput of BASELINE configs[4]'s topology on one MI355X: raw-waveform learned filterbank + PCEN + alpha=1.5 DS-CNN with
squeeze-excite and inverted residuals (seeded random weights, 24 kHz x 2 s chunks as in the reference's deployment geometry).

    python tools/config5_bench.py [batch] [steps] [alpha] [frontend]

alpha / frontend default to 1.5 / raw (configs[4]); `1.0 hybrid` is the reference builder's default topology (hybrid frontend + PWL, 3 s chunks).

One step = per-chunk peak normalisation + the whole plan over `batch` waveform chunks resident in HBM.
"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "birdnet-stm32_amd"))
import torch
from birdnet_stm32.models import build_model
from birdnet_stm32.models._lower_f32 import lower_f32
from birdnet_stm32.models.runners import HipRunner

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
alpha = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
fe = sys.argv[4] if len(sys.argv) > 4 else "raw"
spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2 if fe == "raw" else 3, embeddings_size=256, num_classes=100,
                   audio_frontend=fe, mag_scale="pcen" if fe == "raw" else "pwl", alpha=alpha, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
r = HipRunner(lower_f32(spec), max_batch=B)
if os.environ.get("BN_OPTS"):  # launcher options for experiments: BN_OPTS="f32_tile_slice=12,f32_pwdw=1"
    from birdnet_stm32 import _hip
    for kv in os.environ["BN_OPTS"].split(","):
        k, v = kv.split("=")
        _hip.set_option(k, int(v))
x = torch.randn((B, 48000 if fe == "raw" else 72000), device="cuda")
for _ in range(2):
    r.infer_audio_device(x)
torch.cuda.synchronize()
r.profile(True)
r.infer_audio_device(x)
torch.cuda.synchronize()
rows = [q for q in r.profile_collect() if q["launches"]]
r.profile(False)
t0 = time.perf_counter()
for _ in range(steps):
    r.infer_audio_device(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
if os.environ.get("OPS"):
    for q in rows:
        print(f'{q["kind"]:12s} {q["name"]:28s} {q["ms"]:.3f} {q["p"][:16]}')
top = sorted(rows, key=lambda q: -q["ms"])[:6]
print(json.dumps({"workload": f"{fe} frontend + alpha={alpha} IR/SE DS-CNN, {2 if fe == 'raw' else 3} s @ 24 kHz, seeded weights" + (" (BASELINE configs[4])" if (alpha, fe) == (1.5, "raw") else ""), "batch": B, "ms_per_step": round(dt * 1e3, 3),
                  "chunks_per_s": round(B / dt, 1), "plan_ops": len(r.plan.ops), 
                  "slowest_ops": [{"kind": q["kind"], "name": q["name"], "ms": round(q["ms"], 3)} for q in top]}))
