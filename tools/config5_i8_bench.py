#!/usr/bin/env python3
"""Throughput of BASELINE configs[4]'s backbone in INT8 on one MI355X: alpha = 1.5 DS-CNN with squeeze-excite and inverted residuals
and PCEN magnitude scaling (seeded random weights), quantised by this build's own exporter (conversion/export.py).  Frontend:

    hybrid (default)  the hybrid frontend of current reference code (per-sample max normalisation: REDUCE_MAX -> ADD -> DIV), 3 s @ 24 kHz,
                      spectrogram geometry of the shipped model; one step = STFT + the whole INT8 plan (bn_infer_audio)
    raw               configs[4]'s own frontend: learned 1 x 16 filterbank on the peak-normalised waveform (2 s @ 24 kHz, the geometry the
                      reference's raw frontend builds at); one step = peak normalisation + the whole INT8 plan

    python tools/config5_i8_bench.py [batch] [steps] [hybrid|raw] [alpha]

"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "birdnet-stm32_amd"))
import numpy as np
import torch
from birdnet_stm32.conversion.export import convert_netspec_to_int8
from birdnet_stm32.models import build_model
from birdnet_stm32.models._lower_i8 import lower_i8
from birdnet_stm32.models._tflite_reader import parse_tflite
from birdnet_stm32.models._tflite_writer import write_tflite
from birdnet_stm32.models.runners import HipRunner

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fe = sys.argv[3] if len(sys.argv) > 3 else "hybrid"
alpha = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
raw = fe == "raw"
spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2 if raw else 3, embeddings_size=256, num_classes=100,
                   audio_frontend=fe, mag_scale="pcen", alpha=alpha, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
rng = np.random.default_rng(0)
if raw:
    cal = [rng.standard_normal((1, 48000, 1)).astype(np.float32) for _ in range(8)]
    cal = [c / (np.abs(c).max() + 1e-6) for c in cal]
    model = parse_tflite(write_tflite(convert_netspec_to_int8(spec, lambda: ([c] for c in cal))))
else:
    cal = [rng.random((1, 257, 256, 1), dtype=np.float32) ** 4 for _ in range(8)]
    spec.frontend.attrs["norm"] = True
    model = parse_tflite(write_tflite(convert_netspec_to_int8(spec, lambda: ([c] for c in cal), frontend_norm=True)))
r = HipRunner(lower_i8(model), max_batch=B)
if os.environ.get("BN_OPTS"):  # launcher options for experiments: BN_OPTS="i8_pwdw=0"
    from birdnet_stm32 import _hip
    for kv in os.environ["BN_OPTS"].split(","):
        k, v = kv.split("=")
        _hip.set_option(k, int(v))
x = torch.randn((B, 48000 if raw else 72000), device="cuda")
x = x / x.abs().amax(dim=1, keepdim=True)
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
by_kind = {}
for q in rows:
    by_kind[q["kind"]] = by_kind.get(q["kind"], 0.0) + q["ms"]
top = sorted(rows, key=lambda q: -q["ms"])[:6]
print(json.dumps({"workload": f"alpha={alpha} IR/SE DS-CNN + PCEN in INT8" + (" (configs[4])" if alpha == 1.5 else "") + ", " + ("raw learned-filterbank frontend, 2 s @ 24 kHz" if raw else
                              "hybrid frontend with per-sample max normalisation, 3 s @ 24 kHz") + ", seeded weights, own PTQ",
                  "batch": B, "ms_per_step": round(dt * 1e3, 3), "chunks_per_s": round(B / dt, 1), "tflite_ops": len(model.ops), "plan_ops": len(r.plan.ops),
                  "ms_by_kernel_kind": {k: round(v, 3) for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1])},
                  "slowest_ops": [{"kind": q["kind"], "name": q["name"], "ms": round(q["ms"], 3)} for q in top]}))
