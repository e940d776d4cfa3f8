#!/usr/bin/env python3
"""Per-operator times of the configs[4] INT8 backbone (see config5_i8_bench.py): kind, shape, ms, algorithmic GB/s."""
import os, sys
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
spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=100,
                   audio_frontend="hybrid", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
spec.frontend.attrs["norm"] = True
rng = np.random.default_rng(0)
cal = [rng.random((1, 257, 256, 1), dtype=np.float32) ** 4 for _ in range(8)]
model = parse_tflite(write_tflite(convert_netspec_to_int8(spec, lambda: ([c] for c in cal), frontend_norm=True)))
r = HipRunner(lower_i8(model), max_batch=B)
x = torch.randn((B, 72000), device="cuda")
x = x / x.abs().amax(dim=1, keepdim=True)
for _ in range(2):
    r.infer_audio_device(x)
torch.cuda.synchronize()
r.profile(True)
for _ in range(3):
    r.infer_audio_device(x)
torch.cuda.synchronize()
for q in r.profile_collect():
    if not q["launches"]:
        continue
    p = q["p"]
    ms = q["ms"] / q["launches"]
    if q["kind"] == "i8_dwpw":
        desc = f"H{p[0]} W{p[1]} Cin{p[2]} s{p[3]} -> OH{p[6]} OW{p[7]} Cout{p[14]} dw{p[29]} add{p[18]} strip{p[35]}"
        gb = B * (p[0] * p[1] * p[2] + p[6] * p[7] * p[14]) / ms / 1e6
    elif q["kind"] in ("i8_dw", "i8_stem"):
        desc = f"H{p[0]} W{p[1]} C{p[2]} s{p[3]} -> OH{p[6]} OW{p[7]}"
        gb = B * (p[0] * p[1] * (p[2] if q["kind"] == "i8_dw" else 1) + p[6] * p[7] * p[2]) / ms / 1e6
    elif q["kind"] == "i8_scale":
        desc = f"P{p[0]} C{p[1]}"
        gb = B * 2 * p[0] * p[1] / ms / 1e6
    else:
        desc, gb = str(p[:4]), 0.0
    print(f"{q['kind']:10s} {q['name']:6s} {ms:7.3f} ms  {gb:8.1f} GB/s  {desc}")
