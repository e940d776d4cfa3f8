#!/usr/bin/env python3
"""Where the row-streaming depthwise kernel's (i8_dw_stream_kernel) wave time goes on configs[4] in INT8: in-kernel stamps per launch shape.

    make -C birdnet-stm32_amd/csrc stamps                  # builds lib/libbirdnet_hip_stamps.so (-DBN_TAIL_STAMPS), in the build container
    python tools/dw_stamps.py > profiles/r04_i8_dw_attribution.md          # on the GPU box

For every depthwise shape (C, H) of the network the stamped kernel records, for 4096 waves from the middle of the grid: start, constants and
first rows arrived, row walk done, pooling atomics done (s_memrealtime, 10 ns ticks).
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "birdnet-stm32_amd", "lib", "libbirdnet_hip_stamps.so")
os.environ["BIRDNET_HIP_LIB"] = LIB
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from birdnet_stm32 import _hip  # noqa: E402
from birdnet_stm32.conversion.export import convert_netspec_to_int8  # noqa: E402
from birdnet_stm32.models import build_model  # noqa: E402
from birdnet_stm32.models._lower_i8 import lower_i8  # noqa: E402
from birdnet_stm32.models._tflite_reader import parse_tflite  # noqa: E402
from birdnet_stm32.models._tflite_writer import write_tflite  # noqa: E402
from birdnet_stm32.models.runners import HipRunner  # noqa: E402

B = 1024
spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2, embeddings_size=256, num_classes=100,
                   audio_frontend="raw", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
rng = np.random.default_rng(0)
cal = [rng.standard_normal((1, 48000, 1)).astype(np.float32) for _ in range(8)]
cal = [c / (np.abs(c).max() + 1e-6) for c in cal]
model = parse_tflite(write_tflite(convert_netspec_to_int8(spec, lambda: ([c] for c in cal))))
r = HipRunner(lower_i8(model), max_batch=B)
lib = _hip.load_library()
lib.bn_debug_dw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
x = torch.randn((B, 48000), device="cuda")
x = x / x.abs().amax(dim=1, keepdim=True)
for _ in range(3):
    r.infer_audio_device(x)
torch.cuda.synchronize()
r.profile(True)
r.infer_audio_device(x)
torch.cuda.synchronize()
rows = [q for q in r.profile_collect() if q["launches"] and q["kind"] == "i8_dw"]
r.profile(False)
from birdnet_stm32.models import _pack as pk  # noqa: E402
ops = [op for op in r.plan.ops if op.kind == pk.I8_DW]
print("# `i8_dw_stream_kernel` on configs[4] in INT8 (raw frontend, 1024 chunks): where a wave's time goes (in-kernel stamps, `tools/dw_stamps.py`)\n")
print("| depthwise stage (H x W x C, stride) | launch ms | waves | rows per wave | prologue us (constants + first rows) | row walk us | per output row us | pooling atomics us | a wave's life us |")
print("|---|---|---|---|---|---|---|---|---|")
seen = set()
ms_by_name = {q["name"]: q["ms"] for q in rows}
for op in ops:
    H, Wd, C, sh = op.p[0], op.p[1], op.p[2], op.p[3]
    if (C, H) in seen:
        continue
    seen.add((C, H))
    buf = torch.zeros(4096 * 6, dtype=torch.int64, device="cuda")
    assert lib.bn_debug_dw_stamps(buf.data_ptr(), C, H) == 0
    r.infer_audio_device(x)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(4096, 6).astype(np.float64)
    ok = st[:, 3] > 0
    if not ok.any():
        continue
    t = st[ok, :4] * 0.01
    nrows = st[ok, 4]
    pro, walk, pool, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    print(f"| {H} x {Wd} x {C}, s{sh} | {ms_by_name.get(op.name, float('nan')):.3f} | {int(ok.sum())} | {nrows.mean():.0f} | {pro.mean():.2f} | {walk.mean():.2f} | "
          f"{(walk / nrows).mean():.3f} | {pool.mean():.2f} | {life.mean():.2f} |")
assert lib.bn_debug_dw_stamps(0, 0, 0) == 0
