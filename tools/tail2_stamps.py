#!/usr/bin/env python3
"""Where i8_tail2_kernel's wave time goes: in-kernel time stamps per block and wave (the measurement build of the same sources).

    make -C birdnet-stm32_amd/csrc stamps        # lib/libbirdnet_hip_stamps.so (-DBN_TAIL_STAMPS), in the build container
    python tools/tail2_stamps.py > profiles/r05_i8_tail2_attribution.md      # on the GPU box
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BIRDNET_HIP_LIB"] = os.path.join(REPO, "birdnet-stm32_amd", "lib", "libbirdnet_hip_stamps.so")
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from birdnet_stm32 import _hip  # noqa: E402
from birdnet_stm32.models.runners import load_model_runner  # noqa: E402

B = 4096
dev = torch.device("cuda", 0)
r = load_model_runner(os.path.join(bench.PKG, "checkpoints", "birdnet_stm32n6_100.tflite"), max_batch=B)
lib = _hip.load_library()
lib.bn_debug_tail2_stamps.argtypes = [ctypes.c_void_p]
x = bench.synth_audio_device(torch, B, 0, dev, 42)
out = torch.empty((B, 100), device=dev)
for _ in range(3):
    r.infer_audio_device(x, hop=bench.HOP, out=out)
torch.cuda.synchronize()
WG, GRP, BLK, WAVES = 8, 4, 8, 8
buf = torch.zeros(WG * GRP * BLK * WAVES * 8, dtype=torch.int64, device=dev)
assert lib.bn_debug_tail2_stamps(buf.data_ptr()) == 0
r.profile(True)
r.infer_audio_device(x, hop=bench.HOP, out=out)
torch.cuda.synchronize()
tail_ms = [q["ms"] for q in r.profile_collect() if q["kind"] == "i8_tail" and q["launches"]]
st = buf.cpu().numpy().reshape(WG, GRP, BLK, WAVES, 8).astype(np.float64) * 0.01  # microseconds
names = ["stage3_ds1 (64->128, s2, taps from HBM)", "stage3_ds2 (128->128 + ADD)", "stage3_ds3 (128->128 + ADD)", "stage3_ds4 (128->128 + ADD)",
         "stage4_ds1 (128->256, s2)", "stage4_ds2 (256->256 + ADD)", "head: MEAN (-> barrier), then FULLY_CONNECTED + scores (-> group's end barrier)"]
print("# `i8_tail2_kernel`: where a wave's time goes, per block (in-kernel stamps, `tools/tail2_stamps.py`)\n")
print(f"INT8 B = {B}, stamped build; tail launch {tail_ms[0]:.3f} ms.  Means over 8 workgroups x 4 chunk groups x 8 waves.\n")
print("| block | entry -> behind the middle barrier (depthwise phase) us | -> behind the end barrier (pointwise phase) us | block us |")
print("|---|---|---|---|")
tot = np.zeros(3)
for li, nm in enumerate(names):
    dw = (st[:, :, li, :, 4] - st[:, :, li, :, 0]).mean()
    pw = (st[:, :, li, :, 6] - st[:, :, li, :, 4]).mean()
    print(f"| {nm} | {dw:.2f} | {pw:.2f} | {dw + pw:.2f} |")
    tot += np.array([dw, pw, dw + pw])
print(f"| all six blocks + head | {tot[0]:.2f} | {tot[1]:.2f} | {tot[2]:.2f} |")
per_group = tail_ms[0] * 1e3 / (B / 4 / 256)
print(f"\nA chunk group takes {per_group:.1f} us of the launch; the six blocks account for {tot[2]:.1f} us of it, the rest is MEAN + FULLY_CONNECTED + head.")
# the group's full extent from the first block's entry to the next group's entry
g0 = st[:, :-1, 0, :, 0]
g1 = st[:, 1:, 0, :, 0]
print(f"\nEntry of a group's first block to the entry of the next group's: {(g1 - g0).mean():.2f} us.")
r.close()
