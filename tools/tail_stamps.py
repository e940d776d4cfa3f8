#!/usr/bin/env python3
"""Where the fused tail kernel's wave cycles go: in-kernel time stamps per block and wave.

    make -C birdnet-stm32_amd/csrc stamps                  # builds lib/libbirdnet_hip_stamps.so (-DBN_TAIL_STAMPS), once, in the build container
    python tools/tail_stamps.py > profiles/r04_i8_tail_attribution.md      # on the GPU box

The stamps build records, for every wave of the first 8 workgroups and their first 4 chunk groups, per block: staging (issue of the
constant / weight copies), the wait at the staging barrier, the depthwise phase, the pointwise phase (MFMA + requantisation + ADD + store)
and the wait at the end barrier (s_memrealtime, 10 ns ticks).  The production library is not touched; the stamped kernel is ~1 % slower.
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

import bench  # noqa: E402
from birdnet_stm32 import _hip  # noqa: E402
from birdnet_stm32.models.runners import load_model_runner  # noqa: E402

B = 4096
dev = torch.device("cuda", 0)
r = load_model_runner(os.path.join(bench.PKG, "checkpoints", "birdnet_stm32n6_100.tflite"), max_batch=B)
lib = _hip.load_library()
lib.bn_debug_tail_stamps.argtypes = [ctypes.c_void_p]
x = bench.synth_audio_device(torch, B, 0, dev, 42)
out = torch.empty((B, 100), device=dev)
for _ in range(3):
    r.infer_audio_device(x, hop=bench.HOP, out=out)
torch.cuda.synchronize()
WG, GRP, BLK, WAVES = 8, 4, 8, 16
buf = torch.zeros(WG * GRP * BLK * WAVES * 6, dtype=torch.int64, device=dev)
assert lib.bn_debug_tail_stamps(buf.data_ptr()) == 0
r.profile(True)
r.infer_audio_device(x, hop=bench.HOP, out=out)
torch.cuda.synchronize()
tail_ms = [q["ms"] for q in r.profile_collect() if q["kind"] == "i8_tail" and q["launches"]]
st = buf.cpu().numpy().reshape(WG, GRP, BLK, WAVES, 6).astype(np.float64) * 0.01  # microseconds
names = ["stage3_ds1 (64->128, s2, taps from HBM)", "stage3_ds2 (128->128 + ADD)", "stage3_ds3 (128->128 + ADD)", "stage3_ds4 (128->128 + ADD)",
         "stage4_ds1 (128->256, s2)", "stage4_ds2 (256->256 + ADD)"]
print("# `i8_tail_kernel`: where a wave's time goes, per block (in-kernel stamps, `tools/tail_stamps.py`)\n")
print(f"INT8 B = {B}, the stamped build of the same sources; tail launch {tail_ms[0]:.3f} ms (production {0.43:.2f} ms).  Means over 8 workgroups x 4 chunk "
      "groups x 16 waves; a group = 4 chunks through 6 blocks + MEAN / FC / head.  `busy` = depthwise + pointwise phases; everything else a wave spends "
      "parked: issuing the staging copies and waiting for them at the staging barrier, and waiting at the end barrier for the slowest wave.\n")
print("| block | staging issue us | staging barrier us | depthwise us | pointwise + epilogue us | end barrier us | block us | busy % | slowest wave's busy us |")
print("|---|---|---|---|---|---|---|---|---|")
tot = np.zeros(6)
for li, nm in enumerate(names):
    m = st[:, :, li].reshape(-1, 6).mean(axis=0)
    busy = st[:, :, li, :, 2] + st[:, :, li, :, 3]
    print(f"| {nm} | {m[0]:.2f} | {m[1]:.2f} | {m[2]:.2f} | {m[3]:.2f} | {m[4]:.2f} | {m[5]:.2f} | {100 * (m[2] + m[3]) / m[5]:.0f} | {busy.max(axis=-1).mean():.2f} |")
    tot += m
print(f"| all six blocks | {tot[0]:.2f} | {tot[1]:.2f} | {tot[2]:.2f} | {tot[3]:.2f} | {tot[4]:.2f} | {tot[5]:.2f} | {100 * (tot[2] + tot[3]) / tot[5]:.0f} | |")
per_group = tail_ms[0] * 1e3 / (B / 4 / 256)
print(f"\nA chunk group takes {per_group:.1f} us of the launch ({B // 4} groups over 256 workgroups); the six blocks account for {tot[5]:.1f} us of it, the rest is "
      "MEAN + FULLY_CONNECTED + head and the group loop.")
print(f"\nShare of a wave's block time: busy {100 * (tot[2] + tot[3]) / tot[5]:.0f} %, staging (issue + barrier) {100 * (tot[0] + tot[1]) / tot[5]:.0f} %, "
      f"end barrier {100 * tot[4] / tot[5]:.0f} %.")
sp = st[:, :, 1:4, :, 2:4].sum(axis=-1)  # busy time per wave in the three residual blocks
print(f"\nSpread inside a workgroup (stage-3 residual blocks): fastest wave busy {sp.min(axis=-1).mean():.2f} us, slowest {sp.max(axis=-1).mean():.2f} us, "
      f"mean {sp.mean():.2f} us — all sixteen waves have the same instruction count (two tiles each); the spread is the order in which a SIMD's four waves get "
      "their issue slots, and it IS the end-barrier wait of the early ones.")
r.close()
