#!/usr/bin/env python3
"""Where the guarded mel mixer's (i8_mel_mfma_kernel<true, 1>) wave time goes: in-kernel time stamps per phase (55 % of its
wave cycles parked).

    make -C birdnet-stm32_amd/csrc stamps                  # builds lib/libbirdnet_hip_stamps.so (-DBN_TAIL_STAMPS), in the build container
    python tools/mel_stamps.py > profiles/r04_i8_mel_attribution.md        # on the GPU box

The stamps build records, for every wave of 1024 workgroups from the middle of the grid (steady state: 65 536 workgroups, four resident per CU):
start, block quantised (its 20 float4 loads consumed, bytes in the LDS tile), barrier behind the tile left, float64 settle done, matrix phase done,
epilogue stores issued (s_memrealtime, 10 ns ticks).
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
lib.bn_debug_mel_stamps.argtypes = [ctypes.c_void_p]
x = bench.synth_audio_device(torch, B, 0, dev, 42)
out = torch.empty((B, 100), device=dev)
for _ in range(3):
    r.infer_audio_device(x, hop=bench.HOP, out=out)
torch.cuda.synchronize()
WG = 1024
buf = torch.zeros(WG * 4 * 8, dtype=torch.int64, device=dev)
assert lib.bn_debug_mel_stamps(buf.data_ptr()) == 0
r.profile(True)
r.infer_audio_device(x, hop=bench.HOP, out=out)
torch.cuda.synchronize()
mel_ms = [q["ms"] for q in r.profile_collect() if q["name"] == "t96" and q["launches"]]
st = buf.cpu().numpy().reshape(WG, 4, 8).astype(np.float64)
ok = st[:, :, 5] > 0
t = st[..., :6] * 0.01  # microseconds
d = np.diff(t, axis=-1)[ok]  # [n, 5]: load+quantise, barrier, settle, matrix, epilogue
life = (t[..., 5] - t[..., 0])[ok]
names = ["loads + quantise (20 float4 per lane -> bytes in the LDS tile)", "barrier behind the tile", "float64 settle (tables staged, flagged elements, 2 barriers)",
         "matrix phase (5 B fragments from L2, 20 MFMAs)", "epilogue (requantise, PWL table gathers, stores)"]
print("# `i8_mel_mfma_kernel<true, 1>`: where a wave's time goes (in-kernel stamps, `tools/mel_stamps.py`)\n")
print(f"INT8 B = {B}, the stamped build of the same sources; mixer launch {mel_ms[0]:.3f} ms.  {int(ok.sum())} waves of {WG} workgroups from the middle of the grid; "
      f"flagged elements per workgroup: mean {st[:, 0, 6].mean():.1f}, max {int(st[:, 0, 6].max())}.\n")
print("| phase | mean us | median us | p90 us | share of a wave's life % |")
print("|---|---|---|---|---|")
for i, nm in enumerate(names):
    print(f"| {nm} | {d[:, i].mean():.2f} | {np.median(d[:, i]):.2f} | {np.percentile(d[:, i], 90):.2f} | {100 * d[:, i].mean() / life.mean():.0f} |")
print(f"| a wave's life | {life.mean():.2f} | {np.median(life):.2f} | {np.percentile(life, 90):.2f} | 100 |")
n_tiles = B * 4
print(f"\n{n_tiles} workgroups over 256 CUs with four resident per CU: {n_tiles / 256 / 4:.0f} rounds x {life.mean():.1f} us = {n_tiles / 256 / 4 * life.mean() / 1e3:.3f} ms "
      f"(launch {mel_ms[0]:.3f} ms).")
