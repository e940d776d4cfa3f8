"""Soak of the INT8 exactness pass: the quantised input bytes of the guarded fast path (stft_exact = 2: float32 STFT + float64 pass over the
elements in doubt) against the all-float64 STFT (stft_exact = 1, itself checked against the oracle by tests/test_gpu_sweeps.py) on many batches of
random signals from a dozen families with random parameters — generated on the device, so the CPU oracle's speed does not limit the count.

    python tools/exact_soak.py [batches] [chunks per batch] [seed] [guard] [audit]
        # guard: 0 empirical bound (default), 1 proven worst-case bound (option stft_guard); audit: 1 = option stft_audit (the near misses of the
        # bound are re-evaluated too and the wrong ones counted).  Prints one summary line; exit code 1 on any differing byte or score or audit violation
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
from conftest import TFLITE_PATH  # noqa: E402

from birdnet_stm32 import _hip  # noqa: E402
from birdnet_stm32.models.runners import load_model_runner  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T, sr = 72000, 24000
dev = torch.device("cuda")
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1234
guard = int(sys.argv[4]) if len(sys.argv) > 4 else 0
audit = int(sys.argv[5]) if len(sys.argv) > 5 else 0
g = torch.Generator(device=dev).manual_seed(seed)
sys.path.insert(0, os.path.join(REPO, "tools"))
from signal_families import family_batch  # noqa: E402


def batch(kind: int):
    return family_batch(torch, kind, B, g, dev)


runner = load_model_runner(TFLITE_PATH, max_batch=B)
bad_bytes = bad_scores = total = listed = whole = audited = violations = interval = 0
whole_by_family = {}
for i in range(n_batches):
    x = batch(i % 12)
    with _hip.options(stft_exact=1):
        s1 = runner.infer_audio_device(x).clone()
        q1 = torch.from_numpy(runner.input_bytes(B))
    with _hip.options(stft_guard=guard, stft_audit=audit):
        s2 = runner.infer_audio_device(x)
        q2 = torch.from_numpy(runner.input_bytes(B))
        st = runner.guard_stats(B)
    audited += st["audited"]
    violations += st["audit_violations"]
    d = int((q1 != q2).sum())
    bad_bytes += d
    bad_scores += int((s1 != s2).any(dim=1).sum())
    total += B
    listed += st["listed"]
    whole += st["whole_minmax"] + st["whole_fix"]
    interval += st.get("interval_min", 0)
    f = whole_by_family.setdefault(i % 12, [0, 0, 0, 0])
    f[0] += B; f[1] += st["whole_minmax"]; f[2] += st["whole_fix"]; f[3] += st.get("interval_min", 0)
    if d:
        print(f"batch {i} (family {i % 12}): {d} bytes differ", flush=True)
    if i % 60 == 59:   # (a long soak shows it is alive: the GPU runner takes several silent minutes for a hang)
        print(f"... {total} chunks, {bad_bytes} differing bytes so far", flush=True)
print(f"exactness soak (seed {seed}, {'proven' if guard == 1 else 'empirical'} bound{', audit on' if audit else ''}): {total} chunks of 12 signal families, {bad_bytes} differing input bytes, {bad_scores} chunks with differing scores; "
      f"{listed / (total * 257 * 256):.2e} of the elements re-evaluated in float64, {whole} chunks as whole float64 spectrograms"
      + f", {interval} chunks with the minimum as an interval"
      + (f"; audit: {audited} near misses re-evaluated, {violations} violations of the bound" if audit else ""))
print("per family (chunks, whole float64 behind min/max, behind the mixer, interval minimum):", {k: tuple(v) for k, v in sorted(whole_by_family.items())})
sys.exit(1 if bad_bytes or bad_scores or violations else 0)
