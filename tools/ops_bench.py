import sys, json
sys.path.insert(0, "birdnet-stm32_amd")
import torch, numpy as np
from birdnet_stm32.models.runners import load_model_runner
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if dt == "f32" else 4096)
path = "birdnet-stm32_amd/checkpoints/birdnet_stm32n6_100." + ("keras" if dt == "f32" else "tflite")
r = load_model_runner(path, max_batch=B)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((B, 72000), device="cuda", generator=g)
x = x / x.abs().amax(dim=1, keepdim=True)
for _ in range(3): r.infer_audio_device(x)
torch.cuda.synchronize()
r.profile(True)
for _ in range(20): r.infer_audio_device(x)
torch.cuda.synchronize()
rows = r.profile_collect()
tot = 0
for row in rows:
    if row["launches"]:
        ms = row["ms"] / row["launches"]; tot += ms
        print(f'{row["kind"]:14s} {row["name"]:28s} {ms:.4f}')
print("sum", round(tot, 4))
# whole-step time with and without the per-operator event pairs
import time
for prof in (False, True):
    r.profile(prof)
    for _ in range(3): r.infer_audio_device(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): r.infer_audio_device(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50 * 1e3
    if prof: r.profile_collect()
    print("step ms", "with events" if prof else "plain", round(dt, 4))
