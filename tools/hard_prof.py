import os, sys, torch
REPO = "/root/repo"
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tools")]
import bench
from signal_families import family_batch
from birdnet_stm32.models.runners import load_model_runner
dev = torch.device("cuda", 0)
r = load_model_runner(os.path.join(bench.PKG, "checkpoints", "birdnet_stm32n6_100.tflite"), max_batch=4096)
g = torch.Generator(device=dev).manual_seed(1234)
out = torch.empty((4096, 100), device=dev)
for kind in (0, 9):
    x = family_batch(torch, kind, 4096, g, dev)
    for _ in range(2): r.infer_audio_device(x, hop=bench.HOP, out=out)
    torch.cuda.synchronize()
    r.profile(True)
    for _ in range(3): r.infer_audio_device(x, hop=bench.HOP, out=out)
    torch.cuda.synchronize()
    rows = r.profile_collect(); r.profile(False)
    print("family", kind, r.guard_stats(4096))
    for q in rows:
        if q["launches"]: print("  %-10s %-22s %.4f ms" % (q["kind"], q["name"], q["ms"] / q["launches"] * (1 if q["launches"] <= 3 else q["launches"] / 3)))
