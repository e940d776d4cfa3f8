"""Per-operator times of the INT8 path under one option's values (HIP events around every operator, warm)."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import torch, bench
from birdnet_stm32 import _hip
from birdnet_stm32.models.runners import load_model_runner
dev = torch.device("cuda", 0)
NB = 4096
r = load_model_runner(os.path.join(bench.PKG, "checkpoints", "birdnet_stm32n6_100.tflite"), device=0, max_batch=NB)
x = bench.synth_audio_device(torch, NB, 0, dev, 42)
out = torch.empty((NB, r.num_classes), dtype=torch.float32, device=dev)
name, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
for rep in range(2):
    for v in vals:
        with _hip.options(**{name: v}):
            for _ in range(3): r.infer_audio_device(x, hop=bench.HOP, out=out)
            r.profile(True)
            acc = {}
            for _ in range(10):
                r.infer_audio_device(x, hop=bench.HOP, out=out)
                torch.cuda.synchronize()
                for q in r.profile_collect():
                    if q["launches"]: acc[q["name"] + ":" + q["kind"]] = acc.get(q["name"] + ":" + q["kind"], 0.0) + q["ms"] / 10
            r.profile(False)
        print(name, v, {k: round(t, 4) for k, t in acc.items()}, flush=True)
