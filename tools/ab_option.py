"""A/B of one launcher option inside ONE process (box-to-box differences of 3-7 % hide anything smaller): the shipped INT8 path from
audio, 4096 chunks per step, the values of the option interleaved over six rounds of 20 steps; the scores must be bit-identical.

    python tools/ab_option.py <option> <value,value,...> [f32]       e.g.  python tools/ab_option.py i8_tail 1,0

(`f32` as third argument: the float32 path, 1024 chunks per step.)
"""
import os, sys, time, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd")]
import torch, bench
from birdnet_stm32 import _hip
from birdnet_stm32.models.runners import load_model_runner
dev = torch.device("cuda", 0)
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
NB = 1024 if f32 else 4096
r = load_model_runner(os.path.join(bench.PKG, "checkpoints", "birdnet_stm32n6_100" + (".keras" if f32 else ".tflite")), device=0, max_batch=NB)
x = bench.synth_audio_device(torch, NB, 0, dev, 42)
out = torch.empty((NB, r.num_classes), dtype=torch.float32, device=dev)
name, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
res = {v: [] for v in vals}
ref = None
for rep in range(6):
    for v in vals:
        with _hip.options(**{name: v}):
            for _ in range(3): r.infer_audio_device(x, hop=bench.HOP, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): r.infer_audio_device(x, hop=bench.HOP, out=out)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / 20 * 1e3)
            if ref is None: ref = out.clone()
            assert torch.equal(out, ref), (name, v)
for v in vals: print(name, v, "ms/step median %.4f min %.4f" % (float(np.median(res[v])), min(res[v])))
