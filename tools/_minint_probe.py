import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tools")]
import torch
from conftest import TFLITE_PATH
from birdnet_stm32 import _hip
from birdnet_stm32.models.runners import load_model_runner
from signal_families import family_batch, FAMILIES
B = 4096
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(5)
runner = load_model_runner(TFLITE_PATH, max_batch=B)
for kind in (1, 3, 9, 0):
    x = family_batch(torch, kind, B, g, dev)
    for mi in (0, 1):
        with _hip.options(stft_minint=mi):
            for _ in range(3):
                s = runner.infer_audio_device(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                s = runner.infer_audio_device(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            st = runner.guard_stats(B)
        print(FAMILIES[kind], "minint", mi, "%.3f ms" % (dt * 1e3), st, flush=True)
