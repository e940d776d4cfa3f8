import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/birdnet-stm32_amd')
import torch, bench
from birdnet_stm32 import _hip
from birdnet_stm32.models.runners import load_model_runner
B = 4096
dev = torch.device('cuda', 0)
r = load_model_runner('/root/repo/birdnet-stm32_amd/checkpoints/birdnet_stm32n6_100.tflite', max_batch=B)
pool = [bench.synth_audio_device(torch, B, k * B, dev, 42 + k) for k in range(4)]
out = torch.empty((B, 100), device=dev)
ref = None
for dual in (0, 1, 0, 1):
    with _hip.options(dual_stream=dual):
        for k in range(3):
            r.infer_audio_device(pool[k % 4], hop=281, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            r.infer_audio_device(pool[k % 4], hop=281, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        res = r.infer_audio_device(pool[0], hop=281).clone()
        torch.cuda.synchronize()
        if ref is None: ref = res
        print('dual', dual, 'ms/step', round(dt * 1e3, 4), 'chunks/s', round(B / dt), 'same', bool(torch.equal(res, ref)))
