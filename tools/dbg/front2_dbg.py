import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
from birdnet_stm32 import _hip
from birdnet_stm32.models.runners import load_model_runner
KERAS = os.path.join(REPO, "birdnet-stm32_amd", "checkpoints", "birdnet_stm32n6_100.keras")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
runner = load_model_runner(KERAS, max_batch=B)
rng = np.random.default_rng(33)
spec = torch.from_numpy(rng.random((B, 257 * 256), dtype=np.float32)).cuda()
with _hip.options(f32_front2=0):
    base = runner.predict_device(spec, return_logits=True)[1].clone()
for rep in range(3):
    got = runner.predict_device(spec, return_logits=True)[1].clone()
    d = (got - base).abs().amax(dim=1).cpu().numpy()
    bad = np.nonzero(d > 0)[0]
    print("rep", rep, "bad chunks", len(bad), "of", B, "max diff", d.max(), "first bad", bad[:20], "last bad", bad[-5:] if len(bad) else [])
