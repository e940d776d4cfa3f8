"""Which part of a first `evaluate` call is slow: the process (first use of everything) or the files (first read)?  Two data sets, A then B then A."""
import os, sys, time, shutil, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tools")]
import torch
import evaluate_bench as eb
from birdnet_stm32.evaluation.metrics import evaluate
from birdnet_stm32.models.runners import load_model_runner
from birdnet_stm32.training.config import ModelConfig
ck = os.path.join(REPO, "birdnet-stm32_amd", "checkpoints", "birdnet_stm32n6_100")
cfg = ModelConfig.load(ck + "_model_config.json").to_dict()
classes = cfg["class_names"]
runner = load_model_runner(ck + ".tflite", max_batch=4096)
sets = {}
for name in ("A", "B"):
    root = f"/dev/shm/bn_cold_{name}"
    shutil.rmtree(root, ignore_errors=True)
    sets[name], _ = eb.write_dataset(root, 512, 30.0, 2, 24000, classes[:8], torch)
for name in ("A", "B", "A", "B"):
    st = {}
    t0 = time.perf_counter()
    evaluate(runner, sets[name], classes, cfg, pooling="avg", stats=st)
    print(name, "wall %.3f" % (time.perf_counter() - t0), "read_s %.3f" % st["read_s"], "metrics_s %.4f" % st.get("metrics_s", -1), st["read_s_per_group"][:8], flush=True)
for name in ("A", "B"):
    shutil.rmtree(f"/dev/shm/bn_cold_{name}", ignore_errors=True)
