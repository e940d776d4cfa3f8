"""cProfile of a WARM `evaluate` call (third call of the process): what the host does outside the pipeline's wall time."""
import cProfile, io, os, pstats, shutil, sys, time
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
root = "/dev/shm/bn_warm_prof"
shutil.rmtree(root, ignore_errors=True)
paths, _ = eb.write_dataset(root, 1024, 30.0, 2, 24000, classes[:8], torch)
runner = load_model_runner(ck + ".tflite", max_batch=4096)
for _ in range(3):
    st = {}
    t0 = time.perf_counter()
    evaluate(runner, paths, classes, cfg, pooling="avg", stats=st)
    print("evaluate %.4f  pipeline %.4f  metrics %.4f  pool %.4f probe %.4f" % (time.perf_counter() - t0, st["wall_s"], st["metrics_s"], st["pool_s"], st["probe_s"]))
pr = cProfile.Profile()
pr.enable()
evaluate(runner, paths, classes, cfg, pooling="avg")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
shutil.rmtree(root, ignore_errors=True)
