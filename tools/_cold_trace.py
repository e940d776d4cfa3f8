"""Timeline of a COLD `evaluate` call (first call of a fresh process on freshly written files): BN_PIPELINE_TRACE marks of audio/pipeline.py.
    python tools/_cold_trace.py [--files 1024] [--read_mode mmap|pread]"""
import argparse, json, os, shutil, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "birdnet-stm32_amd"), os.path.join(REPO, "tools")]
os.environ["BN_PIPELINE_TRACE"] = "1"
ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=1024)
ap.add_argument("--read_mode", default=None)
ap.add_argument("--readers", type=int, default=0)
ap.add_argument("--cpu_rank", action="store_true", help="ranking metrics sorted on the host, no warm-up thread (is the sort module's load the pause?)")
ap.add_argument("--quiet", action="store_true")
args = ap.parse_args()
import torch
import evaluate_bench as eb
from birdnet_stm32.evaluation.metrics import evaluate
from birdnet_stm32.models.runners import load_model_runner
from birdnet_stm32.training.config import ModelConfig
ck = os.path.join(REPO, "birdnet-stm32_amd", "checkpoints", "birdnet_stm32n6_100")
cfg = ModelConfig.load(ck + "_model_config.json").to_dict()
classes = cfg["class_names"]
root = "/dev/shm/bn_cold_trace"
shutil.rmtree(root, ignore_errors=True)
paths, _ = eb.write_dataset(root, args.files, 30.0, 2, 24000, classes[:8], torch)
eb.pinned_copy_rate(torch)
runner = load_model_runner(ck + ".tflite", max_batch=4096)
if args.cpu_rank:
    import birdnet_stm32.evaluation.metrics as _m
    import birdnet_stm32.evaluation._ranking as _r
    _orig = _r.descending_orders
    _r.descending_orders = lambda s, ctx=None: _orig(s, None)
opts = {}
if args.read_mode:
    opts["read_mode"] = args.read_mode
if args.readers:
    opts["readers"] = args.readers
for call in range(2):
    st = {}
    t0 = time.perf_counter()
    evaluate(runner, paths, classes, cfg, pooling="avg", stats=st, pipeline_options=opts)
    wall = time.perf_counter() - t0
    print(f"call {call}: evaluate {wall:.4f} s  pipeline {st['wall_s']:.4f}  read_s {st['read_s']:.4f}  ingest_s {st['ingest_s']:.4f}  infer_s {st['infer_s']:.4f}  metrics_s {st.get('metrics_s', -1):.4f}  chunks/s {st['chunks'] / wall:.0f}")
    last = {}
    for what, t, th in ([] if args.quiet else st["trace"]):
        print(f"   {t * 1e3:9.2f} ms  (+{(t - last.get(th, 0.0)) * 1e3:8.2f})  [{th[:18]:18s}] {what}")
        last[th] = t
shutil.rmtree(root, ignore_errors=True)
