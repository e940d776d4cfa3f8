#!/usr/bin/env python3
"""End-to-end throughput of ``evaluate``: files on tmpfs -> metrics, with the per-stage split.

    python tools/evaluate_bench.py [--files 4096] [--seconds 30] [--channels 2] [--sr 24000] [--dtype i8] [--repeats 3] [--serial]

Writes ``--files`` synthetic PCM16 WAVs (tone + noise, distinct per file, generated on the GPU) under ``--dir`` (default
``/dev/shm/bn_eval_bench``; removed afterwards unless ``--keep``), then times ``birdnet_stm32.evaluation.metrics.evaluate`` —
the function behind ``python -m birdnet_stm32 evaluate`` (reference flow: birdnet_stm32/evaluation/metrics.py:117-153,
audio/io.py:63-130,177-213) — from the file list to the metrics dict.  Beside the wall time it reports

* the pipeline's stages (each one's busy time; they overlap): header probing, reading into pinned slabs, H2D, ingest kernels,
  inference kernels, pooling, metric computation (sklearn);
* H2D GB/s against the box's page-locked copy rate measured here (1 GiB, best of 5) — PCIe is this stage's roof;
* the two bounds: PCIe (copy rate / PCM bytes per chunk) and kernels (chunks / (ingest + inference busy time)), and the end-to-end
  rate as a fraction of the smaller one;
* ``--serial``: the same call through the serial device pipeline of rounds 1-3 (A/B), and that both give identical scores.

One JSON object on stdout (the last line); progress on stderr.
"""

from __future__ import annotations

import argparse
import json
import os
import shutil
import struct
import sys
import time
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "birdnet-stm32_amd"))

import numpy as np  # noqa: E402


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def mem_available_bytes() -> int:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def write_dataset(root: str, n_files: int, seconds: float, channels: int, sr: int, classes: list[str], torch) -> tuple[list[str], float]:
    """``n_files`` PCM16 WAVs under ``root/<class>/``; returns (paths in evaluate's order, seconds spent)."""
    t0 = time.perf_counter()
    n = int(sr * seconds)
    dev = torch.device("cuda", 0)
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + n * channels * 2, b"WAVE", b"fmt ", 16, 1, channels, sr, sr * channels * 2, channels * 2, 16,
                      b"data", n * channels * 2)
    for c in classes:
        os.makedirs(os.path.join(root, c), exist_ok=True)
    t = torch.arange(n, device=dev, dtype=torch.float32) / sr
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    paths = [os.path.join(root, classes[i % len(classes)], f"f{i:05d}.wav") for i in range(n_files)]

    def dump(args):
        path, arr = args
        with open(path, "wb") as fh:
            fh.write(hdr)
            fh.write(memoryview(arr))

    block = 64
    with ThreadPoolExecutor(max_workers=8) as pool:
        for b0 in range(0, n_files, block):
            nb = min(block, n_files - b0)
            f = 500.0 + 37.0 * ((torch.arange(b0, b0 + nb, device=dev) % 200).float())
            tone = torch.sin(2 * torch.pi * f[:, None] * t[None, :])                       # [nb, n]
            x = 0.3 * torch.randn((nb, n, channels), device=dev, generator=gen) + tone[:, :, None]
            x = x / x.abs().amax(dim=(1, 2), keepdim=True)
            pcm = torch.clamp(torch.round(x * 30000.0), -32768, 32767).to(torch.int16).cpu().numpy()
            list(pool.map(dump, [(paths[b0 + j], pcm[j]) for j in range(nb)]))
    return paths, time.perf_counter() - t0


def pinned_copy_rate(torch, nbytes: int = 1 << 30) -> float:
    """Best-of-5 GB/s of one ``nbytes`` page-locked host -> device copy (events on the stream)."""
    src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    src.fill_(1)
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    best = 0.0
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.copy_(src, non_blocking=True)
        b.record()
        b.synchronize()
        best = max(best, nbytes / 1e9 / (a.elapsed_time(b) / 1e3))
    return best


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--sr", type=int, default=24000)
    ap.add_argument("--dtype", choices=["i8", "f32"], default="i8")
    ap.add_argument("--dir", default="/dev/shm/bn_eval_bench")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--serial", action="store_true", help="also time the serial device pipeline of rounds 1-3 (on at most --serial_files files)")
    ap.add_argument("--serial_files", type=int, default=512)
    ap.add_argument("--max_batch", type=int, default=4096)
    ap.add_argument("--slab_mb", type=int, default=256)
    ap.add_argument("--no_ramp", action="store_true", help="every group a full slab (A/B of the small first groups)")
    ap.add_argument("--readers", type=int, default=0)
    ap.add_argument("--no_prepare", action="store_true", help="load the runner without prepare_pipeline (the first call then page-locks its slabs itself)")
    ap.add_argument("--read_mode", choices=["mmap", "pread"], default=None, help="A/B of the reader's two ways out of the page cache")
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--latency", action="store_true", help="one more run with measure_latency (slices of --batch_size)")
    args = ap.parse_args(argv)

    import torch

    from birdnet_stm32.evaluation.metrics import evaluate
    from birdnet_stm32.models.runners import load_model_runner
    from birdnet_stm32.training.config import ModelConfig

    ckpt = os.path.join(REPO, "birdnet-stm32_amd", "checkpoints", "birdnet_stm32n6_100")
    cfg = ModelConfig.load(ckpt + "_model_config.json").to_dict()
    cfg.update(sample_rate=24000, hop_length=281)
    classes = cfg["class_names"]
    per_file = int(args.sr * args.seconds) * args.channels * 2 + 44
    want = args.files * per_file
    free = min(shutil.disk_usage(os.path.dirname(args.dir.rstrip("/")) or "/").free, mem_available_bytes() or (1 << 62))
    n_files = args.files
    if want > free // 2:  # never fill tmpfs (it is RAM): at most half of what is free
        n_files = max(64, int(free // 2 // per_file))
        log(f"only {free / 1e9:.1f} GB free: {n_files} files instead of {args.files}")
    if os.path.isdir(args.dir):
        shutil.rmtree(args.dir)
    out: dict = {"files": n_files, "seconds": args.seconds, "channels": args.channels, "native_rate": args.sr, "dtype": args.dtype,
                 "bytes_per_file": per_file, "host_cpus": len(os.sched_getaffinity(0))}
    try:
        paths, gen_s = write_dataset(args.dir, n_files, args.seconds, args.channels, args.sr, classes[:8], torch)
        out["dataset_gb"] = round(n_files * per_file / 1e9, 3)
        out["dataset_write_s"] = round(gen_s, 2)
        log(f"dataset: {n_files} files, {out['dataset_gb']} GB in {gen_s:.1f} s")
        rate = pinned_copy_rate(torch)
        out["pinned_copy_gbps"] = round(rate, 2)
        log(f"pinned H2D copy: {rate:.1f} GB/s")
        # (as the CLI creates it: `python -m birdnet_stm32 evaluate` passes prepare_pipeline=True; --no_prepare times the bare runner)
        runner = load_model_runner(ckpt + (".tflite" if args.dtype == "i8" else ".keras"), max_batch=args.max_batch, prepare_pipeline=not args.no_prepare)
        opts = {"slab_bytes": args.slab_mb << 20}
        if args.no_ramp:
            opts["ramp"] = ()
        if args.readers:
            opts["readers"] = args.readers
        if args.read_mode:
            opts["read_mode"] = args.read_mode
        runs = []
        ys_ref = None
        for r in range(args.repeats):
            st: dict = {}
            t0 = time.perf_counter()
            m, pf, yt, ys = evaluate(runner, paths, classes, cfg, pooling="avg", batch_size=args.batch_size, stats=st, pipeline_options=opts)
            wall = time.perf_counter() - t0
            st = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in st.items()}
            st["evaluate_wall_s"] = round(wall, 4)
            st["files_per_s"] = round(len(pf) / wall, 1)
            st["chunks_per_s"] = round(st["chunks"] / wall, 1)
            runs.append(st)
            log(f"run {r}: {wall:.3f} s, {st['chunks_per_s']:.0f} chunks/s, stages {st}")
            if ys_ref is None:
                ys_ref = ys
            else:
                assert np.array_equal(ys, ys_ref), "two runs of the pipeline disagree"
        best = max(runs[1:] or runs, key=lambda s: s["chunks_per_s"])  # the first run pays for the pinned allocations
        out["cold_run"] = runs[0]
        out["best_warm_run"] = best
        bytes_per_chunk = best["h2d_bytes"] / max(best["chunks"], 1)
        pcie_bound = rate * 1e9 / bytes_per_chunk
        kernel_bound = best["chunks"] / max(best["ingest_s"] + best["infer_s"], 1e-9)
        out.update(metric="evaluate end to end, files on tmpfs -> metrics", value=best["chunks_per_s"], unit="chunks/s",
                   files_per_s=best["files_per_s"], h2d_gbps=round(best["h2d_gbps"], 2),
                   h2d_frac_of_pinned_copy=round(best["h2d_gbps"] / rate, 3), pcm_bytes_per_chunk=round(bytes_per_chunk, 1),
                   pcie_bound_chunks_per_s=round(pcie_bound, 1), kernel_bound_chunks_per_s=round(kernel_bound, 1),
                   frac_of_min_bound=round(best["chunks_per_s"] / min(pcie_bound, kernel_bound), 3),
                   roc_auc=m.get("roc-auc"))
        if args.latency:
            st = {}
            t0 = time.perf_counter()
            m2, *_ = evaluate(runner, paths, classes, cfg, pooling="avg", batch_size=args.batch_size, measure_latency=True, stats=st, pipeline_options=opts)
            out["latency_run"] = {"wall_s": round(time.perf_counter() - t0, 3), "batch_size": args.batch_size,
                                  **{k: round(v, 5) for k, v in m2.items() if k.startswith("latency")}}
        if args.serial:
            sub = paths[: min(args.serial_files, len(paths))]
            t0 = time.perf_counter()
            _, _, _, ys_s = evaluate(runner, sub, classes, cfg, pooling="avg", batch_size=args.batch_size, pipelined=False)
            ws = time.perf_counter() - t0
            t0 = time.perf_counter()
            _, _, _, ys_p = evaluate(runner, sub, classes, cfg, pooling="avg", batch_size=args.batch_size, pipeline_options=opts)
            wp = time.perf_counter() - t0
            n_chunks = int(len(sub) * best["chunks"] / max(best["files"], 1))
            out["serial_ab"] = {"files": len(sub), "serial_wall_s": round(ws, 3), "pipelined_wall_s": round(wp, 3),
                                "serial_chunks_per_s": round(n_chunks / ws, 1), "pipelined_chunks_per_s": round(n_chunks / wp, 1),
                                "scores_identical": bool(np.array_equal(ys_s, ys_p))}
            log("serial A/B:", out["serial_ab"])
        runner.close()
    finally:
        if not args.keep and os.path.isdir(args.dir):
            shutil.rmtree(args.dir, ignore_errors=True)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
