#!/usr/bin/env python3
"""Throughput benchmark of the per-chunk inference hot path on MI355X.

One *step* = one pass of the whole path over one batch of synthetic 3 s @ 24 kHz chunks that are
already resident in HBM:  windowed STFT magnitude -> hybrid mel mixer -> PWL -> DS-CNN -> scores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|i8] [--batch B]

* N = 1 (default): BASELINE.json configs[1] — the shipped birdnet_stm32n6_100 float32 DS-CNN with the
  hybrid+pwl frontend at batch 1024 (``--dtype i8 --batch 4096`` gives configs[2]).
* N > 1: launched by ``python -m torch.distributed.run --nproc-per-node N``; every rank runs the same
  per-GPU batch on its own shard of the chunk stream (weak scaling, no data-path collective) and the
  job ends with the single RCCL all-gather of the scores named by the north star, inside the timed region.

Rank 0 prints ONE JSON line: the throughput contract fields plus ``roofline`` (dominant kernel — named by the
fully profiled warm-up steps, then timed with HIP events on the launch stream during the timed region, where it is the
only bracketed operator; ``stages`` are the warm-up measurements of the other operators) and ``cpu_baseline`` (the numpy oracle of
``oracle/`` timed on this host on a bounded sample — a reported baseline, not a target).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "birdnet-stm32_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

SR, SECONDS, W, NFFT = 24000, 3.0, 256, 512
T = int(SR * SECONDS)
HOP = T // W

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 / 16x16x4 = vector FP32 rate
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 = 2x bf16


def synth_audio_device(torch, batch: int, rank: int, device):
    """peaknorm(0.3 N(0,1) + sin(2 pi f_b t)), f_b = 500 + 37 (g mod 200), g = global chunk index."""
    gen = torch.Generator(device=device)
    gen.manual_seed(42 + rank)
    t = torch.arange(T, device=device, dtype=torch.float64) / SR
    g = torch.arange(batch, device=device, dtype=torch.float64) + rank * batch
    f = 500.0 + 37.0 * torch.remainder(g, 200.0)
    tone = torch.sin(2.0 * np.pi * f[:, None] * t[None, :]).to(torch.float32)
    x = 0.3 * torch.randn((batch, T), generator=gen, device=device, dtype=torch.float32) + tone
    x = x / x.abs().amax(dim=1, keepdim=True)
    return x.contiguous()


def algorithmic_work(row: dict, batch: int, dtype: str) -> tuple[float, float]:
    """(bytes, ops) one launch of plan operator ``row`` must move / execute for ``batch`` chunks."""
    k, p = row["kind"], row["p"]
    e = 4 if dtype == "f32" else 1
    if k == "stft512":
        return batch * (T * 4 + (NFFT // 2 + 1) * W * 4), batch * 3.35e6
    if k == "f32_stftmel":
        return batch * (T * 4 + p[2] * p[1] * 4), batch * 3.35e6
    if k == "f32_melfin":
        return batch * 2.0 * p[0] * p[1] * 4, batch * 10.0 * p[0] * p[1]
    if k == "f32_front":
        macs = 9 * p[0] * (p[1] // 2) * p[2] + p[4] * p[5] * p[2] * (9 + p[3])
        return batch * 4.0 * (p[0] * p[1] + p[4] * p[5] * p[3]), batch * 2.0 * macs
    if k == "i8_front":
        macs = 9 * p[0] * (p[1] // 2) * p[2] + p[4] * p[5] * p[2] * (9 + p[3])
        return batch * 1.0 * (p[0] * p[1] + p[4] * p[5] * p[3]), batch * 2.0 * macs
    if k == "f32_mel":
        return batch * (p[0] * p[1] * 4 + p[2] * p[1] * 4), batch * 2.0 * p[0] * p[1] * p[2]
    if k == "i8_quant":
        return batch * (p[0] * p[1] * 4 + p[1] * p[2]), batch * 2.0 * p[0] * p[1]
    if k == "i8_mel":
        return batch * (p[0] * p[1] + p[2] * p[0]), batch * 2.0 * p[0] * p[1] * p[2]
    if k in ("f32_stem", "i8_stem"):
        return batch * (p[0] * p[1] * e + p[6] * p[7] * p[2] * e), batch * 2.0 * 9 * p[6] * p[7] * p[2]
    if k in ("f32_dw", "i8_dw"):
        return batch * (p[0] * p[1] * p[2] * e + p[6] * p[7] * p[2] * e), batch * 2.0 * 9 * p[6] * p[7] * p[2]
    if k == "f32_dwpw":
        n_in, n_out = p[0] * p[1] * p[2], p[6] * p[7] * p[10]
        macs = p[6] * p[7] * p[2] * (p[10] + (9 if p[15] else 0))
        return batch * 4.0 * (n_in + n_out * (2 if p[12] else 1)), batch * 2.0 * macs
    if k == "i8_dwpw" and p[36]:  # mel mixer with QUANTIZE fused into its load: float32 spectrogram in, int8 [M][W] out
        return batch * (p[5] * p[1] * 4.0 + p[14] * p[1]), batch * 2.0 * p[1] * p[2] * p[14]
    if k == "i8_dwpw":
        n_in, n_out = p[0] * p[1] * p[2], p[6] * p[7] * p[14]
        macs = p[6] * p[7] * p[2] * (p[14] + (9 if p[29] else 0))
        return batch * 1.0 * (n_in + n_out * (2 if p[18] else 1)), batch * 2.0 * macs
    if k == "f32_pw":
        return batch * (p[0] * p[1] * 4 + p[0] * p[2] * 4 * (2 if p[4] else 1)), batch * 2.0 * p[0] * p[1] * p[2]
    if k == "i8_pw":
        return batch * (p[0] * p[1] + p[0] * p[2] * (2 if p[6] else 1)), batch * 2.0 * p[0] * p[1] * p[2]
    if k in ("f32_gap", "i8_mean"):
        return batch * (p[0] * p[1] * e + p[1] * e), batch * 1.0 * p[0] * p[1]
    if k == "f32_gapdense":
        return batch * (p[0] * p[1] * 4 + p[2] * 4), batch * (1.0 * p[0] * p[1] + 2.0 * p[1] * p[2])
    if k in ("f32_dense", "i8_fc"):
        return batch * (p[0] * e + p[1] * 4), batch * 2.0 * p[0] * p[1]
    return 0.0, 0.0


def output_bytes(kind: str, p: list, batch: int, dtype: str):
    """Bytes one launch of the operator writes (what the PMC WRITE_SIZE of that launch shows), or None if not modelled."""
    e = 4 if dtype == "f32" else 1
    if kind == "stft512":
        return batch * (NFFT // 2 + 1) * W * 4
    if kind == "f32_stftmel":
        return batch * p[2] * p[1] * 4
    if kind in ("f32_front", "i8_front"):
        return batch * p[4] * p[5] * p[3] * e
    if kind == "f32_dwpw":
        return batch * p[6] * p[7] * p[10] * 4
    if kind == "i8_dwpw":
        return batch * p[6] * p[7] * p[14]
    return None


def kernel_symbol(kind: str, p: list) -> str:
    """The device kernel an operator launches (the launchers' choices: strip kernels for the wide early blocks)."""
    if kind in ("stft512", "f32_stftmel"):
        return "stft512_mag_kernel"
    if kind == "i8_dwpw" and p[35]:
        return "i8_strip_kernel"
    if kind == "i8_dwpw" and p[30] and p[14] == 64:
        return "i8_mel_mfma_kernel"
    if kind == "i8_front" and p[16]:
        return "i8_front_strip_kernel"
    if kind == "f32_dwpw" and p[15]:
        cin, cout, ow, stride = p[2], p[10], p[7], p[3]
        if ow % 16 == 0 and stride in (1, 2) and ((cin == 32 and cout in (32, 64)) or (cin == 64 and cout in (64, 128)) or (cin, cout) == (128, 128)):
            return "f32_strip_kernel"
        if cin <= 64 and cout <= 64:
            return "f32_dwpw_wave_kernel"
    return kind + "_kernel"


def pmc_traffic(dom: dict, batch: int, dtype: str):
    """HBM bytes per launch of the dominant kernel from the committed PMC digest of this workload (rocprofv3 --pmc
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950;
    produced by tools/profile_digest.py).  Counters cannot be read from inside the benchmark, so this is null when no
    digest of the same workload (dtype, batch, kernel, output size) is on disk."""
    path = os.path.join(REPO, "profiles", f"r01_{dtype}_b{batch}_traffic.json")
    want = output_bytes(dom["kernel"], dom["p"], batch, dtype)
    if want is None or not os.path.isfile(path):
        return None
    base = kernel_symbol(dom["kernel"], dom["p"])
    best = None
    for row in json.load(open(path)):
        name = row["kernel"].split("<")[0]
        if name == base and abs(row["write_bytes"] - want) <= 0.05 * want:
            if best is None or abs(row["write_bytes"] - want) < abs(best["write_bytes"] - want):
                best = row
    return None if best is None else int(best["read_bytes"] + best["write_bytes"])


def roofline_of(rows: list[dict], batch: int, dtype: str, dom_op: int = -1) -> tuple[dict, list[dict]]:
    stages = []
    for r in rows:
        if not r["launches"]:
            continue
        avg_ms = r["ms"] / r["launches"]
        nbytes, nops = algorithmic_work(r, batch, dtype)
        stages.append({"kernel": r["kind"], "layer": r["name"], "avg_ms": round(avg_ms, 4), "GBps": round(nbytes / avg_ms / 1e6, 1),
                       "Tops": round(nops / avg_ms / 1e9, 2), "bytes": nbytes, "ops": nops, "p": r["p"], "op": r["op"]})
    picked = [s for s in stages if s["op"] == dom_op]
    dom = picked[0] if picked else max(stages, key=lambda s: s["avg_ms"])
    peak_compute = F32_MFMA_PEAK_TFLOPS if dtype == "f32" else I8_MFMA_PEAK_TOPS
    ridge = peak_compute * 1e12 / (HBM_PEAK_GBS * 1e9)
    intensity = dom["ops"] / max(dom["bytes"], 1.0)
    if intensity > ridge and dom["kernel"].endswith("pw"):
        roof = {"bound": "mfma", "achieved": dom["Tops"], "peak": peak_compute, "unit": "TFLOP/s" if dtype == "f32" else "TOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["traffic"] = pmc_traffic(dom, batch, dtype)
    roof["kernel"] = kernel_symbol(dom["kernel"], dom["p"])
    roof["layer"] = dom["layer"]
    roof["avg_launch_ms"] = dom["avg_ms"]
    roof["algorithmic_bytes_per_launch"] = dom["bytes"]
    for s in stages:
        s.pop("bytes"), s.pop("ops"), s.pop("p"), s.pop("op")
    return roof, stages


def cpu_baseline(dtype: str, seconds_budget: float = 15.0) -> dict:
    """Time the CPU restatement of the reference path (``oracle/``) on this host, on a bounded sample.

    float32: the plain-C + OpenMP port (``oracle/c/oracle_cpu.c``) on all host threads when it has been built, otherwise the
    numpy oracle on one thread.  INT8: the C + OpenMP port of the TFLite int8 reference kernels (``oracle/c/oracle_i8.c``) when
    built, otherwise the numpy interpreter on one thread.
    """
    import contextlib

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    from oracle import cport, float_graph, stft
    from oracle.int8_graph import Int8Interpreter

    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._tflite_reader import load_tflite

    ckpt = os.path.join(PKG, "checkpoints", "birdnet_stm32n6_100")
    rng = np.random.default_rng(42)
    t = np.arange(T) / SR

    def chunks(n):
        x = 0.3 * rng.standard_normal((n, T)) + np.sin(2 * np.pi * (500 + 37 * (np.arange(n) % 200))[:, None] * t[None, :])
        return (x / np.abs(x).max(axis=1, keepdims=True)).astype(np.float32)

    if dtype == "f32" and os.path.isfile(cport.CPU_LIB):
        path = cport.CpuFloatPath(load_keras_archive(ckpt + ".keras"))
        x = chunks(256)
        path(x[:32])  # warm up the OpenMP pool
        t0 = time.perf_counter()
        path(x)
        per = (time.perf_counter() - t0) / 256
        n = int(max(256, min(16384, seconds_budget / per // 256 * 256)))
        reps, done, t0 = n // 256, 0, time.perf_counter()
        for _ in range(reps):
            path(x)
            done += 256
        dt = time.perf_counter() - t0
        return {"value": round(done / dt, 1), "unit": "chunks/s", "cores": path.threads, "kind": "port",
                "sample": f"{done} synthetic 3 s @ 24 kHz chunks, plain-C + OpenMP port of the float path (oracle/c/oracle_cpu.c), "
                          f"{path.threads} threads, {dt:.1f} s"}

    if dtype == "i8" and os.path.isfile(cport.I8_LIB) and os.path.isfile(cport.CPU_LIB):
        path = cport.CpuInt8Path(load_tflite(ckpt + ".tflite"))
        x = chunks(256)
        path.invoke(path.spectrogram(x[:32], HOP, W))  # warm up the OpenMP pool
        t0 = time.perf_counter()
        path.invoke(path.spectrogram(x, HOP, W))
        per = (time.perf_counter() - t0) / 256
        n = int(max(256, min(16384, seconds_budget / per // 256 * 256)))
        reps, done, t0 = n // 256, 0, time.perf_counter()
        for _ in range(reps):
            path.invoke(path.spectrogram(x, HOP, W))
            done += 256
        dt = time.perf_counter() - t0
        return {"value": round(done / dt, 1), "unit": "chunks/s", "cores": path.threads, "kind": "port",
                "sample": f"{done} synthetic 3 s @ 24 kHz chunks, plain-C + OpenMP port of the TFLite int8 reference kernels "
                          f"(oracle/c/oracle_i8.c under the numpy interpreter's graph walk) + C STFT, {path.threads} threads, {dt:.1f} s"}

    if dtype == "f32":
        spec = load_keras_archive(ckpt + ".keras")
        run = lambda S: float_graph.forward(spec, S, np.float32)  # noqa: E731
    else:
        interp = Int8Interpreter(load_tflite(ckpt + ".tflite"))
        run = lambda S: interp.invoke(S)  # noqa: E731

    def work(xs):
        t0 = time.perf_counter()
        for i in range(0, len(xs), 64):
            S = np.stack([stft.hybrid_spectrogram(a, NFFT, W) for a in xs[i : i + 64]])[..., None]
            run(S)
        return time.perf_counter() - t0

    with threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext():
        per = work(chunks(16)) / 16
        n = int(max(64, min(4096, seconds_budget / per // 64 * 64)))
        dt = work(chunks(n))
    return {"value": round(n / dt, 2), "unit": "chunks/s", "cores": 1, "kind": "port",
            "sample": f"{n} synthetic 3 s @ 24 kHz chunks, numpy oracle (oracle/stft.py + "
                      f"{'float_graph' if dtype == 'f32' else 'int8_graph'}.py), 1 thread, {dt:.1f} s"}


def quick_rate(torch, dtype: str, batch: int, device, local_rank: int, steps: int = 10) -> dict:
    """Whole-path throughput of another BASELINE configuration on this GPU (same synthetic audio, resident in HBM), reported
    beside the main measurement: two warm-up steps, then `steps` timed steps between device synchronisations."""
    from birdnet_stm32.models.runners import load_model_runner

    ckpt = os.path.join(PKG, "checkpoints", "birdnet_stm32n6_100" + (".keras" if dtype == "f32" else ".tflite"))
    runner = load_model_runner(ckpt, device=local_rank, max_batch=batch)
    audio = synth_audio_device(torch, batch, 0, device)
    scores = torch.empty((batch, runner.num_classes), dtype=torch.float32, device=device)
    for _ in range(2):
        runner.infer_audio_device(audio, hop=HOP, out=scores)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.infer_audio_device(audio, hop=HOP, out=scores)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    runner.close()
    del audio, scores
    return {"workload": ("birdnet_stm32n6_100 float32 DS-CNN" if dtype == "f32" else "birdnet_stm32n6_100 INT8 DS-CNN") + ", hybrid+pwl frontend",
            "dtype": dtype, "batch_per_gpu": batch, "value": round(batch * steps / dt, 1), "unit": "chunks/s", "steps": steps,
            "ms_per_step": round(dt / steps * 1e3, 4)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["f32", "i8"], default="f32")
    ap.add_argument("--batch", type=int, default=0, help="chunks per GPU per step (default 1024 f32, 4096 i8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from birdnet_stm32.models.runners import load_model_runner

    batch = args.batch or (1024 if args.dtype == "f32" else 4096)
    ckpt = os.path.join(PKG, "checkpoints", "birdnet_stm32n6_100" + (".keras" if args.dtype == "f32" else ".tflite"))
    runner = load_model_runner(ckpt, device=local_rank, max_batch=batch)
    audio = synth_audio_device(torch, batch, rank, device)
    scores = torch.empty((batch, runner.num_classes), dtype=torch.float32, device=device)
    gathered = torch.empty((world * batch, runner.num_classes), dtype=torch.float32, device=device) if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()

    # Warm-up steps carry an event pair around EVERY operator: they give the per-stage table and name the dominant kernel.
    # In the timed region only that kernel is bracketed (26 event records per step would cost ~6 % of it).
    runner.profile(True)
    for w in range(args.warmup):
        if w == args.warmup - 1 and w > 0:  # first launches carry module loading: the table comes from the last warm-up step
            torch.cuda.synchronize(device)
            runner.profile_collect()
        runner.infer_audio_device(audio, hop=HOP, out=scores)
    if world > 1:
        dist.all_gather_into_tensor(gathered, scores)  # warm the RCCL communicator outside the timed region
    torch.cuda.synchronize(device)
    warm_rows = runner.profile_collect()
    dom_op = max((r for r in warm_rows if r["launches"]), key=lambda r: r["ms"] / r["launches"])["op"] if args.warmup else -1
    runner.profile_only(dom_op)
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.infer_audio_device(audio, hop=HOP, out=scores)
    if world > 1:
        dist.all_gather_into_tensor(gathered, scores)  # the one collective of the path: scores of all shards
    torch.cuda.synchronize(device)
    barrier()
    elapsed = time.perf_counter() - t0
    runner.profile(False)
    rows = runner.profile_collect()
    runner.profile_only(-1)
    if args.warmup:  # stages from the warm-up profile, the dominant kernel's entry replaced by its timed-region measurement
        timed = {r["op"]: r for r in rows if r["launches"]}
        rows = [timed.get(r["op"], r) if r["op"] == dom_op else r for r in warm_rows]

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        roof, stages = roofline_of(rows, batch, args.dtype, dom_op)
        total_chunks = world * batch * args.steps
        out = {
            "metric": "audio chunks/sec (3 s @ 24 kHz)",
            "value": round(total_chunks / elapsed, 1),
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (peak-normalised tone + gaussian noise, generated on device); shipped birdnet_stm32n6_100 weights",
            "config": {
                "workload": ("birdnet_stm32n6_100 float32 DS-CNN, hybrid+pwl frontend" if args.dtype == "f32"
                             else "birdnet_stm32n6_100 INT8 DS-CNN, hybrid+pwl frontend"),
                "batch_per_gpu": batch,
                "global_batch": world * batch,
                "chunk": "3 s @ 24 kHz (72000 samples), n_fft 512, hop 281, 257x256 spectrogram",
                "path": "audio in HBM -> STFT -> mel+PWL -> DS-CNN -> scores in HBM" + (" -> RCCL all-gather" if world > 1 else ""),
            },
            "roofline": roof,
            "stages": stages,
        }
        if world == 1 and not args.batch:  # the other single-GPU BASELINE configuration, for reference (not the reported value)
            runner.close()
            runner = None
            other = "i8" if args.dtype == "f32" else "f32"
            out["also_measured"] = quick_rate(torch, other, 4096 if other == "i8" else 1024, device, local_rank)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.dtype)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if runner is not None:
        runner.close()


if __name__ == "__main__":
    main()
