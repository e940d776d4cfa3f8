"""Chunks per second of the shipped INT8 path from audio against the number of chunks per call (tools; not part of the bench line).

Answers one question: do the tensors between the kernels (263 KB of spectrogram + 190 KB of int8 maps per chunk) pay for leaving the 256 MB
memory-side cache?  A call over few chunks keeps them there, a call over 4096 does not.

    python tools/batch_sweep.py [--dtype i8] [--batches 256,512,...]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "birdnet-stm32_amd"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="i8", choices=["i8", "f32"])
    ap.add_argument("--batches", default="128,256,512,1024,2048,4096,8192")
    ap.add_argument("--chunks", type=int, default=81920, help="chunks per timed repeat (split into calls of the batch size)")
    args = ap.parse_args()
    import torch

    import bench
    from birdnet_stm32.models.runners import load_model_runner

    device = torch.device("cuda", 0)
    ckpt = os.path.join(bench.PKG, "checkpoints", "birdnet_stm32n6_100" + (".keras" if args.dtype == "f32" else ".tflite"))
    print("| chunks per call | calls | chunks/s | ms per 4096 chunks | slowest operators of one profiled call (ms per 4096 chunks) |\n|---|---|---|---|---|")
    for b in [int(x) for x in args.batches.split(",")]:
        runner = load_model_runner(ckpt, device=0, max_batch=b)
        audio = bench.synth_audio_device(torch, b, 0, device, 42)
        steps = max(2, args.chunks // b)
        out = bench.side_measure(torch, runner, audio, args.dtype, b, device, steps=steps, repeats=5, hop=bench.HOP)
        rows = sorted(out.pop("_rows"), key=lambda r: -r["ms"])[:4]
        top = ", ".join(f"{r['name']} {r['ms'] * 4096 / b:.3f}" for r in rows)
        print(f"| {b} | {steps} | {out['value']:.0f} | {out['ms_per_step'] * 4096 / b:.3f} | {top} |", flush=True)
        runner.close()
        del audio


if __name__ == "__main__":
    main()
