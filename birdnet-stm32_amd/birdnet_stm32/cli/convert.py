"""``python -m birdnet_stm32 convert`` — post-training quantisation of a trained ``.keras`` checkpoint to an INT8 ``.tflite``.

Flags and flow follow the reference (reference: birdnet_stm32/cli/convert.py:23-72 flags, :74-200 flow: resolve the model
config next to the checkpoint, stratified representative dataset from ``--data_path_train`` or random inputs without
one, convert, validate float vs INT8 on a different subset, fail below ``--min_cosine_sim``, optional JSON report).
The converter is this build's own (no TensorFlow), in two forms: with ``--template`` (an existing ``.tflite`` of the same
topology, default the shipped ``birdnet_stm32n6_100.tflite``) the output keeps that file's operator structure and frozen frontend
(``conversion.quantize.requantize_like``); with ``--template none`` — and automatically when the checkpoint's topology does not
match the template (squeeze-excite, inverted residuals, another width) — the graph is written from the model itself
(``conversion.export.convert_netspec_to_int8`` + ``models._tflite_writer``).  ``--quantization dynamic`` and ``--export_onnx`` are TensorFlow/tf2onnx features and are
refused.  Validation runs both models on the MI355X (``HipRunner``).
"""

from __future__ import annotations

import argparse
import json
import os
import random

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_DEFAULT_TEMPLATE = os.path.join(_PKG, "checkpoints", "birdnet_stm32n6_100.tflite")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Convert a trained Keras model to quantized TFLite and validate.")
    p.add_argument("--checkpoint_path", type=str, required=True, help="Path to trained .keras model")
    p.add_argument("--model_config", type=str, default="", help="Path to model config JSON")
    p.add_argument("--output_path", type=str, default="", help="Output .tflite path")
    p.add_argument("--data_path_train", type=str, default="", help="Training data directory for rep. dataset")
    p.add_argument("--num_samples", type=int, default=1024, help="Representative dataset samples")
    p.add_argument("--validate_samples", type=int, default=256, help="Validation samples")
    p.add_argument("--min_cosine_sim", type=float, default=0.95, help="Minimum mean cosine similarity threshold. Conversion fails if below (0 to disable).")
    p.add_argument("--quantization", type=str, default="ptq", choices=["ptq", "dynamic"], help="Quantization mode ('ptq' only in this build).")
    p.add_argument("--per_tensor", action="store_true", help="Use per-tensor quantization instead of per-channel (default).")
    p.add_argument("--batch_validate", type=int, default=0, help="Run validation N times with different random seeds and report worst-case metrics (0 = off).")
    p.add_argument("--export_onnx", action="store_true", help="Not available in this build (requires tf2onnx).")
    p.add_argument("--report_json", type=str, default="", help="Path to save a structured JSON conversion report.")
    p.add_argument("--template", type=str, default=_DEFAULT_TEMPLATE,
                   help="Existing .tflite of the same topology (operator structure of the output); 'none' writes the graph from the model itself")
    p.add_argument("--frontend_norm", type=str, default="auto", choices=["auto", "off"],
                   help="'off' exports the hybrid frontend without its per-sample max-normalisation (the shipped checkpoint's form)")
    p.add_argument("--device", type=int, default=0)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.quantization != "ptq":
        raise SystemExit("--quantization dynamic needs the TensorFlow converter; this build implements 'ptq'.")
    if args.export_onnx:
        raise SystemExit("--export_onnx needs tf2onnx; not part of this build.")
    from birdnet_stm32.conversion.quantize import TopologyMismatch, representative_data_gen, requantize_like
    from birdnet_stm32.conversion.validate import validate_models
    from birdnet_stm32.data.dataset import load_file_paths_from_directory
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models._tflite_reader import load_tflite, patch_tflite
    from birdnet_stm32.models.frontend import normalize_frontend_name
    from birdnet_stm32.models.runners import HipRunner, load_model_runner
    from birdnet_stm32.training.config import ModelConfig

    if not args.model_config:
        args.model_config = os.path.splitext(args.checkpoint_path)[0] + "_model_config.json"
    if not os.path.isfile(args.model_config):
        raise FileNotFoundError(f"Model config JSON not found: {args.model_config}")
    cfg = ModelConfig.load(args.model_config).to_dict()
    spec = load_keras_archive(args.checkpoint_path)
    print(f"Loaded model from {args.checkpoint_path}")

    if os.path.isdir(args.data_path_train):
        file_paths, _classes = load_file_paths_from_directory(args.data_path_train)
        by_class: dict[str, list[str]] = {}
        for p in file_paths:
            by_class.setdefault(os.path.basename(os.path.dirname(p)), []).append(p)
        per_class = max(1, args.num_samples // max(len(by_class), 1))
        chosen: list[str] = []
        for paths in by_class.values():
            chosen.extend(random.sample(paths, min(per_class, len(paths))))
        random.shuffle(chosen)
        chosen = chosen[: args.num_samples]
        print(f"Representative dataset: {len(chosen)} stratified samples from {len(by_class)} classes.")
        val_paths = random.sample(file_paths, min(args.validate_samples, len(file_paths)))
        rep = lambda: representative_data_gen(chosen, cfg, num_samples=len(chosen))  # noqa: E731
        rep_val = lambda: representative_data_gen(val_paths, cfg, num_samples=len(val_paths))  # noqa: E731
    else:
        print("No training data directory provided; generating random representative dataset.")
        frontend = normalize_frontend_name(cfg["audio_frontend"])
        T = int(cfg["sample_rate"] * cfg["chunk_duration"])

        def rep(num_samples=args.num_samples):
            for _ in range(num_samples):
                if frontend == "librosa":
                    yield [np.random.rand(1, int(cfg["num_mels"]), int(cfg["spec_width"]), 1).astype(np.float32)]
                elif frontend == "hybrid":
                    yield [np.random.rand(1, int(cfg["fft_length"]) // 2 + 1, int(cfg["spec_width"]), 1).astype(np.float32)]
                else:
                    yield [np.random.randn(1, T, 1).astype(np.float32)]

        rep_val = lambda: rep(num_samples=args.validate_samples)  # noqa: E731

    if not args.output_path:
        args.output_path = os.path.splitext(args.checkpoint_path)[0] + "_quantized.tflite"
    os.makedirs(os.path.dirname(args.output_path) or ".", exist_ok=True)
    new = None
    template_is_default = os.path.abspath(args.template) == os.path.abspath(_DEFAULT_TEMPLATE)
    if args.template.lower() != "none":
        try:
            new = requantize_like(load_tflite(args.template), spec, rep, per_tensor=args.per_tensor)
            with open(args.template, "rb") as fh:
                raw = patch_tflite(fh.read(), new)
        except TopologyMismatch as e:
            # only this, and only for the DEFAULT template: a template the user named must fit (another graph form would be a different
            # artefact under the same exit code), and any other failure (a patch size mismatch, an unsupported operator) is a bug to see
            if not template_is_default:
                raise
            print(f"Template {os.path.basename(args.template)} does not fit this model ({e}); writing the graph from the model itself.")
            new = None
    if new is None:
        from birdnet_stm32.conversion.export import convert_netspec_to_int8
        from birdnet_stm32.models._tflite_writer import write_tflite

        new = convert_netspec_to_int8(spec, rep, per_tensor=args.per_tensor, frontend_norm=False if args.frontend_norm == "off" else None)
        raw = write_tflite(new)
    with open(args.output_path, "wb") as fh:
        fh.write(raw)
    print(f"Quantization mode: {args.quantization}" + (" (per-tensor)" if args.per_tensor else ""))
    print(f"TFLite model saved to {args.output_path}")

    report: dict = {"output_path": args.output_path, "quantization": args.quantization, "per_tensor": args.per_tensor}
    f32 = load_model_runner(args.checkpoint_path, device=args.device, max_batch=8)
    i8 = HipRunner(lower_i8(load_tflite(args.output_path)), device=args.device, max_batch=8)
    runs = max(1, args.batch_validate)
    all_metrics = []
    for k in range(runs):
        if runs > 1:
            print(f"\n--- Validation run {k + 1}/{runs} ---")
            random.seed(k)
            np.random.seed(k)
        m = validate_models(f32, i8, rep_val)
        for key, v in m.items():
            print(f"{key}: {v:.6f}")
        all_metrics.append(m)
    f32.close()
    i8.close()
    metrics = all_metrics[0]
    if runs > 1:
        report["batch_validation"] = {"n_runs": runs, "all_metrics": all_metrics}
        metrics = {"cosine_mean": min(m["cosine_mean"] for m in all_metrics)}
    report["validation"] = metrics
    random.seed(42)
    np.random.seed(42)
    if args.min_cosine_sim > 0:
        if metrics["cosine_mean"] < args.min_cosine_sim:
            raise RuntimeError(f"Quantization quality check failed: mean cosine similarity {metrics['cosine_mean']:.6f} < threshold "
                               f"{args.min_cosine_sim:.4f}. Consider using a more representative calibration dataset or a simpler model.")
        print(f"Cosine similarity check passed: {metrics['cosine_mean']:.6f} >= {args.min_cosine_sim:.4f}")
    if args.report_json:
        with open(args.report_json, "w") as fh:
            json.dump(report, fh, indent=2)
    return report


if __name__ == "__main__":
    main()
