"""``birdnet-stm32 evaluate`` — same flags as the reference CLI (reference: birdnet_stm32/cli/evaluate.py:29-78).

Flow (reference :81-207): resolve ``<model>_model_config.json`` next to the model unless ``--model_config`` is
given, read ``class_names`` from it, collect the test files under ``--data_path_test/<class>/``, load the runner,
run ``evaluate`` and print the metric summary.  ``--benchmark`` writes the reference's JSON report shape
(``model_path``, ``num_classes``, ``num_files`` (= total chunks, as in the reference), ``metrics``, ``config``)
and ``--save_csv`` the per-file scores.  The presentation-only reports of the reference (confusion matrix,
DET curve, species bootstrap CI, threshold optimisation, HTML) are outside the accelerated path: their flags
are accepted and answered with a one-line notice.

Extra flags of this build: ``--device`` (GPU index) and ``--max_batch`` (workspace size in chunks).
"""

from __future__ import annotations

import argparse
import json
import math
import os


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Evaluate model on test audio (file-level pooling).")
    p.add_argument("--model_path", type=str, required=True, help="Path to .keras or .tflite model")
    p.add_argument("--model_config", type=str, default="", help="Path to model config JSON")
    p.add_argument("--data_path_test", type=str, required=True, help="Path to test dataset root")
    p.add_argument("--max_files", type=int, default=-1, help="Max test files per class")
    p.add_argument("--batch_size", type=int, default=16, help="Batch size for chunk inference")
    p.add_argument("--overlap", type=float, default=0.0, help="Chunk overlap (seconds)")
    p.add_argument("--pooling", type=str, default="avg", choices=["avg", "max", "lme"])
    p.add_argument("--save_csv", type=str, default="", help="Optional path to save predictions CSV")
    p.add_argument("--confusion_matrix", action="store_true", default=False)
    p.add_argument("--save_cm_plot", type=str, default="")
    p.add_argument("--optimize_thresholds", action="store_true", default=False)
    p.add_argument("--benchmark", type=str, default="", help="Save structured JSON benchmark report to this path")
    p.add_argument("--benchmark_latency", action="store_true", default=False,
                   help="Measure per-chunk inference latency (mean, median, p95, p99)")
    p.add_argument("--species_report", type=str, default="")
    p.add_argument("--n_bootstrap", type=int, default=1000)
    p.add_argument("--det_curve", action="store_true", default=False)
    p.add_argument("--save_det_plot", type=str, default="")
    p.add_argument("--report_html", type=str, default="")
    p.add_argument("--profile_memory", action="store_true", default=False, help="Report peak memory (RSS) during inference")
    p.add_argument("--device", type=int, default=0, help="MI355X index")
    p.add_argument("--max_batch", type=int, default=1024, help="Workspace size in chunks")
    return p


def get_args(argv=None) -> argparse.Namespace:
    return build_parser().parse_args(argv)


def resolve_config_path(model_path: str, model_config: str = "") -> str:
    path = model_config or os.path.splitext(model_path)[0] + "_model_config.json"
    if not os.path.isfile(path):
        raise FileNotFoundError(f"Model config JSON not found: {path}")
    return path


def save_benchmark_json(metrics: dict, classes: list[str], model_path: str, out_path: str, config: dict | None = None) -> None:
    """Reference report shape (reference: birdnet_stm32/evaluation/reporting.py:192-236)."""
    core = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in metrics.items() if k != "ap_per_class"}
    report = {"model_path": model_path, "num_classes": len(classes), "num_files": metrics.get("total_chunks", 0), "metrics": core}
    if config:
        report["config"] = config
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(report, fh, indent=2, default=str)
    print(f"Benchmark report saved to {out_path}")


def save_predictions_csv(per_file: list[dict], classes: list[str], out_path: str) -> None:
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        fh.write("file,label," + ",".join(c.replace(",", " ") for c in classes) + "\n")
        for row in per_file:
            fh.write(f"{row['file']},{row['label']}," + ",".join(f"{s:.6f}" for s in row["scores"]) + "\n")


def main(argv=None, runner=None):
    """Evaluate a model on a class-structured test set.  ``runner`` lets tests inject a predict()-object."""
    from birdnet_stm32.data.dataset import SUPPORTED_AUDIO_EXTS, load_file_paths_from_directory
    from birdnet_stm32.evaluation.metrics import evaluate
    from birdnet_stm32.training.config import ModelConfig

    args = get_args(argv)
    cfg = ModelConfig.load(resolve_config_path(args.model_path, args.model_config)).to_dict()
    classes = cfg.get("class_names", [])
    if not classes:
        raise ValueError("class_names missing in model config.")
    files, _ = load_file_paths_from_directory(args.data_path_test, classes=classes, exts=SUPPORTED_AUDIO_EXTS, max_samples=args.max_files)
    if not files:
        raise RuntimeError(f"No test audio found in {args.data_path_test}")
    if runner is None:
        from birdnet_stm32.models.runners import load_model_runner

        runner = load_model_runner(args.model_path, device=args.device, max_batch=args.max_batch)

    metrics, per_file, y_true, y_scores = evaluate(
        model_runner=runner, files=files, classes=classes, cfg=cfg, pooling=args.pooling, batch_size=args.batch_size,
        overlap=max(0.0, min(cfg["chunk_duration"] - 0.1, args.overlap)), measure_latency=args.benchmark_latency,
        profile_memory=args.profile_memory,
    )  # fmt: skip

    print(f"\nEvaluated {len(per_file)} files across {len(classes)} classes.")
    for key, value in metrics.items():
        if key == "ap_per_class":
            continue
        print(f"  {key}: {value:.4f}" if isinstance(value, float) else f"  {key}: {value}")
    ranked = sorted(((classes[i], a) for i, a in enumerate(metrics.get("ap_per_class", []))
                     if not (a is None or (isinstance(a, float) and math.isnan(a)))), key=lambda t: t[1], reverse=True)
    if ranked:
        print("\nTop 10 classes by AP:")
        for name, ap in ranked[:10]:
            print(f"  {name}: {ap:.4f}")
    for flag in ("confusion_matrix", "save_cm_plot", "optimize_thresholds", "species_report", "det_curve", "save_det_plot", "report_html"):
        if getattr(args, flag):
            print(f"[notice] --{flag}: presentation report not included in the MI355X hot-path build")
    if args.save_csv:
        save_predictions_csv(per_file, classes, args.save_csv)
        print(f"Predictions saved to {args.save_csv}")
    if args.benchmark:
        save_benchmark_json(metrics, classes, args.model_path, args.benchmark, config=cfg)
    return metrics


if __name__ == "__main__":
    main()
