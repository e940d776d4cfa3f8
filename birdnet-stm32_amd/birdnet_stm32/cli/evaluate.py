"""``birdnet-stm32 evaluate`` — same flags as the reference CLI (reference: birdnet_stm32/cli/evaluate.py:29-78).

Flow (reference :81-207): resolve ``<model>_model_config.json`` next to the model unless ``--model_config`` is
given, read ``class_names`` from it, collect the test files under ``--data_path_test/<class>/``, load the runner,
run ``evaluate`` and print the metric summary.  ``--benchmark`` writes the reference's JSON report shape
(``model_path``, ``num_classes``, ``num_files`` (= total chunks, as in the reference), ``metrics``, ``config``)
and ``--save_csv`` the per-file scores.  ``--confusion_matrix``, ``--det_curve`` (text forms), ``--species_report`` /
``--n_bootstrap`` (bootstrap AP intervals) and ``--optimize_thresholds`` work as in the reference.  The plot / HTML
renderings (``--save_cm_plot``, ``--save_det_plot``, ``--report_html``: matplotlib) are outside the accelerated path:
the command refuses them with a non-zero exit before doing any work, so a script written for the reference fails
loudly instead of missing an output file.

Under ``python -m torch.distributed.run --nproc-per-node N`` (``WORLD_SIZE`` > 1) the test files are sharded over the
N GPUs with one RCCL all-gather of the chunk scores (``evaluation/sharding.py``); rank 0 reports.

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
    p.add_argument("--batch_size", type=int, default=16,
                   help="Batch size for chunk inference. On the GPU pipeline throughput runs use --max_batch slices; --batch_size is the "
                        "slice size of the latency samples taken with --benchmark_latency (per-chunk latency = slice time / slice chunks)")
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
    p.add_argument("--max_batch", type=int, default=4096, help="Workspace size in chunks = inference slice of the device pipeline")
    p.add_argument("--skip_undecodable", action="store_true", default=False,
                   help="Evaluate the decodable files when the data set holds containers this build cannot read (Ogg / MP3 / M4A without soundfile) instead of refusing")
    return p


def get_args(argv=None) -> argparse.Namespace:
    return build_parser().parse_args(argv)


def resolve_config_path(model_path: str, model_config: str = "") -> str:
    path = model_config or os.path.splitext(model_path)[0] + "_model_config.json"
    if not os.path.isfile(path):
        raise FileNotFoundError(f"Model config JSON not found: {path}")
    return path


# The report writers live under the reference's module path (birdnet_stm32/evaluation/reporting.py); re-exported here for callers of earlier rounds.
from birdnet_stm32.evaluation.reporting import (  # noqa: E402,F401
    print_ascii_det_curve,
    print_ascii_histogram,
    print_ascii_pr_curve,
    print_confusion_matrix,
    save_benchmark_json,
    save_predictions_csv,
    save_species_report_csv,
)

print_det_curve = print_ascii_det_curve  # (name of rounds 1-4)


PLOT_FLAGS = ("save_cm_plot", "save_det_plot", "report_html")  # matplotlib / HTML renderings: not part of this build


def _init_distributed(device_arg: int):
    """One process per GPU under ``torch.distributed.run``: join the RCCL group, pick this rank's GPU.  Returns
    ``(rank, world, device index)``; (0, 1, device_arg) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, device_arg
    import torch
    import torch.distributed as dist

    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    return rank, world, local


def main(argv=None, runner=None):
    """Evaluate a model on a class-structured test set.  ``runner`` lets tests inject a predict()-object.

    Launched under ``python -m torch.distributed.run --nproc-per-node N`` the files are sharded over the N GPUs
    (``evaluation/sharding.py``); every rank computes the same metrics, rank 0 prints and writes the reports.
    """
    from birdnet_stm32.data.dataset import SUPPORTED_AUDIO_EXTS, load_file_paths_from_directory
    from birdnet_stm32.evaluation.metrics import bootstrap_ap_ci, compute_det_curve, evaluate, optimize_thresholds
    from birdnet_stm32.training.config import ModelConfig

    args = get_args(argv)
    unsupported = [f for f in PLOT_FLAGS if getattr(args, f)]
    if unsupported:  # refuse before any work instead of silently producing nothing
        raise SystemExit("error: " + ", ".join("--" + f for f in unsupported) + " render plots / HTML, which this MI355X hot-path build "
                         "does not include; use --save_csv / --benchmark / --species_report and the reference's reporting module")
    cfg = ModelConfig.load(resolve_config_path(args.model_path, args.model_config)).to_dict()
    classes = cfg.get("class_names", [])
    if not classes:
        raise ValueError("class_names missing in model config.")
    rank, world, device = (0, 1, args.device) if runner is not None else _init_distributed(args.device)
    if world > 1:  # every rank must walk the same shuffled file list
        import numpy as np

        np.random.seed(0)
    files, _ = load_file_paths_from_directory(args.data_path_test, classes=classes, exts=SUPPORTED_AUDIO_EXTS, max_samples=args.max_files)
    if not files:
        raise RuntimeError(f"No test audio found in {args.data_path_test}")
    # The reference reads whatever libsndfile opens (audio/io.py:90,114-116); this build decodes RIFF/WAVE and FLAC natively and hands every other
    # container to the `soundfile` package when it is installed.  Without it an Ogg / MP3 / M4A data set would silently evaluate to fewer (or
    # zero) files: refuse instead, unless the caller asks for the files to be skipped.
    from birdnet_stm32.audio.io import have_soundfile

    foreign: dict[str, int] = {}
    for path in files:
        ext = os.path.splitext(path)[1].lower()
        if ext not in (".wav", ".flac"):
            foreign[ext] = foreign.get(ext, 0) + 1
    if foreign and not have_soundfile() and not args.skip_undecodable:
        raise SystemExit("error: " + ", ".join(f"{n} x {e}" for e, n in sorted(foreign.items())) + f" of the {len(files)} discovered files cannot be decoded: only RIFF/WAVE and "
                         "FLAC are read natively and the `soundfile` package is not installed.  Convert them (e.g. to FLAC), install soundfile, or pass "
                         "--skip_undecodable to evaluate the remaining files")
    if runner is None:
        from birdnet_stm32.models.runners import load_model_runner

        runner = load_model_runner(args.model_path, device=device, max_batch=args.max_batch, prepare_pipeline=True)

    metrics, per_file, y_true, y_scores = evaluate(
        model_runner=runner, files=files, classes=classes, cfg=cfg, pooling=args.pooling, batch_size=args.batch_size,
        overlap=max(0.0, min(cfg["chunk_duration"] - 0.1, args.overlap)), measure_latency=args.benchmark_latency,
        profile_memory=args.profile_memory,
    )  # fmt: skip
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return metrics

    print(f"\nEvaluated {len(per_file)} files across {len(classes)} classes." + (f" ({world} GPUs)" if world > 1 else ""))
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
        print("\nBottom 10 classes by AP:")
        for name, ap in ranked[-10:]:
            print(f"  {name}: {ap:.4f}")
    # what the reference prints after every run (reference cli/evaluate.py:149-150)
    print_ascii_histogram(y_scores.ravel())
    print_ascii_pr_curve(y_true, y_scores)
    if args.det_curve:
        far, frr, _ = compute_det_curve(y_true, y_scores)
        print_ascii_det_curve(far, frr)
    species = None
    if args.species_report or args.benchmark:
        species = bootstrap_ap_ci(y_true, y_scores, classes, n_bootstrap=args.n_bootstrap)
        if args.species_report:
            save_species_report_csv(species, args.species_report)
    if args.save_csv:
        save_predictions_csv(per_file, classes, args.save_csv)
        print(f"Predictions saved to {args.save_csv}")
    if args.confusion_matrix:
        print_confusion_matrix(y_true, y_scores, classes)
    if args.optimize_thresholds:
        print("\nOptimal per-class thresholds (max F1):")
        for name, thr in sorted(optimize_thresholds(y_true, y_scores, classes).items(), key=lambda t: t[1], reverse=True):
            print(f"  {name}: {thr:.4f}")
    if args.benchmark:
        save_benchmark_json(metrics, classes, args.model_path, args.benchmark, config=cfg, species_data=species)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    return metrics


if __name__ == "__main__":
    main()
