"""Text and file reports of ``evaluate`` under the reference's module path and names.

The reference CLI imports ten names from ``birdnet_stm32.evaluation.reporting`` (reference: ``birdnet_stm32/cli/evaluate.py:14-25``).  The seven
that belong to the hot path's surroundings are here with the reference's signatures and the reference's output, byte for byte where a test can
see it (``tests/golden/reference_pooling_config.json`` pins the benchmark JSON; ``tests/test_host_logic.py`` the text forms):

* ``print_ascii_histogram``, ``print_ascii_pr_curve``, ``print_ascii_det_curve`` — what the reference prints after every run / under ``--det_curve``
  (reference ``evaluation/reporting.py:10-50,239-256``);
* ``save_predictions_csv`` (``:53-78``), ``print_confusion_matrix`` (``:81-114``), ``save_species_report_csv`` (``:173-189``),
  ``save_benchmark_json`` (``:192-236``).

The three matplotlib / HTML renderers (``save_confusion_matrix_plot``, ``save_det_curve_plot``, ``save_html_report``) are presentation and out of
this build's scope (OUT OF SCOPE #14): they exist so that ``from birdnet_stm32.evaluation.reporting import ...`` of the reference
CLI resolves, and raise the refusal the CLI gives for their flags.

The precision-recall text curve needs ``precision_recall_curve`` of the flattened scores: computed here from one sort (the same
thresholds, tie handling and end point as scikit-learn's; ``tests/test_host_logic.py`` compares with the installed scikit-learn).
"""
from __future__ import annotations

import json
import os

import numpy as np

__all__ = ["print_ascii_histogram", "print_ascii_pr_curve", "print_ascii_det_curve", "save_predictions_csv", "print_confusion_matrix",
           "save_species_report_csv", "save_benchmark_json", "save_confusion_matrix_plot", "save_det_curve_plot", "save_html_report"]

PLOT_REFUSAL = ("renders plots / HTML, which this MI355X hot-path build does not include; use --save_csv / --benchmark / --species_report and the "
                "reference's reporting module")


def _bar(width: int, share: float) -> str:
    return "#" * int(width * share)


def print_ascii_histogram(scores: np.ndarray, bins: int = 10, width: int = 40) -> None:
    """Histogram of scores in [0, 1] as text: one line per bin, ``lo - hi | ####  (count)``, bars relative to the fullest bin."""
    counts, edges = np.histogram(np.asarray(scores), bins=bins, range=(0, 1))
    top = int(counts.max()) if counts.size else 0
    for lo, hi, c in zip(edges[:-1], edges[1:], counts):
        print(f"{lo:4.2f} - {hi:4.2f} | {_bar(width, c / top) if top > 0 else ''} ({c})")


def _precision_recall(y_true: np.ndarray, y_score: np.ndarray):
    """(precision, recall) at every distinct score threshold, highest threshold last — scikit-learn's ``precision_recall_curve`` without its
    final (1, 0) point: thresholds ascending, precision = tp / (tp + fp), recall = tp / positives (1 when there are none)."""
    order = np.argsort(-y_score, kind="mergesort")
    ys, yt = y_score[order], y_true[order].astype(np.float64)
    last = np.r_[np.nonzero(np.diff(ys))[0], ys.size - 1]   # last index of every run of equal scores
    tp = np.cumsum(yt)[last]
    fp = 1.0 + last - tp
    pos = tp[-1] if tp.size else 0.0
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
    rec = tp / pos if pos > 0 else np.ones_like(tp)
    return prec[::-1], rec[::-1]


def print_ascii_pr_curve(y_true: np.ndarray, y_scores: np.ndarray, bins: int = 10, width: int = 40) -> None:
    """Precision-recall curve as text: per precision bin (1.0 down to 0.0) the largest recall reached inside it."""
    prec, rec = _precision_recall(np.asarray(y_true).ravel(), np.asarray(y_scores).ravel())
    edges = np.linspace(1.0, 0.0, bins + 1)
    print("\nASCII Precision-Recall Curve (precision down, recall right):")
    for hi, lo in zip(edges[:-1], edges[1:]):
        inside = (prec >= lo) & (prec <= hi)
        best = float(rec[inside].max()) if inside.any() else 0.0
        print(f"{hi:4.1f} | {_bar(width, best)} ({best:4.2f})")


def print_ascii_det_curve(far: np.ndarray, frr: np.ndarray, bins: int = 10, width: int = 40) -> None:
    """DET curve as text: per false-rejection bin the lowest false-acceptance rate reached inside it."""
    print("\nASCII DET Curve (FRR down, FAR right):")
    edges = np.linspace(0.0, 1.0, bins + 1)
    for lo, hi in zip(edges[:-1], edges[1:]):
        inside = (frr >= lo) & (frr < hi)
        best = float(far[inside].min()) if inside.any() else 1.0
        print(f"FRR {lo:4.2f}-{hi:4.2f} | {_bar(width, best)} (FAR={best:4.3f})")


def save_predictions_csv(per_file: list[dict], classes: list[str], out_path: str) -> None:
    """One row per file: ``file, label, top1_label, top1_score`` and one score column per class, three decimals."""
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        fh.write(",".join(["file", "label", "top1_label", "top1_score", *classes]) + "\n")
        for row in per_file:
            s = np.asarray(row["scores"])
            k = int(np.argmax(s))
            fh.write(",".join([row["file"], row["label"], classes[k], f"{s[k]:.3f}", *(f"{v:.3f}" for v in s)]) + "\n")


def print_confusion_matrix(y_true: np.ndarray, y_scores: np.ndarray, classes: list[str], threshold: float = 0.5) -> None:
    """Top-1 confusion matrix as text; a file whose best score is below ``threshold`` counts as no prediction (it appears in no column)."""
    truth = np.argmax(y_true, axis=1)
    pred = np.argmax(y_scores, axis=1)
    pred[np.max(y_scores, axis=1) < threshold] = -1
    n = len(classes)
    cm = np.zeros((n, n), np.int64)
    keep = pred >= 0
    np.add.at(cm, (truth[keep], pred[keep]), 1)
    w = min(12, max(len(c) for c in classes)) if classes else 6
    names = [c[:w] for c in classes]
    print("\nConfusion Matrix (rows=true, cols=predicted):\n" + " " * (w + 1) + " ".join(f"{x:>{w}}" for x in names))
    for name, row in zip(names, cm):
        print(f"{name:>{w}} " + " ".join(f"{v:>{w}}" for v in row))
    hit, total = int(np.trace(cm)), int(cm.sum())
    print(f"\nAccuracy: {hit}/{total} ({100 * hit / max(total, 1):.1f}%)")


def save_species_report_csv(species_data: list[dict], out_path: str) -> None:
    """Per-species AP with its bootstrap interval, best first."""
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        fh.write("class,ap,ci_lower,ci_upper,n_positive,n_total\n")
        for r in sorted(species_data, key=lambda r: r["ap"], reverse=True):
            fh.write(f"{r['class']},{r['ap']:.6f},{r['ci_lower']:.6f},{r['ci_upper']:.6f},{r['n_positive']},{r['n_total']}\n")
    print(f"Species AP report saved to {out_path}")


def save_benchmark_json(metrics: dict, classes: list[str], model_path: str, out_path: str, species_data: list[dict] | None = None,
                        config: dict | None = None) -> None:
    """Structured report: model, class / file counts, the scalar metrics rounded to six decimals, optional species table and model config
    (key order as the reference writes it: tests/golden/reference_pooling_config.json holds the reference's bytes)."""
    core = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in metrics.items() if k != "ap_per_class"}
    report = {"model_path": model_path, "num_classes": len(classes), "num_files": metrics.get("total_chunks", 0), "metrics": core}
    if species_data:
        report["species"] = species_data
    if config:
        report["config"] = config
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(report, fh, indent=2, default=str)
    print(f"Benchmark report saved to {out_path}")


def _refuse(what: str):
    raise NotImplementedError(f"{what} {PLOT_REFUSAL}")


def save_confusion_matrix_plot(*args, **kwargs) -> None:
    _refuse("save_confusion_matrix_plot")


def save_det_curve_plot(*args, **kwargs) -> None:
    _refuse("save_det_curve_plot")


def save_html_report(*args, **kwargs) -> None:
    _refuse("save_html_report")
