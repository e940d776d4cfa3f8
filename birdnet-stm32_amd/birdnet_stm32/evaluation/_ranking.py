"""Ranking metrics of ``evaluate`` without the per-call overhead of the library functions: micro ROC-AUC, per-class average
precision and micro average precision from ONE sort per class and ONE sort of the flattened matrix — optionally done on the GPU.

The reference calls ``sklearn.metrics.roc_auc_score(y, s, average='micro')``, ``average_precision_score(y[:, c], s[:, c])`` for every
class and ``average_precision_score(y, s, average='micro')`` (reference: birdnet_stm32/evaluation/metrics.py:155-190).  On 4096 files ×
100 classes those 102 calls cost ~0.14 s — a third of the whole files -> metrics time once the GPU pipeline feeds them
(``tools/evaluate_bench.py``).  The functions below compute the same numbers, **bit for bit** (``tests/test_host_logic.py`` holds them
against scikit-learn on random, tied, degenerate inputs): the curves are built from integer counts and every floating-point expression
(``tps / (tps + fps)``, ``tps / tps[-1]``, the step-function sum, the trapezoid sum) is evaluated with numpy in the same order on arrays
of the same length, so even the pairwise summation agrees.  What changes is only who sorts: the order of tied scores does not matter
(counts are read at the boundaries between distinct scores), so the argsort may come from the GPU (``bn_rank_orders``).
"""

from __future__ import annotations

import numpy as np


def _counts_at_thresholds(truth_desc: np.ndarray, score_desc: np.ndarray):
    """(false positives, true positives) at every distinct score, scores descending: sample i counts for threshold t iff s_i >= t."""
    last = np.r_[np.flatnonzero(np.diff(score_desc)), truth_desc.size - 1]  # last index of every run of equal scores
    tps = np.cumsum(truth_desc, dtype=np.float64)[last]
    return 1 + last - tps, tps


def average_precision_desc(truth_desc: np.ndarray, score_desc: np.ndarray) -> float:
    """Average precision sum_k (R_k - R_{k-1}) P_k over the distinct thresholds (no interpolation); a column without positives
    gives 0 (recall is taken as one everywhere, the library's convention and warning)."""
    fps, tps = _counts_at_thresholds(truth_desc, score_desc)
    seen = tps + fps
    precision = np.zeros_like(tps)
    np.divide(tps, seen, out=precision, where=seen != 0)
    recall = np.ones_like(tps) if tps[-1] == 0 else tps / tps[-1]
    p = np.hstack((precision[::-1], 1))
    r = np.hstack((recall[::-1], 0))
    return float(max(0.0, -np.sum(np.diff(r) * p[:-1])))


def roc_auc_desc(truth_desc: np.ndarray, score_desc: np.ndarray) -> float:
    """Area under the ROC curve by the trapezoid rule over the curve's corner points; NaN when only one class is present."""
    pos = int(np.count_nonzero(truth_desc))
    if pos == 0 or pos == truth_desc.size:
        return float("nan")
    fps, tps = _counts_at_thresholds(truth_desc, score_desc)
    if fps.size > 2:  # points on a straight segment between two others add nothing to the area; the library drops them, so do we
        corner = np.flatnonzero(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])
        fps, tps = fps[corner], tps[corner]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    fpr = fps / fps[-1]
    tpr = tps / tps[-1]
    return float((np.diff(fpr) * (tpr[1:] + tpr[:-1]) / 2.0).sum())


def descending_orders(scores: np.ndarray, ctx=None):
    """``(per-column order [N, C], order of the flattened matrix [N*C])`` — indices that sort the scores descending.

    With ``ctx`` (the runner's ``_hip.Context``) and float32 scores both sorts run on that context's GPU through ``bn_rank_orders`` (the score
    matrix is 1.6 MB for 4096 files) — the library's own sort kernels, not ``torch.argsort``, whose first call in a process loads PyTorch's sort
    code object for 0.1-0.17 s (round 5); otherwise numpy's stable sort, reversed — what scikit-learn does per call."""
    s = np.ascontiguousarray(scores)
    if ctx is not None and s.dtype == np.float32 and s.ndim == 2 and 0 < s.size < (1 << 30):
        import ctypes

        import torch

        from birdnet_stm32 import _hip

        n, c = s.shape
        dev = torch.device("cuda", ctx.device)
        with torch.cuda.device(dev):
            d = torch.from_numpy(s).to(dev)
            cols = torch.empty((c, n), dtype=torch.int32, device=dev)
            flat = torch.empty(n * c, dtype=torch.int32, device=dev)
            stream = torch.cuda.current_stream(dev)
            _hip.check(ctx.lib.bn_rank_orders(ctx.handle, d.data_ptr(), n, c, cols.data_ptr(), flat.data_ptr(), ctypes.c_void_p(stream.cuda_stream)))
            return cols.cpu().numpy().T, flat.cpu().numpy()
    return np.argsort(s, axis=0, kind="stable")[::-1], np.argsort(s.reshape(-1), kind="stable")[::-1]


def ranking_metrics(y_true: np.ndarray, y_scores: np.ndarray, ctx=None) -> dict:
    """``{'roc-auc', 'ap_per_class', 'mAP'}`` as ``evaluate`` reports them (NaN where the library call raises or is undefined)."""
    yt = np.asarray(y_true)
    ys = np.asarray(y_scores)
    n, n_cls = ys.shape
    out = {"roc-auc": float("nan"), "ap_per_class": [float("nan")] * n_cls, "mAP": float("nan")}
    if n == 0 or not (np.isfinite(ys).all() and np.isfinite(yt).all()):
        return out
    truth = yt == 1
    if not np.logical_or(truth, yt == 0).all():  # not a 0/1 indicator matrix: the library takes other routes
        raise ValueError("y_true must be a 0/1 indicator matrix")
    cols, flat = descending_orders(ys, ctx)
    aps = []
    lacking = 0
    for c in range(n_cls):
        o = cols[:, c]
        t = truth[o, c]
        lacking += not t.any()
        aps.append(average_precision_desc(t, ys[o, c]))
    if lacking:
        import warnings

        warnings.warn(f"No positive class found in y_true for {lacking} of {n_cls} classes, recall is set to one for all thresholds.", UserWarning,
                      stacklevel=2)
    out["ap_per_class"] = aps
    t = truth.reshape(-1)[flat]
    s = ys.reshape(-1)[flat]
    out["roc-auc"] = roc_auc_desc(t, s)
    out["mAP"] = average_precision_desc(t, s)
    return out
