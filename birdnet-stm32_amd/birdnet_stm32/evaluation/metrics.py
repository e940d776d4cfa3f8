"""Per-file evaluation loop: chunk -> predict -> pool -> metrics (reference: birdnet_stm32/evaluation/metrics.py:18-207).

``make_chunks_for_file`` and ``evaluate`` keep the reference's signatures, skip rules (unknown label
directory or unreadable/empty file -> skipped), metric keys (``roc-auc``, ``f1``, ``precision``, ``recall``,
``ap_per_class``, ``cmAP``, ``mAP``; with ``measure_latency`` also ``latency_mean_ms``, ``latency_median_ms``,
``latency_p95_ms``, ``latency_p99_ms``, ``total_chunks``; with ``profile_memory`` ``peak_rss_mb``,
``rss_delta_mb``) and latency accounting (wall time of one ``predict`` divided by the batch size, replicated
per chunk, reference :130-136).

Two execution modes:

* any object with ``predict(x_batch)`` (the reference's duck-typed runner, e.g. the tests' ``FakeRunner``):
  the reference loop, one file at a time, batches never crossing files (reference :117-141);
* a :class:`HipRunner`: the *device pipeline* (``audio/pipeline.py``) — a reader pool ``pread``s the PCM of groups of files, as it
  lies in the files, into pinned slabs; the H2D copy of one group runs on a copy stream under the kernels of the previous one;
  decode, mono mix, resampling, peak normalisation and chunking (``bn_ingest_resample`` / ``bn_ingest_chunks``), STFT + frontend +
  network (``bn_infer_audio``, slices of the runner's ``max_batch`` chunks crossing file boundaries — the reference never batches
  across files, SURVEY.md finding 10) and file-level pooling (``bn_pool_scores``) all run on the GPU; one pooled row per file
  returns.  ``batch_size`` is then only the granularity of the latency samples (``measure_latency``).
"""

from __future__ import annotations

import math
import os
import resource
import time

import numpy as np

from birdnet_stm32.audio.io import load_audio_file
from birdnet_stm32.evaluation.pooling import pool_scores
from birdnet_stm32.models.frontend import normalize_frontend_name


def make_chunks_for_file(path: str, cfg: dict, frontend: str, mag_scale: str, n_fft: int, chunk_overlap: float,
                         spectrogram_fn=None, exact_spectrogram: bool = False) -> list[np.ndarray]:
    """Model-ready inputs for one file: a list of per-chunk float32 arrays (reference :18-72).

    ``spectrogram_fn(chunks [N,T], n_fft, spec_width) -> [N, F, W]`` replaces the GPU STFT (tests inject the
    CPU oracle there; the default is ``birdnet_stm32.audio.spectrogram.spectrograms_from_chunks``).
    """
    sr, cd = int(cfg["sample_rate"]), float(cfg["chunk_duration"])
    width = int(cfg["spec_width"])
    chunks = load_audio_file(path, sample_rate=sr, max_duration=60, chunk_duration=cd, random_offset=False,
                             chunk_overlap=chunk_overlap)
    if len(chunks) == 0:
        return []
    if frontend == "hybrid":
        if spectrogram_fn is None:
            from birdnet_stm32.audio.spectrogram import spectrograms_from_chunks

            # an INT8 runner quantises these values: it gets the reference's float64 arithmetic; float32 models take the float32 FFT
            specs = spectrograms_from_chunks(np.asarray(chunks, np.float32), n_fft, width, exact=exact_spectrogram)
        else:
            specs = spectrogram_fn(np.asarray(chunks, np.float32), n_fft, width)
        specs = np.asarray(specs, np.float32)
        bins = n_fft // 2 + 1
        return [s[:bins, :width, None] for s in specs]
    if frontend == "raw":
        size = int(cfg["chunk_duration"] * cfg["sample_rate"])
        out = []
        for ch in chunks:
            x = np.zeros(size, np.float32)
            x[: min(size, ch.shape[0])] = ch[:size]
            out.append((x / (np.max(np.abs(x)) + 1e-6))[:, None].astype(np.float32))
        return out
    if frontend == "librosa":
        from birdnet_stm32.audio.spectrogram import mel_spectrograms_from_chunks

        specs = mel_spectrograms_from_chunks(np.asarray(chunks, np.float32), sr, n_fft, int(cfg["num_mels"]), width, mag_scale, "mel")
        return [s[:, :, None] for s in specs]
    raise ValueError(f"Invalid audio_frontend: {frontend}")  # the reference's evaluator knows librosa | hybrid | raw only (:72)


def _label_of(path: str) -> str:
    return os.path.basename(os.path.dirname(path))


def _with_known_label(files, classes) -> list[str]:
    """The files whose directory name is one of ``classes`` (reference :104-110), one directory lookup per DIRECTORY instead of two path splits and a
    list search per file (6 ms of a 72 ms warm ``evaluate`` call on 1024 files, tools/_warm_profile.py)."""
    known = set(classes)
    seen: dict[str, bool] = {}
    out = []
    for p in files:
        d = os.path.dirname(p)
        ok = seen.get(d)
        if ok is None:
            ok = seen[d] = os.path.basename(d) in known
        if ok:
            out.append(p)
    return out


def _score_files_reference(model_runner, files, classes, cfg, frontend, mag_scale, n_fft, overlap, batch_size, measure_latency,
                           spectrogram_fn, pooling, beta):
    """The reference loop: per file, batches of at most ``batch_size`` chunks, never across files."""
    lat: list[float] = []
    exact = getattr(model_runner, "dtype", None) == "i8"
    for path in files:
        if _label_of(path) not in classes:
            continue
        chunks = make_chunks_for_file(path, cfg, frontend, mag_scale, n_fft, overlap, spectrogram_fn=spectrogram_fn, exact_spectrogram=exact)
        if not chunks:
            continue
        preds = []
        for i in range(0, len(chunks), batch_size):
            batch = np.stack(chunks[i : i + batch_size], axis=0)
            t0 = time.perf_counter()
            preds.append(model_runner.predict(batch))
            if measure_latency:
                lat.extend([(time.perf_counter() - t0) * 1000.0 / batch.shape[0]] * batch.shape[0])
        chunk_scores = np.concatenate(preds, axis=0)
        yield path, chunk_scores.shape[0], pool_scores(chunk_scores, method=pooling, beta=beta), lat


def _score_files_device_serial(runner, files, classes, cfg, overlap, batch_size, measure_latency, pooling, beta, files_per_group=64):
    """The device pipeline of rounds 1-3, kept as the A/B partner of :func:`_score_files_device` (``evaluate(..., pipelined=False)``,
    ``tools/evaluate_bench.py --serial``): groups of ``files_per_group`` files are read, uploaded, ingested, scored in slices of
    ``batch_size`` and pooled one after another, the host waiting for every step."""
    import torch

    from birdnet_stm32.audio.ingest import load_audio_files_device, pool_scores_device

    sr, cd = int(cfg["sample_rate"]), float(cfg["chunk_duration"])
    todo = _with_known_label(files, classes)
    lat: list[float] = []
    for g0 in range(0, len(todo), files_per_group):
        group = todo[g0 : g0 + files_per_group]
        chunks, counts = load_audio_files_device(runner.ctx, group, sample_rate=sr, max_duration=60, chunk_duration=cd, chunk_overlap=overlap)
        n = chunks.shape[0]
        if n == 0:
            continue
        scores = torch.empty((n, runner.num_classes), dtype=torch.float32, device=chunks.device)
        for b0 in range(0, n, batch_size):
            nb = min(batch_size, n - b0)
            t0 = time.perf_counter()
            scores[b0 : b0 + nb] = runner.infer_audio_device(chunks[b0 : b0 + nb])
            if measure_latency:
                torch.cuda.synchronize(chunks.device)
                lat.extend([(time.perf_counter() - t0) * 1000.0 / nb] * nb)
        pooled = pool_scores_device(runner.ctx, scores, counts, pooling, beta).cpu().numpy()
        for path, c, row in zip(group, counts, pooled):
            if c:
                yield path, c, row, lat


def _score_files_device(runner, files, classes, cfg, overlap, batch_size, measure_latency, pooling, beta, stats: dict | None = None,
                        pipeline_options: dict | None = None):
    """Device pipeline: read -> H2D -> ingest + inference as overlapping stages (``audio/pipeline.py``), pooling at the end.

    The host walks the headers of all files once, a reader pool ``pread``s the PCM of a group of files straight into a ring of
    pinned slabs, one ``non_blocking`` copy per group runs on a copy stream under the kernels of the previous group
    (``bn_ingest_resample`` + ``bn_ingest_chunks`` + ``bn_infer_audio`` in slices of the runner's ``max_batch`` chunks, crossing
    file boundaries — the reference never batches across files, SURVEY.md finding 10), the chunk scores of all files stay on the
    GPU and ONE ``bn_pool_scores`` launch reduces them to a row per file; only those rows travel back.

    ``batch_size`` (the reference CLI's ``--batch_size``, default 16) is the latency-sample granularity only: with
    ``measure_latency`` inference runs in slices of ``batch_size`` chunks, each bracketed by events on the launch stream (time of
    one slice / its chunks, replicated per chunk: the reference's accounting, :130-136, without a host synchronisation per slice);
    without it slices are ``max_batch`` chunks.

    Under ``torch.distributed`` (one process per GPU, ``WORLD_SIZE`` > 1) the files are dealt to the ranks in contiguous blocks of
    equal CHUNK count (every rank probes all headers and derives the same bounds: ``audio.pipeline.balanced_bounds``), every rank
    scores the chunks of its files, the chunk scores meet in ONE all-gather (RCCL over xGMI; ``evaluation/sharding.py``) and every
    rank pools all files from the gathered tensor, so ``evaluate`` returns the same metrics on every rank.  The latency list then
    covers this rank's slices only.
    """
    import torch

    from birdnet_stm32.audio.ingest import pool_scores_device
    from birdnet_stm32.audio.pipeline import EvaluatePipeline, balanced_bounds, plan_files
    from birdnet_stm32.evaluation.sharding import score_files_sharded, world_info

    sr, cd = int(cfg["sample_rate"]), float(cfg["chunk_duration"])
    todo = _with_known_label(files, classes)
    lat: list[float] = []
    if not todo:
        return
    pipe = EvaluatePipeline(runner, sr, cd, overlap, max_duration=60, **(pipeline_options or {}))
    from birdnet_stm32.evaluation import sharding as _sh

    if world_info()[1] > 1 or not _sh._local_only(world_info()[1]):  # (the second clause: the collective path forced at world size 1, tests)
        weights = plan_files(todo, sr, cd, overlap, 60, pipe.readers, keep_decoded=False).n_chunks  # counts only (no decoded window is kept); the same on every rank
        bounds = balanced_bounds(weights, world_info()[1])

        def score_block(lo, hi):
            s, c, st, l = pipe.run(todo[lo:hi], batch_size, measure_latency)
            lat.extend(l)
            if stats is not None:
                stats.update(st)
            return s, c

        scores, counts = score_files_sharded(len(todo), score_block, runner.num_classes, device=runner.device, bounds=bounds)
    else:
        scores, counts, st, l = pipe.run(todo, batch_size, measure_latency)
        lat.extend(l)
        if stats is not None:
            stats.update(st)
    pipe.close()   # (its page-locked slabs go back to the process-wide pool: the next call's pipeline page-locks nothing)
    if scores.shape[0] == 0:
        return
    t0 = time.perf_counter()
    pooled = pool_scores_device(runner.ctx, scores.contiguous(), counts, pooling, beta).cpu().numpy()
    if stats is not None:
        stats["pool_s"] = time.perf_counter() - t0
    for path, c, row in zip(todo, counts, pooled):
        if c:
            yield path, c, row, lat


def evaluate(model_runner, files: list[str], classes: list[str], cfg: dict, pooling: str = "average", batch_size: int = 64,
             overlap: float = 0.0, mep_beta: float = 10.0, measure_latency: bool = False, profile_memory: bool = False,
             spectrogram_fn=None, device_pipeline: bool | None = None, pipelined: bool = True, stats: dict | None = None,
             pipeline_options: dict | None = None):
    """Run inference per chunk, pool to file level and compute metrics (reference :75-207).

    Returns ``(metrics, per_file, y_true [N, C], y_scores [N, C])``.  ``stats`` (a dict) receives the device pipeline's per-stage
    times (read / H2D / ingest / inference / pooling / metrics; ``tools/evaluate_bench.py``); ``pipelined=False`` selects the serial
    device pipeline of earlier rounds (A/B); ``pipeline_options`` are keyword arguments of ``audio.pipeline.EvaluatePipeline``.
    """
    frontend = normalize_frontend_name(cfg["audio_frontend"])
    mag_scale = cfg.get("mag_scale", "none")
    n_fft = int(cfg["fft_length"])
    n_cls = len(classes)
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss if profile_memory else 0

    if device_pipeline is None:
        device_pipeline = (frontend in ("hybrid", "raw", "librosa", "log_mel", "mfcc") and hasattr(model_runner, "infer_audio_device")
                           and spectrogram_fn is None)
    if device_pipeline and frontend not in ("hybrid", "raw"):
        model_runner.configure_precomputed(frontend, int(cfg["sample_rate"]), mag_scale, n_fft, int(cfg["num_mels"]), int(cfg.get("n_mfcc", 20)))
    if device_pipeline:
        if pooling.lower() not in ("avg", "mean", "average", "max", "lme", "log_mean_exp", "log_mean_exponential"):
            raise ValueError(f"Unsupported pooling method: {pooling}")
        if pipelined:
            stream = _score_files_device(model_runner, files, classes, cfg, overlap, batch_size, measure_latency, pooling, mep_beta, stats,
                                         pipeline_options)
        else:
            stream = _score_files_device_serial(model_runner, files, classes, cfg, overlap, batch_size, measure_latency, pooling, mep_beta)
    else:
        stream = _score_files_reference(model_runner, files, classes, cfg, frontend, mag_scale, n_fft, overlap, batch_size,
                                        measure_latency, spectrogram_fn, pooling, mep_beta)

    y_true, y_scores, per_file, lat = [], [], [], []
    total_chunks = 0
    index_of = {}
    for i, name in enumerate(classes):
        index_of.setdefault(name, i)  # (list.index: the first occurrence)
    for path, n_chunks, pooled, lat in stream:
        label = _label_of(path)
        target = np.zeros(n_cls, np.float32)
        target[index_of[label]] = 1.0
        total_chunks += n_chunks
        y_true.append(target)
        y_scores.append(pooled)
        per_file.append({"file": path, "label": label, "scores": pooled.tolist()})
    known = _with_known_label(files, classes)
    if len(per_file) < len(known):  # the reference skips such files silently (:112-116); say which kind they were
        import warnings

        done = {r["file"] for r in per_file}
        by_ext: dict[str, int] = {}
        for p in known:
            if p not in done:
                ext = os.path.splitext(p)[1].lower()
                by_ext[ext] = by_ext.get(ext, 0) + 1
        hint = ""
        if any(e != ".wav" for e in by_ext):
            from birdnet_stm32.audio.io import have_soundfile

            if not have_soundfile():
                hint = " (only RIFF/WAVE and FLAC are decoded natively; other containers need the `soundfile` package, which is not installed)"
        warnings.warn(f"evaluate: {len(known) - len(per_file)} of {len(known)} files were empty or could not be decoded and were skipped: "
                      + ", ".join(f"{n} x {e or '<no extension>'}" for e, n in sorted(by_ext.items())) + hint, RuntimeWarning, stacklevel=2)
    if not y_true:
        raise RuntimeError("No valid test samples found for the provided class set.")

    yt = np.asarray(y_true, np.float32)
    ys = np.asarray(y_scores, np.float32)
    t_metrics = time.perf_counter()
    # ROC-AUC (micro), per-class AP, micro AP: the reference's three scikit-learn calls (:155-190), computed from shared sorts
    # (evaluation/_ranking.py: bit-identical to the library, tests/test_host_logic.py); with the device pipeline the sorts run on the GPU
    from birdnet_stm32.evaluation._ranking import ranking_metrics

    metrics: dict = {}
    # (with the device pipeline the sorts run on the GPU through the library's own sort kernels: bn_rank_orders)
    rank_ctx = getattr(model_runner, "ctx", None) if device_pipeline else None
    try:
        ranked = ranking_metrics(yt, ys, ctx=rank_ctx)
    except Exception:
        ranked = {"roc-auc": float("nan"), "ap_per_class": [float("nan")] * n_cls, "mAP": float("nan")}
    metrics["roc-auc"] = ranked["roc-auc"]
    hit = (ys >= 0.5).astype(np.float32)
    tp, fp, fn = float((yt * hit).sum()), float(((1 - yt) * hit).sum()), float((yt * (1 - hit)).sum())
    precision, recall = tp / (tp + fp + 1e-12), tp / (tp + fn + 1e-12)
    metrics["f1"] = float(2 * precision * recall / (precision + recall)) if precision + recall > 0 else 0.0
    metrics["precision"], metrics["recall"] = float(precision), float(recall)
    aps = ranked["ap_per_class"]
    good = [a for a in aps if not (a is None or (isinstance(a, float) and math.isnan(a)))]
    metrics["ap_per_class"] = aps
    metrics["cmAP"] = float(np.mean(good)) if good else float("nan")
    metrics["mAP"] = ranked["mAP"]
    if stats is not None:
        stats["metrics_s"] = time.perf_counter() - t_metrics
    if measure_latency and lat:
        arr = np.asarray(lat)
        metrics["latency_mean_ms"] = float(arr.mean())
        metrics["latency_median_ms"] = float(np.median(arr))
        metrics["latency_p95_ms"] = float(np.percentile(arr, 95))
        metrics["latency_p99_ms"] = float(np.percentile(arr, 99))
        metrics["total_chunks"] = total_chunks
    if profile_memory:
        rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        metrics["peak_rss_mb"] = round(rss1 / 1024, 1)
        metrics["rss_delta_mb"] = round((rss1 - rss0) / 1024, 1)
    return metrics, per_file, yt, ys


def optimize_thresholds(y_true: np.ndarray, y_scores: np.ndarray, classes: list[str]) -> dict[str, float]:
    """Per-class score threshold with the largest F1 on the precision-recall curve (reference :209-236); a class
    without positives keeps 0.5."""
    from sklearn.metrics import precision_recall_curve

    best: dict[str, float] = {}
    for c, name in enumerate(classes):
        truth = y_true[:, c]
        if truth.sum() == 0:
            best[name] = 0.5
            continue
        prec, rec, thr = precision_recall_curve(truth, y_scores[:, c])
        p, r = prec[:-1], rec[:-1]  # the curve carries one more point than thresholds
        best[name] = float(thr[int(np.argmax(2 * p * r / (p + r + 1e-12)))])
    return best


def bootstrap_ap_ci(y_true: np.ndarray, y_scores: np.ndarray, classes: list[str], n_bootstrap: int = 1000, confidence: float = 0.95,
                    seed: int = 42) -> list[dict]:
    """Per-class average precision with a percentile bootstrap interval (reference :239-318).

    One generator seeded with ``seed`` is consumed class by class, ``n_bootstrap`` draws of ``n`` indices each for every
    class that has both positives and negatives (the reference's draw order, so the intervals agree for equal inputs);
    resamples without both kinds of label are dropped.
    """
    from sklearn.metrics import average_precision_score

    rng = np.random.default_rng(seed)
    n = y_true.shape[0]
    tail = (1.0 - confidence) / 2.0
    rows = []
    for c, name in enumerate(classes):
        truth, score = y_true[:, c], y_scores[:, c]
        pos = int(truth.sum())
        try:
            ap = float(average_precision_score(truth, score))
        except Exception:
            ap = float("nan")
        lo = hi = ap
        if 0 < pos < n:
            draws = []
            for _ in range(n_bootstrap):
                pick = rng.integers(0, n, size=n)
                t = truth[pick]
                k = t.sum()
                if k == 0 or k == len(t):
                    continue
                try:
                    draws.append(float(average_precision_score(t, score[pick])))
                except Exception:
                    continue
            if draws:
                lo, hi = float(np.percentile(draws, 100 * tail)), float(np.percentile(draws, 100 * (1 - tail)))
        rows.append({"class": name, "ap": ap, "ci_lower": lo, "ci_upper": hi, "n_positive": pos, "n_total": n})
    return rows


def compute_det_curve(y_true: np.ndarray, y_scores: np.ndarray) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Detection-error-tradeoff points ``(far, frr, thresholds)`` over the distinct scores, highest threshold first
    (reference :321-372): at threshold t, FAR = negatives scored >= t / negatives, FRR = positives scored < t / positives.
    Degenerate label sets give the single point (0, 0, 0.5)."""
    t = np.asarray(y_true).ravel()
    s = np.asarray(y_scores).ravel()
    n_pos = t.sum()
    n_neg = len(t) - n_pos
    if n_pos == 0 or n_neg == 0:
        return np.array([0.0]), np.array([0.0]), np.array([0.5])
    order = np.argsort(-s, kind="stable")
    s_sorted, t_sorted = s[order], t[order]
    thr, first = np.unique(-s_sorted, return_index=True)  # ascending in -score = descending in score
    last = np.append(first[1:], len(s_sorted)) - 1        # last index whose score is >= the threshold
    tp = np.cumsum(t_sorted)[last]
    fp = (last + 1) - tp
    return fp / n_neg, (n_pos - tp) / n_pos, -thr
