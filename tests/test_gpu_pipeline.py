"""The pipelined ``evaluate`` on the GPU (birdnet_stm32/audio/pipeline.py): reader pool -> pinned slabs -> copy stream -> ingest +
inference, against the serial device pipeline of rounds 1-3 and the reference-style per-file loop.

Reference flow being replaced: birdnet_stm32/evaluation/metrics.py:117-153 (per file: load, spectrograms, predict in batches, pool).
"""

from __future__ import annotations

import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
from conftest import CONFIG_PATH, KERAS_PATH, TFLITE_PATH  # noqa: E402
from test_pipeline_host import dataset  # noqa: E402,F401  (the mixed-format fixture: formats, rates, channels, broken files, FLAC)


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; the product has no CPU path to fall back to")
    return torch


def _cfg():
    from birdnet_stm32.training.config import ModelConfig

    cfg = ModelConfig.load(CONFIG_PATH).to_dict()
    cfg.update(sample_rate=24000, hop_length=281)
    return cfg


def _labelled(paths, tmp_path, classes):
    """Hard-link / copy the fixture's files into <class>/ directories (evaluate takes the label from the directory name)."""
    import shutil

    out = []
    for i, p in enumerate(paths):
        d = tmp_path / classes[i % 5]
        d.mkdir(exist_ok=True)
        q = d / f"{i:03d}_{os.path.basename(p)}"
        if os.path.isfile(p):
            shutil.copy(p, q)
        out.append(str(q))
    return out


@pytest.mark.parametrize("model_path", [TFLITE_PATH, KERAS_PATH])
def test_pipelined_evaluate_equals_the_serial_device_pipeline(torch_mod, dataset, tmp_path, model_path):  # noqa: F811
    """Same kernels, same chunk order -> the pooled scores of every file are identical, whatever the grouping (several small slabs,
    one slab, inference slices of 7 chunks in latency mode), and the skipped files are the same."""
    import warnings

    from birdnet_stm32.evaluation.metrics import evaluate
    from birdnet_stm32.models.runners import load_model_runner

    cfg = _cfg()
    classes = cfg["class_names"]
    files = _labelled(dataset, tmp_path, classes)
    runner = load_model_runner(model_path, max_batch=64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # "n files were empty or could not be decoded"
        m0, pf0, yt0, ys0 = evaluate(runner, files, classes, cfg, pooling="lme", batch_size=16, pipelined=False)
        variants = [dict(slab_bytes=6 << 20, group_chunks=40, readers=3), dict(slab_bytes=256 << 20, group_chunks=100000), dict(pinned_slabs=2, group_chunks=24)]
        for opts in variants:
            st: dict = {}
            m1, pf1, yt1, ys1 = evaluate(runner, files, classes, cfg, pooling="lme", batch_size=16, stats=st, pipeline_options=opts)
            assert [p["file"] for p in pf1] == [p["file"] for p in pf0]
            assert np.array_equal(yt1, yt0) and np.array_equal(ys1, ys0), opts
            assert st["chunks"] >= len(pf1)
            assert st["h2d_bytes"] > 0 and st["groups"] >= 1 and st["read_s"] > 0
        assert st["groups"] > 3  # the last variant really ran as several groups
        m2, pf2, _, ys2 = evaluate(runner, files, classes, cfg, pooling="lme", batch_size=7, measure_latency=True,
                                   pipeline_options=dict(group_chunks=50))
    assert np.array_equal(ys2, ys0)
    assert m2["total_chunks"] == st["chunks"] and m2["latency_mean_ms"] > 0 and m2["latency_p99_ms"] >= m2["latency_median_ms"] > 0
    runner.close()


def test_pipelined_evaluate_equals_the_reference_loop_on_wavs(torch_mod, tmp_path):
    """Per-file loop (host ingest, bn_stft_mag_exact, predict) vs the pipeline on PCM16 / float WAVs at three rates: mean pooling is
    ``array_equal`` (INT8: both routes quantise the reference's bytes)."""
    from birdnet_stm32.audio.io import save_wav
    from birdnet_stm32.evaluation.metrics import evaluate
    from birdnet_stm32.models.runners import load_model_runner

    from conftest import synth_chunks

    cfg = _cfg()
    classes = cfg["class_names"]
    x = synth_chunks(12)
    files = []
    for i in range(12):
        d = tmp_path / classes[i % 3]
        d.mkdir(exist_ok=True)
        n = 72000 if i % 4 else 72000 * 2 + 30000
        wav = np.concatenate([x[i], x[(i + 1) % 12], x[(i + 2) % 12]])[:n]
        save_wav(wav, str(d / f"f{i}.wav"), 24000, subtype="FLOAT" if i % 2 else "PCM_16")
        files.append(str(d / f"f{i}.wav"))
    runner = load_model_runner(TFLITE_PATH, max_batch=16)
    _, pf_dev, _, ys_dev = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=5, pipeline_options=dict(group_chunks=9))
    _, pf_ref, _, ys_ref = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=5, device_pipeline=False)
    assert [p["file"] for p in pf_dev] == [p["file"] for p in pf_ref] == files
    assert np.array_equal(ys_dev, ys_ref)
    runner.close()


def test_pipeline_run_reports_counts_for_unreadable_files(torch_mod, dataset):  # noqa: F811
    from birdnet_stm32.audio.io import load_audio_file
    from birdnet_stm32.audio.pipeline import EvaluatePipeline
    from birdnet_stm32.models.runners import load_model_runner

    runner = load_model_runner(TFLITE_PATH, max_batch=32)
    pipe = EvaluatePipeline(runner, 24000, 3.0, 0.0, group_chunks=30, slab_bytes=8 << 20)
    scores, counts, st, lat = pipe.run(list(dataset))
    want = [len(load_audio_file(p, 24000, 60, 3.0, 0.0)) for p in dataset]
    assert counts == want and scores.shape == (sum(want), 100) and lat == []
    assert st["files"] == len(dataset) and st["readable"] == sum(1 for w in want if w) and st["chunks"] == sum(want)
    # a second run on the same object reuses its rings and gives the same scores
    scores2, counts2, _, _ = pipe.run(list(dataset))
    assert counts2 == counts and torch_mod.equal(scores, scores2)
    # empty input
    s0, c0, st0, _ = pipe.run([])
    assert s0.shape[0] == 0 and c0 == [] and st0["chunks"] == 0
    runner.close()


def test_pinned_slabs_come_from_the_library_and_are_pooled(torch_mod):
    """The pipeline's staging slabs: ``bn_host_alloc_pinned`` memory (PyTorch sees it as page-locked, so H2D copies stay asynchronous), handed back to
    a process-wide pool instead of being freed, reused by the next pipeline, freed by ``release_pinned_slabs``; ``bn_preload_kernels`` is idempotent."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.audio import pipeline as pl

    pl.release_pinned_slabs()
    ctx = _hip.Context(0, 16)
    try:
        ctx.preload_kernels()
        ctx.preload_kernels()
        a = pl._PinnedSlab(ctx, 3 << 20, torch)
        assert a.tensor.is_pinned() and a.tensor.numel() == 3 << 20 and a.tensor.data_ptr() == a.ptr
        a.tensor[:1024] = torch.arange(1024, dtype=torch.int64).to(torch.uint8)
        d = torch.empty(1024, dtype=torch.uint8, device="cuda:0")
        d.copy_(a.tensor[:1024], non_blocking=True)
        torch.cuda.synchronize()
        assert torch.equal(d.cpu(), a.tensor[:1024])
        ptr = a.ptr
        a.release()
        assert a.tensor is None and pl._SLAB_POOL == [(ptr, 3 << 20)]
        b = pl._PinnedSlab(ctx, 1 << 20, torch)      # a pooled slab that is large enough is taken as it is
        assert b.ptr == ptr and b.nbytes == 3 << 20 and pl._SLAB_POOL == []
        c = pl._PinnedSlab(ctx, 1 << 20, torch)
        assert c.ptr != ptr
        b.release()
        c.release()
        assert pl.release_pinned_slabs() == (3 << 20) + (1 << 20) and pl._SLAB_POOL == []
        with pytest.raises(_hip.HipError):
            ctx.alloc_pinned(0)
    finally:
        ctx.close()


def test_rank_orders_sort_like_numpy_and_give_the_librarys_metrics(torch_mod):
    """``bn_rank_orders`` (the sorts behind ROC-AUC / average precision, reference evaluation/metrics.py:155-190): per class and for the flattened
    matrix the scores read through the returned orders are non-increasing and every index occurs once; ``ranking_metrics`` through the device sorts
    equals the host route (numpy's stable sort, which tests/test_host_logic.py holds against scikit-learn) BIT FOR BIT — random scores, heavy ties,
    constant columns, one row, sizes that are no multiple of anything."""
    from birdnet_stm32 import _hip
    from birdnet_stm32.evaluation._ranking import descending_orders, ranking_metrics

    rng = np.random.default_rng(11)
    ctx = _hip.Context(0, 16)
    try:
        for n, c, ties in ((1, 3, False), (257, 7, False), (1024, 100, False), (513, 33, True), (4096, 100, True)):
            ys = rng.random((n, c), dtype=np.float32)
            if ties:
                ys = np.round(ys * 7).astype(np.float32) / 7
                ys[:, 0] = 0.25
            yt = (rng.random((n, c)) < 0.1).astype(np.float32)
            yt[:, -1] = 0                                   # a class without positives
            cols, flat = descending_orders(ys, ctx)
            assert cols.shape == (n, c) and flat.shape == (n * c,)
            for k in range(c):
                assert np.array_equal(np.sort(cols[:, k]), np.arange(n)) and (np.diff(ys[cols[:, k], k]) <= 0).all()
            assert np.array_equal(np.sort(flat), np.arange(n * c)) and (np.diff(ys.reshape(-1)[flat]) <= 0).all()
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                host, dev = ranking_metrics(yt, ys), ranking_metrics(yt, ys, ctx=ctx)
            assert np.array_equal(np.array([host["roc-auc"], host["mAP"], *host["ap_per_class"]]), np.array([dev["roc-auc"], dev["mAP"], *dev["ap_per_class"]]),
                                  equal_nan=True), (n, c, ties)
        assert ctx.lib.bn_rank_orders(ctx.handle, None, 4, 0, None, None, None) != 0 and b"score matrix" in ctx.lib.bn_last_error()
        assert ctx.lib.bn_rank_orders(ctx.handle, None, 4, 3, None, None, None) != 0 and b"null" in ctx.lib.bn_last_error()
    finally:
        ctx.close()


def test_prepare_pipeline_fills_the_pool_while_the_model_is_lowered(torch_mod):
    """``load_model_runner(..., prepare_pipeline=True)`` (what the evaluate CLI passes): the context exists before the model file is parsed, a helper
    thread loads the library's code objects and page-locks three staging slabs into the process-wide pool meanwhile; the runner works as any other,
    a pipeline built afterwards takes its ring from the pool without page-locking anything, closing the runner waits for the helper."""
    torch = torch_mod
    from birdnet_stm32.audio import pipeline as pl
    from birdnet_stm32.models.runners import load_model_runner

    pl.release_pinned_slabs()
    runner = load_model_runner(TFLITE_PATH, max_batch=64, prepare_pipeline=True)
    try:
        while pl._PREPARE:
            pl._PREPARE.pop().join()
        assert len(pl._SLAB_POOL) == 3 and all(size >= 256 << 20 for _, size in pl._SLAB_POOL) and str(runner.device) in pl._PRELOADED
        before = sorted(p for p, _ in pl._SLAB_POOL)
        pipe = pl.EvaluatePipeline(runner, 24000, 3.0)
        pipe._ensure_slabs(1 << 20, 1 << 10)
        for e in pipe._pinned_ready:
            e.wait()
        assert pipe._n_ring == 3 and sorted(sl.ptr for sl in pipe._slabs) == before and pl._SLAB_POOL == []
        pipe.close()
        assert sorted(p for p, _ in pl._SLAB_POOL) == before
        x = torch.zeros((2, 72000), device=runner.device)
        assert torch.isfinite(runner.infer_audio_device(x)).all()
    finally:
        runner.close()
        pl.release_pinned_slabs()
