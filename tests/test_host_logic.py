"""Host-side logic on CPU: registries, config, audio ingest, pooling, the evaluate loop with a FakeRunner
(BASELINE configs[0]: 16 synthetic chunks through evaluate --benchmark_latency plumbing, no GPU), CLI surface."""

import json
import os
import warnings

import numpy as np
import pytest

from conftest import CONFIG_PATH, synth_chunks


# ---------------------------------------------------------------------------------------- registries
def test_frontend_registry_api():
    from birdnet_stm32.models.registry import FrontendInfo, get_frontend_info, is_n6_compatible, is_precomputed, list_frontends, register_frontend

    assert list_frontends() == ["hybrid", "librosa", "log_mel", "mfcc", "raw"]
    info = get_frontend_info("librosa")
    assert (info.name, info.mode, info.precomputed) == ("librosa", "precomputed", True)
    assert [is_precomputed(n) for n in ("librosa", "mfcc", "log_mel", "hybrid", "raw")] == [True, True, True, False, False]
    assert all(is_n6_compatible(n) for n in list_frontends())
    assert get_frontend_info("hybrid").hip_path is True
    with pytest.raises(KeyError, match="not registered"):
        get_frontend_info("nonexistent_frontend")
    with pytest.raises(ValueError, match="already registered"):
        register_frontend(FrontendInfo(name="librosa", mode="precomputed", precomputed=True, n6_compatible=True))


def test_frontend_names_and_aliases():
    from birdnet_stm32.models.frontend import VALID_FRONTENDS, normalize_frontend_name

    for n in VALID_FRONTENDS:
        assert normalize_frontend_name(n) == n
    with pytest.warns(DeprecationWarning, match="deprecated"):
        assert normalize_frontend_name("precomputed") == "librosa"
    with pytest.warns(DeprecationWarning):
        assert normalize_frontend_name("tf") == "raw"
    with pytest.raises(ValueError, match="Invalid audio frontend"):
        normalize_frontend_name("fft")


def test_model_registry_and_builder():
    from birdnet_stm32.models import build_model, list_models, register_model
    from birdnet_stm32.models.blocks import _make_divisible

    assert list_models() == ["dscnn"]
    with pytest.raises(KeyError, match="Unknown model"):
        build_model("resnet")
    with pytest.raises(ValueError, match="already registered"):
        register_model("dscnn")(lambda **kw: None)
    # reference tests/test_dscnn.py:11-30
    assert [_make_divisible(v, 8) for v in (32, 30, 28, 1, 0, 33.5)] == [32, 32, 32, 8, 8, 32]
    kw = dict(num_mels=64, spec_width=256, sample_rate=22050, chunk_duration=3, embeddings_size=256, num_classes=10)
    for fe, shape in (("hybrid", (None, 257, 256, 1)), ("librosa", (None, 64, 256, 1)), ("mfcc", (None, 20, 256, 1)), ("raw", (None, 44100 * 0 + 66150, 1))):
        if fe == "raw":
            with pytest.raises(ValueError, match="STM32N6 constraint"):
                build_model("dscnn", audio_frontend=fe, **kw)
            m = build_model("dscnn", audio_frontend=fe, **{**kw, "sample_rate": 16000, "chunk_duration": 2})
            assert m.input_shape == (None, 32000, 1)
        else:
            m = build_model("dscnn", audio_frontend=fe, **kw)
            assert m.input_shape == shape
        assert m.output_shape == (None, 10)
        for ly in m.layers:
            if ly.filters is not None:
                assert ly.filters % 8 == 0
    small = build_model("dscnn", alpha=0.5, **kw).count_params()
    big = build_model("dscnn", alpha=1.5, **kw).count_params()
    deep = build_model("dscnn", depth_multiplier=2, **kw).count_params()
    base = build_model("dscnn", **kw).count_params()
    assert small < base < big and base < deep
    legacy = build_model("dscnn", **{**kw, "num_classes": 100}, use_se=False, use_inverted_residual=False)
    assert legacy.count_params() == 229508  # the shipped checkpoint's topology


# ---------------------------------------------------------------------------------------- audio ingest
def test_chunking_rules():
    from birdnet_stm32.audio.io import estimate_num_chunks, split_audio_into_chunks

    sr, cd = 1000, 3.0
    ramp = np.arange(7500, dtype=np.float32)
    ch = split_audio_into_chunks(ramp, sr, cd, 0.0)
    assert ch.shape == (3, 3000)
    assert ch[1, 0] == 3000 and ch[2, 0] == 4500  # tail chunk starts at len - chunk
    ch = split_audio_into_chunks(ramp, sr, cd, 1.0)  # step 2000: starts 0,2000,4000 + tail 4500
    assert [int(c[0]) for c in ch] == [0, 2000, 4000, 4500]
    short = split_audio_into_chunks(np.ones(1200, np.float32), sr, cd)
    assert short.shape == (1, 3000) and short[0, :1200].min() == 1 and short[0, 1200:].max() == 0
    assert split_audio_into_chunks(np.zeros(0, np.float32), sr, cd).shape == (0, 3000)
    exact = split_audio_into_chunks(np.ones(6000, np.float32), sr, cd)
    assert exact.shape == (2, 3000)
    # an overlap above chunk_duration - 0.1 is clamped (step = 0.1 s)
    assert split_audio_into_chunks(np.ones(3300, np.float32), sr, cd, 5.0).shape[0] == estimate_num_chunks(3300, sr, cd, 5.0) == 4
    for n in (1, 2999, 3000, 3001, 5999, 6000, 7500, 60000):
        for ov in (0.0, 0.5, 1.5):
            assert split_audio_into_chunks(np.ones(n, np.float32), sr, cd, ov).shape[0] == estimate_num_chunks(n, sr, cd, ov)


def test_wav_decode_mix_resample_normalise(tmp_path):
    import struct

    from birdnet_stm32.audio.io import fast_resample, load_audio_file, load_audio_window, save_wav

    sr = 16000
    x = np.linspace(-1.0, 1.0, sr, dtype=np.float32) * 0.5
    p = tmp_path / "a.wav"
    save_wav(x, str(p), sr, subtype="FLOAT")
    chunks = load_audio_file(str(p), sample_rate=sr, chunk_duration=3.0)  # reference tests/test_audio_io.py:36-48
    assert chunks.shape == (1, 3 * sr)
    np.testing.assert_allclose(chunks[0, :sr], x / np.abs(x).max(), rtol=1e-6, atol=1e-6)
    assert np.all(chunks[0, sr:] == 0)
    # PCM16: libsndfile writes lrint(x * 0x7FFF) and reads int16 / 32768; peak-normalised afterwards
    save_wav(x, str(tmp_path / "b.wav"), sr)
    y = load_audio_window(str(tmp_path / "b.wav"), sample_rate=sr)
    q = np.clip(np.rint(x * 32767), -32768, 32767) / 32768.0
    np.testing.assert_allclose(y, (q / np.abs(q).max()).astype(np.float32), atol=1e-7)
    # stereo PCM16 -> channel mean; 24-bit PCM; resampling on load
    l, r = (np.sin(np.arange(8000) * 0.01) * 12000).astype("<i2"), (np.cos(np.arange(8000) * 0.02) * 9000).astype("<i2")
    inter = np.stack([l, r], axis=1).tobytes()
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(inter), b"WAVE", b"fmt ", 16, 1, 2, 8000, 8000 * 4, 4, 16, b"data", len(inter))
    (tmp_path / "st.wav").write_bytes(hdr + inter)
    y = load_audio_window(str(tmp_path / "st.wav"), sample_rate=8000)
    mono = (l.astype(np.float32) / 32768 + r.astype(np.float32) / 32768) / 2
    np.testing.assert_allclose(y, mono / np.abs(mono).max(), atol=1e-6)
    y16 = load_audio_window(str(tmp_path / "st.wav"), sample_rate=16000)
    assert abs(len(y16) - 16000) <= 1 and abs(np.abs(y16).max() - 1.0) < 1e-6
    v = np.array([0, 1, -1, 8388607, -8388608], np.int32)
    raw24 = b"".join(int(s & 0xFFFFFF).to_bytes(3, "little") for s in v)
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(raw24), b"WAVE", b"fmt ", 16, 1, 1, 8000, 24000, 3, 24, b"data", len(raw24))
    (tmp_path / "p24.wav").write_bytes(hdr + raw24)
    y = load_audio_window(str(tmp_path / "p24.wav"), sample_rate=8000)
    np.testing.assert_allclose(y, v / 8388608.0 / 1.0, atol=1e-7)
    # unreadable / empty files -> empty result, like the reference's blanket except
    (tmp_path / "junk.wav").write_bytes(b"not a wave file")
    assert load_audio_file(str(tmp_path / "junk.wav")) == []
    assert load_audio_file(str(tmp_path / "missing.wav")) == []
    up = fast_resample(np.sin(np.linspace(0, 6.28, 1000, dtype=np.float32)), 16000, 22050)
    assert abs(len(up) - int(1000 * 22050 / 16000)) <= 1
    same = np.ones(100, np.float32)
    assert fast_resample(same, 22050, 22050) is same


def test_pooling_exact_values():
    from birdnet_stm32.evaluation.pooling import lme_pooling, pool_scores

    np.testing.assert_allclose(pool_scores(np.array([[0.2, 0.8], [0.6, 0.4]], np.float32), "avg"), [0.4, 0.6])
    np.testing.assert_allclose(pool_scores(np.array([[0.1, 0.9], [0.7, 0.3]], np.float32), "max"), [0.7, 0.9])
    assert np.array_equal(pool_scores(np.zeros((0, 5), np.float32), "avg"), np.zeros(5))
    with pytest.raises(ValueError, match="Unsupported"):
        pool_scores(np.ones((3, 2), np.float32), "invalid")
    with pytest.raises(ValueError, match="must be"):
        pool_scores(np.ones(5), "avg")
    np.testing.assert_allclose(lme_pooling(np.array([[0.5, 0.3]], np.float32), 10.0), [0.5, 0.3], atol=1e-5)
    np.testing.assert_allclose(lme_pooling(np.array([[0.1, 0.9], [0.8, 0.2]], np.float32), 100.0), [0.8, 0.9], atol=0.05)
    np.testing.assert_allclose(lme_pooling(np.array([[0.1, 0.9], [0.8, 0.2]], np.float32), 1e-4), [0.45, 0.55], atol=1e-3)


# ---------------------------------------------------------------------------------------- evaluate plumbing
class FakeRunner:
    """The reference's mock backend (tests/test_metrics.py:11-22): any object with predict(x) -> [B, C]."""

    def __init__(self, n_classes, key_fn):
        self.n, self.key_fn, self.calls = n_classes, key_fn, []

    def predict(self, x):
        self.calls.append(x.shape)
        out = np.zeros((x.shape[0], self.n), np.float32)
        for i in range(x.shape[0]):
            out[i, self.key_fn(x[i])] = 1.0
        return out


@pytest.fixture
def tiny_dataset(tmp_path):
    """16 synthetic 3 s @ 24 kHz chunks as WAV files under <root>/<class>/ (BASELINE configs[0])."""
    from birdnet_stm32.audio.io import save_wav
    from birdnet_stm32.training.config import ModelConfig

    cfg = ModelConfig.load(CONFIG_PATH).to_dict()
    cfg.update(sample_rate=24000, hop_length=281)
    classes = cfg["class_names"]
    pool = synth_chunks(160)
    x = np.stack([pool[i // 2] if i % 2 == 0 else pool[150 + i // 2] for i in range(16)])  # even files: low tone, odd: high
    files = []
    for i in range(16):
        d = tmp_path / classes[i % 2]
        d.mkdir(exist_ok=True)
        save_wav(x[i], str(d / f"clip_{i}.wav"), 24000, subtype="FLOAT")
        files.append(str(d / f"clip_{i}.wav"))
    long = tmp_path / classes[0] / "long.wav"
    save_wav(np.concatenate([x[0], x[2], x[4][:30000]]), str(long), 24000, subtype="FLOAT")  # 2 full chunks + tail
    files.append(str(long))
    (tmp_path / "not_a_class").mkdir()
    save_wav(x[1], str(tmp_path / "not_a_class" / "x.wav"), 24000)
    files.append(str(tmp_path / "not_a_class" / "x.wav"))
    (tmp_path / classes[1] / "broken.wav").write_bytes(b"junk")
    files.append(str(tmp_path / classes[1] / "broken.wav"))
    return tmp_path, files, classes, cfg


def _oracle_specs(chunks, n_fft, width):
    from oracle import stft

    return np.stack([stft.hybrid_spectrogram(c, n_fft, width) for c in chunks])


def test_evaluate_with_fake_runner_cpu_plumbing(tiny_dataset):
    from birdnet_stm32.evaluation.metrics import evaluate, make_chunks_for_file

    root, files, classes, cfg = tiny_dataset
    chunks = make_chunks_for_file(files[0], cfg, "hybrid", "pwl", 512, 0.0, spectrogram_fn=_oracle_specs)
    assert len(chunks) == 1 and chunks[0].shape == (257, 256, 1) and chunks[0].dtype == np.float32
    assert 0.0 <= chunks[0].min() and chunks[0].max() <= 1.0
    assert len(make_chunks_for_file(files[16], cfg, "hybrid", "pwl", 512, 0.0, spectrogram_fn=_oracle_specs)) == 3
    # the tone's FFT bin tells the class: files of class 0 carry a 0.5-0.8 kHz tone, class 1 a ~6 kHz tone
    def key(spec):
        return int(spec[:, :, 0].mean(axis=1)[5:].argmax() + 5 > 60)

    runner = FakeRunner(len(classes), key)
    metrics, per_file, y_true, y_scores = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=2, measure_latency=True,
                                                   spectrogram_fn=_oracle_specs)
    assert len(per_file) == 17 and y_true.shape == y_scores.shape == (17, 100)  # unknown label + broken file skipped
    assert metrics["total_chunks"] == 16 + 3
    assert max(s[0] for s in runner.calls) <= 2 and (2, 257, 256, 1) in runner.calls  # batches never cross files
    assert metrics["precision"] == pytest.approx(1.0) and metrics["recall"] == pytest.approx(1.0) and metrics["f1"] == pytest.approx(1.0)
    for k in ("latency_mean_ms", "latency_median_ms", "latency_p95_ms", "latency_p99_ms", "total_chunks"):
        assert k in metrics
    for k in ("roc-auc", "cmAP", "mAP", "ap_per_class"):
        assert k in metrics
    m2, *_ = evaluate(FakeRunner(len(classes), key), files, classes, cfg, spectrogram_fn=_oracle_specs, profile_memory=True)
    assert "latency_mean_ms" not in m2 and "total_chunks" not in m2 and "peak_rss_mb" in m2
    with pytest.raises(RuntimeError, match="No valid test samples"):
        evaluate(runner, [files[17]], classes, cfg, spectrogram_fn=_oracle_specs)
    with pytest.raises(Exception, match="(?i)no HIP device|no CPU fallback|not found"):  # the mel modes run on the GPU only
        make_chunks_for_file(files[0], dict(cfg, num_mels=64), "librosa", "none", 512, 0.0)
    with pytest.raises(ValueError, match="Invalid audio_frontend"):  # the reference's evaluator knows librosa | hybrid | raw
        make_chunks_for_file(files[0], dict(cfg, num_mels=64), "mfcc", "none", 512, 0.0)


def test_cli_surface(tiny_dataset, tmp_path, capsys):
    from birdnet_stm32.cli.evaluate import build_parser, main, resolve_config_path
    from birdnet_stm32.data.dataset import load_file_paths_from_directory

    root, files, classes, cfg = tiny_dataset
    flags = {a.dest for a in build_parser()._actions}
    for f in ("model_path", "model_config", "data_path_test", "max_files", "batch_size", "overlap", "pooling", "save_csv", "confusion_matrix",
              "save_cm_plot", "optimize_thresholds", "benchmark", "benchmark_latency", "species_report", "n_bootstrap", "det_curve",
              "save_det_plot", "report_html", "profile_memory"):
        assert f in flags, f
    args = build_parser().parse_args(["--model_path", "m.tflite", "--data_path_test", "d"])
    assert (args.batch_size, args.pooling, args.overlap, args.max_files, args.n_bootstrap) == (16, "avg", 0.0, -1, 1000)
    with pytest.raises(FileNotFoundError, match="Model config JSON not found"):
        resolve_config_path(str(tmp_path / "nope.tflite"))
    assert resolve_config_path(CONFIG_PATH.replace("_model_config.json", ".tflite")) == CONFIG_PATH
    found, cls_out = load_file_paths_from_directory(str(root), classes=classes)
    assert len(found) == 18 and sorted(cls_out) == sorted(classes[:2])  # broken.wav is discovered, skipped later
    capped, _ = load_file_paths_from_directory(str(root), classes=classes, max_samples=3)
    assert len(capped) == 6

    class Const:
        def predict(self, x):
            return np.full((x.shape[0], 100), 0.25, np.float32)

    # CLI flow with an injected runner; the hybrid spectrograms need the GPU, so only the raw-frontend plumbing runs here
    rawcfg = dict(cfg, audio_frontend="raw")
    cfgp = tmp_path / "m_model_config.json"
    cfgp.write_text(json.dumps(rawcfg))
    out_json, out_csv = tmp_path / "bench.json", tmp_path / "pred.csv"
    main(["--model_path", str(tmp_path / "m.keras"), "--data_path_test", str(root), "--benchmark_latency", "--benchmark", str(out_json),
          "--save_csv", str(out_csv), "--confusion_matrix", "--optimize_thresholds", "--det_curve", "--n_bootstrap", "5",
          "--species_report", str(tmp_path / "species.csv")], runner=Const())
    rep = json.loads(out_json.read_text())
    # reference report shape (evaluation/reporting.py:192-236): species rows ride along whenever --benchmark is given
    assert set(rep) == {"model_path", "num_classes", "num_files", "metrics", "config", "species"} and rep["num_classes"] == 100
    assert rep["num_files"] == rep["metrics"]["total_chunks"] == 19
    assert len(rep["species"]) == 100 and set(rep["species"][0]) == {"class", "ap", "ci_lower", "ci_upper", "n_positive", "n_total"}
    assert out_csv.read_text().count("\n") == 18
    assert (tmp_path / "species.csv").read_text().splitlines()[0] == "class,ap,ci_lower,ci_upper,n_positive,n_total"
    out = capsys.readouterr().out
    assert "Confusion Matrix (rows=true, cols=predicted)" in out and "Optimal per-class thresholds (max F1)" in out and "ASCII DET Curve" in out
    # the plot / HTML renderings are not in this build: the command refuses them before doing any work
    for flag in ("--save_cm_plot", "--save_det_plot", "--report_html"):
        with pytest.raises(SystemExit, match="render plots"):
            main(["--model_path", str(tmp_path / "m.keras"), "--data_path_test", str(root), flag, str(tmp_path / "x")], runner=Const())


def test_metric_helpers_match_their_definitions():
    """optimize_thresholds / bootstrap_ap_ci / compute_det_curve (reference evaluation/metrics.py:209-372) against
    direct restatements of their definitions."""
    from sklearn.metrics import average_precision_score, precision_recall_curve

    from birdnet_stm32.evaluation.metrics import bootstrap_ap_ci, compute_det_curve, optimize_thresholds

    rng = np.random.default_rng(3)
    yt = (rng.random((60, 5)) < 0.3).astype(np.float32)
    yt[:, 4] = 0  # a class without positives
    ys = np.round(rng.random((60, 5)), 2).astype(np.float32)  # ties on purpose
    names = list("abcde")
    # DET: one point per distinct score, highest first
    far, frr, thr = compute_det_curve(yt, ys)
    t, s = yt.ravel(), ys.ravel()
    want = [((s >= u)[t == 0].sum() / (t == 0).sum(), 1 - (s >= u)[t == 1].sum() / (t == 1).sum(), u) for u in np.unique(s)[::-1]]
    np.testing.assert_allclose(np.stack([far, frr, thr], 1), np.asarray(want, np.float64), atol=1e-12)
    assert [a.tolist() for a in compute_det_curve(np.zeros(4), np.ones(4))] == [[0.0], [0.0], [0.5]]
    # thresholds: argmax F1 over the PR curve, 0.5 without positives
    got = optimize_thresholds(yt, ys, names)
    assert got["e"] == 0.5
    for c in range(4):
        p, r, th = precision_recall_curve(yt[:, c], ys[:, c])
        assert got[names[c]] == float(th[np.argmax(2 * p[:-1] * r[:-1] / (p[:-1] + r[:-1] + 1e-12))])
    # bootstrap: one generator(seed) consumed class by class, n draws per resample, degenerate resamples dropped
    rows = bootstrap_ap_ci(yt, ys, names, n_bootstrap=25, seed=7)
    g = np.random.default_rng(7)
    for c in range(5):
        pos = int(yt[:, c].sum())
        assert rows[c]["n_positive"] == pos and rows[c]["n_total"] == 60 and rows[c]["class"] == names[c]
        if pos == 0:
            assert rows[c]["ci_lower"] == rows[c]["ci_upper"] or np.isnan(rows[c]["ap"])
            continue
        aps = []
        for _ in range(25):
            idx = g.integers(0, 60, size=60)
            if 0 < yt[idx, c].sum() < 60:
                aps.append(average_precision_score(yt[idx, c], ys[idx, c]))
        assert rows[c]["ci_lower"] == pytest.approx(np.percentile(aps, 2.5), abs=1e-12) and rows[c]["ci_upper"] == pytest.approx(np.percentile(aps, 97.5), abs=1e-12)


def test_hot_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from birdnet_stm32.audio.spectrogram import get_spectrogram_from_audio
    from birdnet_stm32.models.runners import load_model_runner

    from conftest import TFLITE_PATH

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        load_model_runner(TFLITE_PATH)
    with pytest.raises(Exception, match="(?i)no HIP device|no CPU fallback|not found"):
        get_spectrogram_from_audio(np.zeros(72000, np.float32), mel_bins=-1)
    with pytest.raises(Exception, match="(?i)no HIP device|no CPU fallback|not found"):
        get_spectrogram_from_audio(np.zeros(72000, np.float32), mel_bins=64, mode="mfcc")
    with pytest.raises(NotImplementedError):  # mag_scale on a linear spectrogram: no frontend's path uses it
        get_spectrogram_from_audio(np.zeros(72000, np.float32), mel_bins=-1, mag_scale="db")


# ------------------------------------------------------------- device-ingest host side (no GPU needed)
def test_polyphase_filter_layout_and_chunk_table(tmp_path):
    from birdnet_stm32.audio import ingest, io
    from oracle import ingest as oi

    for up, down in [(1, 2), (80, 147), (160, 147), (3, 4), (3, 1)]:
        taps, per_phase, pre = ingest.polyphase_filter(up, down)
        h, pre_o = oi.design_filter(up, down)
        assert pre == pre_o and taps.shape == (up, per_phase) and taps.dtype == np.float32
        full = np.zeros(per_phase * up, np.float32)
        full[: h.shape[0]] = h
        for phase in (0, up // 2, up - 1):
            assert np.array_equal(taps[phase], full[phase::up][::-1])  # oldest input sample first
    # chunk table == split_audio_into_chunks start positions (reference: audio/io.py:155-174)
    for n, overlap in [(72000 * 3 + 5, 0.0), (72000, 0.0), (100, 0.0), (72000 * 2, 1.5), (72001, 2.95), (0, 0.0)]:
        starts, valid, owner, counts, size = ingest.chunk_table([n], 24000, 3.0, overlap)
        y = np.arange(n, dtype=np.float32)
        want = io.split_audio_into_chunks(y, 24000, 3.0, overlap)
        assert counts == [want.shape[0]] and size == 72000
        assert counts[0] == io.estimate_num_chunks(n, 24000, 3.0, overlap)
        for s, v, row in zip(starts, valid, want):
            assert np.array_equal(row[:v], y[s : s + v]) and not row[v:].any()
    # header-only window reader: PCM16 payload passes through untouched, window limited by max_duration
    pcm = (np.arange(48000 * 2 * 2) % 2000 - 1000).astype(np.int16).reshape(-1, 2)
    p = str(tmp_path / "a.wav")
    import struct

    payload = pcm.tobytes()
    with open(p, "wb") as fh:
        fh.write(struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, 1, 2, 48000, 48000 * 4, 4, 16,
                             b"data", len(payload)) + payload)
    w = ingest.read_pcm_window(p, max_duration=1.5)
    assert (w.fmt, w.channels, w.sample_rate, w.frames) == (ingest.PCM_S16, 2, 48000, 72000)
    assert np.array_equal(w.payload.view(np.int16).reshape(-1, 2), pcm[:72000])
    assert ingest.read_pcm_window(str(tmp_path / "nope.wav")) is None
    io.save_wav(np.zeros(10, np.float32), p, 24000, subtype="FLOAT")
    assert ingest.read_pcm_window(p).fmt == ingest.PCM_F32


# ------------------------------------------------------------------------------- FLAC decode (audio/_flac.py, csrc/host/bn_flac.c)
def _flac_cases():
    import flac_writer as fw  # noqa: F401

    rng = np.random.default_rng(12)
    n = 4096 + 1152 + 200 + 16
    t = np.arange(n)
    left = (9000 * np.sin(2 * np.pi * 440 * t / 22050) + rng.normal(0, 300, n)).astype(np.int64)
    right = (0.6 * left + 2500 * np.sin(2 * np.pi * 97 * t / 22050) + rng.normal(0, 200, n)).astype(np.int64)
    stereo = np.clip(np.stack([left, right], axis=1), -32768, 32767)
    frames = [
        {"n": 4096, "mode": "ms", "sub": [dict(kind="fixed", order=2, po=3), dict(kind="fixed", order=1, po=0)]},
        {"n": 1152, "mode": "ls", "sub": [dict(kind="lpc", lpc=([1900, -900], 12, 10), po=2, rice2=True), dict(kind="fixed", order=4, po=1, escape_parts=(1,))]},
        {"n": 200, "mode": "sr", "sub": [dict(kind="fixed", order=3, po=0), dict(kind="verbatim")], "explicit_bps": True},
        {"n": 16, "mode": "indep", "sub": [dict(kind="fixed", order=0, po=2), dict(kind="fixed", order=2, po=0, escape_parts=(0,))]},
    ]
    yield "stereo16", stereo, 22050, 16, frames, {}
    mono = (rng.integers(-(1 << 21), 1 << 21, size=(300 + 4608, 1)) >> 4) << 4  # 24-bit, four wasted bits
    mono[300:900] = 4096  # a constant stretch
    frames = [{"n": 300, "sub": [dict(kind="fixed", order=1, po=0, wasted=4)], "number": 0},
              {"n": 600, "sub": [dict(kind="constant")], "number": 1},
              {"n": 4008, "sub": [dict(kind="lpc", lpc=([7, 3, -2], 5, 4), po=3, wasted=4)], "number": 2}]
    yield "mono24_wasted", mono, 48000, 24, frames, {"id3": True}
    many = np.clip(rng.normal(0, 1000, size=(192 * 140, 1)), -32768, 32767).astype(np.int64)  # frame numbers above 127: two-byte coded numbers
    frames = [{"n": 192, "sub": [dict(kind="fixed", order=2, po=1)]} for _ in range(140)]
    yield "many_frames", many, 16000, 16, frames, {"total_known": False, "with_md5": False}
    # a stream that does not state its length and compresses far below one bit per sample: constant subframes (found by tools/fuzz/flac_fuzz.py —
    # the wrapper used to bound the unknown length by the file size)
    flat = np.full((1000 + 4608, 1), -321, np.int64)
    flat[1000:] = 77
    frames = [{"n": 1000, "sub": [dict(kind="constant")]}, {"n": 4608, "sub": [dict(kind="constant")]}]
    yield "unknown_length_constant", flat, 8000, 24, frames, {"total_known": False}


def test_flac_decoder_matches_the_encoded_samples(tmp_path):
    """csrc/host/bn_flac.c on streams written by tests/flac_writer.py (both from RFC 9639): every subframe type, Rice / Rice2 / escape
    partitions, wasted bits, the three stereo modes, explicit block sizes, multi-byte frame numbers, ID3 prefix, unknown total length,
    window reads, MD5 and CRC checks — then the host loader and the chunker on a .flac file against the same samples in a .wav."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import flac_writer as fw

    from birdnet_stm32.audio import _flac
    from birdnet_stm32.audio.io import load_audio_file, load_audio_window

    for name, x, sr, bps, frames, kw in _flac_cases():
        raw = fw.encode(x, sr, bps, frames, **kw)
        info = _flac.flac_info(raw)
        assert info[:3] == (sr, x.shape[1], bps) and info[3] == (x.shape[0] if kw.get("total_known", True) else 0), name
        ints, sr2, bps2 = _flac.decode_flac(raw)
        assert (sr2, bps2) == (sr, bps) and np.array_equal(ints, x), name
        a, b = 1000, 1500
        part = _flac.decode_flac(raw, a, b, verify_md5=False)[0]
        assert np.array_equal(part, x[a : a + b]), name
        f32, _, _, _ = _flac.read_flac_window(raw, 0, x.shape[0])
        assert f32.dtype == np.float32 and np.array_equal(f32, (x / float(1 << (bps - 1))).astype(np.float32)), name
    # corruption is detected: a flipped payload bit fails the frame CRC, a wrong MD5 the stream check
    name, x, sr, bps, frames, kw = next(_flac_cases())
    raw = bytearray(fw.encode(x, sr, bps, frames))
    raw[len(raw) // 2] ^= 0x10
    with pytest.raises(ValueError, match="checksum|malformed"):
        _flac.decode_flac(bytes(raw))
    good = bytearray(fw.encode(x, sr, bps, frames))
    good[8 + 18 + 4] ^= 0xFF  # first MD5 byte inside STREAMINFO
    with pytest.raises(ValueError, match="MD5"):
        _flac.decode_flac(bytes(good))
    with pytest.raises(ValueError, match="malformed"):
        _flac.flac_info(b"RIFF" + bytes(40))
    # the MD5 is located by the C parser: an ID3v2 tag that happens to contain the bytes "fLaC" does not mislead it, and a stream that does
    # not state its length is checked too once it has been decoded to the end (the length pass of the loaders)
    tagged = fw.encode(x, sr, bps, frames, id3=True)
    assert tagged[:3] == b"ID3"
    body = bytearray(tagged)
    body[10:14] = b"fLaC"  # inside the tag's payload
    assert np.array_equal(_flac.decode_flac(bytes(body))[0], x)
    unknown = bytearray(fw.encode(x, sr, bps, frames, total_known=False))
    assert np.array_equal(_flac.decode_flac(bytes(unknown))[0], x)
    unknown[8 + 18 + 4] ^= 0xFF
    with pytest.raises(ValueError, match="MD5"):
        _flac.decode_flac(bytes(unknown))
    # a frame header that states another bit depth than STREAMINFO is refused (the samples would be mis-scaled silently)
    other = bytearray(fw.encode(x, sr, bps, frames))
    ff = other.index(b"\xff\xf8", 8 + 34)
    other[ff + 3] = (other[ff + 3] & 0xF1) | ((1 if bps != 8 else 4) << 1)  # sample-size code of the first frame: 8 bits (16 for an 8-bit stream)
    with pytest.raises(ValueError, match="malformed|checksum"):
        _flac.decode_flac(bytes(other))
    # a forged STREAMINFO total cannot size the output beyond what the bytes can hold
    forged = bytearray(fw.encode(x, sr, bps, frames, with_md5=False))
    forged[8 + 4 + 13] |= 0x0F
    forged[8 + 4 + 14 : 8 + 4 + 18] = b"\xff\xff\xff\xff"
    assert _flac.decode_flac(bytes(forged), verify_md5=False)[0].shape[0] == x.shape[0]
    # the loader: same audio as .flac and as .wav gives the same chunks (mono mean, resampling, peak normalisation, chunking)
    from birdnet_stm32.audio.io import save_wav

    pcm = x[:, 0].astype(np.int16)
    (tmp_path / "a.flac").write_bytes(fw.encode(pcm[:, None].astype(np.int64), 22050, 16, [{"n": 4096, "sub": [dict(kind="fixed", order=2, po=2)]},
                                                                                          {"n": pcm.size - 4096, "sub": [dict(kind="fixed", order=1, po=0)]}]))
    import struct

    payload = pcm.astype("<i2").tobytes()
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, 1, 1, 22050, 44100, 2, 16, b"data", len(payload))
    (tmp_path / "a.wav").write_bytes(hdr + payload)
    ya, yb = load_audio_window(str(tmp_path / "a.flac"), 24000), load_audio_window(str(tmp_path / "a.wav"), 24000)
    assert ya.size > 0 and np.array_equal(ya, yb)
    ca, cb = load_audio_file(str(tmp_path / "a.flac"), 24000, chunk_duration=0.1), load_audio_file(str(tmp_path / "a.wav"), 24000, chunk_duration=0.1)
    assert len(ca) == len(cb) > 1 and np.array_equal(np.asarray(ca), np.asarray(cb))


# ------------------------------------------------------------------------------- ranking metrics (evaluation/_ranking.py)
def test_ranking_metrics_are_bit_identical_to_scikit_learn():
    """evaluate()'s ROC-AUC (micro), per-class AP and micro AP come from shared sorts instead of 102 library calls (reference:
    birdnet_stm32/evaluation/metrics.py:155-190 calls sklearn): same numbers, bit for bit, on continuous scores, heavy ties (the INT8
    model's 1/256 steps), classes without positives, a class that is all positives, two files only."""
    import warnings

    from sklearn.metrics import average_precision_score, roc_auc_score

    from birdnet_stm32.evaluation._ranking import ranking_metrics

    rng = np.random.default_rng(11)
    cases = []
    for n, c, ties in ((300, 7, False), (513, 12, True), (2, 3, False), (64, 100, True), (1000, 5, False)):
        ys = rng.random((n, c)).astype(np.float32)
        if ties:
            ys = (np.floor(ys * 256) / 256).astype(np.float32)
        yt = np.zeros((n, c), np.float32)
        yt[np.arange(n), rng.integers(0, max(1, c - 2), n)] = 1.0  # the last classes have no positives
        cases.append((yt, ys))
    yt, ys = cases[0]
    yt = yt.copy()
    yt[:, 1] = 1.0  # a class every file belongs to (multi-label y_true is still a 0/1 indicator)
    cases.append((yt, ys))
    cases.append((np.zeros((10, 4), np.float32), rng.random((10, 4)).astype(np.float32)))  # no positives at all: ROC-AUC undefined
    for yt, ys in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = ranking_metrics(yt, ys)
            try:
                want_auc = float(roc_auc_score(yt, ys, average="micro"))
            except Exception:
                want_auc = float("nan")
            want_ap = [float(average_precision_score(yt[:, k], ys[:, k])) for k in range(yt.shape[1])]
            want_map = float(average_precision_score(yt, ys, average="micro"))
        assert got["ap_per_class"] == want_ap
        assert got["mAP"] == want_map
        assert got["roc-auc"] == want_auc or (np.isnan(got["roc-auc"]) and np.isnan(want_auc))
    # non-finite scores: the library raises, evaluate reports NaN
    bad = cases[0][1].copy()
    bad[3, 2] = np.nan
    got = ranking_metrics(cases[0][0], bad)
    assert np.isnan(got["roc-auc"]) and np.isnan(got["mAP"]) and all(np.isnan(a) for a in got["ap_per_class"])


# ------------------------------------------------------------------------------------------ evaluation/reporting.py (reference module path)
def test_reporting_module_exports_the_reference_names_and_forms(tmp_path, capsys):
    """``birdnet_stm32.evaluation.reporting`` resolves every name the reference CLI imports from it (reference cli/evaluate.py:14-25); the text
    curves and the CSV have the reference's form (reporting.py:10-78); the precision-recall points equal scikit-learn's (ties included); the
    three plot / HTML writers refuse."""
    from sklearn.metrics import precision_recall_curve

    from birdnet_stm32.evaluation import reporting as rp

    for name in ("print_ascii_det_curve", "print_ascii_histogram", "print_ascii_pr_curve", "print_confusion_matrix", "save_benchmark_json",
                 "save_confusion_matrix_plot", "save_det_curve_plot", "save_html_report", "save_predictions_csv", "save_species_report_csv"):
        assert callable(getattr(rp, name))
    rng = np.random.default_rng(4)
    for trial in range(6):
        yt = (rng.random(400) < 0.2).astype(np.int64)
        ys = np.round(rng.random(400), 1 if trial % 2 else 6)  # (one decimal: many ties)
        p, r = rp._precision_recall(yt, ys)
        ps, rs, _ = precision_recall_curve(yt, ys)
        assert np.array_equal(p, ps[:-1]) and np.array_equal(r, rs[:-1])
    rp.print_ascii_histogram(np.array([0.05, 0.05, 0.95, 0.5]), bins=10, width=40)
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "0.00 - 0.10 | " + "#" * 40 + " (2)" and out[9] == "0.90 - 1.00 | " + "#" * 20 + " (1)" and len(out) == 10
    y_true = np.eye(3)[[0, 1, 2, 0]]
    y_sc = np.array([[0.9, 0.1, 0.0], [0.2, 0.7, 0.1], [0.3, 0.3, 0.4], [0.4, 0.5, 0.1]])
    rp.print_ascii_pr_curve(y_true, y_sc)
    out = capsys.readouterr().out.splitlines()
    assert out[1] == "ASCII Precision-Recall Curve (precision down, recall right):" and out[2].startswith(" 1.0 | ") and len(out) == 12
    rp.print_ascii_det_curve(np.array([0.0, 0.5, 1.0]), np.array([1.0, 0.05, 0.0]))
    out = capsys.readouterr().out.splitlines()
    assert out[1] == "ASCII DET Curve (FRR down, FAR right):" and out[2] == "FRR 0.00-0.10 | " + "#" * 20 + " (FAR=0.500)"
    rp.print_confusion_matrix(y_true, y_sc, ["aa", "bb", "cc"], threshold=0.45)
    out = capsys.readouterr().out
    assert "Confusion Matrix (rows=true, cols=predicted):" in out and "Accuracy: 2/3 (66.7%)" in out   # (file 3 is below the threshold, file 4 is wrong)
    csv = tmp_path / "p.csv"
    rp.save_predictions_csv([{"file": "a.wav", "label": "aa", "scores": [0.25, 0.75, 0.0]}], ["aa", "bb", "cc"], str(csv))
    assert csv.read_text() == "file,label,top1_label,top1_score,aa,bb,cc\na.wav,aa,bb,0.750,0.250,0.750,0.000\n"
    for fn in (rp.save_confusion_matrix_plot, rp.save_det_curve_plot, rp.save_html_report):
        with pytest.raises(NotImplementedError, match="does not include"):
            fn()


def test_cli_refuses_a_data_set_of_containers_it_cannot_decode(tmp_path):
    """A data set with Ogg / MP3 / M4A files must not silently evaluate to fewer files: without the soundfile package the CLI exits with a message
    naming the containers (before any model is loaded); --skip_undecodable is the explicit way to go on."""
    from birdnet_stm32.audio.io import have_soundfile
    from birdnet_stm32.cli.evaluate import main

    if have_soundfile():
        pytest.skip("soundfile is installed: other containers decode through it")
    from conftest import CKPT_DIR, TFLITE_PATH

    cfg = json.load(open(os.path.join(CKPT_DIR, "birdnet_stm32n6_100_model_config.json")))
    cls = cfg["class_names"][0]
    d = tmp_path / "data" / cls
    d.mkdir(parents=True)
    (d / "a.ogg").write_bytes(b"OggS" + bytes(64))
    (d / "b.mp3").write_bytes(b"ID3" + bytes(64))
    with pytest.raises(SystemExit, match=r"1 x \.mp3, 1 x \.ogg of the 2 discovered files cannot be decoded"):
        main(["--model_path", TFLITE_PATH, "--data_path_test", str(tmp_path / "data")], runner=object())
