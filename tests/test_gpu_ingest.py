"""GPU parity of the steps either side of the path: audio ingest and score pooling.

Integer/bit-level bar: the device chunks must equal, bit for bit, what numpy + scipy produce for the same
PCM (``oracle.ingest`` restates their operation order and is itself pinned to them in
tests/test_oracle_pinning.py); mean and max pooling bit-exact, log-mean-exp within 2e-6 absolute
(``exp``/``log`` differ in the last bits between libm and the device).
"""

import os

import numpy as np
import pytest

from conftest import fixture_signals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; there is no CPU fallback to fall back to")
    from birdnet_stm32 import _hip

    c = _hip.Context(0, 64)
    yield c
    c.close()


def _pcm16(rng, n, ch):
    t = np.arange(n) / 48000.0
    x = 0.4 * np.sin(2 * np.pi * 1500.0 * t)[:, None] + 0.2 * rng.standard_normal((n, ch))
    return np.clip(np.rint(x * 20000.0), -32768, 32767).astype(np.int16)


def _oracle_chunks(frames_f32, sr_in, sr_out, cd, overlap):
    from oracle import ingest as oi

    y = oi.ingest_window(frames_f32, sr_in, sr_out)
    return oi.split_chunks(y, sr_out, cd, overlap), y


@pytest.mark.parametrize("sr_in", [48000, 44100, 22050, 32000, 16000, 24000])
def test_ingest_pcm16_bit_exact_against_scipy_order(ctx, sr_in):
    from birdnet_stm32.audio import ingest

    rng = np.random.default_rng(sr_in)
    lengths = [int(sr_in * 7.3) + 11, int(sr_in * 1.2), int(sr_in * 3.0), 777]  # long, shorter than a chunk, exact, tiny
    for ch in (1, 2):
        pcm = [_pcm16(rng, n, ch) for n in lengths]
        wins = [ingest.window_from_int16(p, sr_in) for p in pcm]
        chunks, counts, mono, off, peak = ingest.ingest_windows_device(ctx, wins, 24000, 3.0, 0.5, return_windows=True)
        chunks = chunks.cpu().numpy()
        mono = mono.cpu().numpy()
        at = 0
        for i, p in enumerate(pcm):
            want, y = _oracle_chunks(p.astype(np.float32) / 32768.0, sr_in, 24000, 3.0, 0.5)
            assert counts[i] == want.shape[0]
            got = chunks[at : at + counts[i]]
            at += counts[i]
            assert got.shape == want.shape
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (sr_in, ch, i, np.abs(got - want).max())
        assert at == chunks.shape[0]


def test_ingest_matches_host_functions_that_call_scipy(ctx, tmp_path):
    """End to end from WAV files, against birdnet_stm32.audio.io (numpy mean + scipy.signal.resample_poly itself)."""
    from birdnet_stm32.audio import ingest, io

    rng = np.random.default_rng(5)
    paths = []
    for k, (sr, secs, ch) in enumerate([(44100, 7.0, 2), (48000, 2.0, 1), (24000, 6.5, 2), (22050, 31.0, 1)]):
        n = int(sr * secs)
        x = _pcm16(rng, n, ch)
        p = str(tmp_path / f"f{k}.wav")
        import struct

        payload = x.tobytes()
        hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, 1, ch, sr, sr * 2 * ch, 2 * ch, 16,
                          b"data", len(payload))
        with open(p, "wb") as fh:
            fh.write(hdr + payload)
        paths.append(p)
    paths.append(str(tmp_path / "missing.wav"))
    chunks, counts = ingest.load_audio_files_device(ctx, paths, 24000, 30, 3.0, 0.0)
    chunks = chunks.cpu().numpy()
    at = 0
    for p, c in zip(paths, counts):
        want = io.load_audio_file(p, 24000, 30, 3.0, 0.0)
        assert c == len(want)
        if c:
            assert np.array_equal(chunks[at : at + c].view(np.uint32), np.asarray(want, np.float32).view(np.uint32)), p
        at += c
    assert counts[-1] == 0 and at == chunks.shape[0]


@pytest.mark.parametrize("fmt", ["s24", "s32", "f32"])
def test_ingest_other_sample_formats_and_channel_counts(ctx, fmt):
    from birdnet_stm32.audio import ingest
    from oracle import ingest as oi

    rng = np.random.default_rng(11)
    for ch in (1, 2, 3, 6, 8):
        n = 48000 + 333
        x = np.clip(0.5 * rng.standard_normal((n, ch)), -0.999, 0.999)
        if fmt == "s24":
            v = np.rint(x * 8388607).astype(np.int32)
            b = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=-1).astype(np.uint8)
            win = ingest.PcmWindow(b.reshape(-1), ingest.PCM_S24, ch, 32000)
            f = (v.astype(np.float32) / np.float32(8388608.0)).astype(np.float32)
        elif fmt == "s32":
            v = np.rint(x * 2147483000).astype(np.int64).astype(np.int32)
            win = ingest.PcmWindow(v.reshape(-1).view(np.uint8), ingest.PCM_S32, ch, 32000)
            f = (v.astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            f = x.astype(np.float32)
            win = ingest.window_from_frames(f, 32000)
        chunks, counts, mono, off, peak = ingest.ingest_windows_device(ctx, [win], 24000, 1.0, 0.0, return_windows=True)
        y = oi.mono_mean(f)
        y = oi.resample_poly_f32(y, 3, 4)
        assert np.array_equal(mono.cpu().numpy().view(np.uint32), y.view(np.uint32)), (fmt, ch)
        assert float(peak.cpu().numpy()[0]) == float(np.abs(y).max())
        want = oi.split_chunks((y / np.float32(np.abs(y).max())).astype(np.float32), 24000, 1.0, 0.0)
        assert np.array_equal(chunks.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_ingest_interleaved_groups_share_one_output_buffer(ctx):
    """Windows of different (format, channels, rate) interleaved in one call: each group launches over its own runs of the
    shared mono / peak buffers, and the chunk gather sees them in the caller's order."""
    from birdnet_stm32.audio import ingest
    from oracle import ingest as oi

    rng = np.random.default_rng(17)
    spec = [(44100, 2, 3.3), (48000, 1, 4.1), (44100, 2, 0.7), (24000, 2, 6.2), (48000, 1, 3.0), (44100, 2, 9.9), (44100, 1, 2.2)]
    pcm = [_pcm16(rng, int(sr * secs), ch) for sr, ch, secs in spec]
    wins = [ingest.window_from_int16(p, sr) for p, (sr, ch, secs) in zip(pcm, spec)]
    wins.insert(3, ingest.window_from_frames(np.zeros((0, 1), np.float32), 32000))  # an empty window in the middle
    chunks, counts = ingest.ingest_windows_device(ctx, wins, 24000, 3.0, 1.0)
    chunks = chunks.cpu().numpy()
    assert counts[3] == 0
    at = 0
    for p, (sr, ch, secs), c in zip(pcm, spec, counts[:3] + counts[4:]):
        want = oi.split_chunks(oi.ingest_window(p.astype(np.float32) / 32768.0, sr, 24000), 24000, 3.0, 1.0)
        assert c == want.shape[0] and np.array_equal(chunks[at : at + c].view(np.uint32), want.view(np.uint32)), (sr, ch, secs)
        at += c
    assert at == chunks.shape[0]


def test_ingest_edge_cases(ctx):
    from birdnet_stm32.audio import ingest

    # silence: peak 0 -> no division; a one-sample window; no windows at all
    wins = [ingest.window_from_int16(np.zeros((5000, 2), np.int16), 44100), ingest.window_from_int16(np.array([[1234]], np.int16), 48000)]
    chunks, counts = ingest.ingest_windows_device(ctx, wins, 24000, 3.0, 0.0)
    assert counts == [1, 1] and chunks.shape == (2, 72000)
    c = chunks.cpu().numpy()
    assert not c[0].any()
    assert c[1, 0] == 1.0 and not c[1, 1:].any()  # the single resampled sample is its own peak
    chunks, counts = ingest.ingest_windows_device(ctx, [], 24000, 3.0, 0.0)
    assert chunks.shape == (0, 72000) and counts == []
    with pytest.raises(Exception, match="channels"):
        ingest.ingest_windows_device(ctx, [ingest.window_from_frames(np.zeros((100, 9), np.float32), 48000)], 24000)


def test_ingest_then_infer_equals_host_ingest_then_infer(ctx, tmp_path):
    """The widened path end to end: device ingest -> bn_infer_audio gives the scores of host ingest -> bn_infer_audio."""
    import torch

    from birdnet_stm32.audio import ingest, io
    from birdnet_stm32.models.runners import load_model_runner
    from conftest import KERAS_PATH

    sig = fixture_signals(44100, 5.0)
    wins, host = [], []
    for name in ("sine", "chirp", "noise"):
        pcm = np.clip(np.rint(sig[name] * 32767.0), -32768, 32767).astype(np.int16)  # save_wav's (libsndfile's) write scaling
        wins.append(ingest.window_from_int16(pcm, 44100))
        p = str(tmp_path / f"{name}.wav")
        io.save_wav(sig[name], p, 44100)
        host.append(np.asarray(io.load_audio_file(p, 24000, 30, 3.0, 0.0), np.float32))
    runner = load_model_runner(KERAS_PATH, max_batch=64)
    chunks, counts = ingest.ingest_windows_device(runner.ctx, wins, 24000, 3.0, 0.0)
    a = runner.infer_audio_device(chunks).cpu().numpy()
    b = runner.infer_audio_device(torch.from_numpy(np.concatenate(host)).cuda()).cpu().numpy()
    assert counts == [len(h) for h in host]
    assert np.array_equal(a, b)
    runner.close()


# ------------------------------------------------------------------------------------- pooling
def test_pool_scores_device_matches_numpy(ctx):
    import torch

    from birdnet_stm32.audio.ingest import pool_scores_device
    from birdnet_stm32.evaluation.pooling import pool_scores

    rng = np.random.default_rng(3)
    counts = [1, 20, 0, 7, 3, 41]
    scores = rng.random((sum(counts), 100)).astype(np.float32) ** 4
    d = torch.from_numpy(scores).cuda()
    off = np.concatenate([[0], np.cumsum(counts)])
    for method in ("avg", "max", "lme"):
        got = pool_scores_device(ctx, d, counts, method, beta=10.0).cpu().numpy()
        for i, n in enumerate(counts):
            want = pool_scores(scores[off[i] : off[i + 1]], method, beta=10.0)
            if method == "lme":
                np.testing.assert_allclose(got[i], want, rtol=0, atol=2e-6)
            else:
                assert np.array_equal(got[i].view(np.uint32), np.asarray(want, np.float32).view(np.uint32)), (method, i)
    with pytest.raises(ValueError, match="Unsupported pooling method"):
        pool_scores_device(ctx, d, counts, "median")
    with pytest.raises(ValueError, match="N_chunks"):
        pool_scores_device(ctx, d[0], counts, "avg")


# ------------------------------------------------------------------- precomputed-frontend spectrograms (§8f rank 3)
@pytest.mark.parametrize("mode,mag", [("mel", "none"), ("mel", "pwl"), ("mel", "db"), ("mel", "pcen"), ("log_mel", "none"), ("mfcc", "none")])
@pytest.mark.parametrize("sr", [24000, 22050])
def test_mel_spectrogram_modes_match_oracle(ctx, mode, mag, sr):
    """get_spectrogram_from_audio(mel_bins=64) for every mode / mag_scale against the librosa restatement (float tolerance).

    Outputs are min-max normalised to [0, 1]; bars are absolute.  The float32 GPU STFT differs from the float64 one by
    ~1e-6 of the peak, which the logarithmic modes amplify for near-silent bins.
    """
    import torch

    from birdnet_stm32.audio.spectrogram import mel_spectrograms_device
    from conftest import synth_chunks
    from oracle import melspec
    from oracle import stft as stft_oracle

    sig = fixture_signals(sr, 3.0)
    x = np.stack([sig["sine"], sig["chirp"], sig["noise"], synth_chunks(1, sr=sr)[0]]).astype(np.float32)
    got, energies = mel_spectrograms_device(ctx, torch.from_numpy(x).cuda(), sr, 512, 64, 256, mag, mode, 20, return_energies=True)
    got, energies = got.cpu().numpy(), energies.cpu().numpy()
    bar = {"none": 2e-5, "pwl": 2e-5, "db": 2e-4, "pcen": 2e-3}[mag] if mode == "mel" else (2e-5 if mode == "log_mel" else 5e-4)
    for b in range(x.shape[0]):
        want, hop = melspec.get_spectrogram(x[b], sr, 512, 64, 256, mag, mode, 20)
        assert got[b].shape == want.shape == ((20, 256) if mode == "mfcc" else (64, 256))
        assert got[b].min() >= 0.0 and got[b].max() <= 1.0
        # un-normalised mel energies (magnitude or power) against the float64-STFT restatement, relative to the chunk's peak
        ref_mel, _ = melspec.mel_spectrogram(x[b], sr, 512, 64, 256, 2.0 if mode == "mfcc" else 1.0)
        ref_mel = ref_mel[:, : energies.shape[2]]
        assert np.abs(energies[b] - ref_mel).max() <= 3e-6 * ref_mel.max()
        if mag == "pcen":
            # PCEN divides every bin by its own smoothed level, so bins that hold nothing but round-off (pure tones) come
            # out O(1) different between a float32 and a float64 STFT: check the finishing pass on the GPU's own energies,
            # and end to end only on the signals with a noise floor
            fin = stft_oracle.minmax_normalize(melspec.pcen(energies[b].astype(np.float32) * (2.0**31), sr, hop))
            assert np.abs(got[b] - fin).max() <= 2e-5, (sr, b)
            if b < 2:
                continue
        err = np.abs(got[b] - want).max()
        assert err <= bar, (mode, mag, sr, b, err)


def test_get_spectrogram_from_audio_signature_and_edge_cases(ctx):
    """The reference entry point itself: every mode's shape/range (reference tests/test_spectrogram.py), silence, mag_scale ignored off 'mel'."""
    from birdnet_stm32.audio.spectrogram import get_spectrogram_from_audio, mel_spectrograms_from_chunks

    sig = fixture_signals(24000, 3.0)
    for kw, shape in [(dict(mel_bins=64), (64, 256)), (dict(mel_bins=64, mag_scale="pwl"), (64, 256)), (dict(mel_bins=64, mode="log_mel"), (64, 256)),
                      (dict(mel_bins=64, mode="mfcc", n_mfcc=20), (20, 256)), (dict(mel_bins=-1), (257, 256)), (dict(mel_bins=32, spec_width=128), (32, 128))]:
        S = get_spectrogram_from_audio(sig["chirp"], sample_rate=24000, n_fft=512, **kw)
        assert S.shape == shape and S.dtype == np.float32 and np.isfinite(S).all()
        assert S.min() >= 0.0 and S.max() <= 1.0 and S.max() > 0.99
    for mode, mag in [("mel", "none"), ("mel", "db"), ("mel", "pcen"), ("log_mel", "none"), ("mfcc", "none")]:
        S = mel_spectrograms_from_chunks(sig["silence"][None], 24000, 512, 64, 256, mag, mode)[0]
        assert np.isfinite(S).all() and not S.any()  # constant map: (S - min) / (0 + 1e-10) = 0
    a = mel_spectrograms_from_chunks(sig["chirp"][None], 24000, 512, 64, 256, "pcen", "log_mel")
    b = mel_spectrograms_from_chunks(sig["chirp"][None], 24000, 512, 64, 256, "none", "log_mel")
    assert np.array_equal(a, b)
    with pytest.raises(ValueError, match="mode"):
        mel_spectrograms_from_chunks(sig["chirp"][None], mode="cqt")
    assert mel_spectrograms_from_chunks(np.zeros((0, 72000), np.float32)).shape == (0, 64, 256)


@pytest.mark.parametrize("frontend,mag", [("librosa", "pwl"), ("librosa", "db"), ("log_mel", "none"), ("mfcc", "none")])
def test_precomputed_frontend_models_from_audio(ctx, frontend, mag):
    """Models built for the precomputed frontends (reference models/dscnn.py:154-168): the graph passes the host-side map
    through, so audio -> bn_mel_spectrogram -> bn_forward must equal oracle spectrogram -> oracle float graph (per layer, 1e-3 bar)."""
    import torch

    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from conftest import cosine, synth_chunks
    from oracle import float_graph, melspec

    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=12,
                       audio_frontend=frontend, mag_scale=mag, n_mfcc=20, use_se=False, use_inverted_residual=False, randomize_bn=True, seed=11)
    audio = np.concatenate([synth_chunks(3), fixture_signals(24000)["chirp"][None]]).astype(np.float32)
    mode = {"librosa": "mel", "log_mel": "log_mel", "mfcc": "mfcc"}[frontend]
    maps = np.stack([melspec.get_spectrogram(a, 24000, 512, 64, 256, mag, mode, 20)[0] for a in audio])[..., None].astype(np.float32)
    ref_scores, ref_logits, acts = float_graph.forward(spec, maps, np.float64, return_all=True, return_logits=True)

    runner = HipRunner(lower_f32(spec, keep_all=True), max_batch=4)
    assert runner.input_kind == 2 and runner.input_elems == maps[0].size
    with pytest.raises(ValueError, match="configure_precomputed"):
        runner.infer_audio_device(torch.from_numpy(audio).cuda())
    runner.configure_precomputed(frontend, 24000, mag, 512, 64, 20)
    got_maps = runner.predict(maps)  # runner boundary: the reference's Runner.predict on precomputed maps
    for oi, op in enumerate(runner.plan.ops):
        if op.out < 0 or op.name not in acts:
            continue
        a = runner.op_output(oi, 4)
        r = acts[op.name].reshape(a.shape)
        assert np.abs(a - r).max() / (np.abs(r).max() + 1e-12) < 5e-4, op.name
    got_audio = runner.infer_audio_device(torch.from_numpy(audio).cuda()).cpu().numpy()
    for b in range(4):
        assert 1.0 - cosine(got_maps[b], ref_scores[b]) < 1e-5
        assert 1.0 - cosine(got_audio[b], ref_scores[b]) < 1e-3  # north-star bar for float frontends
    assert np.abs(got_audio - ref_scores).max() < (2e-2 if frontend == "mfcc" else 2e-3)
    with pytest.raises(ValueError, match="rows"):
        runner.configure_precomputed("mfcc" if frontend != "mfcc" else "librosa", 24000, "none", 512, 64, 20)
    runner.close()


def test_evaluate_librosa_frontend_device_pipeline_matches_reference_loop(tmp_path):
    """evaluate() with audio_frontend='librosa': device pipeline (ingest -> mel spectrogram -> net -> pooling on the GPU) against the
    reference-style per-file loop (make_chunks_for_file -> Runner.predict -> host pooling)."""
    from birdnet_stm32.audio.io import save_wav
    from birdnet_stm32.evaluation.metrics import evaluate
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from conftest import synth_chunks

    classes = [f"sp{i}" for i in range(6)]
    cfg = dict(sample_rate=24000, chunk_duration=3, num_mels=64, spec_width=256, fft_length=512, audio_frontend="precomputed", mag_scale="pwl")
    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=6,
                       audio_frontend="librosa", mag_scale="pwl", use_se=False, use_inverted_residual=False, class_activation="sigmoid", seed=3)
    x = synth_chunks(8)
    files = []
    for i in range(8):
        d = tmp_path / classes[i % 6]
        d.mkdir(exist_ok=True)
        wav = np.concatenate([x[i], x[(i + 3) % 8]])[: 72000 + (40000 if i % 2 else 0)]
        save_wav(wav, str(d / f"f{i}.wav"), 24000)
        files.append(str(d / f"f{i}.wav"))
    runner = HipRunner(lower_f32(spec), max_batch=8)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)  # 'precomputed' is the deprecated alias of 'librosa'
        m_dev, pf_dev, _, ys_dev = evaluate(runner, files, classes, cfg, pooling="max", batch_size=8)
        m_ref, pf_ref, _, ys_ref = evaluate(runner, files, classes, cfg, pooling="max", batch_size=8, device_pipeline=False)
    assert [p["file"] for p in pf_dev] == [p["file"] for p in pf_ref] == files
    assert np.array_equal(ys_dev, ys_ref)  # same kernels either way; only the orchestration differs
    runner.close()


def test_evaluate_raw_frontend_device_pipeline(tmp_path):
    """evaluate() for a raw-frontend model (BASELINE configs[4] topology, 24 kHz x 2 s): device ingest -> per-chunk peak
    normalisation (bn_chunk_peak_normalize) -> network -> pooling on the GPU == the reference-style per-file loop, bit for bit."""
    import torch

    from birdnet_stm32.audio.io import save_wav
    from birdnet_stm32.evaluation.metrics import evaluate
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from conftest import synth_chunks

    classes = [f"sp{i}" for i in range(5)]
    cfg = dict(sample_rate=24000, chunk_duration=2, num_mels=64, spec_width=256, fft_length=512, audio_frontend="raw", mag_scale="pcen")
    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2, embeddings_size=256, num_classes=5,
                       audio_frontend="raw", mag_scale="pcen", alpha=0.5, class_activation="sigmoid", randomize_bn=True, seed=4)
    x = synth_chunks(6, seed=2)
    files = []
    for i in range(6):
        d = tmp_path / classes[i % 5]
        d.mkdir(exist_ok=True)
        save_wav(np.concatenate([x[i], x[(i + 1) % 6]])[: 48000 * (1 + i % 3) + 1234], str(d / f"f{i}.wav"), 24000)
        files.append(str(d / f"f{i}.wav"))
    runner = HipRunner(lower_f32(spec), max_batch=8)
    # the kernel itself: y = x / (max|x| + 1e-6) in float32, exactly numpy's result
    a = torch.from_numpy(x[:, :48000].copy()).cuda()
    got = runner.infer_audio_device(a).cpu().numpy()
    xn = (x[:, :48000] / (np.abs(x[:, :48000]).max(axis=1, keepdims=True) + 1e-6)).astype(np.float32)
    assert np.array_equal(got, runner.predict(xn[..., None]))
    _, pf_dev, _, ys_dev = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=8)
    _, pf_ref, _, ys_ref = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=8, device_pipeline=False)
    assert [p["file"] for p in pf_dev] == [p["file"] for p in pf_ref] == files
    assert np.array_equal(ys_dev, ys_ref)
    runner.close()


def test_device_ingest_reads_flac_like_wav(tmp_path):
    """A FLAC file (16-bit stereo, 44.1 kHz; decoded on the host by csrc/host/bn_flac.c, uploaded as PCM) through the device ingest gives
    the chunks of the same samples in a WAV file, bit for bit — and the 24-bit form the chunks of the float oracle."""
    import struct
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import flac_writer as fw

    from birdnet_stm32 import _hip
    from birdnet_stm32.audio import ingest
    from oracle import ingest as oi

    rng = np.random.default_rng(4)
    n = 44100 * 4
    t = np.arange(n)
    x = np.stack([8000 * np.sin(2 * np.pi * 1000 * t / 44100) + rng.normal(0, 500, n), 6000 * np.sin(2 * np.pi * 300 * t / 44100) + rng.normal(0, 400, n)], axis=1)
    pcm = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    frames = [{"n": 4096, "mode": "ms", "sub": [dict(kind="fixed", order=2, po=3), dict(kind="fixed", order=2, po=3)]} for _ in range(n // 4096)]
    if n % 4096:
        frames.append({"n": n % 4096, "mode": "indep", "sub": [dict(kind="fixed", order=1, po=0), dict(kind="fixed", order=1, po=0)]})
    (tmp_path / "a.flac").write_bytes(fw.encode(pcm.astype(np.int64), 44100, 16, frames))
    payload = pcm.astype("<i2").tobytes()
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, 1, 2, 44100, 44100 * 4, 4, 16, b"data", len(payload))
    (tmp_path / "a.wav").write_bytes(hdr + payload)
    x24 = (pcm.astype(np.int64) << 8) + rng.integers(-100, 100, size=pcm.shape)
    (tmp_path / "b.flac").write_bytes(fw.encode(x24, 44100, 24, frames))
    ctx = _hip.Context(0, 64)
    chunks, counts = ingest.load_audio_files_device(ctx, [str(tmp_path / "a.flac"), str(tmp_path / "a.wav"), str(tmp_path / "b.flac")], 24000, 30, 3.0, 0.0)
    assert counts[0] == counts[1] == counts[2] > 0
    c = chunks.cpu().numpy()
    k = counts[0]
    assert np.array_equal(c[:k].view(np.uint32), c[k : 2 * k].view(np.uint32))
    want = oi.split_chunks(oi.ingest_window((x24.astype(np.float64) / float(1 << 23)).astype(np.float32), 44100, 24000), 24000, 3.0, 0.0)
    assert np.array_equal(c[2 * k :].view(np.uint32), want.view(np.uint32))
    ctx.close()


@pytest.mark.parametrize("sr_in,sr_out", [(96000, 22050), (11025, 32000), (37800, 16000)])
def test_ingest_ratios_whose_filter_needs_more_than_64_kb_of_lds(ctx, sr_in, sr_out):
    """Ratios like 147/640 and 1280/441: the generic polyphase kernel keeps the whole filter (up to ~110 KB) in LDS, the launcher
    raises the kernel's dynamic LDS limit to what the launch needs — chunks bit for bit against the scipy-pinned oracle."""
    from birdnet_stm32.audio import ingest

    rng = np.random.default_rng(sr_in + sr_out)
    pcm = [_pcm16(rng, n, ch) for n, ch in ((int(sr_in * 4.2), 1), (int(sr_in * 0.9), 2))]
    wins = [ingest.window_from_int16(p, sr_in) for p in pcm]
    for rep in range(2):
        chunks, counts = ingest.ingest_windows_device(ctx, wins, sr_out, 3.0, 0.5)[:2]
        chunks = chunks.cpu().numpy()
        at = 0
        for i, p in enumerate(pcm):
            want, _ = _oracle_chunks(p.astype(np.float32) / 32768.0, sr_in, sr_out, 3.0, 0.5)
            got = chunks[at : at + counts[i]]
            at += counts[i]
            assert counts[i] == want.shape[0] and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (sr_in, sr_out, i, rep)


@pytest.mark.gpu
def test_chunk_peak_normalize_lengths_and_alignment(ctx):
    """bn_chunk_peak_normalize (the raw frontend's host-side step, reference evaluation/metrics.py:62-69: x / (max|x| + 1e-6)) bit for bit
    against numpy: chunk lengths that are / are not multiples of 4 (16-byte path + scalar tail), a buffer whose chunks start at odd
    addresses (scalar path), in place."""
    import ctypes

    import torch

    from birdnet_stm32 import _hip

    lib = _hip.load_library()
    rng = np.random.default_rng(17)
    for T, off in ((48000, 0), (47999, 0), (7, 0), (66150, 0), (1028, 1), (4096, 2)):
        host = rng.standard_normal((5, T)).astype(np.float32)
        host[3] = 0.0  # a silent chunk: 0 / 1e-6
        flat = torch.empty(5 * T + off, dtype=torch.float32, device="cuda")  # contiguous [5, T] chunks, `off` floats behind an aligned address
        x = flat[off:].view(5, T)
        x.copy_(torch.from_numpy(host))
        ref = (x.cpu().numpy() / (np.abs(x.cpu().numpy()).max(axis=1, keepdims=True) + np.float32(1e-6))).astype(np.float32)
        y = torch.empty_like(x) if not off else x
        _hip.check(lib.bn_chunk_peak_normalize(ctx.handle, x.data_ptr(), 5, T, ctypes.c_float(1e-6), y.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), ref), (T, off)
