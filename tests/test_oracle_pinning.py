"""Pins the CPU oracle: known answers from the shipped artefacts, the reference's own importable modules
(fixtures), the reference's firmware C (oracle/_ref) and the committed oracle vectors.  CPU only."""

import json
import os
import zipfile

import numpy as np
import pytest

from conftest import GOLDEN, KERAS_PATH, TFLITE_PATH, cosine, fixture_signals, synth_chunks


@pytest.fixture(scope="module")
def keras_vars():
    from birdnet_stm32.models._h5_reader import read_h5_datasets

    with zipfile.ZipFile(KERAS_PATH) as z:
        return read_h5_datasets(z.read("model.weights.h5"))


@pytest.fixture(scope="module")
def netspec():
    from birdnet_stm32.models._keras_loader import load_keras_archive

    return load_keras_archive(KERAS_PATH)


@pytest.fixture(scope="module")
def tfl():
    from birdnet_stm32.models._tflite_reader import load_tflite

    return load_tflite(TFLITE_PATH)


# ---------------------------------------------------------------- known answers in the shipped artefacts
def test_mel_bank_equals_checkpoint_mixer(keras_vars):
    """SURVEY §8c KAT 1: the Slaney restatement == the checkpoint's frozen mel_mixer (librosa-seeded)."""
    from oracle.melbank import hybrid_mel_mixer

    w = keras_vars["/layers/audio_frontend_layer/mel_mixer/vars/0"][0, 0]
    ours = hybrid_mel_mixer(22050, 512, 64, fmin=150.0)
    assert w.shape == ours.shape == (264, 64)
    assert np.abs(w - ours).max() <= 2e-9
    assert np.all(w[257:] == 0.0)


def test_pwl_constants(keras_vars, tfl):
    """KAT 2: PWL break points / slopes exactly as the reference initialises them (magnitude.py:99-130)."""
    base = "/layers/audio_frontend_layer"
    for sfx, t in zip(("", "_1", "_2"), (0.10, 0.35, 0.65)):
        assert np.all(keras_vars[f"{base}/_pwl_shift_dws/depthwise_conv2d{sfx}/vars/0"] == 1.0)
        assert np.allclose(keras_vars[f"{base}/_pwl_shift_dws/depthwise_conv2d{sfx}/vars/1"], -t, atol=1e-7)
    for sfx, k in zip(("", "_1", "_2"), (0.25, 0.15, 0.08)):
        assert np.allclose(keras_vars[f"{base}/_pwl_k_dws/depthwise_conv2d{sfx}/vars/0"], k, atol=1e-7)
    for ti, t in ((65, -0.10), (63, -0.35), (61, -0.65)):
        tt = tfl.tensors[ti]
        deq = (tt.data.astype(np.float64) - tt.zero_point[0]) * tt.scale[0]
        assert np.allclose(deq, t, atol=float(tt.scale[0]))


def test_parameter_count_matches_keras(netspec):
    """229 508 stored parameters, 17 536 of them in the frontend."""
    assert netspec.count_params() == 229508
    assert netspec.frontend.n_params() == 17536
    assert netspec.input_shape == (None, 257, 256, 1) and netspec.output_shape == (None, 100)
    assert netspec.frontend.attrs["norm"] is False  # legacy checkpoint: no per-sample max normalisation


def test_int8_weights_are_folded_keras_weights(netspec, tfl):
    """KAT 3: the .tflite's int8 weights / int32 biases re-derived from the .keras file (two independent readers)."""
    L = {ly.name: ly for ly in netspec.layers}

    def folded(conv, bn):
        s = L[bn].weights["gamma"].astype(np.float64) / np.sqrt(L[bn].weights["var"].astype(np.float64) + 1e-3)
        w = L[conv].weights["kernel"].astype(np.float64) * s
        b = L[bn].weights["beta"].astype(np.float64) - L[bn].weights["mean"].astype(np.float64) * s
        return w, b

    checks = [(22, "stem_conv", "stem_bn", "conv"), (23, "stage1_ds1_dw", "stage1_ds1_dw_bn", "dw"),
              (24, "stage1_ds1_pw", "stage1_ds1_pw_bn", "conv"), (50, "stage4_ds2_pw", "stage4_ds2_pw_bn", "conv")]
    for op_i, conv, bn, kind in checks:
        op = tfl.ops[op_i]
        wt, bt = tfl.tensors[op.inputs[1]], tfl.tensors[op.inputs[2]]
        s_in = float(tfl.tensors[op.inputs[0]].scale[0])
        w, b = folded(conv, bn)
        wq = np.transpose(wt.data, (1, 2, 3, 0)).astype(np.float64) if kind == "conv" else np.transpose(wt.data, (1, 2, 3, 0))[..., 0, :].astype(np.float64)
        if kind == "dw":
            wq = wt.data[0].astype(np.float64)  # [3,3,C]
        sc = wt.scale.astype(np.float64)
        assert np.allclose(sc, np.abs(w).reshape(-1, w.shape[-1]).max(axis=0) / 127.0, rtol=1e-5)
        assert np.abs(wq - w / sc).max() <= 0.5 + 1e-3, conv
        assert np.allclose(bt.scale.astype(np.float64), s_in * sc, rtol=1e-6)
        assert np.abs(bt.data * (s_in * sc) - b).max() <= 0.51 * (s_in * sc).max() + 1e-7
    dense = tfl.tensors[tfl.ops[53].inputs[1]]
    wd = L["pred"].weights["kernel"].astype(np.float64)  # [256,100]
    assert np.allclose(dense.scale, np.abs(wd).max(axis=0) / 127.0, rtol=1e-6)
    assert np.abs(dense.data.T - wd / dense.scale.astype(np.float64)).max() <= 0.5 + 1e-3


def test_float_and_int8_oracles_agree(netspec, tfl):
    """KAT 4: float(norm off) vs INT8 logit cosine ~0.9999 on tone+noise chunks; same top-1."""
    from oracle import float_graph, stft
    from oracle.int8_graph import Int8Interpreter

    x = np.stack([stft.hybrid_spectrogram(a) for a in synth_chunks(3)])[..., None]
    _, lf = float_graph.forward(netspec, x, np.float32, return_logits=True)
    _, env = Int8Interpreter(tfl).invoke(x, return_all=True)
    t = tfl.tensors[128]
    lq = (env[128].astype(np.float32) - t.zero_point[0]) * t.scale[0]
    for b in range(3):
        assert cosine(lf[b], lq[b]) > 0.9995
        assert lf[b].argmax() == lq[b].argmax()


def test_tflite_graph_shape(tfl):
    """Appendix B: 131 tensors, 56 operators, 231 309 constant bytes, the documented operator census."""
    assert len(tfl.tensors) == 131 and len(tfl.ops) == 56 and tfl.constant_bytes() == 231309
    names = [o.name for o in tfl.ops]
    assert names.count("CONV_2D") == 13 and names.count("DEPTHWISE_CONV_2D") == 16 and names.count("ADD") == 13
    assert names[0] == "QUANTIZE" and names[-3:] == ["FULLY_CONNECTED", "LOGISTIC", "DEQUANTIZE"]
    assert abs(tfl.tensors[tfl.ops[0].outputs[0]].scale[0] - 1 / 255) < 1e-9


# ---------------------------------------------------------------- oracle vs its committed vectors
@pytest.mark.parametrize("sr", [22050, 24000])
def test_oracle_reproduces_golden_vectors(netspec, tfl, sr):
    from oracle import float_graph, stft
    from oracle.int8_graph import Int8Interpreter

    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    sig = fixture_signals(sr)
    interp = Int8Interpreter(tfl)
    for name in ("sine", "noise", "chirp"):
        key = f"{name}_{sr}"
        S = stft.hybrid_spectrogram(sig[name])
        assert S.shape == (257, 256) and S.dtype == np.float32 and S.min() >= 0.0 and S.max() <= 1.0
        assert np.abs(S[::16] - g[key + "/spec_rows"]).max() < 1e-6
        assert abs(S.astype(np.float64).sum() - float(g[key + "/spec_sum"])) < 1e-2
        x = S[None, :, :, None]
        probs, logits, acts = float_graph.forward(netspec, x, np.float64, return_all=True, return_logits=True)
        assert np.abs(acts["audio_frontend"][0, :, ::8, 0] - g[key + "/frontend"]).max() < 1e-6
        assert np.abs(logits[0] - g[key + "/logits"]).max() < 1e-4
        assert np.abs(probs[0] - g[key + "/probs"]).max() < 1e-6
        qp, env = interp.invoke(x, return_all=True)
        assert np.array_equal(env[128][0], g[key + "/i8_fc"])
        assert np.array_equal(qp[0], g[key + "/i8_probs"])
        sums = [int(env[t].astype(np.int64).sum()) for t in (83, 96, 97, 102, 110, 121, 126, 127)]
        assert sums == g[key + "/i8_sums"].tolist()


def test_stft_properties():
    """Reference tests/test_spectrogram.py: shape, dtype, silence -> 0, range [0,1]; plus linearity of the raw STFT."""
    from oracle import stft

    sig = fixture_signals(22050)
    assert np.all(stft.hybrid_spectrogram(sig["silence"]) == 0.0)
    a, b = sig["sine"], sig["noise"]
    Sa, Sb = stft.stft_magnitude(a, 512, 258), stft.stft_magnitude(b, 512, 258)
    assert Sa.shape == (257, 1 + len(a) // 258)
    assert np.abs(stft.stft_magnitude(2 * a, 512, 258) - 2 * Sa).max() < 1e-4  # homogeneity
    assert np.all(stft.stft_magnitude(a + b, 512, 258) <= Sa + Sb + 1e-4)  # triangle inequality per bin
    k = int(round(1000 * 512 / 22050))
    assert abs(int(Sa[:, 100].argmax()) - k) <= 1  # the 1 kHz tone sits in its bin


# ---------------------------------------------------------------- the reference's own importable modules
def test_pooling_matches_reference_outputs():
    from birdnet_stm32.evaluation.pooling import lme_pooling, pool_scores

    ref = json.load(open(os.path.join(GOLDEN, "reference_pooling_config.json")))["pooling"]
    for name, row in ref.items():
        if name == "empty":
            assert pool_scores(np.zeros((0, 3), np.float32), "avg").tolist() == row
            continue
        x = np.asarray(row["x"], np.float32)
        for method in ("avg", "mean", "average", "max", "lme", "log_mean_exp"):
            np.testing.assert_allclose(pool_scores(x, method=method), row[method], rtol=1e-6, atol=1e-7)
        for beta in (0.5, 10.0, 50.0):
            np.testing.assert_allclose(lme_pooling(x, beta=beta), row[f"lme_beta_{beta:g}"], rtol=1e-6, atol=1e-7)


def test_model_config_matches_reference_outputs(tmp_path):
    from birdnet_stm32.training.config import ModelConfig

    from conftest import CONFIG_PATH

    ref = json.load(open(os.path.join(GOLDEN, "reference_pooling_config.json")))["config"]
    assert ModelConfig().to_dict() == ref["defaults"]
    assert ModelConfig.load(CONFIG_PATH).to_dict() == ref["shipped"]
    legacy = {"sample_rate": 22050, "num_mels": 64, "spec_width": 256, "fft_length": 512, "chunk_duration": 3, "hop_length": 258,
              "audio_frontend": "hybrid", "mag_scale": "pwl", "embeddings_size": 256, "alpha": 1.0, "depth_multiplier": 1,
              "num_classes": 2, "class_names": ["a", "b"], "some_unknown_key": 1}
    assert ModelConfig.from_dict(legacy).to_dict() == ref["legacy_dict"]
    cases = {"neg_sr": {"sample_rate": -1}, "bad_frontend": {"audio_frontend": "nope"}, "bad_mag": {"mag_scale": "log"},
             "dm0": {"depth_multiplier": 0}, "drop1": {"dropout_rate": 1.0}, "names": {"num_classes": 3, "class_names": ["a"]}}
    for label, kw in cases.items():
        with pytest.raises(ValueError) as e:
            ModelConfig(**kw)
        assert str(e.value) == ref["errors"][label]


# ---------------------------------------------------------------- the reference's firmware C (oracle/_ref)
def _firmware():
    from oracle import cport

    if not os.path.isfile(cport.FW_LIB):
        pytest.skip("oracle/_ref/libfw_ref.so not built (needs /root/reference at build time)")
    return cport.FirmwareRef()


def test_fft_against_reference_firmware():
    """KAT 7: firmware fft_512_real == numpy rfft (butterfly-level pin of the FFT restatement)."""
    fw = _firmware()
    rng = np.random.default_rng(3)
    for _ in range(4):
        x = rng.standard_normal(512).astype(np.float32)
        ref = np.fft.rfft(x.astype(np.float64))
        assert np.abs(fw.fft_512_real(x) - ref).max() < 2e-6 * np.abs(ref).max() + 2e-5


def test_firmware_stft_is_a_different_framing():
    """SURVEY finding 9: the firmware STFT (no centre pad, symmetric Hann) matches a numpy STFT with ITS framing,
    and differs from the evaluate-path framing the oracle implements."""
    fw = _firmware()
    from oracle import stft

    x = fixture_signals(22050)["noise"]
    hop, W = 258, 256
    got = fw.stft_magnitude(x, hop, W)
    # the ORACLE'S function with the firmware's framing (no centre padding, symmetric Hann, W frames) against the reference's C: everything
    # in oracle.stft.stft_magnitude except its two framing switches is thereby reference-checked (the firmware FFT is float32: 5e-6)
    own = stft.stft_magnitude(x, 512, hop, center=False, window="hann_symmetric", n_frames=W)
    assert own.shape == got.shape == (257, W)
    assert np.abs(got - own).max() < 5e-6 * own.max() + 1e-5
    for name, sig in fixture_signals(24000).items():  # the other fixture signals, at the metric's rate
        g2, o2 = fw.stft_magnitude(sig, 281, W), stft.stft_magnitude(sig, 512, 281, center=False, window="hann_symmetric", n_frames=W)
        assert np.abs(g2 - o2).max() < 5e-6 * o2.max() + 1e-5, name
    ev = stft.stft_magnitude(x, 512, hop)[:, :W]
    assert cosine(got, ev) < 0.95


def test_stft_framing_cross_checked_with_scipy_short_time_fft():
    """The evaluate path's framing (librosa defaults: centre padding with zeros, periodic Hann, frame t centred on sample t * hop) written
    by an independent third party: ``scipy.signal.ShortTimeFFT`` puts its slice p on sample p * hop, extends the signal with zeros and takes
    the window as given.  A CROSS-CHECK of ``oracle.stft.stft_magnitude``'s framing (frame count, alignment, padding), not a pin of
    librosa's numerics: librosa itself cannot be imported here."""
    from scipy.signal import ShortTimeFFT, get_window

    from oracle import stft

    for sr, hop in ((24000, 281), (22050, 258)):
        for name, x in fixture_signals(sr).items():
            win = get_window("hann", 512, fftbins=True)  # what librosa passes to its FFT
            assert np.abs(win - stft.hann_periodic(512)).max() < 1e-15
            sft = ShortTimeFFT(win, hop=hop, fs=sr, fft_mode="onesided", mfft=512, scale_to=None)
            n_frames = 1 + len(x) // hop
            Z = sft.stft(x.astype(np.float64), p0=0, p1=n_frames)  # slices 0 .. n_frames - 1: centred on 0, hop, 2 hop, ...
            ours = stft.stft_magnitude(x, 512, hop)
            assert Z.shape == ours.shape == (257, n_frames)
            ref = np.abs(Z)
            assert np.abs(ref - ours).max() <= 2e-7 * ref.max() + 1e-9, (sr, name)



def test_mel_against_reference_firmware():
    fw = _firmware()
    from oracle.melbank import mel_filterbank

    for sr in (22050, 24000):
        ref = fw.mel_matrix(64, sr, 150.0, float(sr // 2))
        ours = mel_filterbank(sr, 512, 64, 150.0, float(sr // 2))
        assert np.abs(ref - ours).max() < 5e-7


def test_c_port_matches_numpy_oracle(netspec):
    from oracle import cport, float_graph, stft

    if not os.path.isfile(cport.CPU_LIB):
        pytest.skip("oracle/_build/liboracle_cpu.so not built")
    x = synth_chunks(4)
    scores, logits, S = cport.CpuFloatPath(netspec)(x)
    Sref = np.stack([stft.hybrid_spectrogram(a) for a in x])
    assert np.abs(S - Sref).max() < 2e-6
    p, l = float_graph.forward(netspec, Sref[..., None], np.float64, return_logits=True)
    assert np.abs(scores - p).max() < 1e-5
    assert min(cosine(logits[b], l[b]) for b in range(4)) > 1 - 1e-6


# ------------------------------------------------------------- audio ingest
@pytest.mark.parametrize("sr_in", [48000, 44100, 22050, 32000, 16000, 8000, 96000, 11025])
def test_ingest_resampler_is_scipy_bit_for_bit(sr_in):
    """oracle.ingest restates scipy.signal.resample_poly (the call the reference makes, audio/io.py:30): same bits."""
    from scipy.signal import resample_poly

    from oracle import ingest

    up, down = ingest.rates_to_ratio(sr_in, 24000)
    sig = fixture_signals(sr_in, 1.0)
    rng = np.random.default_rng(sr_in)
    for x in (sig["sine"], sig["chirp"], sig["noise"], rng.standard_normal(sr_in // 3 + 17).astype(np.float32), np.ones(5, np.float32)):
        want = resample_poly(x, up, down)
        got = ingest.resample_poly_f32(x, up, down)
        assert want.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_ingest_channel_mean_is_numpy_bit_for_bit():
    from oracle import ingest

    rng = np.random.default_rng(8)
    for ch in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 23):
        f = (rng.standard_normal((4000, ch)) * rng.choice([1e-3, 1.0, 1e3], size=(4000, ch))).astype(np.float32)
        assert np.array_equal(ingest.mono_mean(f).view(np.uint32), f.mean(axis=1).astype(np.float32).view(np.uint32)), ch


def test_ingest_window_and_chunks_equal_host_io(tmp_path):
    """Whole ingest restatement vs birdnet_stm32.audio.io (numpy + scipy calls in the reference's order) from WAV files."""
    from birdnet_stm32.audio import io
    from oracle import ingest

    rng = np.random.default_rng(21)
    for sr, secs, ch, overlap in [(44100, 7.4, 2, 0.0), (48000, 2.0, 1, 0.0), (24000, 6.5, 2, 1.0), (22050, 9.1, 1, 2.95)]:
        n = int(sr * secs)
        pcm = np.clip(np.rint(rng.standard_normal((n, ch)) * 6000), -32768, 32767).astype(np.int16)
        p = str(tmp_path / "x.wav")
        payload = pcm.tobytes()
        import struct

        with open(p, "wb") as fh:
            fh.write(struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, 1, ch, sr, sr * 2 * ch,
                                 2 * ch, 16, b"data", len(payload)) + payload)
        want = np.asarray(io.load_audio_file(p, 24000, 30, 3.0, overlap), np.float32)
        y = ingest.ingest_window(pcm.astype(np.float32) / 32768.0, sr, 24000)
        got = ingest.split_chunks(y, 24000, 3.0, overlap)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("native", [False, True])
def test_c_int8_port_is_identical_to_the_numpy_interpreter(native):
    """oracle/c/oracle_i8.c (the INT8 CPU baseline) against oracle/int8_graph.py: every tensor of the shipped graph, bit for bit — the
    portable build (scalar loops) and the -march=native build of the same file (on AVX-512 / VNNI hosts: vpdpbusd 1x1 convolutions,
    sixteen-lane MultiplyByQuantizedMultiplier), which is the one bench.py times."""
    from birdnet_stm32.models._tflite_reader import load_tflite
    from oracle import cport, stft
    from oracle.int8_graph import Int8Interpreter

    from conftest import TFLITE_PATH, synth_chunks

    if not (os.path.isfile(cport.I8_LIB) and os.path.isfile(cport.CPU_LIB)):
        pytest.skip("oracle C libraries not built (python -c 'import __graft_entry__ as g; g.build()')")
    if native and not cport.build_native():
        pytest.skip("no native build of the oracle's C port on this host")
    model = load_tflite(TFLITE_PATH)
    x = synth_chunks(3)
    S = np.stack([stft.hybrid_spectrogram(a) for a in x])[..., None]
    ref, env = Int8Interpreter(model).invoke(S, return_all=True)
    port = cport.CpuInt8Path(model, native=native)
    got, env_c = port.invoke(S, return_all=True)
    assert np.array_equal(got, ref)
    for k, v in env.items():
        assert np.array_equal(np.asarray(v), np.asarray(env_c[k])), f"tensor {k} differs"
    assert np.abs(port.spectrogram(x, 281, 256) - S).max() < 1e-6  # the C STFT of the float port
    # the same graph as ONE C program per chunk (oi_program_run: bench.py's cpu_baseline since round 5): every materialised tensor and the
    # scores, bit for bit, on tone + noise chunks and on random spectrograms, for a batch that gives every thread several chunks
    prog = cport.CpuInt8Program(model, native=native)
    S2 = np.concatenate([S, np.random.default_rng(5).random((29, 257, 256, 1), dtype=np.float32)])
    ref2, env2 = Int8Interpreter(model).invoke(S2, return_all=True)
    got2, env_p = prog.invoke(S2, return_all=True)
    assert np.array_equal(got2, ref2) and len(env_p) >= 40
    for k, v in env_p.items():
        assert np.array_equal(v, np.asarray(env2[k]).reshape(v.shape)), f"tensor {k} differs in the C program"
    assert np.array_equal(prog.invoke(S2[:1]), ref2[:1])


def test_int8_primitives_match_gemmlowp_definitions_on_edge_cases():
    """oracle/int8_graph.py's fixed-point primitives (and the lowering pass's own copies in models/_quant.py) against the published
    gemmlowp definitions written out on Python integers (tests/conftest.py): INT32_MIN x INT32_MIN saturation, negative-half ties of the
    rounding shift, shifts 0..31, left shifts with wrap-around, and QuantizeMultiplier where the mantissa rounds up to 2^31."""
    from birdnet_stm32.models import _quant as qz
    from conftest import I32_MAX, I32_MIN, mbqm_def, rdivpot_def, srdhm_def
    from oracle import int8_graph as ig

    rng = np.random.default_rng(8)
    edge = [0, 1, -1, 2, -2, 3, -3, I32_MAX, I32_MIN, I32_MIN + 1, I32_MAX - 1, 1 << 30, -(1 << 30), (1 << 30) - 1, 12345, -12345]
    pairs = [(a, b) for a in edge for b in edge] + list(zip(rng.integers(I32_MIN, I32_MAX, 4000).tolist(), rng.integers(I32_MIN, I32_MAX, 4000).tolist()))
    a, b = np.array([p[0] for p in pairs], np.int64), np.array([p[1] for p in pairs], np.int64)
    want = np.array([srdhm_def(*p) for p in pairs], np.int64)
    assert np.array_equal(ig.srdhm(a, b), want) and np.array_equal(qz._high_mul(a, b), want)
    assert srdhm_def(I32_MIN, I32_MIN) == I32_MAX and srdhm_def(I32_MIN, I32_MAX) == -I32_MAX
    xs = edge + rng.integers(I32_MIN, I32_MAX, 2000).tolist()
    for e in (0, 1, 2, 3, 7, 8, 20, 30, 31):
        x = np.array(xs, np.int64)
        want = np.array([rdivpot_def(v, e) for v in xs], np.int64)
        assert np.array_equal(ig.rounding_divide_by_pot(x, e), want) and np.array_equal(qz._round_shift(x, e), want), e
    # ties: -(2k+1) * 2^(e-1) rounds away from zero (towards -inf), the positive tie up
    for e in (1, 4, 9):
        h = 1 << (e - 1)
        assert [rdivpot_def(v, e) for v in (-h, h, -3 * h, 3 * h)] == [-1, 1, -2, 2]
        assert ig.rounding_divide_by_pot(np.array([-h, h, -3 * h, 3 * h]), e).tolist() == [-1, 1, -2, 2]
    cases = [(x, m, s) for x in (5, -5, 1000, -1000, (1 << 29) + 3, -(1 << 29) - 3) for m in (1 << 30, 1518500250, I32_MAX, -(1 << 30), -1518500250)
             for s in (-31, -12, -1, 0, 1, 3)]
    for x, m, s in cases:
        want = mbqm_def(x, m, s)
        if abs(x << max(s, 0)) < (1 << 31):  # the numpy forms do not wrap a left shift; TFLite's int32 arithmetic would
            assert int(ig.mbqm(np.array([x]), m, s)[0]) == want == int(qz.requantize(np.array([x]), m, s)[0]), (x, m, s)
    # QuantizeMultiplier: mantissa in [2^30, 2^31), rounding up to 2^31 renormalises, tiny and huge reals saturate
    for real, want in ((1.0, (1 << 30, 1)), (0.5, (1 << 30, 0)), (0.75, (3 << 29, 0)), (1.0 - 2.0**-33, (1 << 30, 1)), (0.0, (0, 0)),
                       (2.0**-40, (0, 0)), (2.0**31, (I32_MAX, 30)), (6.0 / 255.0 * 0.0123 / (6.0 / 255.0), None)):
        got = ig.quantize_multiplier(real)
        assert got == qz.quantize_multiplier(real)
        if want is not None:
            assert got == want, (real, got)
        else:
            assert (1 << 30) <= got[0] < (1 << 31) and abs(got[0] * 2.0 ** (got[1] - 31) - real) <= real * 2.0**-31


def test_numpy_cabs_restatement_is_pinned_to_the_installed_numpy():
    """``oracle.stft.cabs_numpy_simd`` (what csrc/bn_quant_in.h: numpy_cabsf reproduces on the GPU) == ``np.abs`` of complex64 on this
    machine's numpy, over magnitudes spanning 20 octaves, exact zeros and axis-aligned values; and it is NOT the correctly rounded
    magnitude, so the restatement cannot be replaced by one."""
    from oracle.stft import cabs_numpy_simd

    rng = np.random.default_rng(0)
    n = 5_000_000
    z = (rng.standard_normal(n) * np.exp(rng.uniform(-10, 10, n)) + 1j * rng.standard_normal(n) * np.exp(rng.uniform(-10, 10, n))).astype(np.complex64)
    z[:4] = [0, 1, 1j, -3 - 4j]
    got, want = cabs_numpy_simd(z), np.abs(z)
    assert want.dtype == np.float32
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} of {n} differ, first {z[bad[0]]}: {got[bad[0]]} vs {want[bad[0]]}"
    exact = np.sqrt(z.real.astype(np.float64) ** 2 + z.imag.astype(np.float64) ** 2).astype(np.float32)
    assert 0.2 < (want != exact).mean() < 0.5


def test_benchmark_json_report_is_pinned_to_the_reference(tmp_path, capsys):
    """``cli/evaluate.py: save_benchmark_json`` writes, byte for byte, the file the reference's ``evaluation/reporting.py:192-236`` writes for
    the same metric dicts (tests/golden/reference_reporting.json: generated by importing the reference, tests/golden/make_golden.py), and
    prints the same line."""
    import json

    from birdnet_stm32.cli.evaluate import save_benchmark_json

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_reporting.json")))
    assert set(gold["cases"]) == set(gold["reference"]) and len(gold["cases"]) >= 3
    for name, c in gold["cases"].items():
        out = tmp_path / name / "sub" / "report.json"
        save_benchmark_json(c["metrics"], c["classes"], c["model_path"], str(out), config=c["config"], species_data=c["species_data"])
        assert out.read_text() == gold["reference"][name]["file"], name
        assert capsys.readouterr().out.replace(str(out), "<out>") == gold["reference"][name]["stdout"], name


# ----------------------------------------------------------------------- the other published forms of int8 MEAN and LOGISTIC
def _shipped_tail_inputs(n: int):
    """(model, interpreter env up to the MEAN of the shipped graph, index of the MEAN operator) for n spectrograms: synthetic chunks + random ones."""
    from birdnet_stm32.models._tflite_reader import load_tflite
    from oracle import cport, stft
    from oracle.int8_graph import Int8Interpreter

    from conftest import TFLITE_PATH, synth_chunks

    model = load_tflite(TFLITE_PATH)
    rng = np.random.default_rng(3)
    n_syn = min(n // 2, 64)
    S = np.concatenate([np.stack([stft.hybrid_spectrogram(a) for a in synth_chunks(n_syn, seed=5)])[..., None],
                        (rng.random((n - n_syn, 257, 256, 1), dtype=np.float32) ** 3).astype(np.float32)])
    interp = cport.CpuInt8Path(model).interp if os.path.isfile(cport.I8_LIB) else Int8Interpreter(model)
    _, env = interp.invoke(S, return_all=True)
    mean_op = next(i for i, op in enumerate(model.ops) if op.name == "MEAN")
    return model, env, mean_op


def test_int8_logistic_forms_agree_on_every_input_of_the_shipped_head():
    """The builtin int8 LOGISTIC is a float32 table; the fixed-point kernel (gemmlowp, what TFLite-Micro runs) is the other published form.
    A LOGISTIC is a function of ONE byte, so all 256 inputs of the shipped head's quantisation settle it: identical."""
    from birdnet_stm32.models._tflite_reader import load_tflite
    from oracle.int8_graph import Int8Interpreter, logistic_int8_fixed

    from conftest import TFLITE_PATH

    model = load_tflite(TFLITE_PATH)
    it = Int8Interpreter(model)
    ops = [op for op in model.ops if op.name == "LOGISTIC"]
    assert len(ops) == 1
    s_in, z_in = it._q(ops[0].inputs[0])
    q = np.arange(-128, 128)
    lut = it.logistic_lut(ops[0])
    fixed = logistic_int8_fixed(q, s_in, z_in)
    assert np.array_equal(lut[q + 128], fixed)
    # and against the real-valued sigmoid rounded half away from zero, over a range of input scales (both forms are within one step of it)
    for s in (0.01, 0.0473, 0.11, 0.25, 0.6):
        ref = np.clip(np.floor(256.0 / (1.0 + np.exp(-s * (q.astype(np.float64) - z_in))) + 0.5) - 128, -128, 127)
        assert np.abs(logistic_int8_fixed(q, s, z_in).astype(np.int64) - ref).max() <= 1


def test_int8_mean_forms_differ_by_one_step_on_a_few_per_cent_of_the_pooled_bytes():
    """Integer MEAN (count folded into the multiplier: the default, reduce.h) against the float-arithmetic QuantizedMeanOrSum on the shipped graph:
    NOT identical — a few per cent of the pooled bytes differ by one step, which moves some head bytes and a rare top-1 label.  That is why the
    form is a switch of the oracle (``mean_form``) AND of the device plan (``lower_i8(mean_form=...)``); the numbers below are the evidence."""
    from oracle.int8_graph import Int8Interpreter

    n = 384
    model, env, mean_op = _shipped_tail_inputs(n)
    a, ea = Int8Interpreter(model).invoke(None, return_all=True, resume=(env, mean_op))
    b, eb = Int8Interpreter(model, mean_form="float").invoke(None, return_all=True, resume=(env, mean_op))
    pooled_t = model.ops[mean_op].outputs[0]
    pa, pb = np.asarray(ea[pooled_t]).astype(np.int64), np.asarray(eb[pooled_t]).astype(np.int64)
    assert np.abs(pa - pb).max() == 1                      # never more than one step ...
    frac = float((pa != pb).mean())
    assert 0.005 < frac < 0.15, frac                       # ... on a few per cent of the bytes (measured: ~4 %)
    flips = int((a.argmax(axis=1) != b.argmax(axis=1)).sum())
    assert np.abs(a - b).max() <= 8.0 / 256.0              # scores move by a few output steps at most
    assert flips <= n // 20, flips
    # resumed runs are the full runs: the integer form from the resume point equals the env it resumed from
    assert all(np.array_equal(np.asarray(ea[k]), np.asarray(env[k])) for k in env)
    print(f"MEAN forms on {n} spectrograms: {frac:.4f} of the pooled bytes differ by one step, {int((np.asarray(ea[model.outputs[0]]) != np.asarray(eb[model.outputs[0]])).sum())} "
          f"of {a.size} scores differ, top-1 flips {flips}")
