"""Own post-training quantisation: calibrator + INT8 exporter without TensorFlow.

The reference's converter cannot run here, so the checks are known answers and closed loops:

* re-quantising the SHIPPED float checkpoint must reproduce the shipped ``.tflite``'s int8 weights bit for bit (all 30
  weight tensors) — the same per-channel rule, BatchNorm folding and dead-channel floor as the reference converter;
* activation scales calibrated on synthetic chunks land within a factor 2 of the shipped ones, ReLU6 ranges exactly on 6/255;
* the INT8 oracle on the new graph tracks the float oracle like the shipped graph does (reference bar: cosine > 0.8,
  tests/test_conversion.py:112-113 of the reference; asserted > 0.98);
* (GPU) the device plan lowered from the new graph is bit-identical to the INT8 oracle executing the same graph.
"""

import numpy as np
import pytest

from conftest import KERAS_PATH, TFLITE_PATH, cosine, synth_chunks


@pytest.fixture(scope="module")
def ptq():
    from birdnet_stm32.conversion.quantize import requantize_like
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._tflite_reader import load_tflite
    from oracle import stft

    spec, tpl = load_keras_archive(KERAS_PATH), load_tflite(TFLITE_PATH)
    x = np.stack([stft.hybrid_spectrogram(a) for a in synth_chunks(10, seed=3)])[..., None].astype(np.float32)
    new = requantize_like(tpl, spec, lambda: ([x[i : i + 1]] for i in range(6)))
    return spec, tpl, new, x


def test_requantised_weights_equal_the_shipped_tflite(ptq):
    spec, tpl, new, _ = ptq
    n = 0
    for op in tpl.ops:
        if op.name in ("CONV_2D", "DEPTHWISE_CONV_2D", "FULLY_CONNECTED"):
            a, b = tpl.tensors[op.inputs[1]], new.tensors[op.inputs[1]]
            assert np.array_equal(a.data, b.data), (op.index, op.name)
            live = a.scale > 1e-7  # dead channels: the converter reports its own tiny scale, ours is the 5e-7 / 127 floor
            assert np.allclose(a.scale[live], b.scale[live], rtol=1e-6)
            bias = new.tensors[op.inputs[2]]
            assert np.allclose(bias.scale, np.float32(new.tensors[op.inputs[0]].scale[0]) * b.scale, rtol=1e-6)
            n += 1
    assert n == 30


def test_calibrated_activation_parameters(ptq):
    _, tpl, new, _ = ptq
    assert new.tensors[11].scale[0] == pytest.approx(1 / 255, rel=1e-6) and new.tensors[11].zero_point[0] == -128
    relu6 = [op.outputs[0] for op in tpl.ops if op.options.get("activation") == "relu6" and op.index >= 24]
    assert len(relu6) >= 20
    full = 0
    for ti in relu6:  # a ReLU6 output can never exceed [0, 6]; most saturate on the calibration chunks like in the shipped model
        s = float(new.tensors[ti].scale[0])
        assert 0.2 * 6 / 255 < s <= 6 / 255 * (1 + 1e-6) and new.tensors[ti].zero_point[0] == -128
        full += abs(s - 6 / 255) < 1e-8
    assert full >= len(relu6) // 2
    logistic = next(op for op in tpl.ops if op.name == "LOGISTIC").outputs[0]
    assert new.tensors[logistic].scale[0] == pytest.approx(1 / 256) and new.tensors[logistic].zero_point[0] == -128
    for op in tpl.ops:
        if op.name in ("TRANSPOSE", "STRIDED_SLICE", "CONCATENATION") and tpl.tensors[op.outputs[0]].is_quantized:
            assert new.tensors[op.outputs[0]].scale[0] == new.tensors[op.inputs[0]].scale[0]
            assert new.tensors[op.outputs[0]].zero_point[0] == new.tensors[op.inputs[0]].zero_point[0]
    ratios = [float(new.tensors[t.index].scale[0] / t.scale[0]) for t in tpl.tensors if t.data is None and t.is_quantized and t.scale.size == 1]
    assert len(ratios) > 40 and 0.2 < min(ratios) and max(ratios) < 2.0  # same order as the shipped calibration (other data)


def test_requantised_graph_tracks_the_float_model(ptq):
    from oracle import float_graph
    from oracle.int8_graph import Int8Interpreter

    spec, tpl, new, x = ptq
    held_out = x[6:]
    ref, ref_logits = float_graph.forward(spec, held_out, np.float32, return_logits=True)
    got, env = Int8Interpreter(new).invoke(held_out, return_all=True)
    fc = next(op for op in new.ops if op.name == "FULLY_CONNECTED").outputs[0]
    t = new.tensors[fc]
    logits = (env[fc].astype(np.float32) - t.zero_point[0]) * t.scale[0]
    for b in range(held_out.shape[0]):
        assert cosine(got[b], ref[b]) > 0.98
        assert cosine(logits[b], ref_logits[b]) > 0.999
        assert ref_logits[b].argmax() in np.argsort(logits[b])[-3:]  # synthetic noise: the top logits are near ties


def test_fake_quantize_weights_and_metrics():
    from birdnet_stm32.conversion.validate import cosine_similarity, pearson_correlation
    from birdnet_stm32.training.qat import fake_quantize_weights

    rng = np.random.default_rng(0)
    w = rng.standard_normal((3, 3, 8, 16)).astype(np.float32)
    q = fake_quantize_weights(w)
    step = (w.max(axis=(0, 1, 2)) - w.min(axis=(0, 1, 2))) / 255
    assert q.dtype == np.float32 and np.abs(q - w).max() <= step.max() / 2 + 1e-6
    assert np.array_equal(fake_quantize_weights(q), q) or np.abs(fake_quantize_weights(q) - q).max() < 1e-6  # idempotent
    for c in range(16):
        assert len(np.unique(np.round((q[..., c] - w[..., c].min()) / step[c]))) <= 256
    assert fake_quantize_weights(w, per_channel=False).shape == w.shape
    assert cosine_similarity(np.zeros(4), np.zeros(4)) == 1.0 and cosine_similarity(np.zeros(4), np.ones(4)) == 0.0
    assert cosine_similarity(np.ones(4), 2 * np.ones(4)) == pytest.approx(1.0)
    assert pearson_correlation(np.arange(5.0), 3 * np.arange(5.0) + 1) == pytest.approx(1.0) and pearson_correlation(np.ones(3), np.ones(3)) == 1.0


def test_topology_mismatch_and_empty_calibration_are_rejected(ptq):
    from birdnet_stm32.conversion.quantize import requantize_like
    from birdnet_stm32.models import build_model

    _, tpl, _, x = ptq
    other = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=22050, chunk_duration=3, embeddings_size=256, num_classes=100)
    with pytest.raises((ValueError, NotImplementedError)):
        requantize_like(tpl, other, lambda: iter([[x[:1]]]))  # inverted residual + SE: no INT8 kernels / different topology
    spec = ptq[0]
    with pytest.raises(ValueError, match="empty"):
        requantize_like(tpl, spec, lambda: iter([]))


@pytest.mark.gpu
def test_requantised_plan_is_bit_exact_on_the_gpu(ptq):
    """lower_i8(new graph) on the MI355X == the INT8 oracle on the same graph, and it validates against the float runner."""
    from birdnet_stm32.conversion.validate import validate_models
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models.runners import HipRunner, load_model_runner
    from oracle.int8_graph import Int8Interpreter

    spec, tpl, new, x = ptq
    runner = HipRunner(lower_i8(new), max_batch=16)
    got = runner.predict(x)
    assert np.array_equal(got, Int8Interpreter(new).invoke(x))
    f32 = load_model_runner(KERAS_PATH, max_batch=16)
    stats = validate_models(f32, runner, lambda: ([x[i : i + 1]] for i in range(6, 10)))
    assert stats["cosine_mean"] > 0.98 and stats["mae_mean"] < 0.01
    f32.close()
    runner.close()


def test_requantised_graph_round_trips_through_a_tflite_file(ptq, tmp_path):
    """patch_tflite writes a real .tflite (template bytes with new payloads); reading it back gives the same graph, and
    load_model_runner-style lowering accepts it."""
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models._tflite_reader import load_tflite, patch_tflite

    _, tpl, new, _ = ptq
    raw = open(TFLITE_PATH, "rb").read()
    assert patch_tflite(raw, tpl) == raw  # writing the template's own values changes nothing
    out = tmp_path / "requantised.tflite"
    out.write_bytes(patch_tflite(raw, new))
    back = load_tflite(str(out))
    assert len(back.ops) == len(new.ops) and back.constant_bytes() == tpl.constant_bytes()
    for a, b in zip(new.tensors, back.tensors):
        assert np.array_equal(a.scale, b.scale) and np.array_equal(a.zero_point, b.zero_point) and a.quantized_dimension == b.quantized_dimension
        assert (a.data is None) == (b.data is None) and (a.data is None or np.array_equal(a.data, b.data))
    assert lower_i8(back).to_blob() == lower_i8(new).to_blob()
    with pytest.raises(ValueError, match="TFL3"):
        patch_tflite(b"\x00" * 64, new)


def test_template_mismatch_is_its_own_exception():
    """``requantize_like`` signals "this is not the template's topology" with ``TopologyMismatch`` — the only failure on which the convert CLI
    switches to the template-free exporter (and only for its default template); every other failure keeps its own type and propagates."""
    from birdnet_stm32.conversion.quantize import TopologyMismatch, requantize_like
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._tflite_reader import load_tflite

    template = load_tflite(TFLITE_PATH)
    other = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=10, randomize_bn=True, seed=7)
    with pytest.raises(TopologyMismatch):
        requantize_like(template, other, lambda: iter(()))
    assert issubclass(TopologyMismatch, ValueError)
    with pytest.raises(ValueError, match="representative dataset is empty") as e:  # the right topology, no calibration data: NOT a mismatch
        requantize_like(template, load_keras_archive(KERAS_PATH), lambda: iter(()))
    assert not isinstance(e.value, TopologyMismatch)


@pytest.mark.gpu
def test_cli_convert_end_to_end(tmp_path):
    """`python -m birdnet_stm32 convert` on the shipped checkpoint with WAV calibration data: writes a .tflite, validates it on the
    GPU against the float model, and the written file runs through load_model_runner like the shipped one."""
    import json
    import subprocess
    import sys

    from birdnet_stm32.audio.io import save_wav
    from conftest import CONFIG_PATH, PKG

    cfg = json.load(open(CONFIG_PATH))
    cfg.update(sample_rate=24000, hop_length=281)
    (tmp_path / "model_cfg.json").write_text(json.dumps(cfg))
    x = synth_chunks(12, seed=9)
    for i in range(12):
        d = tmp_path / "data" / cfg["class_names"][i % 3]
        d.mkdir(parents=True, exist_ok=True)
        save_wav(np.concatenate([x[i], x[(i + 1) % 12], x[(i + 2) % 12]]), str(d / f"c{i}.wav"), 24000)
    out, rep = tmp_path / "q.tflite", tmp_path / "report.json"
    cmd = [sys.executable, "-m", "birdnet_stm32", "convert", "--checkpoint_path", KERAS_PATH, "--model_config", str(tmp_path / "model_cfg.json"),
           "--data_path_train", str(tmp_path / "data"), "--num_samples", "9", "--validate_samples", "6", "--output_path", str(out),
           "--min_cosine_sim", "0.9", "--report_json", str(rep)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(__import__("os").environ, PYTHONPATH=PKG), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "Cosine similarity check passed" in res.stdout and out.stat().st_size == __import__("os").path.getsize(TFLITE_PATH)
    report = json.loads(rep.read_text())
    assert report["validation"]["cosine_mean"] > 0.9 and report["quantization"] == "ptq"
    from birdnet_stm32.models.runners import load_model_runner
    from oracle import stft
    from oracle.int8_graph import Int8Interpreter
    from birdnet_stm32.models._tflite_reader import load_tflite

    S = np.stack([stft.hybrid_spectrogram(a) for a in x[:3]])[..., None]
    r = load_model_runner(str(out), max_batch=4)
    assert np.array_equal(r.predict(S), Int8Interpreter(load_tflite(str(out))).invoke(S))
    r.close()
    bad = subprocess.run(cmd[:4] + ["--checkpoint_path", KERAS_PATH, "--quantization", "dynamic"], capture_output=True, text=True,
                         env=dict(__import__("os").environ, PYTHONPATH=PKG), timeout=300)
    assert bad.returncode != 0 and "dynamic" in (bad.stderr + bad.stdout)


# ------------------------------------------------------------------ template-free exporter (SE / inverted residuals / softmax)
EXPORT_TOPOLOGIES = {
    "ir_se_softmax": dict(),
    "ds_se_emb_sigmoid": dict(use_inverted_residual=False, use_se=True, embeddings_size=128, class_activation="sigmoid"),
    "alpha1.5_pcen": dict(alpha=1.5, mag_scale="pcen", num_classes=37),
    "ir_deep_narrow_nomag": dict(use_se=False, depth_multiplier=2, alpha=0.5, mag_scale="none"),
    # current hybrid frontends: per-sample max normalisation (REDUCE_MAX -> ADD epsilon -> DIV) in front of the PWL
    "maxnorm_pwl_ds": dict(use_inverted_residual=False, use_se=False, frontend_norm=True),
    "maxnorm_nomag_ir": dict(use_se=False, mag_scale="none", alpha=0.5, frontend_norm=True),
    # raw frontend (QUANTIZE of the waveform -> [PAD] -> RESHAPE -> CONV_2D 1x16 strided VALID): BASELINE configs[4]'s frontend with PCEN, and a
    # geometry whose chunk is shorter than stride * (W - 1) + 16 samples (symmetric zero padding in front of the filterbank)
    "raw_pcen_ir_se": dict(audio_frontend="raw", mag_scale="pcen", chunk_duration=2, alpha=0.5),
    # the same at the metric's chunk length: 3 s @ 24 kHz = 72000 samples, stride 282, no PAD (the reference's T < 65536 guard lifted)
    "raw_pcen_ir_se_3s": dict(audio_frontend="raw", mag_scale="pcen", chunk_duration=3, alpha=0.5, raw_length_limit=None),
    # attention pooling instead of the global average (reference models/blocks.py:136-159): RESHAPE -> FULLY_CONNECTED C -> 1 -> int8 SOFTMAX over
    # the positions -> MUL -> SUM, sigmoid head
    "ds_attnpool_sigmoid": dict(use_inverted_residual=False, use_se=False, use_attention_pooling=True, class_activation="sigmoid", alpha=0.5),
    "raw_pad_nomag_ds": dict(audio_frontend="raw", mag_scale="none", sample_rate=6000, chunk_duration=0.5, spec_width=128, num_mels=32,
                             use_se=False, use_inverted_residual=False),
}


def _export(kw, n_cal=4, seed=0):
    from birdnet_stm32.conversion.export import convert_netspec_to_int8
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._tflite_reader import parse_tflite
    from birdnet_stm32.models._tflite_writer import write_tflite
    from oracle import stft

    args = dict(num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=10, randomize_bn=True, seed=7)
    args.update(kw)
    norm = bool(args.pop("frontend_norm", False))
    spec = build_model("dscnn", **args)
    spec.frontend.attrs["norm"] = norm
    chunks = synth_chunks(n_cal + 3, sr=args["sample_rate"], seconds=args["chunk_duration"], seed=seed + 3)
    if args.get("audio_frontend") == "raw":  # the host side of the raw frontend: x / (max |x| + 1e-6) (reference evaluation/metrics.py:62-69)
        x = np.stack([a / (np.abs(a).max() + 1e-6) for a in chunks])[..., None].astype(np.float32)
    else:
        x = np.stack([stft.hybrid_spectrogram(a, spec_width=args["spec_width"]) for a in chunks])[..., None].astype(np.float32)
    graph = convert_netspec_to_int8(spec, lambda: ([x[i : i + 1]] for i in range(n_cal)), frontend_norm=norm)
    raw = write_tflite(graph)
    return spec, parse_tflite(raw), raw, x


@pytest.mark.parametrize("name", list(EXPORT_TOPOLOGIES))
def test_exported_int8_graph_tracks_the_float_model(name):
    """build_model(...) -> own PTQ -> .tflite bytes -> reader -> INT8 oracle: squeeze-excite (MEAN / FULLY_CONNECTED / LOGISTIC / MUL),
    inverted residuals (ADD without activation), embedding conv, PWL / PCEN / no magnitude scaling, sigmoid and softmax heads.  The
    quantised graph tracks the float oracle (reference bar: cosine > 0.8, tests/test_conversion.py:112-113 of the reference)."""
    from oracle import float_graph
    from oracle.int8_graph import Int8Interpreter

    spec, model, raw, x = _export(EXPORT_TOPOLOGIES[name])
    assert raw[4:8] == b"TFL3" and model.ops[0].name == "QUANTIZE" and model.tensors[model.inputs[0]].dtype == np.float32
    names = {op.name for op in model.ops}
    assert names <= {"QUANTIZE", "TRANSPOSE", "CONV_2D", "DEPTHWISE_CONV_2D", "ADD", "MUL", "MEAN", "FULLY_CONNECTED", "LOGISTIC", "DEQUANTIZE", "SOFTMAX",
                     "REDUCE_MAX", "DIV", "RESHAPE", "PAD", "SUM"}
    if EXPORT_TOPOLOGIES[name].get("use_attention_pooling"):
        assert [op.name for op in model.ops][-10:-3] == ["RESHAPE", "FULLY_CONNECTED", "RESHAPE", "SOFTMAX", "RESHAPE", "MUL", "SUM"]
        sm = next(op for op in model.ops if op.name == "SOFTMAX")
        assert model.tensors[sm.outputs[0]].dtype == np.int8 and model.tensors[sm.outputs[0]].scale[0] == pytest.approx(1 / 256) and model.tensors[sm.outputs[0]].zero_point[0] == -128
        assert np.array_equal(Int8Interpreter(model, softmax_form="lut").invoke(x).argmax(1), Int8Interpreter(model).invoke(x).argmax(1))
    if EXPORT_TOPOLOGIES[name].get("frontend_norm"):
        assert {"REDUCE_MAX", "DIV"} <= names
        div = next(op for op in model.ops if op.name == "DIV")
        rmax = next(op for op in model.ops if op.name == "REDUCE_MAX")
        assert model.tensors[div.outputs[0]].zero_point[0] == -128 and model.tensors[div.outputs[0]].scale[0] == pytest.approx(1 / 255, rel=1e-3)
        assert model.tensors[rmax.outputs[0]].scale[0] == model.tensors[rmax.inputs[0]].scale[0]  # converter rule: same parameters
    if EXPORT_TOPOLOGIES[name].get("use_se", True):
        assert {"MUL", "MEAN", "LOGISTIC"} <= names
    kernels = {op.inputs[1] for op in model.ops if op.name in ("CONV_2D", "DEPTHWISE_CONV_2D", "FULLY_CONNECTED")}
    for t in model.tensors:  # converter rules: int8 activations per tensor, weights symmetric per channel, biases int32 at s_in * s_w
        if t.dtype == np.int8 and t.data is not None and t.index in kernels:  # (other int8 constants — the epsilon — are affine like activations)
            assert np.all(t.zero_point == 0) and np.abs(t.data.astype(np.int32)).max() <= 127
    for op in model.ops:
        if op.name in ("CONV_2D", "DEPTHWISE_CONV_2D", "FULLY_CONNECTED"):
            w, b = model.tensors[op.inputs[1]], model.tensors[op.inputs[2]]
            assert b.dtype == np.int32 and np.allclose(b.scale, np.float32(model.tensors[op.inputs[0]].scale[0]) * w.scale, rtol=1e-6)
    for op in model.ops:
        if op.name == "LOGISTIC":
            assert model.tensors[op.outputs[0]].scale[0] == pytest.approx(1 / 256) and model.tensors[op.outputs[0]].zero_point[0] == -128
    got = Int8Interpreter(model).invoke(x)
    ref = float_graph.forward(spec, x, np.float64)
    assert got.shape == ref.shape and np.all(np.isfinite(got))
    for b in range(x.shape[0]):
        assert cosine(got[b], ref[b]) > 0.98, (name, b, cosine(got[b], ref[b]))
    if spec.layers[-1].attrs["activation"] == "softmax":
        assert np.allclose(got.sum(axis=1), 1.0, atol=1e-5)


def test_int8_softmax_restatements():
    """The int8 SOFTMAX of attention pooling in TFLite's two published forms — the reference kernel (gemmlowp fixed point: exp_on_negative_values,
    Newton-Raphson reciprocal) and the optimized kernel (float32 exponent table): each within one output step of the real softmax, the two
    within one step of each other (they differ on ~1e-4 of the outputs, which is why the form is a switch), the product's tables
    (models/_quant.py: softmax_tables) equal to the oracle's arithmetic, and the fixed-point exponential accurate to 3e-7.
    PARITY UNPINNED: restated from the published kernels (softmax.h, fixedpoint.h), no reference vectors exist."""
    from birdnet_stm32.models import _quant as qz
    from oracle import int8_graph as ig

    rng = np.random.default_rng(0)
    a = -rng.integers(0, 31 << 26, 20000)
    assert np.abs(ig.exp_on_negative_values_q5(a) / 2.0**31 - np.exp(a / 2.0**26)).max() < 3e-7
    assert np.array_equal(ig.exp_on_negative_values_q5(a), qz._exp_q5_to_q31(a))
    for s_in in (0.004, 0.02, 0.1, 0.3, 1.0):
        x = rng.integers(-128, 128, (3000, 32))
        x[:10] = x[:10, :1]  # rows of equal scores
        f = np.exp((x - x.max(-1, keepdims=True)) * np.float64(np.float32(s_in)))
        want = np.clip(np.round(f / f.sum(-1, keepdims=True) * 256) - 128, -128, 127)
        fixed, lut = ig.softmax_int8_fixed(x, s_in), ig.softmax_int8_lut(x, s_in)
        assert np.abs(fixed - want).max() <= 1 and np.abs(lut - want).max() <= 1 and np.abs(fixed.astype(int) - lut).max() <= 1
        tab = qz.softmax_tables(s_in, 1.0, "fixed")
        d = x.max(-1, keepdims=True) - x
        mult, shift, diff_min = ig.softmax_fixed_params(s_in, 1.0)
        assert np.array_equal(tab[0][d] >= 0, -d >= diff_min)
        assert np.array_equal(np.where(tab[0][d] >= 0, tab[0][d], 0), np.where(-d >= diff_min, ig.exp_on_negative_values_q5(ig.srdhm(np.where(-d >= diff_min, -d, 0) << shift, mult)), 0))
        assert np.array_equal(qz.softmax_tables(s_in, 1.0, "lut").view(np.float32), np.exp((np.float32(-np.float32(s_in)) * np.arange(256, dtype=np.float32)).astype(np.float32)))


def test_int8_div_and_reduce_max_restatements():
    """The INT8 DIV of the max-normalised frontend: the product's byte table (models/_quant.py: div_table) equals the oracle's
    restatement of TFLite's DivElementwise on all 65 536 byte pairs, both stay within one output step of real division, and the
    fixed-point reciprocal (gemmlowp's Newton-Raphson, three iterations) is accurate to 2^-27.  PARITY UNPINNED: restated from the
    published kernels (div.h, common.h: GetReciprocal, fixedpoint.h: one_over_one_plus_x_for_x_in_0_1), no reference vectors exist."""
    from birdnet_stm32.models import _quant as qz
    from birdnet_stm32.models._tflite_reader import TfliteModel, TfliteOp, TfliteTensor
    from oracle import int8_graph as ig

    xs = np.concatenate([np.arange(1, 70000), np.random.default_rng(0).integers(1, 2**31 - 1, 50000)])
    inv, e = ig.get_reciprocal(xs)
    assert np.abs(inv.astype(np.float64) / 2.0**31 * 2.0 ** (-e.astype(np.float64)) * xs - 1.0).max() < 2.0**-27
    assert list(ig.count_leading_sign_bits32(np.array([0, 1, -1, 2, -2, 255, -256, 2**30, -(2**31)]))) == [31, 30, 31, 29, 30, 23, 23, 0, 0]
    for s1, z1, s2, z2, so, zo in [(0.05, -128, 0.06, -128, 1 / 255, -128), (0.013, -128, 0.013, -128, 1 / 255, -128), (0.2, -3, 0.07, 5, 0.031, -17)]:
        def tens(i, s, z, shape):
            return TfliteTensor(i, f"t{i}", shape, np.dtype(np.int8), np.asarray([s], np.float32), np.asarray([z], np.int64), 0, None)

        model = TfliteModel(3, "", [tens(0, s1, z1, (256, 256)), tens(1, s2, z2, (256, 1)), tens(2, so, zo, (256, 256))],
                            [TfliteOp(0, 42, "DIV", 2, [0, 1], [2], {"activation": "none"})], [0], [2])
        q1 = np.arange(-128, 128, dtype=np.int8)[None, :].repeat(256, axis=0)
        q2 = np.arange(-128, 128, dtype=np.int8)[:, None]
        got = ig.Int8Interpreter(model)._div(model.ops[0], {0: q1, 1: q2})
        tab = qz.div_table(s1, z1, s2, z2, so, zo)
        assert np.array_equal(tab, got)
        x1, x2 = q1.astype(np.int64) - z1, q2.astype(np.int64) - z2
        x1, x2 = np.where(x2 < 0, -x1, x1), np.where(x2 == 0, 1, np.abs(x2))
        real = np.clip(np.round((x1 * np.float64(np.float32(s1))) / (x2 * np.float64(np.float32(s2))) / np.float64(np.float32(so))) + zo, -128, 127)
        assert np.abs(tab - real).max() <= 1 and (tab != real).mean() < 0.02


def test_squeeze_excite_scale_is_paired_with_its_projection():
    """Production plans of exported squeeze-excite graphs tag every MUL whose only reader is the plain 1x1 projection behind it
    (the library then applies the gate while that convolution loads its input); the pair never shares a slot with the unscaled
    map or the gate it still reads, and debug plans carry no tags."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models._lower_i8 import lower_i8

    _, model, _, _ = _export(EXPORT_TOPOLOGIES["ir_se_softmax"])
    plan = lower_i8(model)
    heads = [i for i, o in enumerate(plan.ops) if o.kind == pk.I8_SCALE and o.p[pk.TAIL_TAG] == pk.SCALE_HEAD]
    assert len(heads) >= 8 and len(heads) == sum(o.kind == pk.I8_SCALE and o.p[1] <= 768 for o in plan.ops)  # (the kernels take up to 768 input channels)
    for i in heads:
        a, b = plan.ops[i], plan.ops[i + 1]
        assert b.kind == pk.I8_DWPW and b.p[pk.TAIL_TAG] == pk.SCALE_COVERED and b.p[29] == 0 and b.in0 == a.out
        assert b.out not in (a.in0, a.in1) and b.p[2] == a.p[1] and b.p[2] % 16 == 0
    # ... and the gate itself (MEAN -> FULLY_CONNECTED -> FULLY_CONNECTED + LOGISTIC table) is tagged to run as one kernel per chunk
    gates = [i for i, o in enumerate(plan.ops) if o.kind == pk.I8_MEAN and o.p[pk.TAIL_TAG] == pk.SEGATE_HEAD]
    assert len(gates) == sum(o.kind == pk.I8_SCALE for o in plan.ops)
    for i in gates:
        m_, f1, f2 = plan.ops[i : i + 3]
        assert f1.kind == f2.kind == pk.I8_FC and f1.p[pk.TAIL_TAG] == f2.p[pk.TAIL_TAG] == pk.SEGATE_COVERED
        assert f1.in0 == m_.out and f2.in0 == f1.out and f2.p[1] == m_.p[1] and f2.p[5] == 1 and f2.out != m_.in0
    assert plan.ops[-3].kind == pk.I8_MEAN and plan.ops[-3].p[pk.TAIL_TAG] != pk.SEGATE_HEAD  # the classifier's pooling is not a gate
    assert not [o for o in lower_i8(model, keep_all=True).ops
                if o.p[pk.TAIL_TAG] in (pk.SCALE_HEAD, pk.SCALE_COVERED, pk.SEGATE_HEAD, pk.SEGATE_COVERED)]


def test_exporter_refuses_what_it_cannot_express():
    from birdnet_stm32.conversion.export import netspec_to_graph
    from birdnet_stm32.models import build_model

    args = dict(num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2, embeddings_size=256, num_classes=10)
    with pytest.raises(NotImplementedError, match="hybrid and raw frontends"):
        netspec_to_graph(build_model("dscnn", audio_frontend="librosa", **args))
    graph, _, _ = netspec_to_graph(build_model("dscnn", use_attention_pooling=True, **args), frontend_norm=False)  # (exported since round 3)
    assert "SOFTMAX" in {op.name for op in graph.ops} and "SUM" in {op.name for op in graph.ops}


def test_tflite_writer_round_trips_the_shipped_file():
    """models/_tflite_writer.write_tflite is the inverse of the reader: the shipped graph written out and read back is identical
    (tensors, shapes, dtypes, quantisation, constant data, operators, options, versions, inputs / outputs)."""
    from birdnet_stm32.models._tflite_reader import load_tflite, parse_tflite
    from birdnet_stm32.models._tflite_writer import write_tflite

    m = load_tflite(TFLITE_PATH)
    raw = write_tflite(m)
    assert raw[4:8] == b"TFL3" and len(raw) % 16 == 0
    m2 = parse_tflite(raw)
    assert (m2.version, m2.inputs, m2.outputs, len(m2.tensors), len(m2.ops)) == (3, m.inputs, m.outputs, len(m.tensors), len(m.ops))
    for a, b in zip(m.tensors, m2.tensors):
        assert (a.name, a.shape, a.dtype, a.quantized_dimension) == (b.name, b.shape, b.dtype, b.quantized_dimension)
        assert np.array_equal(a.scale, b.scale) and np.array_equal(a.zero_point, b.zero_point)
        assert (a.data is None) == (b.data is None) and (a.data is None or np.array_equal(a.data, b.data))
    for a, b in zip(m.ops, m2.ops):
        assert (a.name, a.version, a.inputs, a.outputs, a.options) == (b.name, b.version, b.inputs, b.outputs, b.options)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(EXPORT_TOPOLOGIES))
def test_exported_graphs_are_bit_exact_per_tensor_on_the_gpu(name):
    """The device plan lowered from an exported graph against the INT8 oracle on the same graph: every int8 tensor bit for bit
    (squeeze-excite gates, MUL, ADD without activation, padded FULLY_CONNECTED rows included), sigmoid scores exactly, softmax
    scores (float32 behind DEQUANTIZE) to 1e-6; the production plan (slots recycled, fused blocks) gives the debug plan's scores."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; the product has no CPU path to fall back to")
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models.runners import HipRunner
    from oracle.int8_graph import Int8Interpreter

    spec, model, _, x = _export(EXPORT_TOPOLOGIES[name])
    x = np.concatenate([x, np.zeros_like(x[:1])])  # a silent chunk: with the max normalisation its denominator is the epsilon alone
    B = x.shape[0]
    if EXPORT_TOPOLOGIES[name].get("use_attention_pooling"):  # the other published form of the int8 SOFTMAX (float32 exponent table), oracle and device
        want = Int8Interpreter(model, softmax_form="lut").invoke(x)
        r2 = HipRunner(lower_i8(model, softmax_form="lut"), max_batch=B)
        assert np.array_equal(r2.predict(x), want)
        r2.close()
    ref, env = Int8Interpreter(model).invoke(x, return_all=True)
    for fuse in (True, False):
        runner = HipRunner(lower_i8(model, keep_all=True, fuse=fuse), max_batch=B)
        got = runner.predict(x)
        checked = 0
        for oi, op in enumerate(runner.plan.ops):
            if op.out < 0 or not op.name.startswith("t"):
                continue
            ti = int(op.name[1:])
            a = runner.op_output(oi, B)
            r = env[ti]
            if op.kind == 20:  # quantised, transposed, zero-padded spectrogram
                r = r.reshape(B, a.shape[1], -1)
                a = a[:, :, : r.shape[2]]
            r = r.reshape(a.shape)
            bad = int((a != r).sum())
            assert bad == 0, f"{name} fuse={fuse}: tensor {op.name} (plan op {oi}, kind {op.kind}): {bad} of {a.size} values differ"
            checked += 1
        assert checked >= 12  # (the plain depthwise-separable topologies have the fewest operators)
        softmax = spec.layers[-1].attrs["activation"] == "softmax"
        assert np.allclose(got, ref, atol=1e-6) if softmax else np.array_equal(got, ref)
        runner.close()
    prod = HipRunner(lower_i8(model), max_batch=B)
    assert np.array_equal(prod.predict(x), got)
    for nb in (1, 3):
        assert np.array_equal(prod.predict(x[:nb]), got[:nb])
    from birdnet_stm32 import _hip

    with _hip.options(i8_pwdw=1):  # expand + depthwise of inverted-residual blocks as one kernel (off by default: measured slower), same integers
        assert np.array_equal(prod.predict(x), got)
        assert np.array_equal(prod.predict(x[:3]), got[:3])
    with _hip.options(i8_dw_pool=0):  # the squeeze-excite MEAN reads the depthwise map itself instead of taking the sums the depthwise kernel added up
        assert np.array_equal(prod.predict(x), got)
    with _hip.options(i8_pw_lds=0):  # the dense late 1x1 convolutions (Cin 192 / 384 / 768) through the tile kernel + a separate MUL: same integers
        assert np.array_equal(prod.predict(x), got)
    with _hip.options(i8_add_tab=0):  # residual ADD behind a projection on the vector ALU instead of the 64 KB table in LDS
        assert np.array_equal(prod.predict(x), got)
    with _hip.options(i8_pw_forms=0):  # general requantisation code instead of the compile-time forms of i8_pw_wave_kernel / i8_dw_stream_kernel
        assert np.array_equal(prod.predict(x), got)
    prod.close()


ODD_GEOMETRIES = {
    "mels32_w128_a0.75_2s": dict(num_mels=32, spec_width=128, alpha=0.75, chunk_duration=2),
    "mels48_w192_a1.25": dict(num_mels=48, spec_width=192, alpha=1.25, use_se=True),
    "mels64_w320_ds_se": dict(num_mels=64, spec_width=320, alpha=1.0, use_inverted_residual=False, use_se=True, class_activation="sigmoid"),
    # mel counts that are not a multiple of 16: the mel mixer and most blocks fall back to the one-operator kernels
    "mels40_w128_ds": dict(num_mels=40, spec_width=128, alpha=0.5, use_se=False, use_inverted_residual=False),
    "mels24_w64_se": dict(num_mels=24, spec_width=64, alpha=0.5, use_se=True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ODD_GEOMETRIES))
def test_exported_graphs_of_other_geometries_match_the_oracle_on_the_gpu(name):
    """Spectrogram sizes and width multipliers other than the shipped ones (maps that are not multiples of the kernels' strip
    widths, channel counts like 24 / 40 / 120): the production plan — row-streaming depthwise and stem kernels, the wave-level 1x1
    convolution, the fused squeeze-excite gate and MUL — and the debug plan both reproduce the INT8 oracle."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; the product has no CPU path to fall back to")
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models.runners import HipRunner
    from oracle.int8_graph import Int8Interpreter

    spec, model, _, x = _export(ODD_GEOMETRIES[name], n_cal=3)
    ref, env = Int8Interpreter(model).invoke(x, return_all=True)
    B = x.shape[0]
    softmax = spec.layers[-1].attrs["activation"] == "softmax"
    dbg = HipRunner(lower_i8(model, keep_all=True), max_batch=B)
    got = dbg.predict(x)
    for oi, op in enumerate(dbg.plan.ops):
        if op.out < 0 or not op.name.startswith("t") or op.kind == 20:
            continue
        a = dbg.op_output(oi, B)
        assert np.array_equal(a, env[int(op.name[1:])].reshape(a.shape)), f"{name}: tensor {op.name} (plan op {oi}, kind {op.kind})"
    assert np.allclose(got, ref, atol=1e-6) if softmax else np.array_equal(got, ref)
    dbg.close()
    prod = HipRunner(lower_i8(model), max_batch=B)
    for nb in (B, 1, 2):
        p = prod.predict(x[:nb])
        assert np.allclose(p, ref[:nb], atol=1e-6) if softmax else np.array_equal(p, ref[:nb])
    prod.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [37, 300])
def test_dense_int8_pointwise_kernel_equals_the_tile_kernel_at_batch(batch):
    """i8_pw_lds_kernel (csrc/bn_i8_pw.hip: the 192 / 384 / 768-channel 1x1 convolutions of the alpha = 1.5 inverted-residual net, the
    squeeze-excite MUL applied on load, residual ADD) against the tile kernel + separate MUL (option i8_pw_lds = 0; those are held to
    the INT8 oracle per tensor above) on batches that need several strides of the persistent walkers, odd batch sizes, repeated
    launches; and per tensor against the oracle on the first chunks (debug plan: no gate fusion, the kernel's plain forms)."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; the product has no CPU path to fall back to")
    from birdnet_stm32 import _hip
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models.runners import HipRunner
    from oracle.int8_graph import Int8Interpreter

    spec, model, _, x0 = _export(EXPORT_TOPOLOGIES["alpha1.5_pcen"])
    rng = np.random.default_rng(batch)
    x = (rng.random((batch, 257, 256, 1), dtype=np.float32) ** 3).astype(np.float32)
    x[: x0.shape[0]] = x0
    x[-1] = 0.0
    prod = HipRunner(lower_i8(model), max_batch=batch)
    kinds = [op.kind for op in prod.plan.ops]
    got = prod.predict(x)
    for _ in range(3):
        assert np.array_equal(prod.predict(x), got)
    with _hip.options(i8_pw_lds=0):
        ref = prod.predict(x)
    assert np.array_equal(got, ref)
    assert np.array_equal(prod.predict(x[:5]), got[:5])
    prod.close()
    want = Int8Interpreter(model).invoke(x[:3])
    assert np.allclose(got[:3], want, atol=1e-6)
    assert len(kinds) > 20
