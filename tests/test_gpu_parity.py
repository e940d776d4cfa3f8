"""GPU parity: libbirdnet_hip (through its C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): float paths within 1e-3 cosine distance (asserted far tighter),
INT8 path bit-exact at the runner boundary, exact top-1 and logit cosine >= 0.999 from audio.
"""

import numpy as np
import pytest

from conftest import KERAS_PATH, TFLITE_PATH, cosine, fixture_signals, synth_chunks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; there is no CPU fallback to fall back to")
    return torch


@pytest.fixture(scope="module")
def audio24():
    sig = fixture_signals(24000)
    x = np.concatenate([synth_chunks(5), np.stack([sig["sine"], sig["noise"], sig["chirp"]])]).astype(np.float32)
    return x


@pytest.fixture(scope="module")
def oracle_specs(audio24):
    from oracle import stft

    return np.stack([stft.hybrid_spectrogram(a) for a in audio24])


# ----------------------------------------------------------------------------------------- STFT
@pytest.mark.parametrize("sr,hop", [(24000, 281), (22050, 258)])
def test_stft_matches_oracle(torch_mod, sr, hop):
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models.runners import stft_device
    from oracle import stft

    sig = fixture_signals(sr)
    x = np.stack([sig["sine"], sig["noise"], sig["chirp"], synth_chunks(1, sr=sr)[0]])
    assert x.shape[1] // 256 == hop
    ctx = _hip.Context(0, 16)
    d = torch.from_numpy(x).cuda()
    raw, mm = stft_device(ctx, d, normalize=False, return_minmax=True)
    norm = stft_device(ctx, d, normalize=True)
    torch.cuda.synchronize()
    raw, mm, norm = raw.cpu().numpy(), mm.cpu().numpy(), norm.cpu().numpy()
    for i in range(x.shape[0]):
        ref_raw = stft.stft_magnitude(x[i], 512, hop)[:, :256]
        ref = stft.hybrid_spectrogram(x[i])
        scale = ref_raw.max()
        err = np.abs(raw[i] - ref_raw).max() / scale
        assert err < 2e-6, f"chunk {i}: raw STFT rel-to-peak error {err:.3e}"
        assert abs(mm[i, 0] - ref_raw.min()) <= 2e-6 * scale and abs(mm[i, 1] - ref_raw.max()) <= 2e-6 * scale
        nerr = np.abs(norm[i] - ref).max()
        assert nerr < 5e-6, f"chunk {i}: normalised spectrogram abs error {nerr:.3e}"
    ctx.close()


def test_stft_silence_and_ragged_width(torch_mod):
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models.runners import stft_device
    from oracle import stft

    ctx = _hip.Context(0, 4)
    x = np.zeros((2, 72000), np.float32)
    x[1, 1000] = 1.0  # a single click
    d = torch.from_numpy(x).cuda()
    out = stft_device(ctx, d, normalize=True).cpu().numpy()
    assert np.all(out[0] == 0.0)  # silence: (0 - 0) / (0 + 1e-10) = 0, like the reference (tests/test_spectrogram.py:25-30)
    assert np.abs(out[1] - stft.hybrid_spectrogram(x[1])).max() < 5e-6
    # a spec_width that is not a multiple of the 16-frame tile
    y = synth_chunks(2)[:, :50000]
    w = 100
    got = stft_device(ctx, torch.from_numpy(np.ascontiguousarray(y)).cuda(), spec_width=w, normalize=True).cpu().numpy()
    for i in range(2):
        assert np.abs(got[i] - stft.hybrid_spectrogram(y[i], 512, w)).max() < 5e-6
    # errors: too few frames, unsupported n_fft
    with pytest.raises(_hip.HipError):
        stft_device(ctx, d, hop=20000, normalize=True)
    with pytest.raises(_hip.HipError):
        stft_device(ctx, d, n_fft=1024, normalize=True)
    ctx.close()


# ------------------------------------------------------------------------------------ float32 graph
@pytest.mark.parametrize("fuse", [True, False], ids=["fused_mfma", "baseline_kernels"])
def test_f32_graph_per_layer_and_logits(torch_mod, oracle_specs, fuse):
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models.runners import load_model_runner
    from oracle import float_graph

    spec = load_keras_archive(KERAS_PATH)
    x = oracle_specs[..., None]
    ref_scores, ref_logits, acts = float_graph.forward(spec, x, np.float64, return_all=True, return_logits=True)
    runner = load_model_runner(KERAS_PATH, max_batch=16, keep_all=True, fuse=fuse)
    got = runner.predict(x)
    B = x.shape[0]
    worst = []
    for oi, op in enumerate(runner.plan.ops):
        if op.out < 0 or op.name not in acts:
            continue
        a = runner.op_output(oi, B)
        r = acts[op.name].reshape(a.shape)
        err = np.abs(a - r).max() / (np.abs(r).max() + 1e-12)
        worst.append((err, op.name))
        assert err < 2e-4, f"layer {op.name}: relative-to-peak error {err:.3e}"
    d_scores, d_logits = runner.predict_device(torch_mod.from_numpy(x.reshape(B, -1)).cuda(), return_logits=True)
    d_logits = d_logits.cpu().numpy()
    for b in range(B):
        assert 1.0 - cosine(d_logits[b], ref_logits[b]) < 1e-5
        assert 1.0 - cosine(got[b], ref_scores[b]) < 1e-5
        assert got[b].argmax() == ref_scores[b].argmax()
    assert np.abs(got - ref_scores).max() < 1e-5
    runner.close()


def test_f32_infer_audio_end_to_end(torch_mod, audio24, oracle_specs):
    torch = torch_mod
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models.runners import load_model_runner
    from oracle import float_graph

    spec = load_keras_archive(KERAS_PATH)
    ref_scores, ref_logits = float_graph.forward(spec, oracle_specs[..., None], np.float64, return_logits=True)
    runner = load_model_runner(KERAS_PATH, max_batch=4)  # forces slicing of the batch of 8
    scores, logits = runner.infer_audio_device(torch.from_numpy(audio24).cuda(), return_logits=True)
    scores, logits = scores.cpu().numpy(), logits.cpu().numpy()
    for b in range(audio24.shape[0]):
        assert 1.0 - cosine(logits[b], ref_logits[b]) < 1e-4  # bar: 1e-3 cosine distance
        assert scores[b].argmax() == ref_scores[b].argmax()
    runner.close()


def test_f32_strip_kernel_matches_tile_kernels(torch_mod, oracle_specs):
    """The row-streaming float32 strip kernel (stage 1-3 blocks) against the tile kernels it replaces: per layer within
    float32 round-off of each other (the FMA order differs), for strip heights that move the strip borders around."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models.runners import load_model_runner

    x = np.tile(oracle_specs[..., None], (9, 1, 1, 1))[:130]
    B = x.shape[0]
    runner = load_model_runner(KERAS_PATH, max_batch=B, keep_all=True)
    ops = [oi for oi, op in enumerate(runner.plan.ops) if op.kind == pk.F32_DWPW and op.p[2] <= 128 and op.p[10] <= 128 and op.p[7] % 16 == 0]
    ops += [oi for oi, op in enumerate(runner.plan.ops) if op.kind == pk.F32_FRONT and op.p[pk.OP_PATH] == pk.PATH_INPUT]  # front block
    assert len(ops) == 9
    from birdnet_stm32 import _hip

    with _hip.options(f32_strip=0):
        want_scores = runner.predict(x)
        want = {oi: runner.op_output(oi, B) for oi in ops}
    # repeated: the store-data hazard these kernels guard against (bn_f32_strip.hip: store16) showed up in 1 launch of 50-100
    for th in (0, 1, 3, 5, 7, 64) * 12:
        with _hip.options(f32_strip=1, f32_strip_th=th):
            got_scores = runner.predict(x)
            for oi in ops:
                a = runner.op_output(oi, B)
                err = np.abs(a - want[oi]).max() / np.abs(want[oi]).max()
                assert err < 1e-5, f"rows per strip {th or 'auto'}: layer {runner.plan.ops[oi].name}: relative-to-peak difference {err:.3e}"
            assert np.abs(got_scores - want_scores).max() < 5e-6
    # the audio path: the front block finalises the raw mel energies while loading (its own operator variant)
    import torch

    audio = torch.from_numpy(np.tile(synth_chunks(8), (5, 1))).cuda()
    with _hip.options(f32_strip=0):
        want_audio = runner.infer_audio_device(audio).cpu().numpy()
    for th in (0, 1, 5, 64) * 3:
        with _hip.options(f32_strip=1, f32_strip_th=th):
            assert np.abs(runner.infer_audio_device(audio).cpu().numpy() - want_audio).max() < 5e-6
    runner.close()


# --------------------------------------------------------------------------------------- INT8 graph
@pytest.mark.parametrize("fuse", [True, False], ids=["fused_mfma", "baseline_kernels"])
def test_i8_graph_bit_exact_per_tensor(torch_mod, oracle_specs, fuse):
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import load_model_runner
    from oracle.int8_graph import Int8Interpreter

    model = load_tflite(TFLITE_PATH)
    x = oracle_specs[..., None]
    ref, env = Int8Interpreter(model).invoke(x, return_all=True)
    runner = load_model_runner(TFLITE_PATH, max_batch=16, keep_all=True, fuse=fuse)
    got = runner.predict(x)
    B = x.shape[0]
    for oi, op in enumerate(runner.plan.ops):
        if op.out < 0:
            continue
        ti = int(op.name[1:])
        a = runner.op_output(oi, B)
        r = env[ti]
        if op.kind == 20:  # quantised, transposed, zero-padded spectrogram: compare the graph's 264 columns
            r = r.reshape(B, a.shape[1], -1)
            a = a[:, :, : r.shape[2]]
        r = r.reshape(a.shape)
        bad = int((a != r).sum())
        assert bad == 0, f"tensor {op.name} (plan op {oi}): {bad} of {a.size} int8 values differ, first at {np.argwhere(a != r)[:3].tolist()}"
    assert np.array_equal(got, ref), "dequantised scores differ"
    runner.close()


def test_i8_strip_kernel_matches_generic_block(torch_mod, oracle_specs):
    """The wave-autonomous strip kernels (front block, stage 1-3 blocks) against the generic fused kernels: every tensor bit for bit, for
    rows-per-wave values that put the strip borders everywhere (1, 3, 5, 7 rows, whole map), and a batch large enough for
    the launcher's own choice."""
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models.runners import load_model_runner

    x = np.tile(oracle_specs[..., None], (17, 1, 1, 1))[:260]
    x = np.ascontiguousarray(x[np.random.default_rng(3).permutation(x.shape[0])])
    B = x.shape[0]
    from birdnet_stm32 import _hip

    runner = load_model_runner(TFLITE_PATH, max_batch=B, keep_all=True)
    strip_ops = [oi for oi, op in enumerate(runner.plan.ops) if (op.kind == pk.I8_DWPW and op.p[35]) or (op.kind == pk.I8_FRONT and op.p[16])]
    assert len(strip_ops) == 11
    with _hip.options(i8_strip=0):
        want_scores = runner.predict(x)
        want = {oi: runner.op_output(oi, B) for oi in strip_ops}
    for th in (0, 1, 3, 5, 7, 64) * 4:
        with _hip.options(i8_strip=1, i8_strip_th=th):
            got_scores = runner.predict(x)
            for oi in strip_ops:
                a = runner.op_output(oi, B)
                bad = int((a != want[oi]).sum())
                assert bad == 0, f"rows per wave {th or 'auto'}: tensor {runner.plan.ops[oi].name}: {bad} of {a.size} values differ, first at {np.argwhere(a != want[oi])[:3].tolist()}"
            assert np.array_equal(got_scores, want_scores)
    # odd batch sizes: workgroups of the ADD kernels take 8 / NW strips, the spare ones repeat the last chunk
    for nb in (1, 3, 37):
        assert np.array_equal(runner.predict(x[:nb]), want_scores[:nb])
    runner.close()
    # the production plan (slots recycled, QUANTIZE fused into the mel mixer's load with the three-instruction exact division) gives
    # the same scores bit for bit, on the test spectrograms and on random ones that exercise the rounding of the quantiser
    prod = load_model_runner(TFLITE_PATH, max_batch=B)
    assert prod.plan.ops[0].kind == pk.I8_DWPW and prod.plan.ops[0].p[36] == 1
    assert np.array_equal(prod.predict(x), want_scores)
    dbg = load_model_runner(TFLITE_PATH, max_batch=B, keep_all=True)
    xr = np.random.default_rng(11).random((64, 257, 256, 1), dtype=np.float32)
    xr[:8] *= np.float32(1.0 / 255.0) * np.arange(0, 256, 32, dtype=np.float32)[:, None, None, None]  # values near the quantiser's steps
    assert np.array_equal(prod.predict(xr), dbg.predict(xr))
    prod.close()
    dbg.close()
    # from audio, the spectrogram between the STFT and the fused QUANTIZE is tile-major ([W/16][257][16], the layout the STFT writes
    # fastest); option stft_rowmajor keeps the reference layout: identical scores
    import torch

    audio = torch.from_numpy(np.tile(synth_chunks(8), (9, 1))[:70]).cuda()
    tiled = load_model_runner(TFLITE_PATH, max_batch=70)
    s_tiled = tiled.infer_audio_device(audio).cpu().numpy()
    with _hip.options(stft_rowmajor=1):
        s_row = tiled.infer_audio_device(audio).cpu().numpy()
    tiled.close()
    assert np.array_equal(s_tiled, s_row)


def test_i8_from_audio_top1_and_cosine(torch_mod, audio24, oracle_specs):
    torch = torch_mod
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import load_model_runner
    from oracle.int8_graph import Int8Interpreter

    model = load_tflite(TFLITE_PATH)
    ref, env = Int8Interpreter(model).invoke(oracle_specs[..., None], return_all=True)
    fc = model.ops[53].outputs[0]
    s, z = float(model.tensors[fc].scale[0]), int(model.tensors[fc].zero_point[0])
    ref_logits = (env[fc].astype(np.float32) - z) * s
    runner = load_model_runner(TFLITE_PATH, max_batch=16)
    scores, logits = runner.infer_audio_device(torch.from_numpy(audio24).cuda(), return_logits=True)
    scores, logits = scores.cpu().numpy(), logits.cpu().numpy()
    # bit-exact from audio (float64 pass behind the float32 STFT, csrc/bn_stft_exact.hip): scores, pre-sigmoid outputs, quantised input
    assert np.array_equal(scores, ref)
    assert np.array_equal(logits, ref_logits)
    qin = model.ops[0].outputs[0]
    assert np.array_equal(runner.input_bytes(audio24.shape[0]).reshape(audio24.shape[0], -1), env[qin].reshape(audio24.shape[0], -1))
    runner.close()


# ------------------------------------------------------------------- topologies built by current reference code
@pytest.mark.parametrize(
    "kw",
    [
        dict(),  # defaults: inverted residual + SE, softmax head, per-sample max-normalised hybrid frontend
        dict(use_inverted_residual=False, use_se=True, embeddings_size=128, use_attention_pooling=True, class_activation="sigmoid"),
        dict(alpha=1.5, mag_scale="pcen", num_classes=37),
        dict(use_se=False, depth_multiplier=2, alpha=0.5, mag_scale="none"),
    ],
    ids=["ir_se_default", "ds_se_attnpool_emb", "alpha1.5_pcen", "ir_nose_deep_narrow"],
)
def test_f32_current_code_topologies(torch_mod, oracle_specs, kw):
    """SE, inverted residuals, embedding conv, attention pooling, PCEN/none scaling, frontend max-norm: per layer vs the oracle."""
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from oracle import float_graph

    args = dict(num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=10, randomize_bn=True, seed=7)
    args.update(kw)
    spec = build_model("dscnn", **args)
    x = oracle_specs[:4, ..., None]
    ref_scores, ref_logits, acts = float_graph.forward(spec, x, np.float64, return_all=True, return_logits=True)
    runner = HipRunner(lower_f32(spec, keep_all=True), max_batch=4)
    got = runner.predict(x)
    for oi, op in enumerate(runner.plan.ops):
        if op.out < 0 or op.name not in acts:
            continue
        a = runner.op_output(oi, 4)
        r = acts[op.name].reshape(a.shape)
        err = np.abs(a - r).max() / (np.abs(r).max() + 1e-12)
        assert err < 5e-4, f"layer {op.name}: relative-to-peak error {err:.3e}"
    assert np.abs(got - ref_scores).max() < 1e-4
    for b in range(4):
        assert 1.0 - cosine(got[b], ref_scores[b]) < 1e-5
    # the production plan (slots recycled) gives the same scores as the keep-everything debug plan — bit for bit while the squeeze-excite
    # gates pool the maps themselves (option f32_pwdw = 1: every kernel keeps one summation order); by default the gates add up per-strip
    # sums handed over by the depthwise kernels, whose grouping differs between the plans: float32 noise
    from birdnet_stm32 import _hip

    with _hip.options(f32_pwdw=1):
        got1 = runner.predict(x)
    runner.close()
    runner = HipRunner(lower_f32(spec), max_batch=4)
    with _hip.options(f32_pwdw=1):
        assert np.array_equal(runner.predict(x), got1)
    assert np.abs(runner.predict(x) - got).max() < 1e-6 and np.abs(got1 - got).max() < 1e-6
    runner.close()


F32_GEOMETRIES = {
    "mels32_w128_a0.75_2s": dict(num_mels=32, spec_width=128, alpha=0.75, chunk_duration=2),
    "mels48_w192_a1.25_se": dict(num_mels=48, spec_width=192, alpha=1.25, use_se=True),
    "mels40_w128_ds": dict(num_mels=40, spec_width=128, alpha=0.5, use_se=False, use_inverted_residual=False),
    "mels64_w320_ds_se_norm": dict(num_mels=64, spec_width=320, use_inverted_residual=False, use_se=True, frontend_norm=True),
}


@pytest.mark.parametrize("name", list(F32_GEOMETRIES))
def test_f32_other_geometries_per_layer(torch_mod, name):
    """Spectrogram sizes, mel counts and width multipliers other than the shipped ones (maps the strip kernels do not take, mel
    counts that are not multiples of 16): the float32 plan against the float oracle per layer, debug and production plans."""
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from oracle import float_graph, stft

    args = dict(num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=10, randomize_bn=True, seed=11)
    args.update(F32_GEOMETRIES[name])
    norm = bool(args.pop("frontend_norm", False))
    spec = build_model("dscnn", **args)
    spec.frontend.attrs["norm"] = norm
    chunks = synth_chunks(5, sr=args["sample_rate"], seconds=args["chunk_duration"], seed=4)
    x = np.stack([stft.hybrid_spectrogram(a, spec_width=args["spec_width"]) for a in chunks])[..., None].astype(np.float32)
    ref_scores, _, acts = float_graph.forward(spec, x, np.float64, return_all=True, return_logits=True)
    runner = HipRunner(lower_f32(spec, keep_all=True), max_batch=5)
    got = runner.predict(x)
    checked = 0
    for oi, op in enumerate(runner.plan.ops):
        if op.out < 0 or op.name not in acts:
            continue
        a = runner.op_output(oi, 5)
        r = acts[op.name].reshape(a.shape)
        err = np.abs(a - r).max() / (np.abs(r).max() + 1e-12)
        assert err < 5e-4, f"{name}: layer {op.name}: relative-to-peak error {err:.3e}"
        checked += 1
    assert checked >= 8 and np.abs(got - ref_scores).max() < 1e-4
    runner.close()
    prod = HipRunner(lower_f32(spec), max_batch=5)
    for nb in (5, 1, 2):
        assert np.abs(prod.predict(x[:nb]) - got[:nb]).max() < 2e-6
    prod.close()


# --------------------------------------------------------------------------------- full-size properties
def test_full_batch_properties(torch_mod):
    """BASELINE sizes (B=1024 float32, B=4096 INT8) through size-independent properties: the result of a chunk does not
    depend on its batch neighbours, position or batch slicing, and repeated runs are bit-identical."""
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner

    base = torch.from_numpy(synth_chunks(64)).cuda()
    for path, B in ((KERAS_PATH, 1024), (TFLITE_PATH, 4096)):
        runner = load_model_runner(path, max_batch=B)
        idx = torch.randint(0, 64, (B,), generator=torch.Generator().manual_seed(1)).cuda()
        audio = base[idx].contiguous()
        s1 = runner.infer_audio_device(audio).clone()
        s2 = runner.infer_audio_device(audio)
        assert torch.equal(s1, s2), "run-to-run determinism"
        small = load_model_runner(path, max_batch=96)  # different workspace, batch processed in slices of 96
        ref64 = small.infer_audio_device(base)
        assert torch.equal(s1, ref64[idx]), "a chunk's scores depend on its batch position / neighbours"
        assert torch.isfinite(s1).all() and float(s1.min()) >= 0.0 and float(s1.max()) <= 1.0
        small.close()
        runner.close()


def test_empty_and_single_batches(torch_mod):
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner

    runner = load_model_runner(TFLITE_PATH, max_batch=8)
    assert runner.predict(np.zeros((0, 257, 256, 1), np.float32)).shape == (0, 100)
    one = runner.predict(np.zeros((1, 257, 256, 1), np.float32))
    assert one.shape == (1, 100) and one.dtype == np.float32
    with pytest.raises(ValueError, match="expected input of shape"):
        runner.predict(np.zeros((2, 128, 256, 1), np.float32))
    big = runner.predict(np.tile(np.zeros((1, 257, 256, 1), np.float32), (19, 1, 1, 1)))  # > max_batch: sliced like resize_tensor_input
    assert np.array_equal(big, np.tile(one, (19, 1)))
    runner.close()


# ------------------------------------------------------------------------------------ evaluate on the GPU
def test_evaluate_device_pipeline_matches_reference_loop(torch_mod, tmp_path):
    """evaluate(): the cross-file device pipeline gives the per-file scores of the reference-style per-file loop."""
    from birdnet_stm32.audio.io import save_wav
    from birdnet_stm32.evaluation.metrics import evaluate
    from birdnet_stm32.models.runners import load_model_runner
    from birdnet_stm32.training.config import ModelConfig

    from conftest import CONFIG_PATH

    cfg = ModelConfig.load(CONFIG_PATH).to_dict()
    cfg.update(sample_rate=24000, hop_length=281)
    classes = cfg["class_names"]
    x = synth_chunks(16)
    files = []
    for i in range(16):
        d = tmp_path / classes[i % 3]
        d.mkdir(exist_ok=True)
        n = 72000 if i % 4 else 72000 * 2 + 30000  # some files hold 3 chunks (2 full + tail)
        wav = np.concatenate([x[i], x[(i + 1) % 16], x[(i + 2) % 16]])[:n]
        save_wav(wav, str(d / f"f{i}.wav"), 24000, subtype="FLOAT")
        files.append(str(d / f"f{i}.wav"))
    runner = load_model_runner(TFLITE_PATH, max_batch=16)
    m_dev, pf_dev, yt, ys_dev = evaluate(runner, files, classes, cfg, pooling="lme", batch_size=5, measure_latency=True)
    m_ref, pf_ref, _, ys_ref = evaluate(runner, files, classes, cfg, pooling="lme", batch_size=5, measure_latency=True, device_pipeline=False)
    assert [p["file"] for p in pf_dev] == [p["file"] for p in pf_ref] == files
    assert m_dev["total_chunks"] == m_ref["total_chunks"] == 12 + 4 * 3
    # Both routes quantise the reference's bytes (the loop through bn_stft_mag_exact, the device pipeline through the float64 pass behind
    # the float32 STFT), so the chunk scores are identical; the pooled scores differ only by the device's expf / logf in log-mean-exp
    # (<= 2e-6, tests/test_gpu_ingest.py) and not at all under mean pooling.
    np.testing.assert_allclose(ys_dev, ys_ref, atol=4e-6, rtol=0)
    assert (ys_dev.argmax(axis=1) == ys_ref.argmax(axis=1)).all()
    for k in ("latency_mean_ms", "latency_p99_ms"):
        assert m_dev[k] > 0 and m_ref[k] > 0
    _, _, _, avg_dev = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=5)
    _, _, _, avg_ref = evaluate(runner, files, classes, cfg, pooling="avg", batch_size=5, device_pipeline=False)
    assert np.array_equal(avg_dev, avg_ref)
    runner.close()


def test_cli_evaluate_benchmark_latency(torch_mod, tmp_path):
    """`python -m birdnet_stm32 evaluate --benchmark_latency` on 16 synthetic 3 s @ 24 kHz WAVs (BASELINE configs[0], on the GPU)."""
    import json
    import subprocess
    import sys

    from birdnet_stm32.audio.io import save_wav

    from conftest import CONFIG_PATH, PKG

    cfg = json.load(open(CONFIG_PATH))
    cfg.update(sample_rate=24000, hop_length=281)
    (tmp_path / "model_cfg.json").write_text(json.dumps(cfg))
    x = synth_chunks(16)
    for i in range(16):
        d = tmp_path / "data" / cfg["class_names"][i % 4]
        d.mkdir(parents=True, exist_ok=True)
        save_wav(x[i], str(d / f"c{i}.wav"), 24000)
    out = tmp_path / "bench.json"
    cmd = [sys.executable, "-m", "birdnet_stm32", "evaluate", "--model_path", TFLITE_PATH, "--model_config", str(tmp_path / "model_cfg.json"),
           "--data_path_test", str(tmp_path / "data"), "--benchmark_latency", "--benchmark", str(out), "--batch_size", "16"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(__import__("os").environ, PYTHONPATH=PKG), timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "latency_mean_ms" in res.stdout and "Evaluated 16 files across 100 classes." in res.stdout
    rep = json.loads(out.read_text())
    assert rep["num_files"] == 16 and rep["metrics"]["total_chunks"] == 16 and rep["metrics"]["latency_p95_ms"] > 0


@pytest.mark.parametrize("seconds", [2, 3])
def test_f32_raw_frontend_config5_topology(torch_mod, seconds):
    """BASELINE configs[4]: raw-waveform learned filterbank + PCEN + alpha=1.5 DS-CNN with SE / inverted residuals, seeded weights, per
    layer against the float64 oracle.  2 s: the reference's deployment geometry 24 kHz x 2 s (T = 48000 < 65536, stride 188; SURVEY
    finding 11).  3 s: the METRIC's chunk length (T = 72000, stride ceil(72000 / 256) = 282, pad_total = max(0, 282 * 255 + 16 - 72000) = 0:
    reference models/frontend.py:147-155) with the reference's STM32N6 length guard lifted (``raw_length_limit=None``, models/dscnn.py:144-151)."""
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from oracle import float_graph

    T = 24000 * seconds
    if seconds == 3:
        with pytest.raises(ValueError, match="STM32N6 constraint"):  # the reference's guard, kept by default
            build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=100, audio_frontend="raw")
    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=seconds, embeddings_size=256, num_classes=100,
                       audio_frontend="raw", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42,
                       **({"raw_length_limit": None} if seconds == 3 else {}))
    x = synth_chunks(3)[:, :T]
    x = (x / (np.abs(x).max(axis=1, keepdims=True) + 1e-6)).astype(np.float32)[..., None]  # host prep of evaluation/metrics.py:62-69
    ref_scores, ref_logits, acts = float_graph.forward(spec, x, np.float64, return_all=True, return_logits=True)
    runner = HipRunner(lower_f32(spec, keep_all=True), max_batch=4)
    got = runner.predict(x)
    for oi, op in enumerate(runner.plan.ops):
        if op.out < 0 or op.name not in acts:
            continue
        a = runner.op_output(oi, 3)
        r = acts[op.name].reshape(a.shape)
        err = np.abs(a - r).max() / (np.abs(r).max() + 1e-12)
        assert err < 5e-4, f"layer {op.name}: relative-to-peak error {err:.3e}"
    for b in range(3):
        assert 1.0 - cosine(got[b], ref_scores[b]) < 1e-5
    assert np.abs(got - ref_scores).max() < 1e-4
    with pytest.raises(ValueError, match="expected input of shape"):
        runner.predict(np.zeros((1, 72000 if seconds == 2 else 48000, 1), np.float32))
    runner.close()


def test_f32_raw_frontend_padded_and_odd_widths(torch_mod):
    """The raw filterbank kernel (one frame per thread over all filters) at geometries the configs[4] test does not reach: a chunk
    shorter than stride (W - 1) + 16 samples (symmetric zero padding in front of the filterbank, reference models/frontend.py:147-155),
    a frame count that is not a multiple of the workgroup, filter counts that are not multiples of 16 — frontend map and scores
    against the float64 oracle."""
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from oracle import float_graph

    for sr, cd, W, M, mag in ((6000, 0.5, 128, 32, "none"), (8000, 1.0, 96, 24, "pwl"), (24000, 2.0, 320, 40, "pcen")):
        spec = build_model("dscnn", num_mels=M, spec_width=W, sample_rate=sr, chunk_duration=cd, embeddings_size=64, num_classes=7, alpha=0.5,
                           audio_frontend="raw", mag_scale=mag, use_se=False, use_inverted_residual=False, randomize_bn=True, seed=3)
        T = int(sr * cd)
        rng = np.random.default_rng(T)
        x = rng.standard_normal((3, T)).astype(np.float32)
        x = (x / (np.abs(x).max(axis=1, keepdims=True) + 1e-6)).astype(np.float32)[..., None]
        ref_scores, _, acts = float_graph.forward(spec, x, np.float64, return_all=True, return_logits=True)
        runner = HipRunner(lower_f32(spec, keep_all=True), max_batch=4)
        got = runner.predict(x)
        fe = next(oi for oi, op in enumerate(runner.plan.ops) if op.kind == 3)  # F32_RAWFE
        a = runner.op_output(fe, 3)
        r = acts[runner.plan.ops[fe].name].reshape(a.shape)
        assert np.abs(a - r).max() / (np.abs(r).max() + 1e-12) < 5e-5, (sr, cd, W, M)
        for b in range(3):
            assert 1.0 - cosine(got[b], ref_scores[b]) < 1e-5, (sr, cd, W, M)
        runner.close()


@pytest.mark.parametrize(
    "kw",
    [
        dict(num_mels=32, spec_width=128, sample_rate=16000, chunk_duration=2),
        dict(num_mels=48, spec_width=192, sample_rate=22050, chunk_duration=3, alpha=0.75),
        dict(num_mels=128, spec_width=256, sample_rate=24000, chunk_duration=3, use_se=False, use_inverted_residual=False),
        dict(num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, use_se=False, use_inverted_residual=False, depth_multiplier=2, alpha=1.25),
    ],
    ids=["32mel_w128_16k2s", "48mel_w192_alpha.75", "128mel_legacy", "legacy_deep_alpha1.25"],
)
def test_f32_other_geometries(torch_mod, kw):
    """Mel counts, spectrogram widths, chunk lengths and widths other than the shipped 64 x 256 @ 3 s: maps whose tiles, strips and
    K-slices differ (ragged tiles, the unfused stem, other K / N splits), fused and baseline plans, from spectrograms and from audio."""
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from oracle import float_graph, stft

    args = dict(embeddings_size=256, num_classes=10, randomize_bn=True, seed=5)
    args.update(kw)
    spec = build_model("dscnn", **args)
    T = int(args["sample_rate"] * args["chunk_duration"])
    rng = np.random.default_rng(0)
    audio = (rng.standard_normal((3, T)) * 0.2 + np.sin(2 * np.pi * 900 * np.arange(T) / args["sample_rate"])).astype(np.float32)
    audio /= np.abs(audio).max(axis=1, keepdims=True)
    S = np.stack([stft.hybrid_spectrogram(a, 512, args["spec_width"]) for a in audio])[..., None]
    ref = float_graph.forward(spec, S, np.float64)
    for fuse in (True, False):
        r = HipRunner(lower_f32(spec, fuse=fuse), max_batch=4)
        got = r.predict(S)
        from_audio = r.infer_audio_device(torch_mod.from_numpy(audio).cuda()).cpu().numpy()
        r.close()
        assert np.abs(got - ref).max() < 1e-5 and np.abs(from_audio - ref).max() < 1e-4
        for b in range(3):
            assert 1.0 - cosine(got[b], ref[b]) < 1e-6
