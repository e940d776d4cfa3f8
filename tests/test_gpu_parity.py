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
def test_f32_graph_per_layer_and_logits(torch_mod, oracle_specs):
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models.runners import load_model_runner
    from oracle import float_graph

    spec = load_keras_archive(KERAS_PATH)
    x = oracle_specs[..., None]
    ref_scores, ref_logits, acts = float_graph.forward(spec, x, np.float64, return_all=True, return_logits=True)
    runner = load_model_runner(KERAS_PATH, max_batch=16, keep_all=True)
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


# --------------------------------------------------------------------------------------- INT8 graph
def test_i8_graph_bit_exact_per_tensor(torch_mod, oracle_specs):
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import load_model_runner
    from oracle.int8_graph import Int8Interpreter

    model = load_tflite(TFLITE_PATH)
    x = oracle_specs[..., None]
    ref, env = Int8Interpreter(model).invoke(x, return_all=True)
    runner = load_model_runner(TFLITE_PATH, max_batch=16, keep_all=True)
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
    for b in range(audio24.shape[0]):
        assert cosine(logits[b], ref_logits[b]) >= 0.999
        assert logits[b].argmax() == ref_logits[b].argmax()
    runner.close()
