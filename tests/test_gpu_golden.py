"""The HIP path against the COMMITTED vectors of tests/golden/oracle_vectors.npz (written by tests/golden/make_golden.py from the CPU oracle in
the build container) — not against the live oracle, so that oracle and kernels cannot drift together unnoticed between rounds (VERDICT r3,
weak 1 / item 4).  The reference's own tests hold no vectors for these numerics: the fixtures are oracle-defined.

Covered: the reference's test signals (sine 1 kHz, white noise rng(42), chirp 500 -> 4000 Hz; reference tests/conftest.py:49-81) at 22 050 and
24 000 Hz through the float64-exact STFT, the float32 plan (frontend, pooled features, logits, scores) and the INT8 plan (EVERY int8 tensor of the
shipped graph by checksum, the classifier bytes, the scores); BASELINE configs[4]'s topology in float32 and through this build's INT8 exporter
(the exported file's hash, every int8 tensor, the scores); and the float-arithmetic form of int8 MEAN on the device against the oracle's.
"""

from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
from conftest import GOLDEN, KERAS_PATH, TFLITE_PATH, cosine, fixture_signals  # noqa: E402

sys.path.insert(0, GOLDEN)


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; the product has no CPU path to fall back to")
    return torch


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))


def _checksums(a: np.ndarray) -> tuple[int, int]:
    flat = np.asarray(a).astype(np.int64).reshape(-1)
    return int(flat.sum()), int(((np.arange(flat.size, dtype=np.int64) % 65521 + 1) * flat).sum())


def _check_int8_tensors(runner, B: int, rows: np.ndarray, b: int | None = None, what: str = "") -> int:
    """Every plan operator named after a graph tensor (``t<index>``) against the committed (sum, weighted sum) of that tensor; the QUANTIZE
    operator's output is the transposed, zero-padded spectrogram (not the graph's layout) and is skipped."""
    want = {int(r[0]): (int(r[1]), int(r[2])) for r in rows}
    n = 0
    for oi, op in enumerate(runner.plan.ops):
        if op.out < 0 or not op.name.startswith("t") or op.kind == 20 or int(op.name[1:]) not in want:
            continue
        a = runner.op_output(oi, B)
        if a.dtype != np.int8:
            continue
        a = a if b is None else a[b]
        assert _checksums(a) == want[int(op.name[1:])], f"{what}: tensor {op.name} (plan operator {oi}, kind {op.kind})"
        n += 1
    return n


@pytest.mark.parametrize("sr", [22050, 24000])
def test_shipped_checkpoint_against_committed_vectors(torch_mod, golden, sr):
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner, stft_device

    sig = fixture_signals(sr)
    names = ("sine", "noise", "chirp")
    audio = np.stack([sig[n] for n in names]).astype(np.float32)
    hop = audio.shape[1] // 256
    f32 = load_model_runner(KERAS_PATH, max_batch=4, keep_all=True)
    i8 = load_model_runner(TFLITE_PATH, max_batch=4, keep_all=True)
    d_audio = torch.from_numpy(audio).cuda()
    spec = stft_device(f32.ctx, d_audio, 512, hop, 256, True, exact=True).cpu().numpy()  # bn_stft_mag_exact: the reference's float64 arithmetic
    x = spec[..., None]
    # ---- STFT
    for b, n in enumerate(names):
        rows = golden[f"{n}_{sr}/spec_rows"]
        assert np.abs(spec[b, ::16, :] - rows).max() <= 1.5e-7, n          # one float32 step at most (values in [0, 1])
        assert np.array_equal(np.rint(spec[b, ::16, :] * 255.0), np.rint(rows * 255.0)), n   # ... and the same quantised byte everywhere
        assert abs(float(spec[b].astype(np.float64).sum()) - float(golden[f"{n}_{sr}/spec_sum"])) <= 1e-6 * float(golden[f"{n}_{sr}/spec_sum"])
    # ---- float32 plan
    probs = f32.predict(x)
    _, logits = f32.predict_device(torch.from_numpy(x.reshape(3, -1)).cuda(), return_logits=True)
    logits = logits.cpu().numpy()
    acts = {op.name: oi for oi, op in enumerate(f32.plan.ops) if op.out >= 0}
    for b, n in enumerate(names):
        assert 1.0 - cosine(logits[b], golden[f"{n}_{sr}/logits"]) < 1e-5
        assert np.abs(probs[b] - golden[f"{n}_{sr}/probs"]).max() < 1e-5
        if "audio_frontend" in acts:
            fe = f32.op_output(acts["audio_frontend"], 3)[b].reshape(64, 256)[:, ::8]
            ref = golden[f"{n}_{sr}/frontend"]
            assert np.abs(fe - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-12)
        if "gap" in acts:
            gap = f32.op_output(acts["gap"], 3)[b].reshape(-1)
            ref = golden[f"{n}_{sr}/gap"]
            assert np.abs(gap - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-12)
    # ---- INT8 plan: every tensor, the classifier bytes, the scores — at the runner boundary and from audio (bit-exact input bytes)
    q = i8.predict(x)
    base = load_model_runner(TFLITE_PATH, max_batch=4, keep_all=True, fuse=False)  # one kernel per graph operator: every conv / add tensor exists
    qb = base.predict(x)
    for b, n in enumerate(names):
        assert _check_int8_tensors(i8, 3, golden[f"{n}_{sr}/i8_tensors"], b, f"{n}_{sr} fused") >= 12
        assert _check_int8_tensors(base, 3, golden[f"{n}_{sr}/i8_tensors"], b, f"{n}_{sr} baseline") >= 24
        assert np.array_equal(q[b], golden[f"{n}_{sr}/i8_probs"]) and np.array_equal(qb[b], golden[f"{n}_{sr}/i8_probs"])
    base.close()
    fc = next(oi for oi, op in enumerate(i8.plan.ops) if op.name == "t128")
    assert np.array_equal(i8.op_output(fc, 3).reshape(3, -1), np.stack([golden[f"{n}_{sr}/i8_fc"] for n in names]))
    prod = load_model_runner(TFLITE_PATH, max_batch=4)
    from_audio = prod.infer_audio_device(d_audio, hop=hop).cpu().numpy()
    assert np.array_equal(from_audio, np.stack([golden[f"{n}_{sr}/i8_probs"] for n in names]))
    for r in (f32, i8, prod):
        r.close()


def test_configs4_topology_against_committed_vectors(torch_mod, golden):
    """BASELINE configs[4] (raw frontend + PCEN + alpha = 1.5 IR / SE, seeded weights, 2 s @ 24 kHz): the float32 plan against the float64 oracle's
    committed logits, the INT8 export reproduced byte for byte (its hash) and run bit-exactly (every int8 tensor, the scores)."""
    from make_golden import config4_inputs, config4_model

    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models._tflite_reader import parse_tflite
    from birdnet_stm32.models.runners import HipRunner

    spec, raw = config4_model()
    assert np.array_equal(np.frombuffer(hashlib.sha256(raw).digest(), np.uint8), golden["c4/tflite_sha256"]), "the exporter no longer writes the committed graph"
    x = config4_inputs()
    B = x.shape[0]
    f32 = HipRunner(lower_f32(spec), max_batch=B)
    probs = f32.predict(x)
    _, logits = f32.predict_device(f32._torch.from_numpy(x.reshape(B, -1)).cuda(), return_logits=True)
    logits = logits.cpu().numpy()
    for b in range(B):
        assert 1.0 - cosine(logits[b], golden["c4/f32_logits"][b]) < 1e-4   # bar: 1e-3 cosine distance
    assert np.abs(probs - golden["c4/f32_probs"]).max() < 1e-4
    f32.close()
    model = parse_tflite(raw)
    dbg = HipRunner(lower_i8(model, keep_all=True), max_batch=B)
    got = dbg.predict(x)
    assert _check_int8_tensors(dbg, B, golden["c4/i8_tensors"], None, "configs[4] INT8") >= 40
    assert np.allclose(got, golden["c4/i8_probs"], atol=1e-6)               # (float32 softmax behind DEQUANTIZE)
    dbg.close()
    prod = HipRunner(lower_i8(model), max_batch=B)                           # production plan: fused gates, dense kernels, pooled sums
    assert np.array_equal(prod.predict(x), got)
    prod.close()


@pytest.mark.parametrize("which", ["shipped", "ds_se_emb_sigmoid"])
def test_float_form_of_int8_mean_on_the_device(torch_mod, which):
    """``lower_i8(mean_form='float')``: every MEAN kernel of the device (i8_mean_kernel, the squeeze-excite gate, the fused tail) evaluates TFLite's
    float-arithmetic QuantizedMeanOrSum — bit for bit the oracle's ``mean_form='float'``, and different from the integer form where that one is."""
    from birdnet_stm32 import _hip
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import HipRunner
    from oracle.int8_graph import Int8Interpreter

    if which == "shipped":
        from conftest import synth_chunks
        from oracle import stft

        model = load_tflite(TFLITE_PATH)
        rng = np.random.default_rng(2)
        x = np.concatenate([np.stack([stft.hybrid_spectrogram(a) for a in synth_chunks(8, seed=4)])[..., None],
                            (rng.random((32, 257, 256, 1), dtype=np.float32) ** 3).astype(np.float32)])
    else:
        from test_conversion import EXPORT_TOPOLOGIES, _export

        _, model, _, x = _export(EXPORT_TOPOLOGIES[which])
    B = x.shape[0]
    want, env = Int8Interpreter(model, mean_form="float").invoke(x, return_all=True)
    base = Int8Interpreter(model).invoke(x)
    for fuse in (True, False):
        r = HipRunner(lower_i8(model, keep_all=True, fuse=fuse, mean_form="float"), max_batch=B)
        got = r.predict(x)
        assert np.array_equal(got, want), (which, fuse)
        for oi, op in enumerate(r.plan.ops):
            if op.kind != 25 or op.out < 0 or not op.name.startswith("t"):  # (25 = I8_MEAN)
                continue
            a = r.op_output(oi, B)
            assert np.array_equal(a, np.asarray(env[int(op.name[1:])]).reshape(a.shape)), (which, fuse, op.name)
        r.close()
    prod = HipRunner(lower_i8(model, mean_form="float"), max_batch=B)   # fused tail / gate kernels, sums pooled by the depthwise kernel
    assert np.array_equal(prod.predict(x), want)
    with _hip.options(i8_tail=0, i8_dw_pool=0):
        assert np.array_equal(prod.predict(x), want)
    prod.close()
    if which == "shipped":
        assert not np.array_equal(want, base)  # the two forms really differ on this input set
