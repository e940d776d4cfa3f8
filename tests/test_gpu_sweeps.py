"""GPU parity sweeps at BASELINE sizes and the fixed-point primitives pinned on the device.

* INT8 at the runner boundary: 4096 spectrograms (BASELINE configs[2]'s batch) through the production plan, bit for bit against the C
  port of the TFLite reference kernels (``oracle/c/oracle_i8.c``, itself identical per tensor to ``oracle/int8_graph.py``).
* INT8 from audio: 2048 chunks, bit for bit against the oracle fed with the float64 oracle STFT (quantised input bytes, pre-sigmoid
  bytes, scores); ``bn_stft_mag_exact`` against the oracle STFT.
* ``bn_requant.h`` (every requantisation form the kernels use) against the literal gemmlowp definitions on edge cases.
* float32 row-streaming depthwise kernel (``f32_dw_stream_kernel``) against the baseline depthwise kernel at production batch sizes.
* BASELINE configs[4] (raw + PCEN + alpha 1.5 IR/SE) at its full batch through size-independent properties.

All comparisons go through the C ABI (ctypes -> libbirdnet_hip.so).
"""

import ctypes
import os

import numpy as np
import pytest

from conftest import I32_MAX, I32_MIN, KERAS_PATH, TFLITE_PATH, cosine, mbqm_def, synth_chunks

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; the product has no CPU path to fall back to")
    return torch


def _requant_cases():
    rng = np.random.default_rng(5)
    xs = [0, 1, -1, 2, -2, 3, -3, (1 << 30) - 1, -(1 << 30) + 1, 1 << 29, -(1 << 29), 12345677, -12345677, 255 * 127 * 256, -255 * 127 * 256]
    ms = [1 << 30, (1 << 31) - 1, (1 << 30) + 1, 1518500250, 1073741825, 2000000000, 0]
    shs = [-1, -2, -3, -7, -8, -12, -20, -22]
    cases = [(x, m, s) for x in xs for m in ms for s in shs]
    # negative-half ties of the rounding shift: srdhm result = -(2k + 1) * 2^(e-1)
    for e in (1, 2, 5, 9):
        for k in (0, 1, 7):
            v = -(2 * k + 1) * (1 << (e - 1))
            cases.append((2 * v, 1 << 30, -e))  # srdhm(2v, 2^30) = v exactly
            cases.append((-2 * v, 1 << 30, -e))
    x = rng.integers(-(1 << 30) + 1, 1 << 30, size=20000)
    m = rng.integers(1 << 30, 1 << 31, size=20000)
    s = -rng.integers(1, 23, size=20000)
    cases += list(zip(x.tolist(), m.tolist(), s.tolist()))
    return cases


def test_requant_forms_match_gemmlowp_definitions(torch_mod):
    """bn_requant.h on the device (through bn_debug_requant) against SaturatingRoundingDoublingHighMul + RoundingDivideByPOT written
    out with Python integers: the general form (incl. left shifts, negative multipliers, INT32_MIN x INT32_MIN saturation, shift 0 and
    shifts up to 31), the branch-free right-shift form and the strip kernels' folded-addend form (with zero points at both ends)."""
    torch = torch_mod
    from birdnet_stm32 import _hip

    ctx = _hip.Context(0, 4)

    def run(cases, mode, zp=0):
        x, m, s = (torch.tensor([c[i] for c in cases], dtype=torch.int32, device="cuda") for i in range(3))
        out = torch.empty_like(x)
        _hip.check(ctx.lib.bn_debug_requant(ctx.handle, x.data_ptr(), m.data_ptr(), s.data_ptr(), len(cases), mode, zp, out.data_ptr(), None))
        torch.cuda.synchronize()
        return out.cpu().numpy().astype(np.int64)

    right = _requant_cases()
    want = np.array([mbqm_def(*c) for c in right], np.int64)
    for mode in (0, 1, 2):
        got = run(right, mode)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"mode {mode}: {bad.size} mismatches, first {right[bad[0]]}: got {got[bad[0]]}, want {want[bad[0]]}"
    for zp in (-128, -34, 0, 73, 127, 255):  # + 128 of the ADD blocks included
        got = run(right, 3, zp)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"strip form, zp {zp}: first mismatch {right[bad[0]]}: got {got[bad[0]]}, want {want[bad[0]]}"
    # outside the fast forms' domain (|x| >= 2^30, left shifts, negative multipliers): the literal definitions (mode 1) everywhere,
    # the dispatching form (mode 0) wherever |x| < 2^30 — the bound the lowering pass proves for every accumulator (_lower_i8.py)
    wide = [(I32_MIN, I32_MIN, 0), (I32_MIN, I32_MIN, -1), (I32_MIN, I32_MAX, 0), (I32_MAX, I32_MAX, 0), (I32_MAX, I32_MIN + 1, -3),
            (I32_MIN, 1 << 30, -31), (I32_MAX, (1 << 31) - 1, -31), (-1, 1 << 30, -31), (5, 1 << 30, 0), (-5, 1 << 30, 0),
            (1000, 1 << 30, 3), (-1000, 1 << 30, 3), (1 << 26, 1518500250, 4), (-(1 << 26), 1518500250, 4), (77, -(1 << 30), -2), (-77, -(1 << 30), -2),
            (123456, -1518500250, -5), (-123456, -1518500250, 2), (0, 0, 0), (99, 0, -4), ((1 << 30) - 1, (1 << 31) - 1, -31), (-(1 << 30) + 1, (1 << 31) - 1, -31),
            ((1 << 30) - 1, (1 << 31) - 1, 0), (-(1 << 30) + 1, 1 << 30, -1)]
    want = np.array([mbqm_def(*c) for c in wide], np.int64)
    got = run(wide, 1)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"mode 1: first mismatch {wide[bad[0]]}: got {got[bad[0]]}, want {want[bad[0]]}"
    inside = [i for i, c in enumerate(wide) if abs(c[0]) < (1 << 30)]
    got = run([wide[i] for i in inside], 0)
    bad = np.nonzero(got != want[inside])[0]
    assert bad.size == 0, f"mode 0: first mismatch {wide[inside[bad[0]]]}: got {got[bad[0]]}, want {want[inside[bad[0]]]}"
    ctx.close()


# ------------------------------------------------------------------------------------------- INT8 sweeps
def _c_int8_path():
    from birdnet_stm32.models._tflite_reader import load_tflite
    from oracle import cport

    import os

    if not (os.path.isfile(cport.I8_LIB) and os.path.isfile(cport.CPU_LIB)):
        pytest.fail("oracle/_build is missing: run __graft_entry__.build() (the checker of this test is the C port of the oracle)")
    return cport.CpuInt8Program(load_tflite(TFLITE_PATH))   # (the whole graph per chunk in C: pinned to the numpy interpreter by tests/test_oracle_pinning.py)


def test_i8_runner_boundary_4096_spectrograms_bit_exact(torch_mod):
    """BASELINE configs[2]'s batch at the runner boundary: 3584 spectrograms of the synthetic chunks + 512 random ones (uniform noise,
    values on the quantiser's steps, silence, a single spike) through the PRODUCTION plan == the C port of the TFLite reference kernels,
    score for score (the dequantised int8 sigmoid and the pre-sigmoid outputs)."""
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner

    path = _c_int8_path()
    S = np.empty((4096, 257, 256, 1), np.float32)
    S[:3584] = path.spectrogram(synth_chunks(3584, seed=1234), 281, 256)
    rng = np.random.default_rng(99)
    S[3584:] = rng.random((512, 257, 256, 1), dtype=np.float32)
    S[3584:3592] *= np.float32(1.0 / 255.0) * np.arange(0, 256, 32, dtype=np.float32)[:, None, None, None]
    S[3592:3600] = (np.round(S[3592:3600] * 255.0) / 255.0).astype(np.float32)  # exactly on the quantiser's steps
    S[3600] = 0.0
    S[3601] = 0.0
    S[3601, 40, 100, 0] = 1.0
    S[3602] = 1.0
    want = np.concatenate([path.invoke(S[i : i + 512]) for i in range(0, 4096, 512)])
    runner = load_model_runner(TFLITE_PATH, max_batch=4096)
    x = torch.from_numpy(S.reshape(4096, -1)).cuda()
    scores, logits = runner.predict_device(x, return_logits=True)
    got = scores.cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of 4096 chunks differ from the oracle, first {bad[:5].tolist()}"
    # pre-sigmoid outputs: dequantised FC output of the oracle
    from birdnet_stm32.models._tflite_reader import load_tflite

    model = load_tflite(TFLITE_PATH)
    fc = model.ops[53].outputs[0]
    s, z = float(model.tensors[fc].scale[0]), int(model.tensors[fc].zero_point[0])
    _, env = path.invoke(S[:512], return_all=True)
    assert np.array_equal(logits[:512].cpu().numpy(), (env[fc].astype(np.float32) - z) * np.float32(s))
    runner.close()


def _pathological_chunks(T=72000, sr=24000):
    """Chunks on which the float32 STFT's bound is useless or the extrema are ambiguous: they take the whole-chunk float64 route."""
    t = np.arange(T) / sr
    rng = np.random.default_rng(3)
    out = [np.sin(2 * np.pi * 440.0 * t), np.ones(T), np.zeros(T), np.sign(np.sin(2 * np.pi * 1000.0 * t))]
    imp = np.zeros(T)
    imp[5000] = 1.0
    out.append(imp)
    tail = np.zeros(T)
    tail[:30000] = rng.standard_normal(30000)  # a short file padded with zeros once (split_audio_into_chunks)
    out.append(tail)
    out.append(np.sin(2 * np.pi * 46.875 * 20 * t))  # exactly on a bin
    return np.stack(out).astype(np.float32)


def test_i8_from_audio_2048_chunks_bit_exact(torch_mod):
    """From audio the INT8 path must give the reference's integers: the float32 STFT's magnitudes differ from librosa's float64
    arithmetic by ~1e-6 of the peak (3e-6 of the quantised bytes flipped in round 2), so bn_infer_audio recomputes in float64 every
    element whose byte is in doubt, and the chunk's min / max (csrc/bn_stft_exact.hip).  Asserted on 2048 chunks (synthetic
    tone + noise, silence, quiet chunks, and the pathological ones that take the whole-chunk float64 route): NO quantised input byte
    differs from the oracle's (float64 STFT -> complex64 -> numpy |.| -> float32 min-max -> QUANTIZE), pre-sigmoid bytes and scores
    are identical, hence top-1 agreement is 1.0.  The same bytes come out of stft_exact = 1 (every bin in float64); stft_exact = 0
    (round 2's plain float32 STFT) is shown to flip some, so the test can tell the difference."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import load_model_runner
    from oracle import stft

    path = _c_int8_path()
    N = 2048
    audio = synth_chunks(N, seed=77)
    audio[:4] = 0.0  # silence
    audio[4:8] *= 1e-3  # quiet chunks (the normalisation is scale-free)
    hard = _pathological_chunks()
    audio[8 : 8 + len(hard)] = hard
    S_ref = np.stack([stft.hybrid_spectrogram(a, 512, 256) for a in audio])[..., None].astype(np.float32)
    model = load_tflite(TFLITE_PATH)
    fc = model.ops[53].outputs[0]
    s_fc, z_fc = float(model.tensors[fc].scale[0]), int(model.tensors[fc].zero_point[0])
    qin = model.ops[0].outputs[0]
    ref_scores, ref_q, ref_fc = [], [], []
    for i in range(0, N, 512):
        sc, env = path.invoke(S_ref[i : i + 512], return_all=True)
        ref_scores.append(sc)
        ref_q.append(env[qin].reshape(len(sc), -1))
        ref_fc.append(env[fc].reshape(len(sc), -1).astype(np.int32))
    ref_scores, ref_q, ref_fc = np.concatenate(ref_scores), np.concatenate(ref_q), np.concatenate(ref_fc)

    runner = load_model_runner(TFLITE_PATH, max_batch=N)
    d_audio = torch.from_numpy(audio).cuda()
    for mode in (2, 1):
        with _hip.options(stft_exact=mode):
            scores, logits = runner.infer_audio_device(d_audio, return_logits=True)
            scores, logits = scores.cpu().numpy(), logits.cpu().numpy()
            q_gpu = runner.input_bytes(N).reshape(N, -1)
            if mode == 2:
                st = runner.guard_stats(N)
                print(f"exactness pass, {N} chunks: {st['listed'] / (N * q_gpu.shape[1]):.2e} of the elements recomputed in float64, "
                      f"{st['dirty_blocks']} (chunk, 64-frame block) pairs changed, {st['whole_minmax']} + {st['whole_fix']} chunks as whole float64 spectrograms")
                # (the pathological chunks: their minimum enclosed in an interval since round 5 — the whole-chunk route is taken below, with stft_minint = 0)
                assert st["interval_min"] + st["whole_minmax"] >= 3 and st["listed"] < 2e-3 * N * q_gpu.shape[1]
        flipped = int((q_gpu != ref_q).sum())
        assert flipped == 0, f"stft_exact={mode}: {flipped} quantised input bytes differ from the oracle's"
        got_fc = np.rint(logits / np.float32(s_fc)).astype(np.int32) + z_fc
        assert np.array_equal(got_fc, ref_fc), f"stft_exact={mode}: pre-sigmoid bytes differ"
        assert np.array_equal(scores, ref_scores), f"stft_exact={mode}: scores differ"
        assert (got_fc.argmax(axis=1) == ref_fc.argmax(axis=1)).all()
    with _hip.options(stft_minint=0):  # rounds 3-4: chunks whose extrema have too many candidates as whole float64 spectrograms — same bytes, same scores
        scores = runner.infer_audio_device(d_audio).cpu().numpy()
        st = runner.guard_stats(N)
        assert st["whole_minmax"] >= 3 and st["interval_min"] == 0
        assert int((runner.input_bytes(N).reshape(N, -1) != ref_q).sum()) == 0 and np.array_equal(scores, ref_scores)
    # the give-up route of the mixer (a workgroup with more elements in doubt than it keeps — here forced by keeping at most 2): those chunks
    # are recomputed as whole float64 spectrograms and all their blocks run through the mixer once more; same bytes, same scores
    with _hip.options(stft_flagcap=2):
        scores = runner.infer_audio_device(d_audio).cpu().numpy()
        st = runner.guard_stats(N)
        assert st["whole_fix"] > N // 2 and st["dirty_blocks"] >= 4 * st["whole_fix"]
        assert int((runner.input_bytes(N).reshape(N, -1) != ref_q).sum()) == 0 and np.array_equal(scores, ref_scores)
    with _hip.options(stft_exact=0):  # the plain float32 STFT of round 2: a few bytes off by one (this is what the pass above removes)
        runner.infer_audio_device(d_audio)
        dq = runner.input_bytes(N).reshape(N, -1).astype(np.int32) - ref_q.astype(np.int32)
        assert 0 < (dq != 0).mean() < 2e-3 and np.abs(dq).max() == 1
    runner.close()


def test_stft_mag_exact_equals_the_float64_oracle(torch_mod):
    """bn_stft_mag_exact (every bin a float64 DFT, |.| by numpy's float32 formula) against oracle/stft.py at 24 kHz and 22.05 kHz:
    every float32 value identical on ordinary chunks, at most one ulp off on a <= 2e-5 share of the elements otherwise (see below),
    identical 8-bit quantisation of the normalised spectrogram everywhere."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models.runners import stft_device
    from oracle import stft

    ctx = _hip.Context(0, 32)
    for sr in (24000, 22050):
        chunks = synth_chunks(24, sr=sr, seed=5)
        chunks[0] = 0.0
        if sr == 24000:
            chunks[1 : 1 + 7] = _pathological_chunks()
        T = chunks.shape[1]
        d = torch.from_numpy(chunks).cuda()
        raw = stft_device(ctx, d, normalize=False, exact=True).cpu().numpy()
        nrm = stft_device(ctx, d, normalize=True, exact=True).cpu().numpy()
        n_equal = 0
        for b, a in enumerate(chunks):
            S = stft.stft_magnitude(a, 512, T // 256)[:, :256]
            xp = np.pad(a.astype(np.float64), 256)
            idx = np.arange(512)[None, :] + (T // 256) * np.arange(256)[:, None]
            norm = np.sqrt((xp[idx] ** 2).sum(-1))[None, :]
            # An element whose float64 value lies within the FFT's own rounding noise (~1e-16 of the frame's norm) of a complex64
            # rounding boundary is decided by the summation order — numpy's FFT and the DFT here may then differ by one float32 ulp.
            # For elements of the order of the norm that is a ~1e-8 event; for the far leakage bins of a noise-free tone (1e-6 of the
            # norm and below) it is common, and such elements are the reference's own rounding noise.
            solid = S > 1e-3 * norm
            diff = raw[b][solid] != S[solid]
            assert diff.sum() <= 2e-5 * max(int(solid.sum()), 1) + 1, f"sr={sr} chunk {b}: {int(diff.sum())} of {int(solid.sum())} magnitudes differ"
            assert np.abs(raw[b].astype(np.float64) - S).max() <= 1.2e-7 * max(float(S.max()), 1e-30)
            want = stft.minmax_normalize(S)
            assert np.array_equal(np.rint(nrm[b] * 255.0), np.rint(want * 255.0)), f"sr={sr} chunk {b}: quantised values differ"
            n_equal += int(np.array_equal(raw[b], S) and np.array_equal(nrm[b], want))
        assert n_equal >= 14  # (the ordinary chunks: every float32 value equal)
    ctx.close()


# ----------------------------------------------------------------------- float32 row-streaming depthwise kernel
def test_f32_dw_stream_matches_baseline_kernel(torch_mod):
    """Every stand-alone depthwise 3x3 of an inverted-residual + squeeze-excite model (stride 1 and the stride-2 stage openers):
    ``f32_dw_stream_kernel`` (option f32_strip = 1) against the baseline ``f32_dw_kernel`` (f32_strip = 0), within 1e-6 of the map's
    peak (float32 round-off of a nine-term sum), for rows-per-wave values that move the row-block borders everywhere, at an
    odd batch (130) and a production-size one (1024: the launcher's own choice of 16 rows per wave), 50 launches each — the store-data
    hazard class of these row-streaming kernels showed up once in 50-100 launches."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner

    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=3, embeddings_size=256, num_classes=10,
                       randomize_bn=True, seed=7)
    rng = np.random.default_rng(17)
    for B, ths in ((130, (0, 1, 3, 5, 7, 64) * 8), (1024, (0, 3, 64))):
        runner = HipRunner(lower_f32(spec, keep_all=True), max_batch=B)
        dw_ops = [oi for oi, op in enumerate(runner.plan.ops) if op.kind == pk.F32_DW]
        assert len(dw_ops) >= 8 and {runner.plan.ops[oi].p[3] for oi in dw_ops} == {1, 2}
        x = torch.from_numpy(rng.random((B, 257 * 256), dtype=np.float32)).cuda()
        with _hip.options(f32_strip=0):
            want_scores = runner.predict_device(x).clone()
            want = {oi: torch.from_numpy(runner.op_output(oi, B)) for oi in dw_ops}
        launches = 0
        for th in ths:
            with _hip.options(f32_strip=1, f32_strip_th=th):
                got_scores = runner.predict_device(x)
                launches += 1
                for oi in dw_ops:
                    a = torch.from_numpy(runner.op_output(oi, B))
                    err = float((a - want[oi]).abs().max() / want[oi].abs().max())
                    if err > 1e-6:
                        raise AssertionError(f"B={B}, rows per wave {th or 'auto'}: depthwise {runner.plan.ops[oi].name} differs (relative to peak {err:.3e})")
                assert float((got_scores - want_scores).abs().max()) < 5e-6
        # the remaining launches of the 50: scores only (every depthwise feeds them)
        with _hip.options(f32_strip=1):
            for _ in range(50 - launches):
                assert float((runner.predict_device(x) - want_scores).abs().max()) < 5e-6
        runner.close()


# -------------------------------------------------------------------------------- configs[4] at its full batch
@pytest.mark.parametrize("seconds", [2, 3])
def test_config5_full_batch_properties(torch_mod, seconds):
    """BASELINE configs[4] (raw learned-filterbank frontend + PCEN + alpha 1.5 DS-CNN with SE / inverted residuals, seeded weights,
    24 kHz x 2 s as the reference's raw frontend builds, and x 3 s — the metric's chunk — with its length guard lifted) at B = 1024: a chunk's scores do not depend on its batch neighbours, position or batch slicing (== the same chunks
    through a 96-chunk workspace, where the per-layer parity test against the oracle runs), repeated runs are bit-identical, softmax rows
    sum to one; and the first chunks agree with the float64 oracle."""
    torch = torch_mod
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from oracle import float_graph

    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=seconds, embeddings_size=256, num_classes=100,
                       audio_frontend="raw", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42,
                       **({"raw_length_limit": None} if seconds == 3 else {}))
    base = torch.from_numpy(synth_chunks(64)[:, : 24000 * seconds].copy()).cuda()
    B = 1024
    idx = torch.randint(0, 64, (B,), generator=torch.Generator().manual_seed(3)).cuda()
    audio = base[idx].contiguous()
    big = HipRunner(lower_f32(spec), max_batch=B)
    s1 = big.infer_audio_device(audio).clone()
    s2 = big.infer_audio_device(audio)
    assert torch.equal(s1, s2), "run-to-run determinism"
    small = HipRunner(lower_f32(spec), max_batch=96)
    ref64 = small.infer_audio_device(base)
    assert torch.equal(s1, ref64[idx]), "a chunk's scores depend on its batch position / neighbours"
    assert torch.isfinite(s1).all() and float((s1.sum(dim=1) - 1).abs().max()) < 1e-5
    x = base[:4].cpu().numpy()
    x = (x / (np.abs(x).max(axis=1, keepdims=True) + 1e-6)).astype(np.float32)[..., None]
    ref = float_graph.forward(spec, x, np.float64)
    got = ref64[:4].cpu().numpy()
    for b in range(4):
        assert 1.0 - cosine(got[b], ref[b]) < 1e-5
    big.close()
    small.close()


def test_config5_int8_full_batch_properties(torch_mod):
    """BASELINE configs[4] through this build's INT8 exporter (raw frontend, 2 s @ 24 kHz — the bench line's ``also_measured_configs4_int8``) at
    B = 1024, where the persistent pointwise kernels walk long runs of groups per wave (the per-tensor parity against the INT8 oracle runs at small
    batches in tests/test_conversion.py): repeated runs are bit-identical, a chunk's scores do not depend on its batch position, neighbours or the
    workspace size, and every kernel-selection switch of the exported-graph path reproduces the same bytes at this size."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.conversion.export import convert_netspec_to_int8
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models._tflite_reader import parse_tflite
    from birdnet_stm32.models._tflite_writer import write_tflite
    from birdnet_stm32.models.runners import HipRunner

    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2, embeddings_size=256, num_classes=100,
                       audio_frontend="raw", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
    rng = np.random.default_rng(0)
    cal = [rng.standard_normal((1, 48000, 1)).astype(np.float32) for _ in range(8)]
    cal = [c / (np.abs(c).max() + 1e-6) for c in cal]
    model = parse_tflite(write_tflite(convert_netspec_to_int8(spec, lambda: ([c] for c in cal))))
    base = torch.from_numpy(synth_chunks(64)[:, :48000].copy()).cuda()
    base = base / base.abs().amax(dim=1, keepdim=True)
    B = 1024
    idx = torch.randint(0, 64, (B,), generator=torch.Generator().manual_seed(5)).cuda()
    audio = base[idx].contiguous()
    big = HipRunner(lower_i8(model), max_batch=B)
    s1 = big.infer_audio_device(audio).clone()
    assert torch.equal(s1, big.infer_audio_device(audio)), "run-to-run determinism"
    assert torch.isfinite(s1).all() and float((s1.sum(dim=1) - 1).abs().max()) < 1e-5
    small = HipRunner(lower_i8(model), max_batch=96)
    ref64 = small.infer_audio_device(base)
    assert torch.equal(s1, ref64[idx]), "a chunk's scores depend on its batch position / neighbours"
    small.close()
    for opts in (dict(i8_add_tab=0), dict(i8_pw_forms=0), dict(i8_pw_lds=0), dict(i8_dw_pool=0), dict(i8_pw_forms=0, i8_add_tab=0, i8_pw_lds=0)):
        with _hip.options(**opts):
            assert torch.equal(big.infer_audio_device(audio), s1), opts
    big.close()


# ------------------------------------------------------------------------------------------ fused INT8 tail
def test_i8_fused_tail_matches_per_block_kernels_and_oracle(torch_mod):
    """The back half of the INT8 graph as one kernel (i8_tail_kernel: stage 3-4 + MEAN + FULLY_CONNECTED + head, maps in LDS) against
    the per-block kernels it replaces (option i8_tail = 0) and the oracle: scores and pre-sigmoid outputs bit for bit, for batch sizes
    that leave the last group of four chunks ragged, repeated launches, and from audio."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models.runners import load_model_runner

    path = _c_int8_path()
    rng = np.random.default_rng(21)
    S = np.empty((261, 257, 256, 1), np.float32)
    S[:200] = path.spectrogram(synth_chunks(200, seed=5), 281, 256)
    S[200:] = rng.random((61, 257, 256, 1), dtype=np.float32)
    want = path.invoke(S)
    runner = load_model_runner(TFLITE_PATH, max_batch=261)
    tail = [o for o in runner.plan.ops if o.kind == pk.I8_TAIL]
    assert len(tail) == 1 and sum(o.p[pk.TAIL_TAG] == pk.TAIL_COVERED for o in runner.plan.ops) == 9
    x = torch.from_numpy(S.reshape(261, -1)).cuda()
    with _hip.options(i8_tail=0):
        base_s, base_l = (t.clone() for t in runner.predict_device(x, return_logits=True))
    assert np.array_equal(base_s.cpu().numpy(), want)
    for rep in range(6):
        for nb in (261, 1, 2, 3, 5, 37, 64):
            s, l = runner.predict_device(x[:nb], return_logits=True)
            assert torch.equal(s, base_s[:nb]) and torch.equal(l, base_l[:nb]), f"batch {nb}, launch {rep}"
    # profiling rows: the tail operator is the one that ran
    runner.profile(True)
    runner.predict_device(x)
    rows = {r["kind"]: r["launches"] for r in runner.profile_collect() if r["launches"]}
    runner.profile(False)
    assert rows.get("i8_tail") == 1 and "i8_mean" not in rows
    # both forms of the fused kernel: depthwise stage on the matrix cores (i8_tail2_kernel, the default where the plan carries its
    # constants) and on the vector ALU (i8_tail_kernel)
    assert tail[0].t[2] >= 0 and tail[0].t[3] >= 0, "the shipped graph must take the matrix-core depthwise form"
    assert runner.tail_form()[0] == 2, "the library refused the matrix-core form's LDS plan: the default path would silently be i8_tail_kernel"
    with _hip.options(i8_tail_mfdw=0):
        for nb in (261, 1, 3, 37):
            s, l = runner.predict_device(x[:nb], return_logits=True)
            assert torch.equal(s, base_s[:nb]) and torch.equal(l, base_l[:nb]), f"vector-ALU form, batch {nb}"
    audio = torch.from_numpy(synth_chunks(70, seed=9)).cuda()
    a1 = runner.infer_audio_device(audio).clone()
    with _hip.options(i8_tail=0):
        a0 = runner.infer_audio_device(audio)
    assert torch.equal(a0, a1)
    with _hip.options(i8_tail_mfdw=0):
        assert torch.equal(runner.infer_audio_device(audio), a1)
    runner.close()


def test_i8_fused_stage2_chain_matches_the_strip_kernels_and_oracle(torch_mod):
    """Stage 2 of the shipped INT8 graph as one kernel (i8_mid2_kernel: a stride-2 block with its taps from memory, two residual blocks with
    the maps of two chunks in LDS, depthwise stage on the matrix cores) against the three strip kernels it replaces (option i8_mid = 0):
    the map it hands the tail (operator t110) and the scores bit for bit, for batch sizes that leave the last pair of chunks ragged,
    repeated launches, and from audio; the oracle pins the scores."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models.runners import load_model_runner

    path = _c_int8_path()
    rng = np.random.default_rng(33)
    S = np.empty((131, 257, 256, 1), np.float32)
    S[:100] = path.spectrogram(synth_chunks(100, seed=6), 281, 256)
    S[100:] = rng.random((31, 257, 256, 1), dtype=np.float32)
    want = path.invoke(S)
    runner = load_model_runner(TFLITE_PATH, max_batch=131)
    mids = [i for i, o in enumerate(runner.plan.ops) if o.kind == pk.I8_MID]
    assert len(mids) == 1 and sum(o.p[pk.TAIL_TAG] == pk.MID_COVERED for o in runner.plan.ops) == 3
    assert runner.mid_form()[0] == 1, "the library refused the fused stage-2 chain's LDS plan: the default path would silently be the strip kernels"
    x = torch.from_numpy(S.reshape(131, -1)).cuda()
    with _hip.options(i8_mid=0, i8_tail=0):
        base_s = runner.predict_device(x).clone()
        base_map = runner.op_output(mids[0] - 1, 131)   # stage2_ds3 as its own strip kernel
    assert np.array_equal(base_s.cpu().numpy(), want)
    with _hip.options(i8_tail=0):
        for rep in range(3):
            for nb in (131, 1, 2, 3, 5, 37, 64):
                s = runner.predict_device(x[:nb])
                assert torch.equal(s, base_s[:nb]), f"batch {nb}, launch {rep}"
                assert np.array_equal(runner.op_output(mids[0], nb), base_map[:nb]), f"stage-2 output map, batch {nb}"
    runner.profile(True)
    runner.predict_device(x)
    rows = {r["kind"]: r["launches"] for r in runner.profile_collect() if r["launches"]}
    runner.profile(False)
    assert rows.get("i8_mid") == 1, rows
    assert torch.equal(runner.predict_device(x), base_s)
    audio = torch.from_numpy(synth_chunks(70, seed=9)).cuda()
    a1 = runner.infer_audio_device(audio).clone()
    with _hip.options(i8_mid=0):
        assert torch.equal(runner.infer_audio_device(audio), a1)
    runner.close()


def test_exactness_guard_modes_and_audit(torch_mod):
    """The bound behind "bit-exact from audio" as a switch (option stft_guard) with an audit (option stft_audit):
    * empirical (default) and proven (worst case, docs/exactness.md) bounds give the bytes of the all-float64 STFT (stft_exact = 1) and its scores;
      the proven bound puts many more elements in doubt;
    * the audit re-evaluates the near misses — elements NOT in doubt but within four bounds of a rounding boundary — and finds none wrong
      under either bound;
    * a bound that is far too small on purpose (stft_guard = 2: the empirical constants / 1024, no quantiser slack) leaves wrong bytes behind, and the audit SEES them."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models.runners import load_model_runner

    B = 96
    runner = load_model_runner(TFLITE_PATH, max_batch=B)
    audio = torch.from_numpy(synth_chunks(B, seed=17)).cuda()
    with _hip.options(stft_exact=1):
        want = runner.infer_audio_device(audio).clone()
        want_bytes = runner.input_bytes(B)
    stats = {}
    for mode in (0, 1):
        with _hip.options(stft_guard=mode, stft_audit=1):
            got = runner.infer_audio_device(audio).clone()
            stats[mode] = runner.guard_stats(B)
            assert torch.equal(got, want), f"stft_guard = {mode}"
            assert np.array_equal(runner.input_bytes(B), want_bytes), f"stft_guard = {mode}"
            assert stats[mode]["audited"] > 0 and stats[mode]["audit_violations"] == 0, stats[mode]
        with _hip.options(stft_guard=mode):   # without the audit: the same scores, nothing audited
            assert torch.equal(runner.infer_audio_device(audio), want)
            assert runner.guard_stats(B)["audited"] == 0
    assert stats[1]["listed"] + B * stats[1]["whole_minmax"] > 4 * stats[0]["listed"], stats   # the proven bound doubts far more (or hands whole chunks over)
    with _hip.options(stft_guard=2, stft_audit=1):
        runner.infer_audio_device(audio)
        small = runner.guard_stats(B)
    assert small["audit_violations"] > 0, f"a bound 1024 x too small must leave wrong bytes among its near misses: {small}"
    runner.close()


def test_two_models_in_one_process_run_under_their_own_options(torch_mod):
    """Launcher switches are per context (bn_ctx_set_option) on top of the process default (bn_set_option): two runners of the same model in one
    process, one with the fused kernels switched off for ITS context, give the same scores through different kernels, interleaved; the
    default of the process is untouched; a later change of the process default reaches the context that did not override the switch."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models.runners import load_model_runner

    a = load_model_runner(TFLITE_PATH, max_batch=16)
    b = load_model_runner(TFLITE_PATH, max_batch=16)
    b.ctx.set_option("i8_tail", 0)
    b.ctx.set_option("i8_mid", 0)
    assert (a.ctx.get_option("i8_tail"), b.ctx.get_option("i8_tail"), _hip.get_option("i8_tail")) == (1, 0, 1)
    x = torch.rand((16, 257 * 256), device="cuda")
    for r in (a, b):
        r.profile(True)
    sa, sb = a.predict_device(x).clone(), b.predict_device(x).clone()
    sa2 = a.predict_device(x).clone()
    ka = {q["kind"] for q in a.profile_collect() if q["launches"]}
    kb = {q["kind"] for q in b.profile_collect() if q["launches"]}
    assert torch.equal(sa, sb) and torch.equal(sa, sa2)
    assert {"i8_tail", "i8_mid"} <= ka and not ({"i8_tail", "i8_mid"} & kb) and "i8_mean" in kb
    with _hip.options(i8_tail=0):   # the process default: reaches a (no override), b keeps its own value
        assert (a.ctx.get_option("i8_tail"), b.ctx.get_option("i8_tail")) == (0, 0)
        a.predict_device(x)
        assert "i8_tail" not in {q["kind"] for q in a.profile_collect() if q["launches"]}
    b.ctx.reset_options()
    assert b.ctx.get_option("i8_tail") == 1
    b.predict_device(x)
    assert "i8_tail" in {q["kind"] for q in b.profile_collect() if q["launches"]}
    a.close()
    b.close()


# --------------------------------------------------------------------------------------- float32: front block + stage1_ds2 as one kernel
def test_f32_front2_fused_kernel_matches_the_two_strip_kernels(torch_mod):
    """``f32_front2_kernel`` (front block + the residual block behind it, the 32-channel map in LDS) does the arithmetic of
    ``f32_front_strip_kernel<true>`` followed by ``f32_strip_kernel<2, 32, 1, true>`` in the same order: scores are compared
    BIT FOR BIT with the two-kernel path (option ``f32_front2`` = 0) — from spectrograms and from audio (the finalising variant of
    the front block), batch sizes that leave workgroups idle, repeated launches — and within float32 noise of the tile kernels."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models.runners import load_model_runner

    runner = load_model_runner(KERAS_PATH, max_batch=300)
    heads = [o for o in runner.plan.ops if o.kind == pk.F32_FRONT and o.p[pk.TAIL_TAG] == pk.FRONT2_HEAD]
    covered = [o for o in runner.plan.ops if o.p[pk.TAIL_TAG] == pk.FRONT2_COVERED]
    assert len(heads) == 2 and len(covered) == 1 and covered[0].kind == pk.F32_DWPW  # spectrogram entry, audio entry -> one residual block
    rng = np.random.default_rng(33)
    spec = torch.from_numpy(rng.random((300, 257 * 256), dtype=np.float32)).cuda()
    audio = torch.from_numpy(synth_chunks(300, seed=12)).cuda()
    with _hip.options(f32_front2=0):
        base_s, base_l = (t.clone() for t in runner.predict_device(spec, return_logits=True))
        base_a = runner.infer_audio_device(audio).clone()
    with _hip.options(f32_strip=0):
        tile_s = runner.predict_device(spec).clone()
    assert float((tile_s - base_s).abs().max()) < 5e-6
    for rep in range(8):
        for nb in (300, 1, 2, 7, 255, 256, 257):
            s, l = runner.predict_device(spec[:nb], return_logits=True)
            assert torch.equal(s, base_s[:nb]) and torch.equal(l, base_l[:nb]), f"spectrogram path, batch {nb}, launch {rep}"
            assert torch.equal(runner.infer_audio_device(audio[:nb]), base_a[:nb]), f"audio path, batch {nb}, launch {rep}"
    # the profile shows the pair as one launch on the front operator
    runner.profile(True)
    runner.predict_device(spec)
    rows = [r for r in runner.profile_collect() if r["launches"]]
    runner.profile(False)
    assert not any(r["name"] == covered[0].name and r["kind"] == "f32_dwpw" for r in rows)
    with _hip.options(f32_front2=0):
        runner.profile(True)
        runner.predict_device(spec)
        rows0 = [r for r in runner.profile_collect() if r["launches"]]
        runner.profile(False)
    assert len(rows0) == len(rows) + 1
    runner.close()


# --------------------------------------------------------------------------------------- float32: expand 1x1 + depthwise 3x3 as one kernel
@pytest.mark.parametrize("alpha", [1.5, 1.0])
def test_f32_pwdw_fused_kernel_matches_the_two_kernels(torch_mod, alpha):
    """``f32_pwdw_kernel`` (inverted-residual blocks: the expand convolution runs inside the depthwise kernel, the expanded map stays in
    LDS) against the two-kernel path (option ``f32_pwdw`` = 0) on BASELINE configs[4]'s topology (alpha = 1.5: all three fused shapes,
    stride 1 and 2, the zero-padded k-step of Cin = 24) and on the reference builder's default width (alpha = 1: two channel tiles per
    producer wave, 256 depthwise threads): BIT FOR BIT (same summation orders), odd batch sizes, repeated launches; and against
    the float64 oracle."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner
    from oracle import float_graph

    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2, embeddings_size=256, num_classes=100,
                       audio_frontend="raw", mag_scale="pcen", alpha=alpha, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
    runner = HipRunner(lower_f32(spec), max_batch=70)
    ops = runner.plan.ops
    heads = [i for i, o in enumerate(ops) if o.p[pk.TAIL_TAG] == pk.PWDW_HEAD]
    assert len(heads) == 11 and all(ops[i].kind == pk.F32_DWPW and ops[i + 1].kind == pk.F32_DW and ops[i + 1].p[pk.TAIL_TAG] == pk.PWDW_COVERED
                                    and ops[i + 1].out != ops[i].in0 for i in heads)  # every inverted-residual block is tagged
    rng = np.random.default_rng(5)
    x = rng.standard_normal((70, 48000)).astype(np.float32)
    x /= np.abs(x).max(axis=1, keepdims=True) + 1e-6
    xd = torch.from_numpy(x).cuda()
    with _hip.options(f32_pwdw=0):
        base_s, base_l = (t.clone() for t in runner.predict_device(xd, return_logits=True))
        runner.profile(True)
        runner.predict_device(xd)
        rows0 = [r for r in runner.profile_collect() if r["launches"]]
        runner.profile(False)
    runner.profile(True)
    runner.predict_device(xd)
    rows = [r for r in runner.profile_collect() if r["launches"]]
    runner.profile(False)
    fused = len(rows0) - len(rows)
    assert fused == 7, fused  # stages 1-2 and the first block of stage 3 (input maps 32+ columns wide; the narrow late stages keep two kernels) + the stem
    assert sum(1 for o in ops if o.kind == pk.F32_STEM and o.p[pk.TAIL_TAG] == pk.PWDW_STEM) == 1
    with _hip.options(f32_pwdw=1):  # fused kernels, the squeeze-excite gates pool the maps themselves: same summation orders, bit for bit
        for rep in range(4):
            for nb in (70, 1, 3, 64, 65):
                s, l = runner.predict_device(xd[:nb], return_logits=True)
                assert torch.equal(s, base_s[:nb]) and torch.equal(l, base_l[:nb]), f"batch {nb}, launch {rep}"
    # default (2): the gates behind fused pairs pool per-row-block channel sums handed over by the fused kernel (another summation order)
    first = {}
    for rep in range(3):
        for nb in (70, 1, 3, 64, 65):
            l = runner.predict_device(xd[:nb], return_logits=True)[1]
            assert float((l - base_l[:nb]).abs().max()) <= 2e-6 * float(base_l.abs().max()), f"batch {nb}, launch {rep}"
            if rep:
                assert torch.equal(l, first[nb]), f"batch {nb}: launch {rep} differs from the first (the sums are added in a fixed order)"
            else:
                first[nb] = l.clone()
    ref = float_graph.forward(spec, x[:6, :, None], np.float64)
    got = runner.predict_device(xd[:6]).cpu().numpy()
    for b in range(6):
        assert cosine(got[b], ref[b]) > 1 - 1e-6
    runner.close()


# --------------------------------------------------------------------------------------- INT8 from audio: chunks that go through the float64 STFT as a whole
def test_i8_from_audio_whole_float64_chunks_keep_the_exact_bytes(torch_mod):
    """Chunks whose exact min / max the guarded pass cannot settle within its budget (digital silence in front of an onset, sparse impulses, almost
    noise-free tones) are recomputed as whole float64 spectrograms; their frames then carry a bound of 0.  The mel mixer's kept bytes come from a
    folded multiply-add whose own rounding can cross a quantisation boundary, so exact frames still need that band (a first version listed nothing
    for them: 304 of 245 812 such chunks ended with different scores in tools/exact_soak.py).  6144 chunks of the three families, guarded path
    against the all-float64 path: identical bytes (the debug view) AND identical scores (what the network consumed)."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models.runners import load_model_runner

    B, T, sr = 2048, 72000, 24000
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(77)
    t = torch.arange(T, device=dev, dtype=torch.float64) / sr

    def rnd(*shape, lo=0.0, hi=1.0):
        return lo + (hi - lo) * torch.rand(shape, generator=g, device=dev, dtype=torch.float64)

    runner = load_model_runner(TFLITE_PATH, max_batch=B)
    whole = 0
    for kind in range(3):
        tone = torch.sin(2 * np.pi * rnd(B, 1, lo=60.0, hi=11500.0) * t[None, :] + rnd(B, 1, hi=6.28))
        noise = torch.randn((B, T), generator=g, device=dev, dtype=torch.float64)
        if kind == 0:
            x = torch.where(t[None, :] > rnd(B, 1, hi=2.5), 0.2 * noise + tone, torch.zeros_like(tone))
        elif kind == 1:
            x = (torch.rand((B, T), generator=g, device=dev) < 2e-3).double() * noise + 1e-3 * noise
        else:
            x = 1e-4 * noise + tone
        x = (x * 10.0 ** rnd(B, 1, lo=-4.0, hi=0.0)).to(torch.float32).contiguous()
        with _hip.options(stft_exact=1):
            s1 = runner.infer_audio_device(x).clone()
            q1 = runner.input_bytes(B)
        with _hip.options(stft_minint=0):   # (the whole-chunk route is what is under test: the interval minimum would keep family 2 off it)
            s2 = runner.infer_audio_device(x)
            q2 = runner.input_bytes(B)
            st = runner.guard_stats(B)
        whole += st["whole_minmax"] + st["whole_fix"]
        assert np.array_equal(q1, q2), f"family {kind}: {int((q1 != q2).sum())} input bytes differ"
        assert torch.equal(s1, s2), f"family {kind}: {int((s1 != s2).any(dim=1).sum())} chunks with different scores"
    assert whole > 500, whole  # (the case under test occurred)
    runner.close()


def test_interval_minimum_keeps_noise_free_chunks_on_the_fast_path(torch_mod):
    """Chunks whose minimum has hundreds of candidates (noise-free chirps and tones: every near-zero bin is one; a tone 80 dB above its noise) went
    to the float64 STFT as a whole in rounds 3-4.  With option ``stft_minint`` (default) ``stft_minmax_exact_kernel`` ENCLOSES the minimum
    (lower end from the error bound, upper end = the smallest exact value among every lane's best candidate), the mel mixer widens its band
    of doubt by what that interval can move a quantiser argument and decides the elements it re-evaluates at the four corner values of the
    reference's monotone float32 chain; an element the corners disagree on hands the chunk over (docs/exactness.md).  Checked here: identical
    input bytes and scores against the all-float64 STFT with the option on and off, the interval form taken by (nearly) every such chunk, and
    the whole-chunk route by a few per cent of them at most — while ordinary audio never sees an interval."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models.runners import load_model_runner

    B, T, sr = 2048, 72000, 24000
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(2025)
    t = torch.arange(T, device=dev, dtype=torch.float64) / sr

    def rnd(*shape, lo=0.0, hi=1.0):
        return lo + (hi - lo) * torch.rand(shape, generator=g, device=dev, dtype=torch.float64)

    runner = load_model_runner(TFLITE_PATH, max_batch=B)
    for kind in range(3):
        f0 = rnd(B, 1, lo=60.0, hi=11000.0)
        noise = torch.randn((B, T), generator=g, device=dev, dtype=torch.float64)
        if kind == 0:    # linear chirp, noise-free
            f1 = rnd(B, 1, lo=60.0, hi=11000.0)
            x = torch.sin(2 * np.pi * (f0 * t[None, :] + (f1 - f0) / (2 * 3.0) * t[None, :] ** 2) + rnd(B, 1, hi=6.28))
        elif kind == 1:  # tone 80 dB above white noise
            x = torch.sin(2 * np.pi * f0 * t[None, :] + rnd(B, 1, hi=6.28)) + 1e-4 * noise
        else:            # ordinary: tone in noise
            x = torch.sin(2 * np.pi * f0 * t[None, :]) + 0.3 * noise
        x = (x * 10.0 ** rnd(B, 1, lo=-4.0, hi=0.0)).to(torch.float32).contiguous()
        with _hip.options(stft_exact=1):
            s1 = runner.infer_audio_device(x).clone()
            q1 = runner.input_bytes(B)
        for minint in (1, 0):
            with _hip.options(stft_minint=minint):
                s2 = runner.infer_audio_device(x)
                q2 = runner.input_bytes(B)
                st = runner.guard_stats(B)
            assert np.array_equal(q1, q2), f"family {kind}, stft_minint {minint}: {int((q1 != q2).sum())} input bytes differ"
            assert torch.equal(s1, s2), f"family {kind}, stft_minint {minint}: {int((s1 != s2).any(dim=1).sum())} chunks with different scores"
            whole = st["whole_minmax"] + st["whole_fix"]
            if minint == 0:
                assert st["interval_min"] == 0
                assert (whole > 0.9 * B) if kind < 2 else (whole < 0.01 * B), (kind, st)
            elif kind < 2:
                assert st["interval_min"] > 0.9 * B and whole < 0.08 * B, (kind, st)
            else:
                assert st["interval_min"] < 0.01 * B and whole < 0.01 * B, (kind, st)
    runner.close()


# --------------------------------------------------------------------------------------- float32: plain 1x1 convolutions, three-role persistent kernel
@pytest.mark.parametrize("alpha", [1.5, 1.0])
def test_f32_pw_ws_kernel_matches_the_tile_kernel(torch_mod, alpha):
    """``f32_pw_ws_kernel`` (bn_f32_pw.hip: plain 1x1 convolutions with Cin > 128 — the projections with squeeze-excite gate and residual,
    the expansions of the late stages, the embedding convolution) against ``f32_dwpw_kernel<.., false>`` (option ``f32_pw_ws`` = 0): the
    same k order, bias, residual, activation — scores and logits BIT FOR BIT; batch sizes that leave a partial 64-position tile (one chunk
    = 32 positions in stage 4), an odd number of steps per workgroup, one and several tiles per workgroup; repeated launches."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._lower_f32 import lower_f32
    from birdnet_stm32.models.runners import HipRunner

    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=2, embeddings_size=256, num_classes=100,
                       audio_frontend="raw", mag_scale="pcen", alpha=alpha, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
    B = 600  # stage 3: 1200 tiles over 256 workgroups (4-5 tiles each), stage 4: 300 tiles (1-2 each)
    runner = HipRunner(lower_f32(spec), max_batch=B)
    from birdnet_stm32.models import _pack as pk

    wide = [o for o in runner.plan.ops if o.kind == pk.F32_DWPW and o.p[2] > 128 and o.p[2] % 64 == 0 and o.p[10] % 128 in (0, 64)]
    assert len(wide) >= 6, len(wide)  # (the shapes the kernel takes: Cin > 128, Cout a multiple of 192 or 128)
    rng = np.random.default_rng(9)
    x = rng.standard_normal((B, 48000)).astype(np.float32)
    x /= np.abs(x).max(axis=1, keepdims=True) + 1e-6
    xd = torch.from_numpy(x).cuda()
    with _hip.options(f32_pw_ws=0):
        base_s, base_l = (t.clone() for t in runner.predict_device(xd, return_logits=True))
        runner.profile(True)
        runner.predict_device(xd[:8])
        t0 = sum(r["ms"] for r in runner.profile_collect())
        runner.profile(False)
    assert t0 > 0
    for rep in range(3):
        for nb in (B, 1, 3, 64, 65, 257, 511):
            s, l = runner.predict_device(xd[:nb], return_logits=True)
            assert torch.equal(s, base_s[:nb]) and torch.equal(l, base_l[:nb]), f"batch {nb}, launch {rep}"
    runner.close()


# --------------------------------------------------------------------------------------- INT8: row-streaming depthwise kernel
def test_i8_dw_stream_kernel_matches_the_baseline_kernel(torch_mod):
    """``i8_dw_stream_kernel`` (stand-alone depthwise 3x3 of exported inverted-residual graphs, stride 1 and 2, channel counts that
    are not multiples of 16) and ``i8_stem_stream_kernel`` against ``i8_dw_kernel`` / ``i8_stem_kernel``: every such tensor of a debug plan bit for bit, for strip heights that
    move the strip borders, odd batch sizes and repeated launches (the store pattern is one dword per lane)."""
    torch = torch_mod
    from test_conversion import EXPORT_TOPOLOGIES, _export

    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models.runners import HipRunner

    for name in ("alpha1.5_pcen", "ir_se_softmax"):  # (without squeeze-excite the depthwise stage fuses with the projection)
        _, model, _, x = _export(EXPORT_TOPOLOGIES[name])
        x = np.concatenate([x] * 3)[:19]
        B = x.shape[0]
        runner = HipRunner(lower_i8(model, keep_all=True), max_batch=B)
        dw_ops = [oi for oi, op in enumerate(runner.plan.ops) if op.kind in (pk.I8_DW, pk.I8_STEM)]  # the stem streams its rows the same way
        assert len(dw_ops) >= 5 and {runner.plan.ops[oi].p[3] for oi in dw_ops} == {1, 2} and pk.I8_STEM in {runner.plan.ops[oi].kind for oi in dw_ops}
        with _hip.options(i8_strip=0):
            want_scores = runner.predict(x)
            want = {oi: runner.op_output(oi, B) for oi in dw_ops}
        for th in (0, 1, 3, 5, 64) * 4:
            with _hip.options(i8_strip=1, i8_strip_th=th):
                got_scores = runner.predict(x)
                for oi in dw_ops:
                    assert np.array_equal(runner.op_output(oi, B), want[oi]), f"{name}: rows per strip {th or 'auto'}: {runner.plan.ops[oi].name}"
                assert np.array_equal(got_scores, want_scores)
        for nb in (1, 2, 7):
            assert np.array_equal(runner.predict(x[:nb]), want_scores[:nb])
        runner.close()


# --------------------------------------------------------------------------------------- INT8: strip kernel with the depthwise stage on the matrix cores
def test_i8_strip_mf_kernel_matches_the_strip_kernel_and_the_oracle(torch_mod):
    """``i8_strip_mf_kernel`` (stage1_ds2 of the shipped graph: 32 -> 32 channels, stride 1, residual ADD; depthwise 3x3 as block-diagonal
    matrix products straight from the NHWC map) against ``i8_strip_kernel<32, 1, 32, 1, true>`` (option ``i8_strip_mfdw`` = 0) and the oracle's
    tensor: bit for bit, for strip heights that move the row-block borders (the top / bottom padding rows become a register of zero points, the
    border columns of the outer strips a select), odd batch sizes, repeated launches."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models._lower_i8 import lower_i8
    from birdnet_stm32.models._tflite_reader import load_tflite
    from birdnet_stm32.models.runners import HipRunner
    from oracle.int8_graph import Int8Interpreter

    model = load_tflite(TFLITE_PATH)
    rng = np.random.default_rng(12)
    B = 21
    x = rng.random((B, 257, 256, 1), dtype=np.float32)
    runner = HipRunner(lower_i8(model, keep_all=True), max_batch=B)
    ops = [oi for oi, op in enumerate(runner.plan.ops) if op.kind == pk.I8_DWPW and op.p[35] and op.p[2] == 32 and op.p[14] == 32 and op.p[3] == 1 and op.p[18]]
    assert len(ops) == 1, ops
    oi = ops[0]
    _, env = Int8Interpreter(model).invoke(x, return_all=True)
    with _hip.options(i8_strip_mfdw=0):
        want_scores = runner.predict(x)
        want = runner.op_output(oi, B)
    tensor = [t for t, v in env.items() if np.asarray(v).size == want.size and np.array_equal(np.asarray(v).reshape(-1), want.reshape(-1))]
    assert tensor, "the strip kernel's output is none of the oracle's tensors"
    for th in (0, 1, 3, 5, 7, 32) * 3:
        with _hip.options(i8_strip_mfdw=1, i8_strip_th=th):
            got_scores = runner.predict(x)
            assert np.array_equal(runner.op_output(oi, B), want), f"rows per strip {th or 'auto'}"
            assert np.array_equal(got_scores, want_scores)
    for nb in (1, 2, 9):
        assert np.array_equal(runner.predict(x[:nb]), want_scores[:nb])
    runner.close()


# --------------------------------------------------------------------------------------- the remaining launcher options
def test_every_other_launcher_option_reproduces_the_default_results(torch_mod):
    """The A/B switches that no other test flips (older kernel variants kept for measurements): ``f32_front_staged``, ``front_tpw``,
    ``wave_dwpw`` on the float32 plan (float32 round-off of each other: the FMA order differs), ``i8_mel_generic`` on
    the INT8 plan and ``ingest_blk`` / ``ingest_generic`` on the ingest kernels (bit-identical)."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.audio import ingest
    from birdnet_stm32.models.runners import load_model_runner

    audio = torch.from_numpy(synth_chunks(130, seed=23)).cuda()
    f32 = load_model_runner(KERAS_PATH, max_batch=130)
    spec = f32.stft_device(audio)
    want_a, want_s = f32.infer_audio_device(audio).clone(), f32.predict_device(spec.reshape(130, -1)).clone()
    for opts in (dict(f32_front_staged=0), dict(f32_strip=0, front_tpw=1), dict(f32_strip=0, front_tpw=4), dict(f32_strip=0, wave_dwpw=0),
                 dict(wave_dwpw=0)):
        with _hip.options(**opts):
            for nb in (130, 3):
                assert float((f32.infer_audio_device(audio[:nb]) - want_a[:nb]).abs().max()) < 5e-6, opts
                assert float((f32.predict_device(spec[:nb].reshape(nb, -1)) - want_s[:nb]).abs().max()) < 5e-6, opts
    f32.close()
    i8 = load_model_runner(TFLITE_PATH, max_batch=130)
    want_a, want_s = i8.infer_audio_device(audio).clone(), i8.predict_device(spec.reshape(130, -1)).clone()
    with _hip.options(i8_mel_generic=1):
        for nb in (130, 3):
            assert torch.equal(i8.predict_device(spec[:nb].reshape(nb, -1)), want_s[:nb])
            assert torch.equal(i8.infer_audio_device(audio[:nb]), want_a[:nb])
    i8.close()
    ctx = _hip.Context(0, 64)
    rng = np.random.default_rng(5)
    for sr_in in (48000, 44100, 32000):
        wins = [ingest.window_from_int16(rng.integers(-20000, 20000, (n, ch)).astype(np.int16), sr_in)
                for n, ch in ((int(sr_in * 4.1), 1), (int(sr_in * 0.7), 2), (999, 1))]
        base = ingest.ingest_windows_device(ctx, wins, 24000, 3.0, 0.5)[0].clone()
        for opts in (dict(ingest_generic=1), dict(ingest_blk=1024), dict(ingest_blk=2048, ingest_generic=1)):
            with _hip.options(**opts):
                assert torch.equal(ingest.ingest_windows_device(ctx, wins, 24000, 3.0, 0.5)[0], base), (sr_in, opts)
    ctx.close()


# --------------------------------------------------------------------------------------- batches beyond one launch group
def test_batches_beyond_one_launch_group(torch_mod):
    """More chunks than one launch group takes (32 768: the grid's y limit): ``bn_infer_audio`` and ``bn_forward`` walk the batch in
    slices; every chunk still gets the scores it gets in a small batch, for the float32 and the INT8 plan."""
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner

    B = 32768 + 1500
    base = torch.from_numpy(synth_chunks(48, seed=31)).cuda()
    idx = torch.randint(0, 48, (B,), generator=torch.Generator().manual_seed(3)).cuda()
    audio = base[idx].contiguous()
    for path in (KERAS_PATH, TFLITE_PATH):
        small = load_model_runner(path, max_batch=48)
        ref = small.infer_audio_device(base).clone()
        ref_spec_scores = small.predict_device(small.stft_device(base).reshape(48, -1)).clone()
        small.close()
        big = load_model_runner(path, max_batch=B)
        got = big.infer_audio_device(audio)
        assert torch.equal(got, ref[idx]), os.path.basename(path)
        if path == TFLITE_PATH:  # the exactness pass's give-up route in both launch groups (their lists and counters are separate)
            from birdnet_stm32 import _hip

            with _hip.options(stft_flagcap=2):
                assert torch.equal(big.infer_audio_device(audio), ref[idx])
        spec = big.stft_device(audio).reshape(B, -1)
        assert torch.equal(big.predict_device(spec), ref_spec_scores[idx]), os.path.basename(path)
        del spec, got
        big.close()
        torch.cuda.empty_cache()


# --------------------------------------------------------------------------------------- other chunk lengths from audio
def test_from_audio_at_the_training_sample_rate(torch_mod):
    """22.05 kHz x 3 s chunks (66 150 samples, hop 258: the rate the shipped network was trained at) through ``bn_infer_audio``:
    the INT8 plan gives exactly what it gives for the spectrogram ``bn_stft_mag_exact`` writes AND for the oracle's spectrogram
    (bit-exact from audio: the float64 pass behind the float32 STFT), the float32
    audio path (STFT + mel mixer fused, normalisation applied behind the mixer) stays within 1e-5 of its spectrogram path and
    both track the CPU oracle's spectrogram path."""
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner
    from oracle import stft

    chunks = synth_chunks(24, sr=22050, seed=8)
    audio = torch.from_numpy(chunks).cuda()
    S = np.stack([stft.hybrid_spectrogram(a) for a in chunks])[..., None].astype(np.float32)
    i8 = load_model_runner(TFLITE_PATH, max_batch=24)
    spec = i8.stft_device(audio, exact=True)
    assert tuple(spec.shape) == (24, 257, 256)
    assert torch.equal(i8.infer_audio_device(audio), i8.predict_device(spec.reshape(24, -1)))
    assert np.array_equal(i8.infer_audio_device(audio).cpu().numpy(), i8.predict(S))
    i8.close()
    f32 = load_model_runner(KERAS_PATH, max_batch=24)
    a, b = f32.infer_audio_device(audio), f32.predict_device(f32.stft_device(audio).reshape(24, -1))
    assert float((a - b).abs().max()) < 1e-5
    assert np.abs(a.cpu().numpy() - f32.predict(S)).max() < 1e-4
    f32.close()


# --------------------------------------------------------------------------------------- launches on the caller's stream
def test_everything_runs_on_the_callers_stream(torch_mod):
    """The C ABI takes the stream from the caller: on a non-default torch stream, with the input produced on that stream right
    before the call (no synchronisation in between) and the default stream kept busy, the results equal those of the default
    stream — a kernel launched on the wrong stream would read the input before it exists."""
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner

    base = torch.from_numpy(synth_chunks(200, seed=41)).cuda()
    for path in (KERAS_PATH, TFLITE_PATH):
        runner = load_model_runner(path, max_batch=200)
        want = runner.infer_audio_device(base).clone()
        want_spec = runner.predict_device(runner.stft_device(base).reshape(200, -1)).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        busy = torch.randn(4096, 4096, device="cuda")
        for rep in range(3):
            for _ in range(4):
                busy = busy @ busy * 1e-3  # keeps the default stream occupied
            with torch.cuda.stream(side):
                x = torch.zeros_like(base)
                x.copy_(base.flip(0)).mul_(1.0)  # the input comes into being on the side stream
                x = x.flip(0).contiguous()
                got = runner.infer_audio_device(x)
                got_spec = runner.predict_device(runner.stft_device(x).reshape(200, -1))
            side.synchronize()
            assert torch.equal(got, want) and torch.equal(got_spec, want_spec), f"{os.path.basename(path)}, repetition {rep}"
        torch.cuda.synchronize()
        runner.close()


# --------------------------------------------------------------------------------------- load / free cycles
def test_contexts_and_models_give_their_memory_back(torch_mod):
    """Thirty load -> infer -> free cycles of both model files (own context each time): the device's free memory returns to where it
    was (bn_ctx_create / bn_model_load allocate with hipMalloc outside torch's allocator; a leak would show here) and the scores of
    the last cycle equal those of the first."""
    torch = torch_mod
    from birdnet_stm32.models.runners import load_model_runner

    audio = torch.from_numpy(synth_chunks(16, seed=2)).cuda()
    first = {}

    def cycles(n):
        for cycle in range(n):
            for path in (KERAS_PATH, TFLITE_PATH):
                runner = load_model_runner(path, max_batch=64 + 8 * (cycle % 3))
                s = runner.infer_audio_device(audio).clone()
                first.setdefault(path, s)
                assert torch.equal(s, first[path])
                del s
                runner.close()
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    free0 = cycles(5)   # code objects, torch's cached blocks and the HIP runtime's pools are in place after the first cycles
    free1 = cycles(30)
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) >> 20} MiB of device memory did not come back over 30 cycles"


# --------------------------------------------------------------------------------------- the C ABI refuses bad calls
def test_c_abi_refuses_bad_calls_with_an_error_code(torch_mod):
    """Straight through ctypes: batches beyond max_batch, null pointers, audio geometries that give too few frames, a damaged blob —
    every call returns its error code with a message in bn_last_error() and leaves the context usable."""
    torch = torch_mod
    from birdnet_stm32 import _hip
    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models.runners import load_model_runner, lower_model_file

    runner = load_model_runner(TFLITE_PATH, max_batch=8)
    lib, mh = runner.lib, runner.model.handle
    audio = torch.from_numpy(synth_chunks(9, seed=1)).cuda()
    scores = torch.empty((9, runner.num_classes), device="cuda")
    spec = torch.rand((9, 257 * 256), device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def err():
        return lib.bn_last_error().decode()

    assert lib.bn_infer_audio(mh, audio.data_ptr(), 9, 72000, 281, scores.data_ptr(), None, stream) != 0 and "max_batch" in err()
    assert lib.bn_forward(mh, spec.data_ptr(), None, 9, scores.data_ptr(), None, stream) != 0 and "max_batch" in err()
    assert lib.bn_infer_audio(mh, None, 4, 72000, 281, scores.data_ptr(), None, stream) != 0 and "null" in err()
    assert lib.bn_forward(mh, spec.data_ptr(), None, 4, None, None, stream) != 0 and "null" in err()
    assert lib.bn_infer_audio(mh, audio.data_ptr(), 4, 72000, 20000, scores.data_ptr(), None, stream) != 0  # 4 frames, the model wants 256
    assert lib.bn_infer_audio(mh, audio.data_ptr(), 4, 0, 281, scores.data_ptr(), None, stream) != 0
    assert lib.bn_infer_audio(None, audio.data_ptr(), 4, 72000, 281, scores.data_ptr(), None, stream) != 0 and "null model" in err()
    blob = bytearray(pk.pack_plan(lower_model_file(TFLITE_PATH)))
    blob[40:44] = (0x7FFFFFFF).to_bytes(4, "little")  # a header count far beyond the blob
    out = ctypes.c_void_p()
    assert lib.bn_model_load(runner.ctx.handle, bytes(blob), len(blob), ctypes.byref(out)) != 0 and not out.value
    assert lib.bn_model_load(runner.ctx.handle, bytes(blob[:100]), 100, ctypes.byref(out)) != 0 and not out.value
    # the context and the model still work
    want = runner.infer_audio_device(audio[:8])
    assert torch.isfinite(want).all() and lib.bn_infer_audio(mh, audio.data_ptr(), 8, 72000, 281, scores.data_ptr(), None, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(scores[:8], want)
    runner.close()


def test_hostile_tail_descriptor_falls_back_to_the_block_kernels(torch_mod):
    """A fused-tail descriptor with a NEGATIVE constant-block offset or a clamp outside int8 (a stale or hostile blob; the size check bounds
    offsets from above only) must not reach i8_tail_kernel: tail_plan refuses it at load and the per-block operators run instead — same
    scores, bit for bit."""
    import copy

    from birdnet_stm32.models import _pack as pk
    from birdnet_stm32.models.runners import HipRunner, lower_model_file

    plan = lower_model_file(TFLITE_PATH)
    ti = next(i for i, o in enumerate(plan.ops) if o.kind == pk.I8_TAIL)
    S = np.random.default_rng(3).random((8, 257, 256, 1), dtype=np.float32) ** 3
    good = HipRunner(plan, max_batch=8)
    want = good.predict(S)
    assert good.profile_collect() is not None
    good.close()
    for word, value in ((21, -4), (13, -300), (24 + 19, 400), (24 * 6 + 7, -8)):  # g_dwc / pw_lo of block 0, add_hi of block 1 (the first with an ADD), g_fcw of the head
        bad = copy.deepcopy(plan)
        desc = bad.tensors[bad.ops[ti].t[1]].copy()
        desc.reshape(-1)[word] = value
        bad.tensors[bad.ops[ti].t[1]] = desc
        r = HipRunner(bad, max_batch=8)
        r.profile(True)
        got = r.predict(S)
        kinds = {q["kind"] for q in r.profile_collect() if q["launches"]}
        r.close()
        assert "i8_tail" not in kinds, f"descriptor word {word} = {value} was accepted by the fused kernel"
        assert np.array_equal(got, want)
