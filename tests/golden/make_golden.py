#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/.  Run from the repo root in the BUILD container:

    python tests/golden/make_golden.py

Two kinds of fixture are written:

1. ``reference_*.json|npz`` — outputs of the REFERENCE's own code for the parts of the hot path that import
   cleanly here (no TensorFlow / librosa / soundfile needed): ``birdnet_stm32.evaluation.pooling`` and
   ``birdnet_stm32.training.config``.  They are produced in a subprocess whose ``sys.path`` holds only
   ``/root/reference`` (our package has the same top-level name).  These pin the oracle/host logic to the
   real reference.
2. ``oracle_vectors.npz`` — stage-by-stage outputs of the CPU oracle (``oracle/``) on the reference's test
   signals (sine 1 kHz / white noise rng(42) / chirp 500->4000 Hz, at 22 050 Hz and 24 000 Hz; formulas from
   the reference's tests/conftest.py:49-81 and tests/fixtures/generate_fixtures.py:17-32).  The reference's
   tests hold no expected values for STFT / Keras / TFLite numerics, so these are oracle-defined: they guard
   the oracle against regressions and let the GPU box check the HIP path without recomputing the oracle.

Only data is written (inputs are regenerated from seeds by tests/conftest.py); no reference source is copied.
"""

import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

_REF_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from birdnet_stm32.evaluation.pooling import pool_scores, lme_pooling
from birdnet_stm32.training.config import ModelConfig
out = {}
rng = np.random.default_rng(7)
cases = {"rand_5x4": rng.uniform(0, 1, (5, 4)), "one_chunk": rng.uniform(0, 1, (1, 6)), "saturated": np.array([[0.0, 1.0], [1.0, 0.0], [1.0, 1.0]]),
         "tiny": rng.uniform(0, 1e-6, (7, 3)), "twenty": rng.uniform(0, 1, (20, 100))}
pool = {}
for name, x in cases.items():
    x = x.astype(np.float32)
    row = {"x": x.tolist()}
    for method in ("avg", "mean", "average", "max", "lme", "log_mean_exp"):
        row[method] = np.asarray(pool_scores(x, method=method), dtype=np.float64).tolist()
    for beta in (0.5, 10.0, 50.0):
        row["lme_beta_%%g" %% beta] = np.asarray(lme_pooling(x, beta=beta), dtype=np.float64).tolist()
    pool[name] = row
pool["empty"] = np.asarray(pool_scores(np.zeros((0, 3), np.float32), "avg")).tolist()
out["pooling"] = pool
cfg = {}
cfg["defaults"] = ModelConfig().to_dict()
cfg["shipped"] = ModelConfig.load(%r).to_dict()
cfg["legacy_dict"] = ModelConfig.from_dict({"sample_rate": 22050, "num_mels": 64, "spec_width": 256, "fft_length": 512,
    "chunk_duration": 3, "hop_length": 258, "audio_frontend": "hybrid", "mag_scale": "pwl", "embeddings_size": 256, "alpha": 1.0,
    "depth_multiplier": 1, "num_classes": 2, "class_names": ["a", "b"], "some_unknown_key": 1}).to_dict()
errors = {}
for label, kw in {"neg_sr": {"sample_rate": -1}, "bad_frontend": {"audio_frontend": "nope"}, "bad_mag": {"mag_scale": "log"},
                  "dm0": {"depth_multiplier": 0}, "drop1": {"dropout_rate": 1.0}, "names": {"num_classes": 3, "class_names": ["a"]}}.items():
    try:
        ModelConfig(**kw); errors[label] = None
    except ValueError as e:
        errors[label] = str(e)
cfg["errors"] = errors
out["config"] = cfg
json.dump(out, sys.stdout)
"""


_REPORT_CASES = {
    "full": dict(metrics={"roc-auc": 0.9123456789, "cmAP": 0.5, "mAP": 0.123456749, "f1": 1.0, "precision": 0.3333333333, "recall": 2 / 3,
                          "total_chunks": 28, "ap_per_class": [0.1, 0.2], "latency_mean_ms": 1.23456789, "latency_p99_ms": 3, "note": "x", "flag": True,
                          "nested": {"a": 0.1234567891}},
                 classes=["alpha", "beta", "gamma"], model_path="checkpoints/m.tflite",
                 species_data=[{"class": "alpha", "ap": 0.25, "ci_lower": 0.1, "ci_upper": 0.4, "n_positive": 3, "n_total": 9}],
                 config={"sample_rate": 24000, "class_names": ["alpha", "beta", "gamma"], "alpha": 1.0}),
    "bare": dict(metrics={"roc-auc": 0.0, "total_chunks": 0}, classes=[], model_path="m.keras", species_data=None, config=None),
    "no_chunks_key": dict(metrics={"f1": 0.7071067811865476}, classes=["a"], model_path="x", species_data=[], config={}),
}

_REPORT_SCRIPT = r"""
import io, json, os, sys, tempfile, contextlib
sys.path.insert(0, %r)
from birdnet_stm32.evaluation.reporting import save_benchmark_json
cases = json.loads(%r)
out = {}
for name, c in cases.items():
    d = tempfile.mkdtemp()
    path = os.path.join(d, "sub", "report.json")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        save_benchmark_json(c["metrics"], c["classes"], c["model_path"], path, species_data=c["species_data"], config=c["config"])
    out[name] = {"file": open(path).read(), "stdout": buf.getvalue().replace(path, "<out>")}
json.dump(out, sys.stdout)
"""


def reference_reporting() -> None:
    """``save_benchmark_json`` of the reference (evaluation/reporting.py:192-236) on three metric dicts: the text of the file it writes."""
    res = subprocess.run([sys.executable, "-c", _REPORT_SCRIPT % (REF, json.dumps(_REPORT_CASES))], capture_output=True, text=True, check=True, cwd="/tmp")
    with open(os.path.join(HERE, "reference_reporting.json"), "w") as fh:
        json.dump({"cases": _REPORT_CASES, "reference": json.loads(res.stdout)}, fh, indent=1)
    print("wrote reference_reporting.json")


def reference_fixtures() -> None:
    shipped = os.path.join(REF, "checkpoints", "birdnet_stm32n6_100_model_config.json")
    code = _REF_SCRIPT % (REF, shipped)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, cwd="/tmp")
    data = json.loads(res.stdout)
    with open(os.path.join(HERE, "reference_pooling_config.json"), "w") as fh:
        json.dump(data, fh, indent=1)
    print("wrote reference_pooling_config.json")


def oracle_fixtures() -> None:
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "birdnet-stm32_amd"))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import KERAS_PATH, TFLITE_PATH, fixture_signals
    from oracle import float_graph, stft
    from oracle.int8_graph import Int8Interpreter

    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._tflite_reader import load_tflite

    spec = load_keras_archive(KERAS_PATH)
    tfl = load_tflite(TFLITE_PATH)
    interp = Int8Interpreter(tfl)
    out = {}
    for sr in (22050, 24000):
        sig = fixture_signals(sr)
        for name in ("sine", "noise", "chirp"):
            key = f"{name}_{sr}"
            S = stft.hybrid_spectrogram(sig[name])
            x = S[None, :, :, None]
            probs, logits, acts = float_graph.forward(spec, x, np.float64, return_all=True, return_logits=True)
            q_probs, env = interp.invoke(x, return_all=True)
            out[key + "/spec_rows"] = S[::16, :].astype(np.float32)  # every 16th frequency row
            out[key + "/spec_sum"] = np.float64(S.astype(np.float64).sum())
            out[key + "/frontend"] = acts["audio_frontend"][0, :, :, 0].astype(np.float32)[:, ::8]
            out[key + "/gap"] = acts["gap"][0].astype(np.float32)
            out[key + "/logits"] = logits[0].astype(np.float32)
            out[key + "/probs"] = probs[0].astype(np.float32)
            out[key + "/i8_fc"] = env[128][0].astype(np.int8)
            out[key + "/i8_probs"] = q_probs[0].astype(np.float32)
            out[key + "/i8_sums"] = np.array([int(env[t].astype(np.int64).sum()) for t in (83, 96, 97, 102, 110, 121, 126, 127)], np.int64)
            # every int8 tensor of the graph: (tensor index, sum, position-weighted checksum) — a GPU test holds each plan operator's output to these
            out[key + "/i8_tensors"] = tensor_checksums(env)
    out.update(config4_fixtures())
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote oracle_vectors.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def tensor_checksums(env) -> np.ndarray:
    """``[n, 3]`` int64 rows (tensor index, sum of the bytes, sum of (i mod 65521 + 1) * byte) over the int8 tensors of an interpreter run."""
    rows = []
    for t in sorted(env):
        v = np.asarray(env[t])
        if v.dtype != np.int8:
            continue
        flat = v.astype(np.int64).reshape(-1)
        rows.append((int(t), int(flat.sum()), int(((np.arange(flat.size, dtype=np.int64) % 65521 + 1) * flat).sum())))
    return np.array(rows, np.int64)


C4_SECONDS, C4_CHUNKS = 2, 3


def config4_model():
    """BASELINE configs[4]'s topology as bench.py builds it (raw frontend + PCEN + alpha = 1.5 DS-CNN with squeeze-excite / inverted residuals,
    seeded weights, the reference's 2 s deployment geometry) and its INT8 export (own PTQ, seeded calibration)."""
    from birdnet_stm32.conversion.export import convert_netspec_to_int8
    from birdnet_stm32.models import build_model
    from birdnet_stm32.models._tflite_writer import write_tflite

    T4 = 24000 * C4_SECONDS
    spec = build_model("dscnn", num_mels=64, spec_width=256, sample_rate=24000, chunk_duration=C4_SECONDS, embeddings_size=256, num_classes=100,
                       audio_frontend="raw", mag_scale="pcen", alpha=1.5, use_se=True, use_inverted_residual=True, randomize_bn=True, seed=42)
    rng = np.random.default_rng(0)
    cal = [rng.standard_normal((1, T4, 1)).astype(np.float32) for _ in range(8)]
    cal = [c / (np.abs(c).max() + 1e-6) for c in cal]
    raw = write_tflite(convert_netspec_to_int8(spec, lambda: ([c] for c in cal)))
    return spec, raw


def config4_inputs() -> np.ndarray:
    """``[C4_CHUNKS, T, 1]`` model inputs of the raw frontend: peak-normalised noise + tone, seeded."""
    T4 = 24000 * C4_SECONDS
    rng = np.random.default_rng(11)
    t = np.arange(T4) / 24000.0
    x = np.stack([0.3 * rng.standard_normal(T4) + np.sin(2 * np.pi * (700 + 900 * b) * t) for b in range(C4_CHUNKS)])
    return (x / (np.abs(x).max(axis=1, keepdims=True) + 1e-6)).astype(np.float32)[..., None]


def config4_fixtures() -> dict:
    import hashlib

    from oracle import float_graph
    from oracle.int8_graph import Int8Interpreter

    from birdnet_stm32.models._tflite_reader import parse_tflite

    spec, raw = config4_model()
    x = config4_inputs()
    probs, logits = float_graph.forward(spec, x, np.float64, return_logits=True)
    q_probs, env = Int8Interpreter(parse_tflite(raw)).invoke(x, return_all=True)
    return {"c4/tflite_sha256": np.frombuffer(hashlib.sha256(raw).digest(), np.uint8).copy(), "c4/f32_logits": logits.astype(np.float32),
            "c4/f32_probs": probs.astype(np.float32), "c4/i8_probs": q_probs.astype(np.float32), "c4/i8_tensors": tensor_checksums(env)}


if __name__ == "__main__":
    if os.path.isdir(REF):
        reference_fixtures()
        reference_reporting()
    else:
        print("no /root/reference here: keeping the committed reference_* fixtures")
    oracle_fixtures()
