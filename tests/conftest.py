"""Shared pytest configuration.

* ``-m gpu`` tests are the parity tests proper: they drive ``libbirdnet_hip.so`` through its C ABI on
  a real MI355X and compare with the CPU oracle (``oracle/``) on the same seeded inputs.
* ``-m "not gpu"`` tests cover the oracle against golden vectors / known answers, the host-side
  logic (readers, lowering, chunking, pooling, registry, config, CLI plumbing) and the C-ABI surface.

Nothing here reads ``/root/reference``: that tree does not exist on the GPU box.
"""

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "birdnet-stm32_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

CKPT_DIR = os.path.join(PKG, "checkpoints")
KERAS_PATH = os.path.join(CKPT_DIR, "birdnet_stm32n6_100.keras")
TFLITE_PATH = os.path.join(CKPT_DIR, "birdnet_stm32n6_100.tflite")
CONFIG_PATH = os.path.join(CKPT_DIR, "birdnet_stm32n6_100_model_config.json")
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def synth_chunks(n: int, sr: int = 24000, seconds: float = 3.0, seed: int = 42) -> np.ndarray:
    """Synthetic chunks of BASELINE.md §4: peaknorm(0.3 N(0,1) + sin(2 pi f_b t)), f_b = 500 + 37 (b mod 200)."""
    rng = np.random.default_rng(seed)
    T = int(sr * seconds)
    t = np.arange(T, dtype=np.float64) / sr
    out = np.empty((n, T), np.float32)
    for b in range(n):
        x = 0.3 * rng.standard_normal(T) + np.sin(2 * np.pi * (500 + 37 * (b % 200)) * t)
        out[b] = (x / np.max(np.abs(x))).astype(np.float32)
    return out


def fixture_signals(sr: int, seconds: float = 3.0) -> dict:
    """The reference's test signals (formulas: reference tests/conftest.py:49-81, tests/fixtures/generate_fixtures.py:17-32)."""
    T = int(sr * seconds)
    t = np.linspace(0, seconds, T, endpoint=False)
    sine = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    noise = np.random.default_rng(42).standard_normal(T).astype(np.float32) * 0.1
    f0, f1 = 500.0, 4000.0
    chirp = (0.5 * np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / seconds * t**2))).astype(np.float32)
    silence = np.zeros(T, np.float32)
    return {"sine": sine, "noise": noise, "chirp": chirp, "silence": silence}


def cosine(a, b) -> float:
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if na < 1e-12 and nb < 1e-12:
        return 1.0
    if na < 1e-12 or nb < 1e-12:
        return 0.0
    return float(a @ b / (na * nb))


I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1


# gemmlowp fixed-point definitions, literally, on Python integers (the yardstick for oracle/int8_graph.py and csrc/bn_requant.h)
def srdhm_def(a: int, b: int) -> int:
    """SaturatingRoundingDoublingHighMul (gemmlowp fixedpoint.h): saturate only for INT32_MIN x INT32_MIN, nudge by +-2^30, divide by
    2^31 truncating toward zero."""
    if a == b == I32_MIN:
        return I32_MAX
    ab = a * b
    nudge = (1 << 30) if ab >= 0 else 1 - (1 << 30)
    v = ab + nudge
    return v // (1 << 31) if v >= 0 else -((-v) // (1 << 31))


def rdivpot_def(x: int, e: int) -> int:
    """RoundingDivideByPOT: arithmetic shift, ties away from zero."""
    mask = (1 << e) - 1
    rem = x & mask
    thr = (mask >> 1) + (1 if x < 0 else 0)
    return (x >> e) + (1 if rem > thr else 0)


def wrap32(v: int) -> int:
    return ((v + (1 << 31)) % (1 << 32)) - (1 << 31)


def mbqm_def(x: int, m: int, shift: int) -> int:
    left, right = max(shift, 0), max(-shift, 0)
    return rdivpot_def(srdhm_def(wrap32(x * (1 << left)), m), right)



@pytest.fixture(scope="session")
def has_gpu():
    import torch

    return torch.cuda.is_available()
