"""Float vs INT8 agreement metrics (reference: birdnet_stm32/conversion/validate.py:7-118)."""

from __future__ import annotations

import numpy as np


def cosine_similarity(a: np.ndarray, b: np.ndarray, eps: float = 1e-8) -> float:
    """Cosine of two flattened prediction vectors; two (near-)zero vectors agree (1.0), one zero vector disagrees (0.0)."""
    an, bn = np.linalg.norm(a), np.linalg.norm(b)
    if an < eps and bn < eps:
        return 1.0
    if an < eps or bn < eps:
        return 0.0
    return float(np.dot(a, b) / (an * bn))


def pearson_correlation(a: np.ndarray, b: np.ndarray, eps: float = 1e-12) -> float:
    a = a - np.mean(a)
    b = b - np.mean(b)
    denom = np.linalg.norm(a) * np.linalg.norm(b)
    return 1.0 if denom < eps else float(np.dot(a, b) / denom)


def validate_models(float_runner, int8_runner, rep_data_gen) -> dict[str, float]:
    """Compare two runners sample by sample over ``rep_data_gen()`` (same keys as the reference's summary dict)."""
    cos, mse, mae, pcc = [], [], [], []
    for sample in rep_data_gen():
        x = np.asarray(sample[0], np.float32)
        a = np.asarray(float_runner.predict(x), np.float64).reshape(-1)
        b = np.asarray(int8_runner.predict(x), np.float64).reshape(-1)
        cos.append(cosine_similarity(a, b))
        mse.append(float(np.mean((a - b) ** 2)))
        mae.append(float(np.mean(np.abs(a - b))))
        pcc.append(pearson_correlation(a, b))
    return {"cosine_mean": float(np.mean(cos)) if cos else 0.0, "mse_mean": float(np.mean(mse)) if mse else float("inf"),
            "mae_mean": float(np.mean(mae)) if mae else float("inf"), "pearson_mean": float(np.mean(pcc)) if pcc else 0.0}
