"""Chunk-score -> file-score pooling (reference: birdnet_stm32/evaluation/pooling.py:6-47).

``pool_scores(chunk_scores [N, C], method, beta) -> [C]`` with the reference's method names:
'avg' | 'mean' | 'average', 'max', 'lme' | 'log_mean_exp' | 'log_mean_exponential'.
``lme`` is the numerically stabilised log-mean-exp ``(m + log(mean(exp(beta s - m)) + 1e-12)) / beta``.
An empty chunk list pools to zeros; a non-2-D input or an unknown method raises ``ValueError``.
"""

from __future__ import annotations

import numpy as np

_MEAN = {"avg", "mean", "average"}
_LME = {"lme", "log_mean_exp", "log_mean_exponential"}


def lme_pooling(scores: np.ndarray, beta: float = 10.0) -> np.ndarray:
    """Log-mean-exp over the chunk axis; beta -> 0 tends to the mean, beta -> inf to the max."""
    if scores.size == 0:
        return scores
    z = beta * scores
    peak = z.max(axis=0, keepdims=True)
    pooled = peak + np.log(np.exp(z - peak).mean(axis=0, keepdims=True) + 1e-12)
    return (pooled / beta).ravel()


def pool_scores(chunk_scores: np.ndarray, method: str = "average", beta: float = 10.0) -> np.ndarray:
    key = method.lower()
    if chunk_scores.ndim != 2:
        raise ValueError("chunk_scores must be [N_chunks, C]")
    n, c = chunk_scores.shape
    if n == 0:
        return np.zeros((c,), dtype=np.float32)
    if key in _MEAN:
        return chunk_scores.mean(axis=0)
    if key == "max":
        return chunk_scores.max(axis=0)
    if key in _LME:
        return lme_pooling(chunk_scores, beta=beta)
    raise ValueError(f"Unsupported pooling method: {method}")
