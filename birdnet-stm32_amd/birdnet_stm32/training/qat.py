"""Weight fake-quantisation used by the reference's QAT fine-tuning (reference: birdnet_stm32/training/qat.py:30-61).

Only ``fake_quantize_weights`` is mirrored: it is pure array arithmetic (asymmetric min/max grid of ``2**num_bits - 1``
steps per output channel, quantise then dequantise) and serves the conversion checks as the "what survives 8 bits"
reference point.  The Keras callback around it belongs to training, which is out of scope here.
"""

from __future__ import annotations

import numpy as np


def fake_quantize_weights(w: np.ndarray, num_bits: int = 8, per_channel: bool = True, channel_axis: int = -1) -> np.ndarray:
    """Quantise-dequantise ``w`` on a ``[min, max]`` grid with ``2**num_bits - 1`` steps (per channel along ``channel_axis``)."""
    qmax = (1 << num_bits) - 1
    w = np.asarray(w)
    if per_channel and w.ndim > 1:
        axes = tuple(i for i in range(w.ndim) if i != channel_axis % w.ndim)
        lo = w.min(axis=axes, keepdims=True)
        hi = w.max(axis=axes, keepdims=True)
    else:
        lo, hi = w.min(), w.max()
    scale = np.maximum((hi - lo) / qmax, 1e-10)
    return (np.round((w - lo) / scale) * scale + lo).astype(np.float32)
