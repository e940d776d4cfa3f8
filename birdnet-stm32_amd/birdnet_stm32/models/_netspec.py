"""Framework-free description of a DS-CNN graph: the hand-off between model files and kernels.

The reference keeps its topology in a ``tf.keras.Model`` (reference:
birdnet_stm32/models/dscnn.py:87-262).  TensorFlow does not exist on the MI355X box, so
the graph is held here as a plain list of :class:`Layer` records in execution order, one
per Keras layer, each naming its producer(s).  Three consumers read it:

* ``_lower.py`` fuses it into the device plan executed by ``libbirdnet_hip.so``;
* ``oracle/float_graph.py`` (tests only) evaluates it layer by layer with Keras semantics;
* the reference's own model tests read ``input_shape``, ``output_shape``, ``count_params()``
  and ``layers[i].filters`` (reference: tests/test_dscnn.py:73,105,133-135,161,187), which
  :class:`NetSpec` exposes with the same meaning.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# Layer kinds (Keras class it stands for)
INPUT = "input"
FRONTEND = "frontend"  # AudioFrontendLayer
CONV = "conv2d"  # Conv2D, no bias
DWCONV = "dwconv2d"  # DepthwiseConv2D, no bias
BN = "batchnorm"
RELU = "relu"  # ReLU(max_value)
ADD = "add"
MUL = "multiply"
GAP = "gap"  # GlobalAveragePooling2D
DENSE = "dense"
ATTNPOOL = "attnpool"
IDENTITY = "identity"  # Dropout / SpatialDropout2D at inference


@dataclass
class Layer:
    """One Keras layer: kind, producers, static attributes and weights."""

    name: str
    kind: str
    inputs: list[str] = field(default_factory=list)
    attrs: dict = field(default_factory=dict)
    weights: dict[str, np.ndarray] = field(default_factory=dict)
    out_shape: tuple = ()  # without batch

    # attribute the reference's tests read on conv layers
    @property
    def filters(self):
        return self.attrs.get("filters")

    def n_params(self) -> int:
        return int(sum(int(w.size) for w in self.weights.values()))


@dataclass
class NetSpec:
    """Ordered layer list plus the model-level facts the runners and tests need."""

    layers: list[Layer]
    name: str = "dscnn_audio"
    meta: dict = field(default_factory=dict)

    def layer(self, name: str) -> Layer:
        for ly in self.layers:
            if ly.name == name:
                return ly
        raise KeyError(name)

    @property
    def input_shape(self) -> tuple:
        return (None, *self.layers[0].out_shape)

    @property
    def output_shape(self) -> tuple:
        return (None, *self.layers[-1].out_shape)

    @property
    def frontend(self) -> Layer:
        for ly in self.layers:
            if ly.kind == FRONTEND:
                return ly
        raise KeyError("graph has no AudioFrontendLayer")

    @property
    def num_classes(self) -> int:
        return int(self.layers[-1].out_shape[-1])

    def count_params(self) -> int:
        """Stored parameter count, BN moving statistics included (Keras ``count_params``)."""
        return int(sum(ly.n_params() for ly in self.layers))


def same_pad(in_size: int, k: int, stride: int) -> tuple[int, int, int]:
    """TensorFlow ``SAME`` geometry: ``(out_size, pad_before, pad_after)``; the extra cell goes after."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2, total - total // 2


def make_divisible(v, divisor: int = 8) -> int:
    """Channel rounding rule of the reference (reference: birdnet_stm32/models/blocks.py:13-24)."""
    return max(divisor, int(v + divisor / 2) // divisor * divisor)
