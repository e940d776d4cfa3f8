"""DS-CNN topology builder producing a :class:`NetSpec` (no TensorFlow).

Same signature, defaults, layer naming and channel arithmetic as the reference's
``build_dscnn_model`` (reference: birdnet_stm32/models/dscnn.py:87-262; blocks:
birdnet_stm32/models/blocks.py:27-133), so models described by a reference ``ModelConfig`` map
one-to-one:

* stem ``Conv2D(md(16 a), 3x3, stride (1,2)) -> BN -> ReLU6``;
* four stages of widths ``md(int(bf*a))`` for bf in 32/64/128/256, repeats ``ceil(br*dm)`` for br in
  2/3/4/2, the first block of a stage with stride (2,2);
* DS block ``DW3x3 -> BN -> ReLU6 -> PW -> BN [-> +x] -> ReLU6`` (+ optional SE after the block), or
  inverted residual ``PW expand -> BN -> ReLU6 -> DW3x3 -> BN -> ReLU6 [-> SE] -> PW project -> BN [-> +x]``;
* optional 1x1 embedding conv, GAP or attention pooling, Dense(num_classes, class_activation).

Weights are freshly initialised (Keras defaults: Glorot-uniform kernels, identity BatchNorm, PWL/PCEN
constants of the reference) from ``seed`` — there are no trained weights for anything but the shipped
checkpoint, which is loaded from its `.keras` file instead (``_keras_loader``).  The returned object
answers ``input_shape``, ``output_shape``, ``count_params()`` and ``layers[i].filters`` like the Keras
model the reference's tests inspect (reference: tests/test_dscnn.py:59-187).
"""

from __future__ import annotations

import math

import numpy as np

from birdnet_stm32.models import _netspec as ns
from birdnet_stm32.models.frontend import normalize_frontend_name

_md = ns.make_divisible


class _Init:
    def __init__(self, seed: int, randomize_bn: bool):
        self.rng = np.random.default_rng(seed)
        self.randomize_bn = randomize_bn

    def glorot(self, shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return self.rng.uniform(-lim, lim, size=shape).astype(np.float32)

    def bn(self, c):
        if self.randomize_bn:
            return {
                "gamma": self.rng.uniform(0.5, 1.5, c).astype(np.float32),
                "beta": (0.1 * self.rng.standard_normal(c)).astype(np.float32),
                "mean": (0.1 * self.rng.standard_normal(c)).astype(np.float32),
                "var": self.rng.uniform(0.5, 1.5, c).astype(np.float32),
            }
        return {"gamma": np.ones(c, np.float32), "beta": np.zeros(c, np.float32), "mean": np.zeros(c, np.float32), "var": np.ones(c, np.float32)}


class _Graph:
    def __init__(self, init: _Init):
        self.layers: list[ns.Layer] = []
        self.init = init

    def add(self, layer: ns.Layer) -> str:
        self.layers.append(layer)
        return layer.name

    def shape(self, name: str) -> tuple:
        for ly in self.layers:
            if ly.name == name:
                return ly.out_shape
        raise KeyError(name)

    def conv(self, src, name, cout, k=(1, 1), s=(1, 1)):
        h, w, cin = self.shape(src)
        oh, ow = ns.same_pad(h, k[0], s[0])[0], ns.same_pad(w, k[1], s[1])[0]
        kern = self.init.glorot((k[0], k[1], cin, cout), k[0] * k[1] * cin, k[0] * k[1] * cout)
        return self.add(ns.Layer(name, ns.CONV, [src], {"filters": cout, "kernel": k, "strides": s}, {"kernel": kern}, (oh, ow, cout)))

    def dw(self, src, name, s=(1, 1)):
        h, w, c = self.shape(src)
        oh, ow = ns.same_pad(h, 3, s[0])[0], ns.same_pad(w, 3, s[1])[0]
        kern = self.init.glorot((3, 3, c), 9 * c, 9)
        return self.add(ns.Layer(name, ns.DWCONV, [src], {"kernel": (3, 3), "strides": s}, {"kernel": kern}, (oh, ow, c)))

    def bn(self, src, name):
        shp = self.shape(src)
        return self.add(ns.Layer(name, ns.BN, [src], {"eps": 1e-3}, self.init.bn(shp[-1]), shp))

    def relu6(self, src, name):
        return self.add(ns.Layer(name, ns.RELU, [src], {"max_value": 6}, {}, self.shape(src)))

    def identity(self, src, name):
        return self.add(ns.Layer(name, ns.IDENTITY, [src], {}, {}, self.shape(src)))

    def addl(self, a, b, name):
        return self.add(ns.Layer(name, ns.ADD, [a, b], {}, {}, self.shape(a)))

    def se(self, src, name, reduction):
        h, w, c = self.shape(src)
        cr = max(1, c // reduction)
        sq = self.add(ns.Layer(f"{name}_squeeze", ns.GAP, [src], {"keepdims": True}, {}, (1, 1, c)))
        r = self.add(ns.Layer(f"{name}_reduce", ns.DENSE, [sq], {"units": cr, "activation": "relu"}, {"kernel": self.init.glorot((c, cr), c, cr)}, (1, 1, cr)))
        e = self.add(ns.Layer(f"{name}_expand", ns.DENSE, [r], {"units": c, "activation": "sigmoid"}, {"kernel": self.init.glorot((cr, c), cr, c)}, (1, 1, c)))
        return self.add(ns.Layer(f"{name}_scale", ns.MUL, [src, e], {}, {}, (h, w, c)))


def _frontend_weights(mode, mel_bins, sample_rate, fft_length, mag_scale, init: _Init, chunk_T, spec_width):
    from birdnet_stm32.audio.melbank import hybrid_mel_mixer

    w: dict[str, np.ndarray] = {}
    ones = np.ones(mel_bins, np.float32)
    if mode == "hybrid":
        w["mel"] = hybrid_mel_mixer(sample_rate, fft_length, mel_bins)
    elif mode == "raw":
        w["fb"] = init.glorot((16, mel_bins), 16, 16 * mel_bins)
        for k, v in init.bn(mel_bins).items():
            w[f"fb_{k}"] = v
    if mag_scale == "pwl":  # reference: magnitude.py:99-130
        w["pwl_k0"] = 0.40 * ones
        w["pwl_k"] = np.stack([0.25 * ones, 0.15 * ones, 0.08 * ones])
        w["pwl_w"] = np.stack([ones, ones, ones])
        w["pwl_b"] = np.stack([-0.10 * ones, -0.35 * ones, -0.65 * ones])
    elif mag_scale == "pcen":  # reference: magnitude.py:53-90
        w.update(pcen_agc=0.6 * ones, pcen_k1=0.15 * ones, pcen_sw=ones.copy(), pcen_sb=-0.2 * ones, pcen_k2=0.45 * ones)
    return w


def build_dscnn_model(
    num_mels: int,
    spec_width: int,
    sample_rate: int,
    chunk_duration: int,
    embeddings_size: int,
    num_classes: int,
    audio_frontend: str = "hybrid",
    alpha: float = 1.0,
    depth_multiplier: int = 1,
    fft_length: int = 512,
    mag_scale: str = "pwl",
    frontend_trainable: bool = False,
    class_activation: str = "softmax",
    dropout_rate: float = 0.5,
    n_mfcc: int = 20,
    weight_decay: float = 1e-4,
    use_se: bool = True,
    se_reduction: int = 8,
    use_inverted_residual: bool = True,
    expansion_factor: int = 2,
    use_attention_pooling: bool = False,
    seed: int = 42,
    randomize_bn: bool = False,
    raw_length_limit: int | None = 1 << 16,
) -> ns.NetSpec:
    """Build the topology; see the module docstring.  ``seed``/``randomize_bn`` control the fresh weights.

    ``raw_length_limit`` is the reference's STM32N6 guard on the raw frontend (``sample_rate *
    chunk_duration`` must stay below 65536, reference dscnn.py:144-151); pass ``None`` to lift it on MI355X.
    """
    audio_frontend = normalize_frontend_name(audio_frontend)
    T = int(sample_rate * chunk_duration)
    if audio_frontend == "raw" and raw_length_limit is not None and T >= raw_length_limit:
        raise ValueError(
            f"STM32N6 constraint: raw input length (sample_rate*chunk_duration={T}) must be < {raw_length_limit}.\n"
            "Use --sample_rate 16000, --chunk_duration 2, or --audio_frontend hybrid/librosa."
        )
    init = _Init(seed, randomize_bn)
    g = _Graph(init)

    if audio_frontend in ("librosa", "mfcc", "log_mel"):
        bins = n_mfcc if audio_frontend == "mfcc" else num_mels
        mode, in_name, in_shape = "precomputed", "mel_spectrogram_input", (bins, spec_width, 1)
        mag = mag_scale if audio_frontend == "librosa" else "none"
    elif audio_frontend == "hybrid":
        bins, mode, in_name, in_shape, mag = num_mels, "hybrid", "linear_spectrogram_input", (fft_length // 2 + 1, spec_width, 1), mag_scale
    else:
        bins, mode, in_name, in_shape, mag = num_mels, "raw", "raw_audio_input", (T, 1), mag_scale
    g.add(ns.Layer(in_name, ns.INPUT, [], {}, {}, in_shape))
    fattrs = {"mode": mode, "mel_bins": bins, "spec_width": spec_width, "sample_rate": int(sample_rate), "chunk_duration": float(chunk_duration),
              "fft_length": fft_length, "mag_scale": mag, "norm": mode == "hybrid", "fb_eps": 1e-3}
    x = g.add(ns.Layer("audio_frontend", ns.FRONTEND, [in_name], fattrs,
                       _frontend_weights(mode, bins, sample_rate, fft_length, mag, init, T, spec_width), (bins, spec_width, 1)))

    stem_ch = _md(int(16 * alpha), 8)
    x = g.relu6(g.bn(g.conv(x, "stem_conv", stem_ch, (3, 3), (1, 2)), "stem_bn"), "stem_relu")

    for si, (bf, br) in enumerate(zip((32, 64, 128, 256), (2, 3, 4, 2)), start=1):
        out_ch = _md(int(bf * alpha), 8)
        reps = max(1, int(math.ceil(br * depth_multiplier)))
        for bi in range(1, reps + 1):
            stride = (2, 2) if bi == 1 else (1, 1)
            in_ch = g.shape(x)[-1]
            if use_inverted_residual:
                n = f"stage{si}_ir{bi}"
                hidden = _md(int(in_ch) * expansion_factor, 8)
                y = g.relu6(g.bn(g.conv(x, f"{n}_expand", hidden), f"{n}_expand_bn"), f"{n}_expand_relu")
                y = g.relu6(g.bn(g.dw(y, f"{n}_dw", stride), f"{n}_dw_bn"), f"{n}_dw_relu")
                if use_se:
                    y = g.se(y, f"{n}_se", se_reduction)
                y = g.identity(g.bn(g.conv(y, f"{n}_project", out_ch), f"{n}_project_bn"), f"{n}_drop")
                if stride == (1, 1) and in_ch == out_ch:
                    y = g.addl(x, y, f"{n}_add")
                x = y
            else:
                n = f"stage{si}_ds{bi}"
                y = g.relu6(g.bn(g.dw(x, f"{n}_dw", stride), f"{n}_dw_bn"), f"{n}_dw_relu")
                y = g.identity(g.bn(g.conv(y, f"{n}_pw", out_ch), f"{n}_pw_bn"), f"{n}_drop")
                if stride == (1, 1) and in_ch == out_ch:
                    y = g.addl(x, y, f"{n}_add")
                x = g.relu6(y, f"{n}_pw_relu")
                if use_se:
                    x = g.se(x, f"stage{si}_se{bi}", se_reduction)

    emb_ch = _md(int(embeddings_size), 8)
    if g.shape(x)[-1] != emb_ch:
        x = g.relu6(g.bn(g.conv(x, "emb_conv", emb_ch), "emb_bn"), "emb_relu")

    c = g.shape(x)[-1]
    if use_attention_pooling:
        x = g.add(ns.Layer("attn_pool", ns.ATTNPOOL, [x], {}, {"score": init.glorot((c,), c, 1)}, (c,)))
    else:
        x = g.add(ns.Layer("gap", ns.GAP, [x], {"keepdims": False}, {}, (c,)))
    x = g.identity(x, "dropout")
    g.add(ns.Layer("pred", ns.DENSE, [x], {"units": num_classes, "activation": class_activation},
                   {"kernel": init.glorot((c, num_classes), c, num_classes), "bias": np.zeros(num_classes, np.float32)}, (num_classes,)))
    return ns.NetSpec(g.layers, name="dscnn_audio", meta={"source": "build_dscnn_model", "seed": seed})
