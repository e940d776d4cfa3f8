"""Load a Keras-3 `.keras` archive into a :class:`NetSpec` without TensorFlow.

The reference does ``tf.keras.models.load_model(path, custom_objects={AudioFrontendLayer,
MagnitudeScalingLayer})`` (reference: birdnet_stm32/models/runners.py:109-113).  Here the
archive's ``config.json`` (functional graph) is walked in layer order and each layer's
variables are pulled from ``model.weights.h5``.  Keras 3 stores them under
``/layers/<snake_case(class)>[_k]/vars/<i>`` where *k* counts layers of the same class in
``model.layers`` order.
"""

from __future__ import annotations

import json
import re
import zipfile

import numpy as np

from birdnet_stm32.models import _netspec as ns
from birdnet_stm32.models._h5_reader import read_h5_datasets


def _snake(name: str) -> str:
    s = re.sub(r"(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub(r"([a-z])([A-Z])", r"\1_\2", s).lower()


class _Vars:
    """Variable lookup for one layer directory of the weights file."""

    def __init__(self, datasets: dict[str, np.ndarray], prefix: str):
        self.prefix = prefix
        self.sub = {k[len(prefix) :]: v for k, v in datasets.items() if k.startswith(prefix + "/")}

    def get(self, rel: str) -> np.ndarray:
        key = "/" + rel.strip("/")
        if key not in self.sub:
            raise KeyError(f"{self.prefix}{key} missing; have {sorted(self.sub)[:8]}...")
        return np.asarray(self.sub[key], dtype=np.float32)

    def find(self, fragment: str) -> list[str]:
        return sorted(k for k in self.sub if fragment in k)

    def __bool__(self):
        return bool(self.sub)


def _inbound_names(layer_cfg: dict) -> list[str]:
    names: list[str] = []

    def visit(obj):
        if isinstance(obj, dict):
            if obj.get("class_name") == "__keras_tensor__":
                names.append(obj["config"]["keras_history"][0])
            else:
                for v in obj.values():
                    visit(v)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                visit(v)

    for node in layer_cfg.get("inbound_nodes", []):
        visit(node.get("args", []))
    return names


def _frontend_layer(cfg: dict, v: _Vars, name: str, inputs: list[str], frontend_norm) -> ns.Layer:
    mode = cfg["mode"]
    mel_bins = int(cfg["mel_bins"])
    mag = cfg.get("mag_scale", "none")
    attrs = {
        "mode": mode,
        "mel_bins": mel_bins,
        "spec_width": int(cfg["spec_width"]),
        "sample_rate": int(cfg["sample_rate"]),
        "chunk_duration": float(cfg["chunk_duration"]),
        "fft_length": int(cfg.get("fft_length", 512)),
        "mag_scale": mag,
    }
    # Variables sit directly under the layer in checkpoints saved before the reference's
    # magnitude refactor and under mag_layer/ after it (SURVEY.md finding 6).
    nested = bool(v.find("/mag_layer/"))
    base = "mag_layer/" if nested else ""
    # Per-sample max normalisation exists only in the refactored frontend
    # (reference: birdnet_stm32/models/frontend.py:338-343; SURVEY.md finding 4).
    attrs["norm"] = bool(nested) if frontend_norm is None else bool(frontend_norm)
    w: dict[str, np.ndarray] = {}
    if mode == "hybrid":
        w["mel"] = v.get("mel_mixer/vars/0")[0, 0]  # [F_pad, M]
    elif mode == "raw":
        w["fb"] = v.get("fb2d/vars/0")[0, :, 0, :]  # [16, M]
        for i, key in enumerate(("gamma", "beta", "mean", "var")):
            w[f"fb_{key}"] = v.get(f"fb_bn/vars/{i}")
        attrs["fb_eps"] = 1e-3
    if mag == "pwl":
        sfx = ["", "_1", "_2"]
        w["pwl_k0"] = v.get(f"{base}_pwl_k0_dw/vars/0").reshape(-1)
        w["pwl_k"] = np.stack([v.get(f"{base}_pwl_k_dws/depthwise_conv2d{s}/vars/0").reshape(-1) for s in sfx])
        w["pwl_w"] = np.stack([v.get(f"{base}_pwl_shift_dws/depthwise_conv2d{s}/vars/0").reshape(-1) for s in sfx])
        w["pwl_b"] = np.stack([v.get(f"{base}_pwl_shift_dws/depthwise_conv2d{s}/vars/1").reshape(-1) for s in sfx])
    elif mag == "pcen":
        w["pcen_agc"] = v.get(f"{base}_pcen_agc_dw/vars/0").reshape(-1)
        w["pcen_k1"] = v.get(f"{base}_pcen_k1_dw/vars/0").reshape(-1)
        w["pcen_sw"] = v.get(f"{base}_pcen_shift_dw/vars/0").reshape(-1)
        w["pcen_sb"] = v.get(f"{base}_pcen_shift_dw/vars/1").reshape(-1)
        w["pcen_k2"] = v.get(f"{base}_pcen_k2mk1_dw/vars/0").reshape(-1)
    return ns.Layer(name, ns.FRONTEND, inputs, attrs, w, (mel_bins, int(cfg["spec_width"]), 1))


def load_keras_archive(path: str, frontend_norm: bool | None = None) -> ns.NetSpec:
    """Decode ``path`` (a `.keras` zip) into a NetSpec.

    Args:
        path: `.keras` file.
        frontend_norm: force the hybrid frontend's per-sample max normalisation on/off;
            ``None`` infers it from the checkpoint generation (off for the shipped legacy model).
    """
    with zipfile.ZipFile(path) as z:
        config = json.loads(z.read("config.json"))
        meta = json.loads(z.read("metadata.json")) if "metadata.json" in z.namelist() else {}
        datasets = read_h5_datasets(z.read("model.weights.h5"))
    if config.get("class_name") != "Functional":
        raise ValueError(f"unsupported Keras model class {config.get('class_name')!r}")

    counters: dict[str, int] = {}
    layers: list[ns.Layer] = []
    shapes: dict[str, tuple] = {}

    for lc in config["config"]["layers"]:
        cls, cfg, name = lc["class_name"], lc["config"], lc["config"]["name"]
        snake = _snake(cls)
        k = counters.get(snake, 0)
        counters[snake] = k + 1
        v = _Vars(datasets, f"/layers/{snake}" + (f"_{k}" if k else ""))
        inputs = _inbound_names(lc)
        src_shape = shapes.get(inputs[0]) if inputs else None

        if cls == "InputLayer":
            shape = tuple(cfg["batch_shape"][1:])
            ly = ns.Layer(name, ns.INPUT, [], {}, {}, shape)
        elif cls == "AudioFrontendLayer":
            ly = _frontend_layer(cfg, v, name, inputs, frontend_norm)
        elif cls in ("Conv2D", "DepthwiseConv2D"):
            if cfg.get("use_bias"):
                raise ValueError(f"{name}: biased convolutions are not produced by the reference builder")
            if str(cfg.get("padding", "same")).lower() != "same":
                raise ValueError(f"{name}: only SAME padding is supported")
            kh, kw = cfg["kernel_size"]
            sh, sw = cfg["strides"]
            h, w_, c = src_shape
            oh, _, _ = ns.same_pad(h, kh, sh)
            ow, _, _ = ns.same_pad(w_, kw, sw)
            kern = v.get("vars/0")
            if cls == "Conv2D":
                cout = int(cfg["filters"])
                ly = ns.Layer(name, ns.CONV, inputs, {"filters": cout, "kernel": (kh, kw), "strides": (sh, sw)}, {"kernel": kern}, (oh, ow, cout))
            else:
                ly = ns.Layer(name, ns.DWCONV, inputs, {"kernel": (kh, kw), "strides": (sh, sw)}, {"kernel": kern[..., 0]}, (oh, ow, c))
        elif cls == "BatchNormalization":
            w = {key: v.get(f"vars/{i}") for i, key in enumerate(("gamma", "beta", "mean", "var"))}
            ly = ns.Layer(name, ns.BN, inputs, {"eps": float(cfg.get("epsilon", 1e-3))}, w, src_shape)
        elif cls == "ReLU":
            ly = ns.Layer(name, ns.RELU, inputs, {"max_value": cfg.get("max_value")}, {}, src_shape)
        elif cls == "Add":
            ly = ns.Layer(name, ns.ADD, inputs, {}, {}, src_shape)
        elif cls == "Multiply":
            ly = ns.Layer(name, ns.MUL, inputs, {}, {}, max((shapes[i] for i in inputs), key=lambda s: int(np.prod(s))))
        elif cls == "GlobalAveragePooling2D":
            keep = bool(cfg.get("keepdims", False))
            c = src_shape[-1]
            ly = ns.Layer(name, ns.GAP, inputs, {"keepdims": keep}, {}, (1, 1, c) if keep else (c,))
        elif cls == "Dense":
            w = {"kernel": v.get("vars/0")}
            if cfg.get("use_bias", True):
                w["bias"] = v.get("vars/1")
            units = int(cfg["units"])
            ly = ns.Layer(name, ns.DENSE, inputs, {"units": units, "activation": cfg.get("activation", "linear")}, w, (*src_shape[:-1], units))
        elif cls == "AttentionPooling":
            keys = v.find("vars/0")
            if not keys:
                raise ValueError(f"{name}: attention-pooling score weights not found")
            ly = ns.Layer(name, ns.ATTNPOOL, inputs, {}, {"score": np.asarray(v.sub[keys[0]], np.float32).reshape(-1)}, (src_shape[-1],))
        elif cls in ("Dropout", "SpatialDropout2D"):
            ly = ns.Layer(name, ns.IDENTITY, inputs, {}, {}, src_shape)
        else:
            raise ValueError(f"unsupported Keras layer class {cls!r} ({name})")
        layers.append(ly)
        shapes[name] = ly.out_shape

    return ns.NetSpec(layers, name=config["config"].get("name", "dscnn_audio"), meta={"source": path, **meta})
