"""Layer-by-layer evaluation of a DS-CNN graph with Keras inference semantics (numpy).

ORACLE — test infrastructure only (see oracle/__init__.py).  **Parity unpinned**: Keras
cannot run here; the semantics below are TensorFlow's documented ones, anchored on the
reference's layer definitions.

Follows, layer for layer and *unfused* (conv, then BatchNorm, then ReLU as separate
steps — the device plan folds them, so the two computations differ in rounding):

* hybrid frontend: birdnet_stm32/models/frontend.py:299-345 (transpose, zero-pad F to a
  multiple of 8, 1x1 mel mixer, ReLU, optional per-sample max normalisation
  ``y / (max + 1e-6)`` (:338-343), magnitude scaling, transpose back);
* raw frontend: frontend.py:347-358 (symmetric zero pad, VALID 1x16 strided conv, BN,
  ReLU6, magnitude scaling, transpose);
* PWL / PCEN: birdnet_stm32/models/magnitude.py:166-192;
* stem / ds_conv_block: birdnet_stm32/models/dscnn.py:28-84,198-202; inverted residual and
  squeeze-excite: birdnet_stm32/models/blocks.py:27-133; attention pooling: blocks.py:136-159;
* head: dscnn.py:248-261.

TensorFlow conventions restated: NHWC; ``SAME`` padding puts the odd cell after
(pad_before = total // 2); BatchNorm inference ``gamma * (x - mean) / sqrt(var + eps) + beta``;
``ReLU(max_value=6)``; Dense contracts the last axis.
"""

from __future__ import annotations

import numpy as np


def _same_pad(size: int, k: int, s: int):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, kernel, strides):
    """x [B,H,W,Cin], kernel [kh,kw,Cin,Cout] -> [B,OH,OW,Cout]."""
    kh, kw, cin, cout = kernel.shape
    sh, sw = strides
    B, H, W, _ = x.shape
    oh, pt, pb = _same_pad(H, kh, sh)
    ow, pl, pr = _same_pad(W, kw, sw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((B, oh, ow, cout), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i : i + (oh - 1) * sh + 1 : sh, j : j + (ow - 1) * sw + 1 : sw, :]
            out += patch @ kernel[i, j].astype(x.dtype)
    return out


def dwconv2d_same(x, kernel, strides):
    """x [B,H,W,C], kernel [kh,kw,C] -> [B,OH,OW,C]."""
    kh, kw, _ = kernel.shape
    sh, sw = strides
    B, H, W, C = x.shape
    oh, pt, pb = _same_pad(H, kh, sh)
    ow, pl, pr = _same_pad(W, kw, sw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((B, oh, ow, C), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            out += xp[:, i : i + (oh - 1) * sh + 1 : sh, j : j + (ow - 1) * sw + 1 : sw, :] * kernel[i, j].astype(x.dtype)
    return out


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def magnitude_scale(y, attrs, w, dt):
    """Channel-wise magnitude scaling on [..., C] (magnitude.py:166-192)."""
    kind = attrs.get("mag_scale", "none")
    relu = lambda z: np.maximum(z, 0)  # noqa: E731
    if kind == "pwl":
        out = y * w["pwl_k0"].astype(dt)
        for i in range(3):
            out = out + w["pwl_k"][i].astype(dt) * relu(w["pwl_w"][i].astype(dt) * y + w["pwl_b"][i].astype(dt))
        return out
    if kind == "pcen":
        y0 = relu(y - w["pcen_agc"].astype(dt) * y)  # the K "EMA" pools are (1,1) average pools = identity
        b1 = w["pcen_k1"].astype(dt) * y0
        b2 = w["pcen_k2"].astype(dt) * relu(w["pcen_sw"].astype(dt) * y0 + w["pcen_sb"].astype(dt))
        return relu(b1 + b2)
    if kind == "db":
        return dt(10.0) * np.log(np.maximum(y, dt(1e-6))) / np.log(dt(10.0))
    return y


def frontend_forward(x, layer, dt=np.float32):
    """AudioFrontendLayer.call for the 'hybrid', 'raw' and 'precomputed' modes."""
    a, w = layer.attrs, layer.weights
    mode, W = a["mode"], a["spec_width"]
    if mode == "precomputed":
        return x[:, :, :W, :].astype(dt)
    if mode == "hybrid":
        y = np.transpose(x.astype(dt), (0, 3, 2, 1))[:, :, :W, :]  # [B,1,T,F]
        mel = w["mel"].astype(dt)  # [F_pad, M]
        pad = mel.shape[0] - y.shape[-1]
        if pad:
            y = np.concatenate([y, np.zeros((*y.shape[:-1], pad), dt)], axis=-1)
        y = np.maximum(y @ mel, 0)
        if a.get("norm", False):
            y = y / (y.max(axis=(1, 2, 3), keepdims=True) + dt(1e-6))
        y = magnitude_scale(y, a, w, dt)
        return np.transpose(y, (0, 3, 2, 1))[:, :, :W, :]
    if mode == "raw":
        T = int(a["sample_rate"] * a["chunk_duration"])
        stride = -(-T // W)
        pad_total = max(0, stride * (W - 1) + 16 - T)
        left = pad_total // 2
        sig = x[:, :T, 0].astype(dt)
        sig = np.pad(sig, ((0, 0), (left, pad_total - left)))
        idx = np.arange(16)[None, :] + stride * np.arange(W)[:, None]
        frames = sig[:, idx]  # [B,W,16]
        y = frames @ w["fb"].astype(dt)  # [B,W,M]
        y = w["fb_gamma"].astype(dt) * (y - w["fb_mean"].astype(dt)) / np.sqrt(w["fb_var"].astype(dt) + dt(a.get("fb_eps", 1e-3))) + w["fb_beta"].astype(dt)
        y = np.minimum(np.maximum(y, 0), dt(6))
        y = magnitude_scale(y, a, w, dt)
        return np.transpose(y, (0, 2, 1))[..., None]  # [B,M,W,1]
    raise ValueError(mode)


def forward(spec, x, dtype=np.float32, return_all: bool = False, return_logits: bool = False):
    """Run ``spec`` (a NetSpec) on the batch ``x``; returns the class scores [B, C].

    With ``return_all`` a dict ``{layer_name: activation}`` is returned as well; with
    ``return_logits`` the pre-activation output of the final Dense is returned too.
    """
    dt = np.dtype(dtype).type
    acts: dict[str, np.ndarray] = {}
    logits = None
    for ly in spec.layers:
        k = ly.kind
        src = [acts[n] for n in ly.inputs]
        if k == "input":
            y = np.asarray(x).astype(dt)
        elif k == "frontend":
            y = frontend_forward(src[0], ly, dt)
        elif k == "conv2d":
            y = conv2d_same(src[0], ly.weights["kernel"].astype(dt), ly.attrs["strides"])
        elif k == "dwconv2d":
            y = dwconv2d_same(src[0], ly.weights["kernel"].astype(dt), ly.attrs["strides"])
        elif k == "batchnorm":
            w = ly.weights
            y = w["gamma"].astype(dt) * (src[0] - w["mean"].astype(dt)) / np.sqrt(w["var"].astype(dt) + dt(ly.attrs["eps"])) + w["beta"].astype(dt)
        elif k == "relu":
            y = np.maximum(src[0], 0)
            if ly.attrs.get("max_value") is not None:
                y = np.minimum(y, dt(ly.attrs["max_value"]))
        elif k == "add":
            y = src[0] + src[1]
        elif k == "multiply":
            y = src[0] * src[1]
        elif k == "gap":
            y = src[0].mean(axis=(1, 2), keepdims=bool(ly.attrs.get("keepdims")))
        elif k == "dense":
            y = src[0] @ ly.weights["kernel"].astype(dt)
            if "bias" in ly.weights:
                y = y + ly.weights["bias"].astype(dt)
            act = ly.attrs.get("activation", "linear")
            if ly is spec.layers[-1]:
                logits = y
            if act == "relu":
                y = np.maximum(y, 0)
            elif act == "sigmoid":
                y = _sigmoid(y)
            elif act == "softmax":
                e = np.exp(y - y.max(axis=-1, keepdims=True))
                y = e / e.sum(axis=-1, keepdims=True)
            elif act != "linear":
                raise ValueError(act)
        elif k == "attnpool":
            B, H, W, C = src[0].shape
            flat = src[0].reshape(B, H * W, C)
            s = flat @ ly.weights["score"].astype(dt)
            e = np.exp(s - s.max(axis=1, keepdims=True))
            y = (flat * (e / e.sum(axis=1, keepdims=True))[..., None]).sum(axis=1)
        elif k == "identity":
            y = src[0]
        else:
            raise ValueError(f"unknown layer kind {k}")
        acts[ly.name] = y
    out = acts[spec.layers[-1].name]
    res = [out]
    if return_logits:
        res.append(logits)
    if return_all:
        res.append(acts)
    return res[0] if len(res) == 1 else tuple(res)
