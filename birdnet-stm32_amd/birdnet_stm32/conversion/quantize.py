"""Post-training quantisation of a float DS-CNN to the INT8 graph family the runners execute — without TensorFlow.

The reference converts with ``tf.lite.TFLiteConverter`` (reference: birdnet_stm32/conversion/quantize.py:113-168:
``Optimize.DEFAULT``, a representative dataset, ``TFLITE_BUILTINS_INT8``, float32 input/output, per-channel weights
unless ``per_tensor``).  TensorFlow is not available on the MI355X image, so the INT8 path used to be limited to the one
shipped ``.tflite``.  This module is the own calibrator + exporter of SURVEY.md §8f rank 4, in the form the rest of the
build can consume directly:

* :func:`representative_data_gen` — the reference's calibration-sample selection (:19-110): one centre chunk per file,
  near-silent samples skipped, inputs in the exact model-input shape;
* :func:`requantize_like` — given an existing INT8 graph of the SAME topology as a structural template (operator list,
  tensor wiring, data-movement constants, the frozen frontend's constants) and a float model (``NetSpec`` from a
  ``.keras`` archive), it (1) folds BatchNorm and maps the float kernels onto the template's weight tensors in operator
  order, (2) runs the graph in float32 over the representative samples and records every activation tensor's range,
  (3) chooses TFLite's affine int8 parameters for the activations (range widened to include 0, ``scale = (max - min) /
  255``, nudged zero point; tensors tied by data-movement operators share one range; LOGISTIC output fixed at 1/256,
  -128), symmetric per-channel int8 weights (``scale_c = max|w_c| / 127``) and int32 biases at ``s_in * s_w[c]``,
  and returns a new :class:`TfliteModel`.  ``lower_i8`` turns that into a device plan like any parsed ``.tflite``.

Scope: the hybrid-frontend DS-CNN family without squeeze-excite / inverted residuals (the operators the INT8 kernels
implement); the frontend's constants (mel mixer, PWL) are taken from the template — the reference trains with
``frontend_trainable=False`` by default.  Calibration is host-side numpy, as the reference's converter is a host tool.
"""

from __future__ import annotations

import copy
import random

import numpy as np

from birdnet_stm32.models import _netspec as ns
from birdnet_stm32.models._tflite_reader import TfliteModel, TfliteTensor

_MOVERS = ("TRANSPOSE", "STRIDED_SLICE", "CONCATENATION", "RESHAPE", "REDUCE_MAX", "PAD")  # output shares the input's quantisation
_PLAIN_KINDS = (ns.INPUT, ns.FRONTEND, ns.CONV, ns.DWCONV, ns.BN, ns.RELU, ns.ADD, ns.GAP, ns.DENSE, ns.IDENTITY)


# --------------------------------------------------------------------------------------- representative data
def representative_data_gen(file_paths: list[str], cfg: dict, num_samples: int = 100, snr_threshold: float = 0.01, spectrogram_fn=None):
    """Yield ``[x]`` with ``x`` in the model's input shape, one centre chunk per sampled file (reference :19-110).

    ``spectrogram_fn(chunks [N,T], n_fft, spec_width) -> [N, F, W]`` replaces the GPU STFT (tests); the default computes
    hybrid spectrograms with ``bn_stft_mag`` and mel maps with ``bn_mel_spectrogram``.
    """
    from birdnet_stm32.audio.io import load_audio_file
    from birdnet_stm32.models.frontend import normalize_frontend_name

    sr, cd = int(cfg["sample_rate"]), float(cfg["chunk_duration"])
    width, n_fft = int(cfg["spec_width"]), int(cfg["fft_length"])
    frontend = normalize_frontend_name(cfg["audio_frontend"])
    T = int(sr * cd)
    if len(file_paths) == 0:
        raise ValueError("No audio files found for representative dataset generation.")
    chosen = random.sample(list(file_paths), min(num_samples, len(file_paths)))
    read_s = float(cfg.get("max_duration", 0)) or max(30.0, cd * 5.0)
    for path in chosen:
        chunks = load_audio_file(path, sample_rate=sr, max_duration=read_s, chunk_duration=cd)
        if len(chunks) == 0:
            continue
        chunks = np.asarray(chunks, np.float32)
        if chunks.shape[0] > 1:
            chunks = chunks[chunks.shape[0] // 2][None]  # centre chunk: avoids silence-only calibration
        if frontend == "hybrid":
            if spectrogram_fn is None:
                from birdnet_stm32.audio.spectrogram import spectrograms_from_chunks as spectrogram_fn
            pool = [np.asarray(s, np.float32)[None, :, :, None] for s in spectrogram_fn(chunks, n_fft, width)]
        elif frontend == "librosa":
            from birdnet_stm32.audio.spectrogram import mel_spectrograms_from_chunks

            pool = [s[None, :, :, None] for s in mel_spectrograms_from_chunks(chunks, sr, n_fft, int(cfg["num_mels"]), width,
                                                                               cfg.get("mag_scale", "none"), "mel")]
        elif frontend == "raw":
            pool = []
            for c in chunks:
                x = np.zeros(T, np.float32)
                x[: min(T, c.shape[0])] = c[:T]
                if snr_threshold > 0 and float(np.sqrt(np.mean(x**2))) < snr_threshold:
                    continue
                pool.append((x / (np.max(np.abs(x)) + 1e-6)).astype(np.float32)[None, :, None])
        else:
            raise ValueError(f"Invalid audio frontend: {frontend}")
        for x in pool:
            if frontend != "raw" and snr_threshold > 0 and float(np.mean(np.abs(x))) < snr_threshold:
                continue
            yield [x]


# --------------------------------------------------------------------------------------- float execution of a TFLite graph
def _same_pad(size: int, k: int, s: int):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def _act(y: np.ndarray, kind: str) -> np.ndarray:
    if kind == "relu":
        return np.maximum(y, 0.0)
    if kind == "relu6":
        return np.clip(y, 0.0, 6.0)
    if kind == "none":
        return y
    raise NotImplementedError(f"fused activation {kind}")


def _windows(x: np.ndarray, kh: int, kw: int, sh: int, sw: int, padding: str = "SAME"):
    """Sliding windows ``[B, OH, OW, C, kh, kw]`` of an NHWC array (SAME: TensorFlow's asymmetric zero padding; VALID: none)."""
    _, H, W, _ = x.shape
    xp = x
    if padding == "SAME":
        _, pt, pb = _same_pad(H, kh, sh)
        _, pl, pr = _same_pad(W, kw, sw)
        xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))
    return win[:, ::sh, ::sw]


def _conv2d(x, w, b, sh, sw, act, padding="SAME"):  # w [Cout, kh, kw, Cin] (TFLite layout)
    cout, kh, kw, _ = w.shape
    if (kh, kw) == (1, 1) and (sh, sw) == (1, 1):
        y = x @ w[:, 0, 0, :].T
    else:
        y = np.einsum("bhwcij,oijc->bhwo", _windows(x, kh, kw, sh, sw, padding), w, optimize=True)
    return _act(y + b, act)


def _dwconv2d(x, w, b, sh, sw, act, padding="SAME"):  # w [1, kh, kw, C]
    _, kh, kw, _ = w.shape
    if (kh, kw) == (1, 1) and (sh, sw) == (1, 1):
        y = x * w[0, 0, 0]
    else:
        y = np.einsum("bhwcij,ijc->bhwc", _windows(x, kh, kw, sh, sw, padding), w[0], optimize=True)
    return _act(y + b, act)


def _strided_slice(x, begin, end, strides, o):
    if o["ellipsis_mask"] or o["new_axis_mask"]:
        raise NotImplementedError("ellipsis / new-axis masks")
    sl = []
    for d in range(len(begin)):
        if o["shrink_axis_mask"] >> d & 1:
            sl.append(int(begin[d]))
            continue
        sl.append(slice(None if o["begin_mask"] >> d & 1 else int(begin[d]), None if o["end_mask"] >> d & 1 else int(end[d]), int(strides[d])))
    return x[tuple(sl)]


def run_float(model: TfliteModel, consts: dict[int, np.ndarray], x: np.ndarray) -> dict[int, np.ndarray]:
    """Execute ``model`` in float32: ``consts`` maps constant tensor index -> real-valued array (integer shape constants
    stay integers); returns every tensor's value."""
    env: dict[int, np.ndarray] = dict(consts)
    env[model.inputs[0]] = np.asarray(x, np.float32)
    for op in model.ops:
        i, n = op.inputs, op.name
        if n in ("QUANTIZE", "DEQUANTIZE"):
            y = env[i[0]]
        elif n == "TRANSPOSE":
            y = np.transpose(env[i[0]], [int(v) for v in env[i[1]]])
        elif n == "STRIDED_SLICE":
            y = _strided_slice(env[i[0]], env[i[1]], env[i[2]], env[i[3]], op.options)
        elif n == "SHAPE":
            y = np.asarray(env[i[0]].shape, np.int32)
        elif n == "PACK":
            y = np.stack([np.asarray(env[k]) for k in i], axis=op.options["axis"]).astype(np.int32)
        elif n == "FILL":
            y = np.full([int(v) for v in env[i[0]]], float(np.asarray(env[i[1]]).reshape(-1)[0]), np.float32)
        elif n == "CONCATENATION":
            y = np.concatenate([env[k] for k in i], axis=op.options["axis"])
        elif n == "RESHAPE":
            y = env[i[0]].reshape([env[i[0]].shape[0]] + [int(v) for v in env[i[1]]][1:])
        elif n == "PAD":
            y = np.pad(env[i[0]], [(int(a), int(b)) for a, b in np.asarray(env[i[1]]).reshape(-1, 2)])
        elif n == "CONV_2D":
            y = _conv2d(env[i[0]], env[i[1]], env[i[2]], op.options["stride_h"], op.options["stride_w"], op.options["activation"], op.options.get("padding", "SAME"))
        elif n == "DEPTHWISE_CONV_2D":
            y = _dwconv2d(env[i[0]], env[i[1]], env[i[2]], op.options["stride_h"], op.options["stride_w"], op.options["activation"], op.options.get("padding", "SAME"))
        elif n == "ADD":
            y = _act(env[i[0]] + env[i[1]], op.options["activation"])
        elif n == "MEAN":
            axes = tuple(int(a) % env[i[0]].ndim for a in np.atleast_1d(env[i[1]]))
            y = env[i[0]].mean(axis=axes, keepdims=bool(op.options.get("keep_dims")))
        elif n == "FULLY_CONNECTED":
            xin = env[i[0]]
            flat = xin.reshape(-1, env[i[1]].shape[1])
            y = _act(flat @ env[i[1]].T + (env[i[2]] if len(i) > 2 and i[2] >= 0 else 0.0), op.options.get("activation", "none"))
            if op.options.get("keep_num_dims"):
                y = y.reshape(*xin.shape[:-1], -1)
        elif n == "LOGISTIC":
            y = 1.0 / (1.0 + np.exp(-env[i[0]]))
        elif n == "MUL":
            y = _act(env[i[0]] * env[i[1]], op.options["activation"])
        elif n == "REDUCE_MAX":
            axes = tuple(int(a) % env[i[0]].ndim for a in np.atleast_1d(env[i[1]]))
            y = env[i[0]].max(axis=axes, keepdims=bool(op.options.get("keep_dims")))
        elif n == "DIV":
            y = _act(env[i[0]] / env[i[1]], op.options.get("activation", "none"))
        elif n == "SOFTMAX":
            z = (env[i[0]] - env[i[0]].max(axis=-1, keepdims=True)) * op.options.get("beta", 1.0)
            y = np.exp(z) / np.exp(z).sum(axis=-1, keepdims=True)
        elif n == "SUM":
            axes = tuple(int(a) % env[i[0]].ndim for a in np.atleast_1d(env[i[1]]))
            y = env[i[0]].sum(axis=axes, keepdims=bool(op.options.get("keep_dims")))
        else:
            raise NotImplementedError(f"operator {n} in float calibration")
        env[op.outputs[0]] = y.astype(np.float32) if y.dtype.kind == "f" else y
    return env


# --------------------------------------------------------------------------------------- quantisation parameters
def choose_activation_params(rmin: float, rmax: float) -> tuple[float, int]:
    """TFLite's affine int8 parameters for an observed range (widened to contain 0; zero point nudged onto the grid)."""
    rmin, rmax = min(float(rmin), 0.0), max(float(rmax), 0.0)
    if rmax - rmin < 2e-6:  # a tensor that never left zero on the calibration data: the converter assigns [-1e-6, 1e-6]
        rmin, rmax = -1e-6, 1e-6
    scale = (rmax - rmin) / 255.0
    zp = int(np.clip(np.round(-128.0 - rmin / scale), -128, 127))
    return float(np.float32(scale)), zp


def quantize_weights(w: np.ndarray, channel_axis: int, per_tensor: bool = False):
    """Symmetric int8 weights: ``scale_c = max(max|w_c|, 5e-7) / 127`` (one common scale with ``per_tensor``), values in [-127, 127]."""
    w = np.asarray(w, np.float64)
    axes = tuple(a for a in range(w.ndim) if a != channel_axis % w.ndim)
    amax = np.abs(w).max(axis=axes)
    if per_tensor:
        amax = np.full_like(amax, amax.max())
    scale = np.maximum(amax, 5e-7) / 127.0  # TFLite floors a channel's magnitude at 5e-7: dead channels quantise to zeros, not to +-127 at a subnormal scale
    shape = [1] * w.ndim
    shape[channel_axis % w.ndim] = -1
    q = np.clip(np.round(w / scale.reshape(shape)), -127, 127).astype(np.int8)
    return q, scale.astype(np.float32)


def _dequant(t: TfliteTensor) -> np.ndarray:
    if t.data is None:
        raise ValueError(f"tensor {t.index} is not constant")
    if not t.is_quantized:
        return np.asarray(t.data)
    s = t.scale.astype(np.float64)
    z = t.zero_point.astype(np.float64)
    if s.size > 1:
        shape = [1] * t.data.ndim
        shape[t.quantized_dimension] = -1
        s, z = s.reshape(shape), z.reshape(shape)
    return ((t.data.astype(np.float64) - z) * s).astype(np.float32)


def _backbone_kernels(spec: ns.NetSpec):
    """BatchNorm-folded (kernel, bias, kind) of every conv / depthwise / dense layer behind the frontend, in graph order."""
    from birdnet_stm32.models._lower_f32 import fold_bn

    by_input: dict[str, list[ns.Layer]] = {}
    for ly in spec.layers:
        for src in ly.inputs:
            by_input.setdefault(src, []).append(ly)
    out = []
    for ly in spec.layers:
        if ly.kind in (ns.CONV, ns.DWCONV):
            nxt = by_input.get(ly.name, [])
            bn = nxt[0] if len(nxt) == 1 and nxt[0].kind == ns.BN else None
            k, b = fold_bn(ly.weights["kernel"], bn)
            if "bias" in ly.weights:
                raise TopologyMismatch(f"{ly.name}: convolution with its own bias")
            out.append((ly.name, ly.kind, k, b))
        elif ly.kind == ns.DENSE:
            if "bias" not in ly.weights or ly is not spec.layers[-1]:
                raise TopologyMismatch(f"{ly.name}: a Dense layer that is not the classifier (squeeze-excite): not the template's layer family")
            out.append((ly.name, ly.kind, ly.weights["kernel"].astype(np.float32), ly.weights["bias"].astype(np.float32)))
        elif ly.kind not in _PLAIN_KINDS:
            raise TopologyMismatch(f"{ly.name}: layer kind {ly.kind!r} (squeeze-excite / attention pooling) is not in the template's layer family")
    return out


class TopologyMismatch(ValueError):
    """The float model is not the template's topology (another layer list, kernel shape or layer family): the template cannot describe it.
    Distinct from every other failure of :func:`requantize_like` so that a caller may fall back to the template-free exporter on THIS only."""


def requantize_like(template: TfliteModel, spec: ns.NetSpec, rep_data_gen, per_tensor: bool = False) -> TfliteModel:
    """Quantise the float model ``spec`` into a new INT8 graph with the operator structure of ``template`` (module docstring)."""
    fa = spec.frontend.attrs
    if fa["mode"] != "hybrid":
        raise TopologyMismatch("the template graph has the hybrid frontend, the float model does not")
    T = template.tensors
    conv_ops = [op for op in template.ops if op.name in ("CONV_2D", "DEPTHWISE_CONV_2D", "FULLY_CONNECTED")]
    # the backbone starts at the first 3x3 CONV_2D (the stem); everything before it is the frozen frontend
    first = next(k for k, op in enumerate(conv_ops) if op.name == "CONV_2D" and T[op.inputs[1]].shape[1:3] == (3, 3))
    kernels = _backbone_kernels(spec)
    if len(kernels) != len(conv_ops) - first:
        raise TopologyMismatch(f"template has {len(conv_ops) - first} backbone convolutions, the float model {len(kernels)}: not the same topology")

    consts: dict[int, np.ndarray] = {t.index: (_dequant(t) if t.is_quantized else np.asarray(t.data)) for t in T if t.data is not None}
    float_wb: dict[int, tuple[np.ndarray, np.ndarray]] = {}  # operator index -> (kernel in TFLite layout, bias), real-valued
    for op, (name, kind, k, b) in zip(conv_ops[first:], kernels):
        want = T[op.inputs[1]].shape
        if op.name == "CONV_2D" and kind == ns.CONV:
            w = np.transpose(k, (3, 0, 1, 2))
        elif op.name == "DEPTHWISE_CONV_2D" and kind == ns.DWCONV:
            w = k.reshape(k.shape[0], k.shape[1], -1)[None]
        elif op.name == "FULLY_CONNECTED" and kind == ns.DENSE:
            w = k.T
        else:
            raise TopologyMismatch(f"operator {op.index} ({op.name}) does not line up with layer {name} ({kind})")
        if tuple(w.shape) != tuple(want):
            raise TopologyMismatch(f"layer {name}: kernel {w.shape} vs template {want}")
        float_wb[op.index] = (w.astype(np.float32), b.astype(np.float32))
        consts[op.inputs[1]], consts[op.inputs[2]] = float_wb[op.index]
    for op in conv_ops[:first]:  # frontend: constants of the template, real-valued
        float_wb[op.index] = (consts[op.inputs[1]], consts[op.inputs[2]])

    new = quantize_graph(template, consts, float_wb, rep_data_gen, per_tensor=per_tensor, fresh_weights={op.index for op in conv_ops[first:]})
    new.description = (template.description or "") + " | requantised without TensorFlow (birdnet_stm32.conversion.quantize)"
    return new


def quantize_graph(template: TfliteModel, consts: dict[int, np.ndarray], float_wb: dict[int, tuple[np.ndarray, np.ndarray]], rep_data_gen,
                   per_tensor: bool = False, fresh_weights: set[int] | None = None) -> TfliteModel:
    """Calibrate and quantise: ``template`` gives the operator structure (its quantised tensors are re-parameterised), ``consts`` the
    real-valued constants for the float run, ``float_wb[op.index]`` the real-valued (kernel in TFLite layout, bias) of every
    convolution / fully-connected operator.  Operators in ``fresh_weights`` get new symmetric per-channel int8 weights; the others keep
    the template's weight tensors (a frozen frontend) and only their biases are re-scaled."""
    T = template.tensors
    conv_ops = [op for op in template.ops if op.name in ("CONV_2D", "DEPTHWISE_CONV_2D", "FULLY_CONNECTED")]
    if fresh_weights is None:
        fresh_weights = {op.index for op in conv_ops}
    # ---- calibration: ranges of every activation tensor over the representative samples
    lo: dict[int, float] = {}
    hi: dict[int, float] = {}
    n_seen = 0
    for sample in rep_data_gen():
        env = run_float(template, consts, sample[0])
        for ti, v in env.items():
            if T[ti].data is None and np.asarray(v).dtype.kind == "f":
                lo[ti] = min(lo.get(ti, np.inf), float(np.min(v)))
                hi[ti] = max(hi.get(ti, -np.inf), float(np.max(v)))
        n_seen += 1
    if n_seen == 0:
        raise ValueError("the representative dataset is empty")

    # tensors tied by data movement share one range (TFLite requires equal parameters across these operators)
    parent = {ti: ti for ti in lo}

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for op in template.ops:
        if op.name in _MOVERS:
            members = [k for k in (op.inputs if op.name == "CONCATENATION" else op.inputs[:1]) if k in parent] + [op.outputs[0]]
            for k in members[1:]:
                parent[find(k)] = find(members[0])
    glo: dict[int, float] = {}
    ghi: dict[int, float] = {}
    for ti in lo:
        r = find(ti)
        glo[r] = min(glo.get(r, np.inf), lo[ti])
        ghi[r] = max(ghi.get(r, -np.inf), hi[ti])

    new = copy.deepcopy(template)
    N = new.tensors
    act_q: dict[int, tuple[float, int]] = {}
    logistic_out = {op.outputs[0] for op in template.ops if op.name in ("LOGISTIC", "SOFTMAX")}  # fixed output parameters 1 / 256, -128
    float_io = {template.inputs[0], template.outputs[0]}
    fixed_roots = {find(ti) for ti in logistic_out if ti in parent}  # tensors tied to such an output by data movement (RESHAPE behind SOFTMAX) share it
    for ti in lo:
        if ti in float_io or not T[ti].is_quantized:
            continue
        s, z = (1.0 / 256.0, -128) if (ti in logistic_out or find(ti) in fixed_roots) else choose_activation_params(glo[find(ti)], ghi[find(ti)])
        act_q[ti] = (s, z)
        N[ti].scale = np.asarray([s], np.float32)
        N[ti].zero_point = np.asarray([z], np.int64)
    for op in template.ops:  # constants produced for a quantised data path (FILL values) follow the tensor they join
        if op.name == "FILL" and T[op.inputs[1]].is_quantized:
            tgt = next(o for o in template.ops if o.name == "CONCATENATION" and op.outputs[0] in o.inputs).outputs[0]
            s, z = act_q[tgt]
            val = float(_dequant(T[op.inputs[1]]).reshape(-1)[0])
            for k in (op.inputs[1], op.outputs[0]):
                N[k].scale, N[k].zero_point = np.asarray([s], np.float32), np.asarray([z], np.int64)
            N[op.inputs[1]].data = np.full(T[op.inputs[1]].data.shape, np.clip(np.round(val / s) + z, -128, 127), np.int8)

    # quantised scalar / vector constants of element-wise operators (the epsilon of the frontend's max normalisation): own range
    weight_like = {k for op in conv_ops for k in op.inputs[1:3]}
    for op in template.ops:
        if op.name in ("ADD", "MUL", "DIV"):
            for k in op.inputs:
                if T[k].data is not None and T[k].is_quantized and k not in weight_like and k in consts:
                    v = np.asarray(consts[k], np.float64)
                    sc, z = choose_activation_params(float(v.min()), float(v.max()))
                    N[k].scale, N[k].zero_point = np.asarray([sc], np.float32), np.asarray([z], np.int64)
                    N[k].data = np.clip(np.round(v / sc) + z, -128, 127).astype(np.int8).reshape(T[k].data.shape)

    # ---- weights and biases
    for op in conv_ops:
        w, b = float_wb[op.index]
        wt, bt = N[op.inputs[1]], N[op.inputs[2]]
        s_in = float(N[op.inputs[0]].scale[0])
        if op.index in fresh_weights:
            axis = 3 if op.name == "DEPTHWISE_CONV_2D" else 0
            q, sw = quantize_weights(w, axis, per_tensor)
            wt.data, wt.scale, wt.zero_point, wt.quantized_dimension = q, sw, np.zeros(sw.shape, np.int64), axis
        sw = wt.scale.astype(np.float64)
        sb = (np.float64(np.float32(s_in)) * sw).astype(np.float64)
        if sb.size == 1:
            sb = np.full(b.shape, sb.reshape(-1)[0])
        bt.data = np.clip(np.round(b.astype(np.float64) / sb), -(2**30), 2**30).astype(np.int32)  # the converter clamps biases to +-2^30
        bt.scale, bt.zero_point = sb.astype(np.float32), np.zeros(sb.shape, np.int64)
    return new




def convert_to_int8(spec: ns.NetSpec, template_path: str, rep_data_gen, per_tensor: bool = False) -> TfliteModel:
    """Counterpart of the reference's ``convert_to_tflite(model, rep_data_gen, ..., quantization='ptq', per_tensor)``: returns the
    in-memory INT8 graph (``lower_i8`` / ``HipRunner`` consume it; there is no flatbuffer writer)."""
    from birdnet_stm32.models._tflite_reader import load_tflite

    return requantize_like(load_tflite(template_path), spec, rep_data_gen, per_tensor=per_tensor)
