"""Own INT8 exporter for models built by current reference code (squeeze-excite, inverted residuals, softmax head) — without a
template graph and without TensorFlow.

The reference converts a Keras model with ``tf.lite.TFLiteConverter`` (reference: birdnet_stm32/conversion/quantize.py:111-168:
``Optimize.DEFAULT``, representative dataset, ``TFLITE_BUILTINS_INT8``, float32 input / output, per-channel weights).  Here the
float model is a :class:`NetSpec` (``build_model('dscnn', ...)`` or a loaded ``.keras`` archive); :func:`netspec_to_graph` writes
it down as a TFLite operator graph, :func:`convert_netspec_to_int8` calibrates it on representative samples and quantises it with
the converter's rules (``conversion/quantize.py: quantize_graph``), and ``models/_tflite_writer.write_tflite`` serialises the result
as an ordinary `.tflite` file that ``load_model_runner`` reads back.

Operator vocabulary of the emitted graphs (all int8 between QUANTIZE and DEQUANTIZE):

* frontend (raw): QUANTIZE of the waveform -> [PAD] -> RESHAPE [1, 1, T, 1] -> CONV_2D 1x16, stride ceil(T / W), VALID, BatchNorm folded,
  ReLU6 (reference models/frontend.py:138-164,347-358) -> magnitude scaling as below -> TRANSPOSE;
* frontend (hybrid): QUANTIZE -> TRANSPOSE -> CONV_2D 1x1 (mel mixer, ReLU) -> [per-sample max normalisation of current hybrid
  frontends: REDUCE_MAX -> ADD 1e-6 -> DIV, reference models/frontend.py:338-342] -> element-wise magnitude scaling as 1x1
  DEPTHWISE_CONV_2D / ADD operators (PWL: ``k0 x + sum_i k_i relu(w_i x + b_i)``, reference models/magnitude.py:179-192; PCEN:
  ``relu(k1 y0 + k2 relu(w y0 + b))`` with ``y0 = relu((1 - a) x)``, :166-177) -> TRANSPOSE;
* backbone: CONV_2D / DEPTHWISE_CONV_2D with BatchNorm folded and ReLU6 fused (reference models/dscnn.py:28-84,198-246,
  models/blocks.py:49-133), residual ADD (fused ReLU6 in DS blocks, none in inverted-residual blocks), squeeze-excite as
  MEAN(keep_dims) -> FULLY_CONNECTED(ReLU) -> FULLY_CONNECTED -> LOGISTIC -> MUL (reference models/blocks.py:27-46);
* head: MEAN -> FULLY_CONNECTED -> LOGISTIC -> DEQUANTIZE (sigmoid) or FULLY_CONNECTED -> DEQUANTIZE -> SOFTMAX in float32
  (softmax; the TFLite converter would keep the softmax in int8 — a deliberate difference: the float softmax is exact on the
  dequantised logits and needs no fixed-point exponential).

* attention pooling (reference models/blocks.py:136-159): RESHAPE -> FULLY_CONNECTED (C -> 1 per position) -> SOFTMAX over the positions (int8) ->
  MUL -> SUM.

Not emitted (``NotImplementedError``): precomputed frontends (their graphs start at the stem: nothing to quantise in front).
"""

from __future__ import annotations

import numpy as np

from birdnet_stm32.models import _netspec as ns
from birdnet_stm32.models._tflite_reader import TfliteModel, TfliteOp, TfliteTensor
from birdnet_stm32.models._tflite_writer import _CODE_OF, _VERSION

_Q = (np.ones(1, np.float32), np.zeros(1, np.int64))  # placeholder quantisation of an int8 activation (set by quantize_graph)
_NOQ = (np.zeros(0, np.float32), np.zeros(0, np.int64))


class _GraphBuilder:
    def __init__(self):
        self.tensors: list[TfliteTensor] = []
        self.ops: list[TfliteOp] = []
        self.consts: dict[int, np.ndarray] = {}      # real-valued constants for the float calibration run
        self.float_wb: dict[int, tuple[np.ndarray, np.ndarray]] = {}

    def act(self, name: str, shape, quantized: bool = True) -> int:
        sc, zp = _Q if quantized else _NOQ
        t = TfliteTensor(len(self.tensors), name, tuple(int(v) for v in shape), np.dtype(np.int8 if quantized else np.float32), sc.copy(), zp.copy(), 0, None)
        self.tensors.append(t)
        return t.index

    def const(self, name: str, arr: np.ndarray, dtype, quantized: bool, real: np.ndarray | None = None) -> int:
        a = np.asarray(arr)
        sc, zp = _Q if quantized else _NOQ
        t = TfliteTensor(len(self.tensors), name, tuple(a.shape), np.dtype(dtype), sc.copy(), zp.copy(), 0, np.zeros(a.shape, dtype) if quantized else a.astype(dtype))
        self.tensors.append(t)
        self.consts[t.index] = np.asarray(real if real is not None else a)
        return t.index

    def op(self, name: str, inputs: list[int], out: int, options: dict | None = None) -> int:
        o = TfliteOp(len(self.ops), _CODE_OF[name], name, _VERSION.get(name, 1), list(inputs), [out], dict(options or {}))
        self.ops.append(o)
        return o.index

    def conv(self, kind: str, src: int, name: str, w: np.ndarray, b: np.ndarray, out_shape, options: dict) -> int:
        """CONV_2D (w [Cout, kh, kw, Cin]) / DEPTHWISE_CONV_2D (w [1, kh, kw, C]) / FULLY_CONNECTED (w [out, in])."""
        wi = self.const(name + "/w", w, np.int8, True, real=w.astype(np.float32))
        bi = self.const(name + "/b", b, np.int32, True, real=b.astype(np.float32))
        y = self.act(name, out_shape)
        oi = self.op(kind, [src, wi, bi], y, options)
        self.float_wb[oi] = (w.astype(np.float32), b.astype(np.float32))
        return y


def _conv_opts(stride=(1, 1), act="none", depthwise=False) -> dict:
    o = {"padding": "SAME", "stride_w": int(stride[1]), "stride_h": int(stride[0]), "activation": act, "dilation_w": 1, "dilation_h": 1}
    if depthwise:
        o["depth_multiplier"] = 1
    return o


def netspec_to_graph(spec: ns.NetSpec, frontend_norm: bool | None = None):
    """Write ``spec`` down as a TFLite operator graph with placeholder quantisation: ``(TfliteModel, consts, float_wb)`` as
    ``quantize_graph`` takes them (consts: real-valued constants; float_wb: real-valued kernel / bias per convolution operator)."""
    from birdnet_stm32.models._lower_f32 import fold_bn

    fe = spec.frontend
    fa = fe.attrs
    if fa["mode"] not in ("hybrid", "raw"):
        raise NotImplementedError(f"own INT8 export covers the hybrid and raw frontends, not '{fa['mode']}'")
    norm = fa.get("norm", False) if frontend_norm is None else frontend_norm
    g = _GraphBuilder()
    W = int(fa["spec_width"])
    M = int(fa["mel_bins"])
    perm = g.const("perm_0321", np.array([0, 3, 2, 1], np.int32), np.int32, False)
    if fa["mode"] == "raw":
        # reference models/frontend.py:347-358: [B, T, 1] -> symmetric zero pad -> expand_dims -> Conv2D (1 x 16, stride ceil(T / W), VALID,
        # no bias) -> BatchNorm -> ReLU6; the converter writes expand_dims as RESHAPE and folds the BatchNorm into the convolution
        T = int(fa["sample_rate"] * fa["chunk_duration"])
        stride = -(-T // W)
        pad_total = max(0, stride * (W - 1) + 16 - T)
        x_in = g.act(spec.layers[0].name, (1, T, 1), quantized=False)
        q = g.act("quantized_input", (1, T, 1))
        g.op("QUANTIZE", [x_in], q)
        cur = q
        if pad_total:
            pads = g.const("frontend/paddings", np.array([[0, 0], [pad_total // 2, pad_total - pad_total // 2], [0, 0]], np.int32), np.int32, False)
            xp = g.act("frontend/pad", (1, T + pad_total, 1))
            g.op("PAD", [cur, pads], xp)
            cur = xp
        shp = g.const("frontend/expand_shape", np.array([1, 1, T + pad_total, 1], np.int32), np.int32, False)
        xr = g.act("frontend/expand_dims", (1, 1, T + pad_total, 1))
        g.op("RESHAPE", [cur, shp], xr)
        fwt = fe.weights
        g_ = fwt["fb_gamma"].astype(np.float64) / np.sqrt(fwt["fb_var"].astype(np.float64) + float(fa.get("fb_eps", 1e-3)))
        fb = (fwt["fb"].astype(np.float64) * g_).astype(np.float32)                      # [16, M], BatchNorm folded
        fbias = (fwt["fb_beta"].astype(np.float64) - fwt["fb_mean"].astype(np.float64) * g_).astype(np.float32)
        opts = _conv_opts((1, stride), "relu6")
        opts["padding"] = "VALID"
        y = g.conv("CONV_2D", xr, "frontend/raw_fb2d", np.transpose(fb, (1, 0)).reshape(M, 1, 16, 1), fbias, (1, 1, W, M), opts)
        norm = False
    else:
        F = spec.layers[0].out_shape[0]
        x_in = g.act(spec.layers[0].name, (1, F, W, 1), quantized=False)
        q = g.act("quantized_input", (1, F, W, 1))
        g.op("QUANTIZE", [x_in], q)
        xt = g.act("frontend/transpose_in", (1, 1, W, F))
        g.op("TRANSPOSE", [q, perm], xt)
        mel = np.asarray(fe.weights["mel"], np.float32)          # [F (padded), M]
        w_mel = np.transpose(mel[:F], (1, 0)).reshape(M, 1, 1, F)
        y = g.conv("CONV_2D", xt, "frontend/mel_mixer", w_mel, np.zeros(M, np.float32), (1, 1, W, M), _conv_opts(act="relu"))
    zeros = np.zeros(M, np.float32)

    def dw1(src, name, w, b=zeros, act="none"):
        return g.conv("DEPTHWISE_CONV_2D", src, name, np.asarray(w, np.float32).reshape(1, 1, 1, M), np.asarray(b, np.float32), (1, 1, W, M),
                      _conv_opts(act=act, depthwise=True))

    def add(a, b, name, act="none", shape=(1, 1, W, M)):
        o = g.act(name, shape)
        g.op("ADD", [a, b], o, {"activation": act})
        return o

    if norm:
        # per-sample normalisation to [0, 1] (reference models/frontend.py:338-342: y / (reduce_max(y, [1, 2, 3]) + 1e-6)) as the
        # three operators the TFLite converter writes for it: REDUCE_MAX (keep_dims) -> ADD epsilon -> DIV with the scalar broadcast
        axes = g.const("frontend/norm_axes", np.array([1, 2, 3], np.int32), np.int32, False)
        ymax = g.act("frontend/norm_max", (1, 1, 1, 1))
        g.op("REDUCE_MAX", [y, axes], ymax, {"keep_dims": True})
        eps = g.const("frontend/norm_eps", np.full((1, 1, 1, 1), 1e-6, np.float32), np.int8, True, real=np.full((1, 1, 1, 1), 1e-6, np.float32))
        den = add(ymax, eps, "frontend/norm_den", shape=(1, 1, 1, 1))
        yn = g.act("frontend/norm", (1, 1, W, M))
        g.op("DIV", [y, den], yn, {"activation": "none"})
        y = yn
    mag = fa.get("mag_scale", "none")
    fw = fe.weights
    if mag == "pwl":
        acc = dw1(y, "frontend/pwl_k0", fw["pwl_k0"])
        for i in range(3):
            r = dw1(y, f"frontend/pwl_relu{i + 1}", fw["pwl_w"][i], fw["pwl_b"][i], act="relu")
            acc = add(acc, dw1(r, f"frontend/pwl_k{i + 1}", fw["pwl_k"][i]), f"frontend/pwl_sum{i + 1}")
        y = acc
    elif mag == "pcen":
        y0 = dw1(y, "frontend/pcen_agc", 1.0 - np.asarray(fw["pcen_agc"], np.float32), act="relu")
        t = dw1(y0, "frontend/pcen_inner", fw["pcen_sw"], fw["pcen_sb"], act="relu")
        y = add(dw1(y0, "frontend/pcen_k1", fw["pcen_k1"]), dw1(t, "frontend/pcen_k2", fw["pcen_k2"]), "frontend/pcen_out", act="relu")
    elif mag != "none":
        raise NotImplementedError(f"mag_scale '{mag}' in the INT8 export")
    yt = g.act(fe.name, (1, M, W, 1))
    g.op("TRANSPOSE", [y, perm], yt)

    # ---- backbone: walk the layer list behind the frontend ------------------------------------------------------------------
    by_input: dict[str, list[ns.Layer]] = {}
    for ly in spec.layers:
        for src in ly.inputs:
            by_input.setdefault(src, []).append(ly)
    val: dict[str, int] = {fe.name: yt}
    shape: dict[str, tuple] = {fe.name: (1, M, W, 1)}
    done: set[str] = set()
    axes12 = g.const("axes_hw", np.array([1, 2], np.int32), np.int32, False)
    layers = spec.layers[spec.layers.index(fe) + 1:]

    def sole_next(name: str, kind: str):
        nx = by_input.get(name, [])
        return nx[0] if len(nx) == 1 and nx[0].kind == kind else None

    def fused_act(ly: ns.Layer):
        """(activation name, last layer of the fused chain, identity layers skipped on the way) when a ReLU(6) is the only consumer of
        `ly` (Dropout layers in between are identities at inference)."""
        cur, skipped = ly, []
        while True:
            nx = by_input.get(cur.name, [])
            if len(nx) == 1 and nx[0].kind == ns.IDENTITY:
                skipped.append(nx[0])
                cur = nx[0]
                continue
            break
        r = nx[0] if len(nx) == 1 and nx[0].kind == ns.RELU else None
        if r is None:
            return "none", ly, []
        mv = r.attrs.get("max_value")
        return ("relu6" if mv == 6 else "relu" if mv is None else None), r, skipped

    for ly in layers:
        if ly.name in done:
            continue
        src = ly.inputs[0] if ly.inputs else None
        if ly.kind in (ns.CONV, ns.DWCONV):
            bn = sole_next(ly.name, ns.BN)
            k, b = fold_bn(ly.weights["kernel"], bn)
            if "bias" in ly.weights:
                b = b + np.asarray(ly.weights["bias"], np.float32)
            last = bn if bn is not None else ly
            act, end, skipped = fused_act(last)
            if act is None:
                raise NotImplementedError(f"{ly.name}: activation {end.attrs}")
            _, H, Wd, Cin = shape[src]
            sh, sw = ly.attrs["strides"]
            kh, kw = ly.attrs["kernel"]
            OH, OW = ns.same_pad(H, kh, sh)[0], ns.same_pad(Wd, kw, sw)[0]
            if ly.kind == ns.CONV:
                w = np.transpose(np.asarray(k, np.float32), (3, 0, 1, 2))
                out = g.conv("CONV_2D", val[src], ly.name, w, b, (1, OH, OW, w.shape[0]), _conv_opts((sh, sw), act))
                oshape = (1, OH, OW, w.shape[0])
            else:
                w = np.asarray(k, np.float32).reshape(kh, kw, -1)[None]
                out = g.conv("DEPTHWISE_CONV_2D", val[src], ly.name, w, b, (1, OH, OW, Cin), _conv_opts((sh, sw), act, depthwise=True))
                oshape = (1, OH, OW, Cin)
            for l2 in (bn, end, *skipped):
                if l2 is not None:
                    done.add(l2.name)
                    val[l2.name], shape[l2.name] = out, oshape
            val[ly.name], shape[ly.name] = out, oshape
        elif ly.kind == ns.IDENTITY:
            val[ly.name], shape[ly.name] = val[src], shape[src]
        elif ly.kind == ns.ADD:
            act, end, skipped = fused_act(ly)
            if act is None:
                raise NotImplementedError(f"{ly.name}: activation after the ADD")
            o = g.act(ly.name, shape[ly.inputs[0]])
            g.op("ADD", [val[ly.inputs[0]], val[ly.inputs[1]]], o, {"activation": act})
            val[ly.name], shape[ly.name] = o, shape[ly.inputs[0]]
            for l2 in ([end] if end is not ly else []) + skipped:
                done.add(l2.name)
                val[l2.name], shape[l2.name] = o, shape[ly.name]
        elif ly.kind == ns.RELU:
            raise NotImplementedError(f"{ly.name}: a ReLU that is not fused into its producer")
        elif ly.kind == ns.GAP:
            shp = shape[src]
            keep = bool(ly.attrs.get("keepdims"))
            oshape = (1, 1, 1, shp[3]) if keep else (1, shp[3])
            o = g.act(ly.name, oshape)
            g.op("MEAN", [val[src], axes12], o, {"keep_dims": keep})
            val[ly.name], shape[ly.name] = o, oshape
        elif ly.kind == ns.DENSE:
            shp = shape[src]
            kern = np.asarray(ly.weights["kernel"], np.float32)           # [in, out]
            bias = np.asarray(ly.weights["bias"], np.float32) if "bias" in ly.weights else np.zeros(kern.shape[1], np.float32)
            a = ly.attrs.get("activation", "linear")
            oshape = (*shp[:-1], kern.shape[1])
            is_last = ly is spec.layers[-1]
            fc = g.conv("FULLY_CONNECTED", val[src], ly.name + ("/logits" if a in ("sigmoid", "softmax") else ""), kern.T.copy(), bias, oshape,
                        {"activation": "relu" if a == "relu" else "none", "keep_num_dims": len(shp) > 2})
            out = fc
            if a == "sigmoid":
                out = g.act(ly.name + ("/sigmoid" if is_last else ""), oshape)
                g.op("LOGISTIC", [fc], out)
            elif a == "softmax":
                if not is_last:
                    raise NotImplementedError(f"{ly.name}: softmax inside the graph")
            elif a not in ("relu", "linear", None):
                raise NotImplementedError(f"{ly.name}: Dense activation {a}")
            val[ly.name], shape[ly.name] = out, oshape
            if is_last:
                deq = g.act(ly.name + "/dequantized", oshape, quantized=False)
                g.op("DEQUANTIZE", [out], deq)
                final = deq
                if a == "softmax":
                    final = g.act(ly.name, oshape, quantized=False)
                    g.op("SOFTMAX", [deq], final, {"beta": 1.0})
                model_out = final
        elif ly.kind == ns.ATTNPOOL:
            # attention pooling (reference models/blocks.py:136-159): Dense(1, no bias) over the channels of every position -> softmax over the
            # positions -> weighted sum.  As the converter writes it: RESHAPE [1, P, C] -> FULLY_CONNECTED (keep_num_dims) -> RESHAPE [1, P] ->
            # SOFTMAX (int8, 1 / 256) -> RESHAPE [1, P, 1] -> MUL (broadcast over the channels) -> SUM over the positions
            _, H, Wd, C = shape[src]
            P = H * Wd
            flat = g.act(ly.name + "/flat", (1, P, C))
            g.op("RESHAPE", [val[src], g.const(ly.name + "/flat_shape", np.array([1, P, C], np.int32), np.int32, False)], flat)
            score = g.conv("FULLY_CONNECTED", flat, ly.name + "/score", np.asarray(ly.weights["score"], np.float32).reshape(1, C), np.zeros(1, np.float32), (1, P, 1),
                           {"activation": "none", "keep_num_dims": True})
            row = g.act(ly.name + "/score_row", (1, P))
            g.op("RESHAPE", [score, g.const(ly.name + "/row_shape", np.array([1, P], np.int32), np.int32, False)], row)
            attn = g.act(ly.name + "/softmax", (1, P))
            g.op("SOFTMAX", [row], attn, {"beta": 1.0})
            col = g.act(ly.name + "/attn", (1, P, 1))
            g.op("RESHAPE", [attn, g.const(ly.name + "/col_shape", np.array([1, P, 1], np.int32), np.int32, False)], col)
            wmap = g.act(ly.name + "/weighted", (1, P, C))
            g.op("MUL", [flat, col], wmap, {"activation": "none"})
            o = g.act(ly.name, (1, C))
            g.op("SUM", [wmap, g.const(ly.name + "/axis", np.array([1], np.int32), np.int32, False)], o, {"keep_dims": False})
            val[ly.name], shape[ly.name] = o, (1, C)
        elif ly.kind == ns.MUL:
            o = g.act(ly.name, shape[ly.inputs[0]])
            g.op("MUL", [val[ly.inputs[0]], val[ly.inputs[1]]], o, {"activation": "none"})
            val[ly.name], shape[ly.name] = o, shape[ly.inputs[0]]
        elif ly.kind == ns.BN:
            raise NotImplementedError(f"{ly.name}: BatchNorm that does not follow a convolution")
        else:
            raise NotImplementedError(f"{ly.name}: layer kind {ly.kind!r} has no INT8 export (nested frontends)")
    model = TfliteModel(3, "birdnet_stm32.conversion.export (own PTQ, no TensorFlow)", g.tensors, g.ops, [x_in], [model_out])
    return model, g.consts, g.float_wb


def convert_netspec_to_int8(spec: ns.NetSpec, rep_data_gen, per_tensor: bool = False, frontend_norm: bool | None = None) -> TfliteModel:
    """Counterpart of the reference's ``convert_to_tflite(model, rep_data_gen, ..., quantization='ptq', per_tensor)`` for any
    hybrid-frontend DS-CNN this build can express (squeeze-excite, inverted residuals, embedding conv, sigmoid / softmax head)."""
    from birdnet_stm32.conversion.quantize import quantize_graph

    graph, consts, float_wb = netspec_to_graph(spec, frontend_norm=frontend_norm)
    return quantize_graph(graph, consts, float_wb, rep_data_gen, per_tensor=per_tensor)
