"""Lower a :class:`NetSpec` (Keras layer list) to the float32 device plan.

Fusions performed here, all algebraically exact up to float rounding:

* ``Conv2D/DepthwiseConv2D -> BatchNormalization`` becomes one convolution with
  ``w' = w * gamma / sqrt(var + eps)`` and ``b' = beta - mean * gamma / sqrt(var + eps)``
  (folded in float64, stored float32);
* a following ``Add`` (residual) and ``ReLU(max_value=6)`` ride in the pointwise kernel's epilogue;
* squeeze-excite (reference: birdnet_stm32/models/blocks.py:27-46) becomes a gate vector that
  the next pointwise convolution multiplies into its input channels (inverted-residual
  blocks) or an explicit scale pass (DS blocks, where the scaled tensor is also a residual);
* the hybrid frontend's mel mixer is stored band-sparse: every Slaney triangle touches only a
  few FFT bins (reference seeds the mixer from ``librosa.filters.mel`` and freezes it,
  birdnet_stm32/models/frontend.py:121-129,257-276); exact zeros are skipped, any other
  weight is kept, so a dense mixer degrades to full-length bands, never to a wrong result.

Layer semantics follow birdnet_stm32/models/dscnn.py:28-84,198-261 and blocks.py:49-133.
"""

from __future__ import annotations

import numpy as np

from birdnet_stm32.models import _netspec as ns
from birdnet_stm32.models import _pack as pk


def fold_bn(kernel: np.ndarray, bn: ns.Layer | None, cout_axis_last: bool = True):
    """Return (kernel', bias') in float32 with the BatchNorm folded in (float64 arithmetic)."""
    k = kernel.astype(np.float64)
    cout = k.shape[-1]
    if bn is None:
        return k.astype(np.float32), np.zeros(cout, np.float32)
    g = bn.weights["gamma"].astype(np.float64)
    b = bn.weights["beta"].astype(np.float64)
    mu = bn.weights["mean"].astype(np.float64)
    var = bn.weights["var"].astype(np.float64)
    s = g / np.sqrt(var + float(bn.attrs["eps"]))
    return (k * s).astype(np.float32), (b - mu * s).astype(np.float32)


def mel_bands(mel: np.ndarray, n_bins: int):
    """Band-sparse form of a ``[F_pad, M]`` mixer: (values, int32 [3, M] = start, len, offset)."""
    M = mel.shape[1]
    vals: list[np.ndarray] = []
    bands = np.zeros((3, M), np.int32)
    off = 0
    for m in range(M):
        col = mel[:n_bins, m]
        nz = np.nonzero(col)[0]
        if nz.size == 0:
            start, length = 0, 0
        else:
            start, length = int(nz[0]), int(nz[-1] - nz[0] + 1)
        bands[:, m] = (start, length, off)
        vals.append(col[start : start + length])
        off += length
    flat = np.concatenate(vals) if off else np.zeros(1, np.float32)
    return flat.astype(np.float32), bands


def mag_params(front: ns.Layer) -> np.ndarray:
    """Pack magnitude-scaling weights as rows of ``[NP][M]`` in the order the kernels index them."""
    w, M = front.weights, int(front.attrs["mel_bins"])
    kind = front.attrs.get("mag_scale", "none")
    if kind == "pwl":
        rows = [w["pwl_k0"], *w["pwl_k"], *w["pwl_w"], *w["pwl_b"]]
    elif kind == "pcen":
        rows = [w["pcen_agc"], w["pcen_k1"], w["pcen_sw"], w["pcen_sb"], w["pcen_k2"]]
    else:
        rows = [np.zeros(M, np.float32)]
    return np.stack([np.asarray(r, np.float32).reshape(M) for r in rows])


def pick_tile(oh: int, ow: int):
    """64 output positions per workgroup: a TH x TW tile of one chunk, or whole small maps of NB chunks."""
    if oh * ow >= 64:
        for th, tw in ((8, 8), (4, 16), (16, 4), (2, 32), (32, 2), (1, 64), (64, 1)):
            if oh % th == 0 and ow % tw == 0:
                return th, tw, 1
        return None
    if 64 % (oh * ow) == 0:
        return oh, ow, 64 // (oh * ow)
    return None


def pack_pw_fragments(w: np.ndarray) -> np.ndarray:
    """``[Cin][Cout]`` -> ``[Cin/16][Cout/16][64 lanes][4]``: lane (q = l >> 4, c = l & 15) of k-step j holds
    ``W[16 j + 4 q + e][16 ct + c]`` for e = 0..3 — the B operands of four consecutive ``mfma_f32_16x16x4``."""
    cin, cout = w.shape
    if cin % 16:  # zero rows up to a whole number of k-steps (the kernel zero-fills the matching tile columns)
        w = np.concatenate([w, np.zeros((-cin % 16, cout), w.dtype)], axis=0)
        cin = w.shape[0]
    t = w.reshape(cin // 16, 4, 4, cout // 16, 16)  # [j][q][e][ct][c]
    return np.ascontiguousarray(t.transpose(0, 3, 1, 4, 2)).reshape(cin // 16, cout // 16, 64, 4).astype(np.float32)


def _gap_dense_tail(layers, i, only_consumer):
    """Index of the classifier Dense if layer i (a GAP) is followed only by identities and that Dense, else None."""
    cur = i
    while True:
        nxt = only_consumer(layers[cur].name)
        if nxt is None:
            return None
        if layers[nxt].kind == ns.IDENTITY:
            cur = nxt
            continue
        if layers[nxt].kind == ns.DENSE and nxt == len(layers) - 1:
            return nxt
        return None


def lower_f32(spec: ns.NetSpec, keep_all: bool = False, fuse: bool = True) -> pk.Plan:
    """Build the float32 plan for ``spec``.  ``keep_all`` disables slot reuse (debug/tests); ``fuse=False`` keeps
    depthwise and pointwise convolutions as separate baseline kernels instead of the fused matrix-core block."""
    layers = spec.layers
    consumers: dict[str, list[int]] = {}
    index_of = {ly.name: i for i, ly in enumerate(layers)}
    for i, ly in enumerate(layers):
        for src in ly.inputs:
            consumers.setdefault(src, []).append(i)

    front = spec.frontend
    fa = front.attrs
    if fa["mode"] == "hybrid":
        in_kind, F, W = pk.INPUT_SPECTROGRAM, layers[0].out_shape[0], int(fa["spec_width"])
        in_elems = int(np.prod(layers[0].out_shape))
    elif fa["mode"] == "raw":
        in_kind, F, W = pk.INPUT_WAVEFORM, 0, int(fa["spec_width"])
        in_elems = int(fa["sample_rate"] * fa["chunk_duration"])
    elif fa["mode"] == "precomputed":
        # the graph passes a host-side (here: bn_mel_spectrogram) mel / log-mel / MFCC map through, cut to spec_width
        # (reference: models/frontend.py:296-297); the plan starts at the stem
        in_kind, F, W = pk.INPUT_MEL, 0, int(fa["spec_width"])
        if tuple(layers[0].out_shape[:2]) != (int(fa["mel_bins"]), W):
            raise NotImplementedError(f"precomputed input {layers[0].out_shape} wider than spec_width {W}")
        in_elems = int(np.prod(layers[0].out_shape))
    else:
        raise NotImplementedError(f"frontend mode {fa['mode']!r} is not lowered to HIP yet")

    plan = pk.Plan(pk.DTYPE_F32, in_kind, in_elems, F, W, spec.num_classes,
                   meta={"source": spec.meta.get("source", ""),
                         "frontend": {k: fa.get(k) for k in ("mode", "mel_bins", "spec_width", "sample_rate", "fft_length", "mag_scale")}})
    pb = pk.PlanBuilder(plan)
    val: dict[str, int] = {}  # layer name -> value id holding its output
    shape: dict[str, tuple] = {}
    gate_of: dict[str, tuple[int, int]] = {}  # multiply output name -> (value of x, value of gate)
    audio_raw: dict[str, dict] = {}  # frontend name -> un-normalised mel value + finalisation tensors (audio path, norm off)
    done: set[int] = set()

    def only_consumer(name: str):
        c = consumers.get(name, [])
        return c[0] if len(c) == 1 else None

    def chain_after(i: int, produced: str):
        """Follow BN / identity / Add / ReLU after layer i; return (bn, res_name, act, last_idx)."""
        bn = res = None
        act = "none"
        cur, j = produced, i
        nxt = only_consumer(cur)
        if nxt is not None and layers[nxt].kind == ns.BN:
            bn, cur, j = layers[nxt], layers[nxt].name, nxt
            nxt = only_consumer(cur)
        while nxt is not None and layers[nxt].kind == ns.IDENTITY:
            cur, j = layers[nxt].name, nxt
            nxt = only_consumer(cur)
        if nxt is not None and layers[nxt].kind == ns.ADD:
            other = [n for n in layers[nxt].inputs if n != cur]
            if len(other) == 1:
                res, cur, j = other[0], layers[nxt].name, nxt
                nxt = only_consumer(cur)
        if nxt is not None and layers[nxt].kind == ns.RELU:
            mv = layers[nxt].attrs.get("max_value")
            act = "relu6" if mv is not None and float(mv) == 6.0 else "relu"
            if mv is not None and float(mv) != 6.0:
                raise NotImplementedError("ReLU max_value other than 6")
            cur, j = layers[nxt].name, nxt
        return bn, res, act, cur

    for i, ly in enumerate(layers):
        if i in done:
            continue
        k = ly.kind
        if k == ns.INPUT:
            val[ly.name] = pk.SLOT_INPUT
            shape[ly.name] = ly.out_shape
        elif k == ns.FRONTEND and fa["mode"] == "precomputed":
            val[ly.name], shape[ly.name] = val[ly.inputs[0]], (int(fa["mel_bins"]), W, 1)
        elif k == ns.FRONTEND and fa["mode"] == "raw":
            M, T = int(fa["mel_bins"]), in_elems
            if M % 4:
                raise NotImplementedError("raw frontend needs a multiple of 4 filters")
            stride = -(-T // W)
            pad_total = max(0, stride * (W - 1) + 16 - T)
            g_ = ly.weights["fb_gamma"].astype(np.float64) / np.sqrt(ly.weights["fb_var"].astype(np.float64) + float(fa.get("fb_eps", 1e-3)))
            fb = (ly.weights["fb"].astype(np.float64) * g_).astype(np.float32)  # [16][M], BatchNorm folded
            fbias = (ly.weights["fb_beta"].astype(np.float64) - ly.weights["fb_mean"].astype(np.float64) * g_).astype(np.float32)
            mag = pk.MAG_CODES[fa.get("mag_scale", "none")]
            v = pb.value(M * W * 4)
            pb.op(pk.F32_RAWFE, val[ly.inputs[0]], v, p=[T, W, M, stride, pad_total // 2, mag],
                  t=[pb.tensor(fb, np.float32), pb.tensor(fbias, np.float32), pb.tensor(mag_params(ly), np.float32)], name=ly.name,
                  out_shape=(M, W, 1))
            val[ly.name], shape[ly.name] = v, (M, W, 1)
        elif k == ns.FRONTEND:
            M = int(fa["mel_bins"])
            wv, bands = mel_bands(ly.weights["mel"], F)
            t_w, t_b = pb.tensor(wv, np.float32), pb.tensor(bands, np.int32)
            t_m = pb.tensor(mag_params(ly), np.float32)
            mag, norm = pk.MAG_CODES[fa.get("mag_scale", "none")], int(bool(fa.get("norm", False)))
            v = pb.value(M * W * 4)
            both = pk.PATH_BOTH if not fuse else pk.PATH_INPUT
            pb.op(pk.F32_MEL, val[ly.inputs[0]], v, p=[F, W, M, mag, norm], t=[t_w, t_b, t_m],
                  name=ly.name if not norm else ly.name + ":mel", out_shape=(M, W, 1), path=both)
            if norm:
                pb.op(pk.F32_MAG, v, v, p=[M, W, mag], t=[-1, -1, t_m], name=ly.name, out_shape=(M, W, 1), path=both)
            if fuse and F == 257:
                # audio entry point: STFT with the mixer fused (no spectrogram in HBM), then normalise + scale
                wsum = ly.weights["mel"][:F].astype(np.float64).sum(axis=0).astype(np.float32)
                raw = pb.value(M * W * 4)
                pb.op(pk.F32_STFTMEL, pk.SLOT_AUDIO, raw, p=[0, W, M], t=[t_w, t_b], name=ly.name + ":melraw", out_shape=(M, W, 1),
                      path=pk.PATH_AUDIO)
                t_ws = pb.tensor(wsum, np.float32)
                nxt_i = only_consumer(ly.name)
                front_next = (not norm and nxt_i is not None and layers[nxt_i].kind == ns.CONV and tuple(layers[nxt_i].attrs["kernel"]) == (3, 3)
                              and tuple(layers[nxt_i].attrs["strides"]) == (1, 2) and int(layers[nxt_i].attrs["filters"]) == 16)
                if front_next:
                    audio_raw[ly.name] = {"value": raw, "wsum": t_ws, "magp": t_m, "mag": mag}  # the front block finalises while loading
                else:
                    pb.op(pk.F32_MELFIN, raw, v, p=[M, W, mag, norm], t=[t_ws, -1, t_m], name=ly.name, out_shape=(M, W, 1), path=pk.PATH_AUDIO)
            val[ly.name], shape[ly.name] = v, (M, W, 1)
        elif k in (ns.CONV, ns.DWCONV):
            src = ly.inputs[0]
            H, Wd, Cin = shape[src]
            kh, kw = ly.attrs["kernel"]
            sh, sw = ly.attrs["strides"]
            bn, res, act, last = chain_after(i, ly.name)
            OH, pt, _ = ns.same_pad(H, kh, sh)
            OW, pl, _ = ns.same_pad(Wd, kw, sw)
            if k == ns.DWCONV:
                if (kh, kw) != (3, 3) or res is not None:
                    raise NotImplementedError(f"{ly.name}: only 3x3 depthwise convolutions without residual")
                w, b = fold_bn(ly.weights["kernel"], bn)  # [3,3,C]
                C = Cin
                if C % 4:
                    raise NotImplementedError("channel counts must be multiples of 4")
                nxt_i = only_consumer(last)
                nxt = layers[nxt_i] if nxt_i is not None else None
                tile = pick_tile(OH, OW)
                fusable = (fuse and nxt is not None and nxt.kind == ns.CONV and tuple(nxt.attrs["kernel"]) == (1, 1)
                           and tuple(nxt.attrs["strides"]) == (1, 1) and tile is not None and C % 16 == 0
                           and int(nxt.attrs["filters"]) % 16 == 0)
                if fusable:
                    bn2, res2, act2, last2 = chain_after(nxt_i, nxt.name)
                    w2, b2 = fold_bn(nxt.weights["kernel"], bn2)
                    Cout = w2.shape[-1]
                    v = pb.value(OH * OW * Cout * 4)
                    p = [H, Wd, C, sh, sw, pk.ACT_CODES[act], OH, OW, pt, pl, Cout, pk.ACT_CODES[act2], int(res2 is not None), 0, 0, 1, *tile]
                    pb.op(pk.F32_DWPW, val[src], v, p=p,
                          t=[pb.tensor(w, np.float32), pb.tensor(b, np.float32), pb.tensor(pack_pw_fragments(w2[0, 0]), np.float32),
                             pb.tensor(b2, np.float32)],
                          in1=val[res2] if res2 is not None else pk.SLOT_NONE, name=last2, out_shape=(OH, OW, Cout))
                    out_shape = (OH, OW, Cout)
                    last = last2
                else:
                    v = pb.value(OH * OW * C * 4)
                    pb.op(pk.F32_DW, val[src], v, p=[H, Wd, C, sh, sw, pk.ACT_CODES[act], OH, OW, pt, pl],
                          t=[pb.tensor(w, np.float32), pb.tensor(b, np.float32)], name=last, out_shape=(OH, OW, C))
                    out_shape = (OH, OW, C)
            elif (kh, kw) == (3, 3):
                if Cin != 1 or res is not None:
                    raise NotImplementedError(f"{ly.name}: 3x3 convolutions are only lowered for the 1-channel stem")
                w, b = fold_bn(ly.weights["kernel"], bn)  # [3,3,1,Cout]
                Cout = w.shape[-1]
                # front block: stem -> depthwise stride 2 -> pointwise in one kernel (the stem activation stays in LDS)
                front = None
                dw_i = only_consumer(last)
                if fuse and dw_i is not None and layers[dw_i].kind == ns.DWCONV and (sh, sw) == (1, 2) and Cout == 16:
                    dwl = layers[dw_i]
                    bn_d, res_d, act_d, last_d = chain_after(dw_i, dwl.name)
                    pw_i = only_consumer(last_d)
                    if (tuple(dwl.attrs["strides"]) == (2, 2) and res_d is None and pw_i is not None and layers[pw_i].kind == ns.CONV
                            and tuple(layers[pw_i].attrs["kernel"]) == (1, 1) and int(layers[pw_i].attrs["filters"]) == 32):
                        pwl = layers[pw_i]
                        bn_p, res_p, act_p, last_p = chain_after(pw_i, pwl.name)
                        BH, BW = ns.same_pad(OH, 3, 2)[0], ns.same_pad(OW, 3, 2)[0]
                        if res_p is None and BH % 8 == 0 and BW % 8 == 0 and H == 2 * BH and Wd == 4 * BW:
                            front = (dwl, bn_d, act_d, pwl, bn_p, act_p, last_p, BH, BW)
                if front is not None:
                    dwl, bn_d, act_d, pwl, bn_p, act_p, last_p, BH, BW = front
                    wd, bd = fold_bn(dwl.weights["kernel"], bn_d)
                    wp_, bp = fold_bn(pwl.weights["kernel"], bn_p)
                    N = wp_.shape[-1]
                    v = pb.value(BH * BW * N * 4)
                    tens = [pb.tensor(w[:, :, 0, :], np.float32), pb.tensor(b, np.float32), pb.tensor(wd, np.float32), pb.tensor(bd, np.float32),
                            pb.tensor(pack_pw_fragments(wp_[0, 0]), np.float32), pb.tensor(bp, np.float32)]
                    base_p = [H, Wd, Cout, N, BH, BW, pk.ACT_CODES[act], pk.ACT_CODES[act_d], pk.ACT_CODES[act_p]]
                    raw = audio_raw.get(src)
                    if raw is not None:
                        # audio entry point: read the un-normalised mel energies and finalise them while loading the patch
                        pb.op(pk.F32_FRONT, val[src], v, p=base_p + [0, 0], t=tens, name=last_p, out_shape=(BH, BW, N), path=pk.PATH_INPUT)
                        pb.op(pk.F32_FRONT, raw["value"], v, p=base_p + [1, raw["mag"]], t=tens + [raw["wsum"], raw["magp"]], name=last_p,
                              out_shape=(BH, BW, N), path=pk.PATH_AUDIO)
                    else:
                        pb.op(pk.F32_FRONT, val[src], v, p=base_p + [0, 0], t=tens, name=last_p, out_shape=(BH, BW, N))
                    out_shape = (BH, BW, N)
                    last = last_p
                else:
                    raw = audio_raw.pop(src, None)
                    if raw is not None:  # no front block after all: finalise the mel energies in their own pass
                        Ms, Ws, _ = shape[src]
                        pb.op(pk.F32_MELFIN, raw["value"], val[src], p=[Ms, Ws, raw["mag"], 0], t=[raw["wsum"], -1, raw["magp"]], name=src,
                              out_shape=(Ms, Ws, 1), path=pk.PATH_AUDIO)
                    v = pb.value(OH * OW * Cout * 4)
                    pb.op(pk.F32_STEM, val[src], v, p=[H, Wd, Cout, sh, sw, pk.ACT_CODES[act], OH, OW, pt, pl],
                          t=[pb.tensor(w[:, :, 0, :], np.float32), pb.tensor(b, np.float32)], name=last, out_shape=(OH, OW, Cout))
                    out_shape = (OH, OW, Cout)
            elif (kh, kw) == (1, 1) and (sh, sw) == (1, 1):
                w, b = fold_bn(ly.weights["kernel"], bn)  # [1,1,Cin,Cout]
                Cout = w.shape[-1]
                P = H * Wd
                if src in gate_of:
                    x_val, gate_val = gate_of[src]
                else:
                    x_val, gate_val = val[src], None
                v = pb.value(P * Cout * 4)
                tile = pick_tile(H, Wd)
                if fuse and tile is not None and Cin % 4 == 0 and Cout % 16 == 0:
                    p = [H, Wd, Cin, 1, 1, 0, H, Wd, 0, 0, Cout, pk.ACT_CODES[act], int(res is not None), int(gate_val is not None),
                         gate_val if gate_val is not None else 0, 0, *tile]
                    zero = pb.tensor(np.zeros(4, np.float32), np.float32)
                    pb.op(pk.F32_DWPW, x_val, v, p=p, t=[zero, zero, pb.tensor(pack_pw_fragments(w[0, 0]), np.float32), pb.tensor(b, np.float32)],
                          in1=val[res] if res is not None else pk.SLOT_NONE, name=last, out_shape=(H, Wd, Cout),
                          value_params=(14,) if gate_val is not None else ())
                else:
                    p = [P, Cin, Cout, pk.ACT_CODES[act], int(res is not None), int(gate_val is not None),
                         gate_val if gate_val is not None else 0]
                    pb.op(pk.F32_PW, x_val, v, p=p, t=[pb.tensor(w[0, 0], np.float32), pb.tensor(b, np.float32)],
                          in1=val[res] if res is not None else pk.SLOT_NONE, name=last, out_shape=(H, Wd, Cout),
                          value_params=(6,) if gate_val is not None else ())
                out_shape = (H, Wd, Cout)
            else:
                raise NotImplementedError(f"{ly.name}: kernel {kh}x{kw} stride {sh}x{sw}")
            # every layer swallowed by the fusion aliases the fused output
            j = i
            names = [ly.name]
            cur = ly.name
            while cur != last:
                j = only_consumer(cur)
                cur = layers[j].name
                names.append(cur)
                done.add(j)
            for n in names:
                val[n], shape[n] = v, out_shape
        elif k == ns.GAP and ly.attrs.get("keepdims"):
            # squeeze-excite: GAP(keepdims) -> Dense(relu) -> Dense(sigmoid) -> Multiply
            x = ly.inputs[0]
            d1 = layers[only_consumer(ly.name)]
            d2 = layers[only_consumer(d1.name)]
            mul_i = only_consumer(d2.name)
            mul = layers[mul_i]
            if not (d1.kind == ns.DENSE and d2.kind == ns.DENSE and mul.kind == ns.MUL and x in mul.inputs):
                raise NotImplementedError("unrecognised squeeze-excite pattern")
            H, Wd, C = shape[x]
            Cr = int(d1.attrs["units"])
            g = pb.value(C * 4)
            pb.op(pk.F32_SEGATE, val[x], g, p=[H * Wd, C, Cr],
                  t=[pb.tensor(d1.weights["kernel"], np.float32), pb.tensor(d2.weights["kernel"], np.float32)],
                  name=d2.name, out_shape=(C,))
            done.update({index_of[d1.name], index_of[d2.name], mul_i})
            nxt = only_consumer(mul.name)
            if nxt is not None and layers[nxt].kind == ns.CONV and tuple(layers[nxt].attrs["kernel"]) == (1, 1):
                gate_of[mul.name] = (val[x], g)
                shape[mul.name] = shape[x]
            else:
                v = pb.value(H * Wd * C * 4)
                pb.op(pk.F32_SCALE, val[x], v, p=[H * Wd, C], in1=g, name=mul.name, out_shape=(H, Wd, C))
                val[mul.name], shape[mul.name] = v, shape[x]
        elif k == ns.GAP and fuse and _gap_dense_tail(layers, i, only_consumer) is not None:
            head_i = _gap_dense_tail(layers, i, only_consumer)
            head = layers[head_i]
            H, Wd, C = shape[ly.inputs[0]]
            cout = int(head.attrs["units"])
            act = {"linear": 0, "sigmoid": 1, "softmax": 2}[head.attrs.get("activation", "linear")]
            bias = head.weights.get("bias", np.zeros(cout, np.float32))
            pb.op(pk.F32_GAPDENSE, val[ly.inputs[0]], pk.SLOT_SCORES, p=[H * Wd, C, cout, act],
                  t=[pb.tensor(head.weights["kernel"], np.float32), pb.tensor(bias, np.float32)], name=head.name, out_shape=(cout,))
            done.update(range(i + 1, head_i + 1))
        elif k == ns.GAP:
            H, Wd, C = shape[ly.inputs[0]]
            v = pb.value(C * 4)
            pb.op(pk.F32_GAP, val[ly.inputs[0]], v, p=[H * Wd, C], name=ly.name, out_shape=(C,))
            val[ly.name], shape[ly.name] = v, (C,)
        elif k == ns.ATTNPOOL:
            H, Wd, C = shape[ly.inputs[0]]
            v = pb.value(C * 4)
            pb.op(pk.F32_ATTNPOOL, val[ly.inputs[0]], v, p=[H * Wd, C], t=[pb.tensor(ly.weights["score"], np.float32)],
                  name=ly.name, out_shape=(C,))
            val[ly.name], shape[ly.name] = v, (C,)
        elif k == ns.IDENTITY:
            val[ly.name], shape[ly.name] = val[ly.inputs[0]], shape[ly.inputs[0]]
        elif k == ns.DENSE:
            if ly is not layers[-1]:
                raise NotImplementedError("Dense layers are only lowered as the classifier head or inside squeeze-excite")
            (cin,) = shape[ly.inputs[0]]
            cout = int(ly.attrs["units"])
            act = {"linear": 0, "sigmoid": 1, "softmax": 2}[ly.attrs.get("activation", "linear")]
            bias = ly.weights.get("bias", np.zeros(cout, np.float32))
            pb.op(pk.F32_DENSE, val[ly.inputs[0]], pk.SLOT_SCORES, p=[cin, cout, act],
                  t=[pb.tensor(ly.weights["kernel"], np.float32), pb.tensor(bias, np.float32)], name=ly.name, out_shape=(cout,))
        else:
            raise NotImplementedError(f"layer {ly.name} of kind {k} cannot be lowered on its own")
    if fuse and not keep_all:
        _tag_front2(pb)
        _tag_pwdw(pb)
    return pb.finalize(reuse=not keep_all)


def _tag_pwdw(pb: pk.PlanBuilder) -> None:
    """Mark (expand 1x1, depthwise 3x3) pairs of inverted-residual blocks (reference models/blocks.py:88-110) that the library may run as
    ONE kernel (``f32_pwdw_kernel``): the depthwise stage directly follows, is the only reader of the expanded map, and nothing gates
    or adds to the expand convolution.  The fused kernel reads the block input while it writes the depthwise output: no slot sharing."""
    ops = pb.plan.ops
    gate_slots = {ops[oi].p[pi] for oi, pi in pb._gate_refs}
    for i in range(len(ops) - 1):
        e, d = ops[i], ops[i + 1]
        if not (e.kind == pk.F32_DWPW and e.p[15] == 0 and e.p[12] == 0 and e.p[13] == 0 and d.kind == pk.F32_DW and d.in0 == e.out and e.out >= 0):
            continue
        if e.p[pk.TAIL_TAG] or d.p[pk.TAIL_TAG] or e.p[pk.OP_PATH] != d.p[pk.OP_PATH]:
            continue
        v = e.out
        readers = [k for k, r in enumerate(ops) if k != i + 1 and (r.in0 == v or r.in1 == v)]
        writers = [k for k, r in enumerate(ops) if r.out == v]
        if readers or writers != [i] or v in gate_slots or d.p[2] != e.p[10] or (d.p[0], d.p[1]) != (e.p[6], e.p[7]):
            continue
        e.p[pk.TAIL_TAG] = pk.PWDW_HEAD
        d.p[pk.TAIL_TAG] = pk.PWDW_COVERED
        if e.in0 >= 0:
            pb._extra_uses.append((i + 1, e.in0))
        # a stem convolution right in front whose map nothing else reads: the fused kernel computes its rows too
        if i and ops[i - 1].kind == pk.F32_STEM and ops[i - 1].out == e.in0 and e.in0 >= 0 and not ops[i - 1].p[pk.TAIL_TAG]:
            st = ops[i - 1]
            others = [k for k, r in enumerate(ops) if k != i and (r.in0 == e.in0 or r.in1 == e.in0)]
            if not others and [k for k, r in enumerate(ops) if r.out == e.in0] == [i - 1] and e.in0 not in gate_slots \
                    and (st.p[6], st.p[7], st.p[2]) == (e.p[0], e.p[1], e.p[2]) and st.p[pk.OP_PATH] == e.p[pk.OP_PATH]:
                st.p[pk.TAIL_TAG] = pk.PWDW_STEM
                if st.in0 >= 0:
                    pb._extra_uses.append((i + 1, st.in0))


def _tag_front2(pb: pk.PlanBuilder) -> None:
    """Mark (front block, residual block 32 -> 32 behind it) pairs the library may run as ONE kernel (``f32_front2_kernel``).

    The pair qualifies when the residual block (reference: the second ``ds_conv_block`` of stage 1, ``models/dscnn.py:209-226``)
    reads nothing but the front block's output — as its input and as its residual — and no other operator reads that map:
    the fused kernel keeps it in LDS and never writes it.  Debug plans (``keep_all``) are left alone, every tensor stays readable.
    """
    ops = pb.plan.ops
    gate_slots = {ops[oi].p[pi] for oi, pi in pb._gate_refs}
    for i, o in enumerate(ops):
        q = o.p
        if not (o.kind == pk.F32_DWPW and q[15] == 1 and q[2] == 32 and q[10] == 32 and q[3] == 1 and q[4] == 1 and q[12] == 1
                and q[13] == 0 and o.in0 == o.in1 and o.in0 >= 0):
            continue
        v = o.in0
        fronts = [j for j in range(i) if ops[j].kind == pk.F32_FRONT and ops[j].out == v]
        readers = [k for k, r in enumerate(ops) if k != i and (r.in0 == v or r.in1 == v)]
        writers = [k for k, r in enumerate(ops) if r.out == v]
        if not fronts or readers or writers != fronts or v in gate_slots:
            continue
        if any(ops[j].p[2] != 16 or ops[j].p[3] != 32 or ops[j].p[4] != q[0] or ops[j].p[5] != q[1] or q[1] % 16 or q[1] // 16 > 4
               or i - j >= 8 for j in fronts):
            continue
        for j in fronts:
            ops[j].p[pk.TAIL_TAG] = pk.FRONT2_HEAD
            ops[j].p[pk.FRONT2_DIST] = i - j
            if ops[j].in0 >= 0:  # the fused kernel reads the front block's input while it writes this block's output: no slot sharing
                pb._extra_uses.append((i, ops[j].in0))
        q[pk.TAIL_TAG] = pk.FRONT2_COVERED
