"""Lower a decoded INT8 `.tflite` graph to the INT8 device plan.

The reference executes the flatbuffer operator by operator through ``tf.lite.Interpreter``
(reference: birdnet_stm32/models/runners.py:51-95).  The graphs its converter emits for
hybrid-frontend DS-CNNs (reference: birdnet_stm32/conversion/quantize.py:131-152) all have
the shape decoded in SURVEY.md Appendix B:

    QUANTIZE -> TRANSPOSE -> STRIDED_SLICE -> [SHAPE, STRIDED_SLICE, PACK, FILL, CONCATENATION]
    -> CONV_2D (mel mixer, ReLU) -> element-wise PWL sub-graph (1x1 DEPTHWISE_CONV_2D / ADD)
    -> TRANSPOSE -> STRIDED_SLICE -> CONV_2D 3x3 (stem)
    -> { DEPTHWISE_CONV_2D 3x3 -> CONV_2D 1x1 [-> ADD residual] } * n
    -> MEAN -> FULLY_CONNECTED -> LOGISTIC -> DEQUANTIZE

This pass walks that shape, checks every assumption it relies on (tensor shapes, permutations,
quantisation of both sides of data-movement operators) and raises ``NotImplementedError`` on
anything else.  Data movement is absorbed into kernel addressing; the PWL sub-graph is
evaluated once for all 256 int8 inputs per channel and stored as a table, which is exact.
"""

from __future__ import annotations

import numpy as np

from birdnet_stm32.models import _pack as pk
from birdnet_stm32.models import _quant as qz
from birdnet_stm32.models._netspec import same_pad

K_ALIGN = 64  # int8 MFMA contracts 64 channels per instruction


class _Graph:
    def __init__(self, model):
        self.m = model
        self.t = model.tensors
        self.consumers: dict[int, list[int]] = {}
        for op in model.ops:
            for ti in op.inputs:
                if ti >= 0:
                    self.consumers.setdefault(ti, []).append(op.index)

    def q(self, ti: int) -> tuple[float, int]:
        t = self.t[ti]
        if t.scale.size != 1:
            raise NotImplementedError(f"tensor {ti} ({t.name}) is not per-tensor quantised")
        return float(t.scale[0]), int(t.zero_point[0])

    def const(self, ti: int) -> np.ndarray:
        d = self.t[ti].data
        if d is None:
            raise NotImplementedError(f"tensor {ti} ({self.t[ti].name}) is expected to be constant")
        return d


def _expect(cond: bool, what: str) -> None:
    if not cond:
        raise NotImplementedError(f"unsupported INT8 graph: {what}")


def _expect_acc_range(w, folded_bias, axes, what: str, mult=None, shift=None) -> None:
    """Domain proof of the device's fast requantisation forms (csrc/bn_requant.h: ``(x*m + 2^30) >> 31`` on a 64-bit product, then
    ``(v + 2^(e-1) + sign) >> e`` in 32 bits): every int32 accumulator ``x = sum(q * w) + folded_bias`` over int8 inputs ``q`` must fit
    int32, and per channel ``|x| * m / 2^31 + 2^(e-1) + 1`` must stay below 2^31 (the folded bias already holds ``-zp * sum(w)``).
    Channels with a negative multiplier or a left shift take the literal reference path on the device, which only needs int32 inputs."""
    w64 = np.asarray(w, np.int64)
    b = np.asarray(folded_bias, np.int64)
    lo = np.minimum(-128 * w64, 127 * w64).sum(axis=axes) + b
    hi = np.maximum(-128 * w64, 127 * w64).sum(axis=axes) + b
    top = np.maximum(np.abs(lo), np.abs(hi))
    _expect(int(top.max()) < 2**31, f"{what}: accumulators overflow int32")
    if mult is not None:
        m, e = np.asarray(mult, np.int64), -np.asarray(shift, np.int64)
        fast = (m >= 0) & (e >= 0)
        half = np.where(e > 0, np.int64(1) << np.maximum(e - 1, 0), 0)
        _expect(bool((((top * m) >> 31) + half + 1 < 2**31)[fast].all()), f"{what}: the rounding shift of the requantisation can overflow int32")


def _pwl_table(g: _Graph, ops: list, src: int, dst: int, channels: int) -> np.ndarray:
    """Evaluate the element-wise sub-graph ``ops`` for every int8 value of ``src``: int8 [C][256]."""
    env = {src: np.repeat(np.arange(-128, 128, dtype=np.int64)[:, None], channels, axis=1)}  # [256, C]

    def value(ti: int) -> np.ndarray:
        if ti in env:
            return env[ti]
        return g.const(ti).astype(np.int64).reshape(1, -1)  # per-channel constant, broadcast over the 256 rows

    for op in ops:
        if op.name == "DEPTHWISE_CONV_2D":
            w = g.t[op.inputs[1]]
            _expect(tuple(w.shape[1:3]) == (1, 1) and op.options["stride_w"] == 1 and op.options["stride_h"] == 1, "PWL depthwise must be 1x1")
            s_in, z_in = g.q(op.inputs[0])
            s_out, z_out = g.q(op.outputs[0])
            mult, shift = qz.channel_multipliers(s_in, w.scale, s_out, channels)
            bias = g.const(op.inputs[2]).astype(np.int64) if len(op.inputs) > 2 and op.inputs[2] >= 0 else 0
            acc = (value(op.inputs[0]) - z_in) * w.data.astype(np.int64).reshape(1, channels) + bias
            lo, hi = qz.activation_bounds(op.options["activation"], s_out, z_out)
            env[op.outputs[0]] = np.clip(qz.requantize(acc, mult, shift) + z_out, lo, hi)
        elif op.name == "ADD":
            s1, z1 = g.q(op.inputs[0])
            s2, z2 = g.q(op.inputs[1])
            so, zo = g.q(op.outputs[0])
            ap = qz.AddParams(s1, z1, s2, z2, so, zo, op.options["activation"])
            a, b = np.broadcast_arrays(value(op.inputs[0]), value(op.inputs[1]))
            env[op.outputs[0]] = ap.apply(a, b)
        else:
            _expect(False, f"{op.name} inside the element-wise frontend region")
    return env[dst].T.astype(np.int8).copy()  # [C][256]


def pack_i8_fragments(w: np.ndarray) -> np.ndarray:
    """``[Cout][Cin]`` int8 -> ``[Kp/64][Cout/16][64 lanes][16]``: lane (q = l >> 4, c = l & 15) of k-step s holds the 16
    bytes ``W[16 ct + c][64 s + 16 q : +16]`` — the B operand of one ``v_mfma_i32_16x16x64_i8`` (K zero-padded to 64)."""
    cout, cin = w.shape
    kp = (cin + 63) // 64 * 64
    wz = np.zeros((cout, kp), np.int8)
    wz[:, :cin] = w
    t = wz.reshape(cout // 16, 16, kp // 64, 4, 16)  # [ct][c][s][q][16]
    return np.ascontiguousarray(t.transpose(2, 0, 3, 1, 4)).reshape(kp // 64, cout // 16, 64, 16)


STRIP_MAX_SHIFT = 22  # c1 = 2^(e-1) + (zp << e) and the rounded product must stay inside int32


def _strip_requant(mult, shift, zp: int, acc_lo, acc_hi) -> tuple[np.ndarray, np.ndarray, np.ndarray] | None:
    """Per-channel (multiplier, c1, e) of the strip kernel's requantisation ``((v + c1 + (v >> 31)) >> e)`` with
    ``v = (x * multiplier + 2^30) >> 31`` — ``RoundingDivideByPOT(SRDHM(x, M0), e) + zp`` with the rounding offset and the
    zero point in one addend.  ``[acc_lo[c], acc_hi[c]]`` contains every accumulator the channel can see.  Channels whose
    shift exceeds ``STRIP_MAX_SHIFT`` are dead (vanishing weight scale, the bias is all that is left): requantisation is
    monotone in x, so when both ends of the range give the same value the channel is constant and is rewritten as
    (multiplier 0, e = 1, c1 = 1 + 2 (value + zp)), which produces exactly that.  ``None`` = not expressible (left shifts,
    negative multipliers, a dead channel that is not constant)."""
    m = np.asarray(mult, np.int64).copy()
    sh = np.asarray(shift, np.int64)
    e = -sh
    lo, hi = np.asarray(acc_lo, np.int64), np.asarray(acc_hi, np.int64)
    if (m < 0).any() or (e < 1).any():
        return None
    c1 = (np.int64(1) << (e - 1)) + (np.int64(zp) << e)
    for c in np.nonzero(e > STRIP_MAX_SHIFT)[0]:
        q_lo, q_hi = (int(qz.requantize(np.array([v]), m[c], sh[c])[0]) for v in (lo[c], hi[c]))
        if q_lo != q_hi:
            return None
        m[c], e[c], c1[c] = 0, 1, 1 + 2 * (q_lo + zp)
    if (np.maximum(np.abs(lo), np.abs(hi))[m != 0] >= 2**30).any():  # |v| <= |x| and |c1| < 2^30 must add inside int32
        return None
    return m.astype(np.int32), c1.astype(np.int32), e.astype(np.int32)


def strip_waves(cin: int, cout: int, stride: int, ow: int, add: bool) -> int:
    """Waves sharing one strip (channel split) — mirrors ``i8_strip_waves`` in csrc/bn_i8_strip.hip; 0 = no strip kernel."""
    if stride not in (1, 2) or (add and (stride != 1 or cin != cout)):
        return 0
    if ow == 8:  # stage 4: two row blocks x 8 columns per wave (needs an even map height, checked by the caller)
        if add:
            return 8 if cin == 256 else 0
        return 4 if (cin, cout, stride) == (128, 256, 2) else 0
    if ow % 16:
        return 0
    if (cin, cout) == (64, 64):
        return 2
    if cin in (32, 64) and cout in (32, 64):
        return 1
    if add:
        return 4 if cin == 128 else 0
    if (cin, cout) == (64, 128):
        return 2
    if (cin, cout) == (128, 128):
        return 4
    return 0


def add_table(add_p, z_own: int) -> np.ndarray:
    """The whole TFLite int8 ADD as a function of two bytes: int8 ``[256][256]``, row = byte pattern of the residual, column =
    the block's own value + 128.  ``add_p`` = [1, z_res, m_res, s_res, m_own, s_own, m_out, s_out, z_out, amin, amax]."""
    _, z1, m1, s1, m2, s2, mo, so, zo, amin, amax = add_p
    res = np.arange(256).astype(np.uint8).view(np.int8).astype(np.int64)
    own = np.arange(256, dtype=np.int64) - 128
    sa = qz.requantize((res - z1) << 20, m1, s1)
    sb = qz.requantize((own - z_own) << 20, m2, s2)
    out = np.clip(qz.requantize(sa[:, None] + sb[None, :], mo, so) + zo, amin, amax)
    return out.astype(np.int8)


def strip_constants(wd, bdw, mu, sh, z_dw_out, w2, b2, mu2, sh2, z_pw_out, add: bool, nw: int = 1) -> np.ndarray | None:
    """Constant block of ``i8_strip_kernel`` (csrc/bn_i8_strip.hip) for one DW 3x3 -> PW 1x1 [-> ADD] block, int32 words.

    ``nw`` waves share a strip: wave w runs the depthwise stage for the CW = Cin/nw input channels ``CW w ..`` and produces
    the CWO = Cout/nw output channels ``CWO w ..``.  Lane (n, kq) of wave w holds the CL = CW/4 channels ``CW w + CL kq ..``
    of column n (QL = CL/4 quads) and ends up with the COL = CWO/4 output channels ``CWO w + COL kq ..``.  Sections, each
    with the wave index first: depthwise weights ``[w][kq][ql][row][e]`` as bytes (tap0, tap1, tap2, 0) of channel
    ``CW w + CL kq + 4 ql + e``; folded depthwise bias ``[w][kq][ql][e]``; depthwise (multiplier, c1, e) ``[w][kq][ql][3][e]``;
    pointwise A fragments ``[w][t][ks][lane][QL]`` = ``W[CWO w + COL (m >> 2) + 4 t + (m & 3)][CW ks + CL kq : + CL]`` for lane
    (m, kq) and channel slice ks; folded pointwise bias ``[w][q][t][reg]`` for channel ``CWO w + COL q + 4 t + reg``;
    pointwise (multiplier, c1, e) ``[w][q][t][3][reg]``.  With the ADD the block's own value is produced + 128 (it indexes a
    table).  ``None`` if the block's requantisation cannot take the kernel's form."""
    wd = np.asarray(wd, np.int8)  # [3][3][C]
    C, (N, K) = wd.shape[2], w2.shape
    if K != C or nw < 1 or C % (16 * nw) or N % (16 * nw):
        return None
    CW, CWO = C // nw, N // nw
    CL, COL = CW // 4, CWO // 4
    QL, NT = CL // 4, CWO // 16
    if (CW, QL) not in ((32, 2), (64, 4)) or NT not in (2, 4) or (nw > 1 and CW != 32):
        return None
    # accumulator ranges over every int8 input (the folded biases already hold -zp * sum(w); padded taps read zp)
    wd64, w264 = wd.astype(np.int64), np.asarray(w2, np.int64)
    lo_dw = np.minimum(-128 * wd64, 127 * wd64).sum(axis=(0, 1)) + np.asarray(bdw, np.int64)
    hi_dw = np.maximum(-128 * wd64, 127 * wd64).sum(axis=(0, 1)) + np.asarray(bdw, np.int64)
    lo_pw = np.minimum(-128 * w264, 127 * w264).sum(axis=1) + np.asarray(b2, np.int64)
    hi_pw = np.maximum(-128 * w264, 127 * w264).sum(axis=1) + np.asarray(b2, np.int64)
    rq_dw = _strip_requant(mu, sh, z_dw_out, lo_dw, hi_dw)
    rq_pw = _strip_requant(mu2, sh2, z_pw_out + (128 if add else 0), lo_pw, hi_pw)
    if rq_dw is None or rq_pw is None:
        return None
    ax = np.arange
    ch_in = (CW * ax(nw)[:, None, None, None] + CL * ax(4)[None, :, None, None] + 4 * ax(QL)[None, None, :, None] + ax(4)[None, None, None, :])  # [w][kq][ql][e]
    dww = np.zeros((nw, 4, QL, 3, 4, 4), np.uint8)  # [...][row][e][byte]
    for i in range(3):
        for j in range(3):
            dww[..., i, :, j] = wd[i, j][ch_in].view(np.uint8)
    sec = [dww.view(np.int32).reshape(-1), np.asarray(bdw, np.int32)[ch_in].reshape(-1),
           np.stack([r[ch_in] for r in rq_dw], axis=3).reshape(-1)]  # [w][kq][ql][3][e]
    lane = ax(64)
    m_, kq_ = lane & 15, lane >> 4
    pwa = np.zeros((nw, NT, nw, 64, CL), np.int8)
    for w in range(nw):
        for t in range(NT):
            ch = CWO * w + COL * (m_ >> 2) + 4 * t + (m_ & 3)
            for ks in range(nw):
                for k in range(CL):
                    pwa[w, t, ks, :, k] = w2[ch, CW * ks + CL * kq_ + k]
    ch_out = (CWO * ax(nw)[:, None, None, None] + COL * ax(4)[None, :, None, None] + 4 * ax(NT)[None, None, :, None] + ax(4)[None, None, None, :])  # [w][q][t][reg]
    sec += [pwa.view(np.int32).reshape(-1), np.asarray(b2, np.int32)[ch_out].reshape(-1),
            np.stack([r[ch_out] for r in rq_pw], axis=3).reshape(-1)]
    return np.concatenate(sec).astype(np.int32)


def front_strip_constants(ws, bs, mus, shs, z_fe, z_st, wd, bdw, mud, shd, z_dw, w2, b2, mu2, sh2, z_pw) -> np.ndarray | None:
    """Constant block of ``i8_front_strip_kernel`` (csrc/bn_i8_strip.hip), int32 words.  ``ws`` [3][3][16] stem weights,
    ``bs`` stem bias (unfolded), ``wd`` [3][3][16] depthwise weights with folded bias ``bdw``, ``w2`` [32][16] pointwise
    weights with folded bias ``b2``.  Lane (n, kq) holds channel quad kq.  Sections: stem A fragments ``[lane]`` = bytes
    (w[kq][0][m], w[kq][1][m], w[kq][2][m], 0) for lane (m, kq), 0 for kq = 3 (contraction index 8 * window row + column);
    stem bias with ``-zp_fe * sum(w)`` folded ``[q][reg]``; stem (multiplier, c1, e) ``[q][3][reg]``; depthwise weights
    ``[kq][row][e]`` as bytes (tap0, tap1, tap2, 0); depthwise bias ``[kq][e]``, (multiplier, c1, e) ``[kq][3][e]``; pointwise A
    fragments ``[t][lane]`` = ``W[8 (m >> 2) + 4 t + (m & 3)][4 kq : 4 kq + 4]``; pointwise bias ``[q][t][reg]`` and
    (multiplier, c1, e) ``[q][t][3][reg]`` for channel ``8 q + 4 t + reg``."""
    ws, wd, w2 = np.asarray(ws, np.int8), np.asarray(wd, np.int8), np.asarray(w2, np.int8)
    if ws.shape != (3, 3, 16) or wd.shape != (3, 3, 16) or w2.shape != (32, 16):
        return None
    ws64, wd64, w264 = ws.astype(np.int64), wd.astype(np.int64), w2.astype(np.int64)
    bsf = np.asarray(bs, np.int64) - z_fe * ws64.sum(axis=(0, 1))
    rng = lambda w, b, ax: (np.minimum(-128 * w, 127 * w).sum(axis=ax) + b, np.maximum(-128 * w, 127 * w).sum(axis=ax) + b)  # noqa: E731
    rq_st = _strip_requant(mus, shs, z_st, *rng(ws64, bsf, (0, 1)))
    rq_dw = _strip_requant(mud, shd, z_dw, *rng(wd64, np.asarray(bdw, np.int64), (0, 1)))
    rq_pw = _strip_requant(mu2, sh2, z_pw, *rng(w264, np.asarray(b2, np.int64), 1))
    if rq_st is None or rq_dw is None or rq_pw is None or np.abs(bsf).max() >= 2**31:
        return None
    lane = np.arange(64)
    m_, kq_ = lane & 15, lane >> 4
    sta = np.zeros((64, 4), np.int8)
    for j in range(3):
        sta[:, j] = np.where(kq_ < 3, ws[np.minimum(kq_, 2), j, m_], 0)
    quad = 4 * np.arange(4)[:, None] + np.arange(4)[None, :]  # [q][reg] -> channel
    dww = np.zeros((4, 3, 4, 4), np.uint8)
    for i in range(3):
        for j in range(3):
            dww[:, i, :, j] = wd[i, j][quad].view(np.uint8)
    pwa = np.zeros((2, 64, 4), np.int8)
    for t in range(2):
        ch = 8 * (m_ >> 2) + 4 * t + (m_ & 3)
        for k in range(4):
            pwa[t, :, k] = w2[ch, 4 * kq_ + k]
    ch_out = 8 * np.arange(4)[:, None, None] + 4 * np.arange(2)[None, :, None] + np.arange(4)[None, None, :]  # [q][t][reg]
    sec = [sta.view(np.int32).reshape(-1), bsf.astype(np.int32)[quad].reshape(-1), np.stack([r[quad] for r in rq_st], axis=1).reshape(-1),
           dww.view(np.int32).reshape(-1), np.asarray(bdw, np.int32)[quad].reshape(-1), np.stack([r[quad] for r in rq_dw], axis=1).reshape(-1),
           pwa.view(np.int32).reshape(-1), np.asarray(b2, np.int32)[ch_out].reshape(-1), np.stack([r[ch_out] for r in rq_pw], axis=2).reshape(-1)]
    out = np.concatenate(sec).astype(np.int32)
    assert out.size == 496
    return out



TAIL_G = 4          # chunks a workgroup of i8_tail_kernel holds in LDS at a time (csrc/bn_i8_tail.hip)
TAIL_LAYER_WORDS = 24
TAIL_HEAD_WORDS = 16


def _tail_quad_base(cin: int) -> list[int]:
    """First channel quad of lane group kq in i8_tail_kernel's depthwise stage (mirrors pq_base() in csrc/bn_i8_tail.hip): lane
    (n, kq) takes the quads ``base[kq] + 4 ks + j`` (k-step ks, j = 0..3).  With a map pitch of Cin + 4 bytes the groups kq and
    kq ^ 1 are 16 quads apart, so the 32 lanes of one LDS access cycle hit 32 different banks."""
    return {64: [0, 4, 8, 12], 128: [0, 16, 8, 24], 256: [0, 16, 32, 48]}[cin]


def tail_layer_sections(wd, bdw, rq_dw, w2, b2, rq_pw, add_luts):
    """Constant sections of one block of the fused tail, int32 words: pointwise A fragments ``[nt][ks][lane][16 bytes]`` = bytes
    ``W[16 nt + m][4 pq(kq, ks, b >> 2) + (b & 3)]`` for lane (m, kq); depthwise constants ``[quad][7][4]`` (three tap rows as bytes
    (tap0, tap1, tap2, 0) per channel, folded bias, multiplier, c1, shift); pointwise constants ``[nt][q][4][4]`` (folded bias,
    multiplier, c1, shift of channel 16 nt + 4 q + r); the two 256-entry ADD tables or an empty array."""
    wd = np.asarray(wd, np.int8)
    C, (N, K) = wd.shape[2], w2.shape
    assert K == C and C in (64, 128, 256) and N % 16 == 0
    base = _tail_quad_base(C)
    KS = C // 64
    lane = np.arange(64)
    m_, kq_ = lane & 15, lane >> 4
    frag = np.zeros((N // 16, KS, 64, 16), np.int8)
    for nt in range(N // 16):
        for ks in range(KS):
            for b in range(16):
                pq = np.asarray(base)[kq_] + 4 * ks + (b >> 2)
                frag[nt, ks, :, b] = w2[16 * nt + m_, 4 * pq + (b & 3)]
    dwc = np.zeros((C // 4, 7, 4), np.int32)
    ch = 4 * np.arange(C // 4)[:, None] + np.arange(4)[None, :]  # [quad][e]
    rows = np.zeros((C // 4, 3, 4, 4), np.uint8)
    for i in range(3):
        for j in range(3):
            rows[:, i, :, j] = wd[i, j][ch].view(np.uint8)
    dwc[:, 0:3, :] = rows.view(np.int32).reshape(C // 4, 3, 4)
    dwc[:, 3, :] = np.asarray(bdw, np.int32)[ch]
    for k in range(3):
        dwc[:, 4 + k, :] = rq_dw[k][ch]
    cho = 16 * np.arange(N // 16)[:, None, None] + 4 * np.arange(4)[None, :, None] + np.arange(4)[None, None, :]  # [nt][q][r]
    pwc = np.stack([np.asarray(b2, np.int32)[cho], rq_pw[0][cho], rq_pw[1][cho], rq_pw[2][cho]], axis=2)  # [nt][q][kind][r]
    lut = np.concatenate(add_luts).astype(np.int32) if add_luts is not None else np.zeros(0, np.int32)
    return frag.view(np.int32).reshape(-1), dwc.reshape(-1), pwc.astype(np.int32).reshape(-1), lut


def tail_constants(blocks: list[dict], head: dict):
    """Constant block and descriptor table of ``i8_tail_kernel`` (csrc/bn_i8_tail.hip) for the blocks behind stage 2 of the shipped
    topology plus MEAN / FULLY_CONNECTED / LOGISTIC, or ``None`` when a block's arithmetic cannot take the kernel's forms.

    Every block is a dict with the fields of the I8_DWPW operator it replaces (geometry, quantisation, weights with folded biases).
    Descriptor words per block: H W Cin Cout S OH OW pt pl has_add zp_in dw_lo dw_hi pw_lo pw_hi add_m add_c1 add_e add_lo add_hi
    g_w g_dwc g_pwc g_lut (the g_* are word offsets into the constant block; with the ADD the pointwise stage produces its value + 128,
    the index of the second table).  Head words: mean zp_in mult shift zp_out | fc zp_out lo hi g_w g_bias g_mult g_shift | g_lut (-1 =
    none) zp_fc zp_head P C."""
    sections, desc = [], []
    pos = 0

    def put(arr):
        nonlocal pos
        a = np.ascontiguousarray(np.asarray(arr, np.int32).reshape(-1))
        pad = (-a.size) % 4  # 16-byte pieces for the staging copies
        a = np.concatenate([a, np.zeros(pad, np.int32)])
        off = pos
        sections.append(a)
        pos += a.size
        return off

    for b in blocks:
        C, N = b["C"], b["N"]
        if C not in (64, 128, 256) or N % 64 or b["sh"] != b["sw"] or b["sh"] not in (1, 2) or (b["OH"] * b["OW"]) % 16 or b["OW"] not in (8, 16):
            return None
        wd64, w264 = np.asarray(b["wd"], np.int64), np.asarray(b["w2"], np.int64)
        lo_dw = np.minimum(-128 * wd64, 127 * wd64).sum(axis=(0, 1)) + np.asarray(b["bdw"], np.int64)
        hi_dw = np.maximum(-128 * wd64, 127 * wd64).sum(axis=(0, 1)) + np.asarray(b["bdw"], np.int64)
        lo_pw = np.minimum(-128 * w264, 127 * w264).sum(axis=1) + np.asarray(b["b2"], np.int64)
        hi_pw = np.maximum(-128 * w264, 127 * w264).sum(axis=1) + np.asarray(b["b2"], np.int64)
        add = b["add"]
        rq_dw = _strip_requant(b["mu"], b["sh_dw"], b["z_dw"], lo_dw, hi_dw)
        rq_pw = _strip_requant(b["mu2"], b["sh2"], b["z_pw"] + (128 if add[0] else 0), lo_pw, hi_pw)
        if rq_dw is None or rq_pw is None or b["dw_lo"] < b["z_dw"] or (not add[0] and b["pw_lo"] < b["z_pw"]):
            return None  # (the kernel drops the sign term of the rounding shift where a negative result clamps to the zero point anyway)
        luts, add_m, add_c1, add_e, add_lo, add_hi = None, 0, 0, 1, 0, 0
        if add[0]:
            _, z1, m1, s1, m2, s2, mo, so, zo, amin, amax = add
            if mo < 0 or so >= 0 or -so > STRIP_MAX_SHIFT or amin < zo:
                return None  # (amin >= zo: the kernel's rescale of the sum drops the sign term of the rounding shift like the other stages)
            res = np.arange(256).astype(np.uint8).view(np.int8).astype(np.int64)
            own = np.arange(256, dtype=np.int64) - 128
            sa = qz.requantize((res - z1) << 20, m1, s1)
            sb = qz.requantize((own - b["z_pw"]) << 20, m2, s2)
            if max(np.abs(sa).max(), np.abs(sb).max()) >= 2**29:
                return None
            luts = (sa, sb)
            add_m, add_e = int(mo), int(-so)
            add_c1 = (1 << (add_e - 1)) + (int(zo) << add_e)
            add_lo, add_hi = int(amin), int(amax)
        frag, dwc, pwc, lut = tail_layer_sections(b["wd"], b["bdw"], rq_dw, b["w2"], b["b2"], rq_pw, luts)
        off = 128 if add[0] else 0
        desc += [b["H"], b["W"], C, N, b["sh"], b["OH"], b["OW"], b["pt"], b["pl"], int(bool(add[0])), b["z_in"], b["dw_lo"], b["dw_hi"],
                 b["pw_lo"] + off, b["pw_hi"] + off, add_m, add_c1, add_e, add_lo, add_hi, put(frag), put(dwc), put(pwc), put(lut) if lut.size else -1]
    h = head
    if h["C"] != 256 or h["C"] != blocks[-1]["N"] or h["P"] != blocks[-1]["OH"] * blocks[-1]["OW"] or TAIL_G * h["NC"] > 1024:
        return None
    desc += [h["mean_zp_in"], h["mean_mult"], h["mean_shift"], h["mean_zp_out"], h["fc_zp_out"], h["fc_lo"], h["fc_hi"],
             put(np.ascontiguousarray(np.asarray(h["fc_w"], np.int8)).view(np.int32)), put(h["fc_b"]), put(h["fc_m"]), put(h["fc_s"]),
             put(np.asarray(h["lut"], np.int8).view(np.int32)) if h["lut"] is not None else -1, h["zp_fc"], h["zp_head"], h["P"], h["C"]]
    assert len(desc) == TAIL_LAYER_WORDS * len(blocks) + TAIL_HEAD_WORDS
    return np.concatenate(sections).astype(np.int32), np.asarray(desc, np.int32)


TAIL2_LAYER_WORDS = 32
TAIL2_DW_KINDS = 5   # v4i per (channel tile, lane group) of the depthwise constants: bias, multiplier, C01, C23, packed shifts
TAIL2_DW_KINDS_FIRST = 8   # ... + three border biases in the first block (taps from memory, zero-filled outside the map)
TAIL2_PW_KINDS = 5


def _rq_hi_consts(rq):
    """(multiplier, c1, e) per channel -> the operands of the sign-free one-multiply-add form ``((x m + C) >> 32) >> (e - 1)`` with
    ``C = (c1 << 31) + 2^30``: multiplier, C low dword, C high dword, e - 1 (csrc/bn_i8_tail2.hip: rq_hi)."""
    m, c1, e = (np.asarray(v, np.int64) for v in rq)
    c = (c1 << 31) + (1 << 30)
    return m.astype(np.int32), (c & 0xFFFFFFFF).astype(np.uint32).view(np.int32), (c >> 32).astype(np.int32), (e - 1).astype(np.int32)


def tail2_layer_section(b: dict, rq_dw, rq_pw, first: bool) -> np.ndarray:
    """Constants of one block of ``i8_tail2_kernel`` (csrc/bn_i8_tail2.hip) as ONE run of int32 words: the depthwise part (depthwise A
    fragments, depthwise constants) followed by the pointwise part (pointwise A fragments, pointwise constants) — the kernel stages the
    two parts at different times.  The pieces:

    * pointwise A fragments ``[nt][ks][lane][16 bytes]``: byte ``4 j + r`` of lane (m, g) = ``W[16 nt + m][16 (4 ks + j) + 4 g + r]`` —
      the contraction order in which the depthwise stage leaves its results in registers (dword j of k-step ks = channel tile
      ``4 ks + j``, lane group g holds its channels ``4 g ..``);
    * depthwise A fragments ``[ct][dy][lane][16 bytes]``: the 3x3 weights of channel tile ct as a block-diagonal 16 x 64 matrix per
      window row dy: lane (m, g), byte c' = ``wd[dy][g][16 ct + m]`` where c' = m and g < 3, else 0 (contraction index = 16 g + c':
      window column g, channel c' of the tile);
    * depthwise constants ``[ct][kind][g][4]`` for channel ``16 ct + 4 g + r``: folded bias, multiplier, (C low, C high) of r = 0, 1,
      of r = 2, 3, packed shifts - 1, then the bias for positions whose window leaves the map on the right / at the bottom / both when
      the taps outside read 0 instead of the zero point (first block: taps from memory, range-checked loads);
    * pointwise constants ``[nt][kind][g][4]``: the same five kinds.  With the ADD the block's own term may be negative, so the form keeps
      the sign term of the rounding shift: kinds (bias, M' = 2 m as a signed dword, (2^31, 2^(e-1)) pairs, packed shifts e) of
      ``v = (hi + x + (x >> 31)) >> e`` with ``hi = (x M' + C) >> 32`` — because M' is read as 2 m - 2^32, ``hi = SRDHM(x, m) + 2^(e-1) - x``
      (csrc/bn_i8_tail2.hip: rq_signed; ``tail2_constants`` checks the conditions).  A dead channel (vanishing scale, constant output q)
      gets zero weights and bias, M' = 0, e = 1 and 2 q in place of 2^(e-1)."""
    wd, w2 = np.asarray(b["wd"], np.int8), np.asarray(b["w2"], np.int8).copy()
    b2 = np.asarray(b["b2"], np.int64).copy()
    C, N = b["C"], b["N"]
    if b["add"][0]:
        dead = np.asarray(rq_pw[0]) == 0
        w2[dead, :] = 0
        b2[dead] = 0
    KS, NCT, NT = (C + 63) // 64, C // 16, N // 16   # (32 input channels: half of the k-step's contraction index stays zero)
    lane = np.arange(64)
    m_, g_ = lane & 15, lane >> 4
    frag = np.zeros((NT, KS, 64, 16), np.int8)
    for nt in range(NT):
        for ks in range(KS):
            for j in range(4):
                if 4 * ks + j >= NCT:
                    continue
                for r in range(4):
                    frag[nt, ks, :, 4 * j + r] = w2[16 * nt + m_, 16 * (4 * ks + j) + 4 * g_ + r]
    dwa = np.zeros((NCT, 3, 64, 16), np.int8)
    for ct in range(NCT):
        for dy in range(3):
            for ln in range(64):
                if g_[ln] < 3:
                    dwa[ct, dy, ln, m_[ln]] = wd[dy, g_[ln], 16 * ct + m_[ln]]
    ch = 16 * np.arange(NCT)[:, None, None] + 4 * np.arange(4)[None, :, None] + np.arange(4)[None, None, :]  # [ct][g][r]
    m, clo, chi, e1 = _rq_hi_consts(rq_dw)
    dwc = np.zeros((NCT, TAIL2_DW_KINDS_FIRST if first else TAIL2_DW_KINDS, 4, 4), np.int32)
    bdw = np.asarray(b["bdw"], np.int64)
    dwc[:, 0] = bdw[ch]
    dwc[:, 1] = m[ch]
    dwc[:, 2] = np.stack([clo[ch][..., 0], chi[ch][..., 0], clo[ch][..., 1], chi[ch][..., 1]], axis=-1)
    dwc[:, 3] = np.stack([clo[ch][..., 2], chi[ch][..., 2], clo[ch][..., 3], chi[ch][..., 3]], axis=-1)
    e1c = e1[ch].astype(np.int64)
    dwc[:, 4, :, 0] = (e1c[..., 0] | (e1c[..., 1] << 8) | (e1c[..., 2] << 16) | (e1c[..., 3] << 24)).astype(np.int32)
    if first:  # a tap outside the map contributes (0 - zp) w instead of 0: take it back through the bias
        wd64 = wd.astype(np.int64)
        right, bottom = wd64[:, 2, :].sum(axis=0), wd64[2, :, :].sum(axis=0)
        corner = right + bottom - wd64[2, 2, :]
        for k, lost in ((5, right), (6, bottom), (7, corner)):
            dwc[:, k] = (bdw + b["z_in"] * lost)[ch]
    cho = 16 * np.arange(NT)[:, None, None] + 4 * np.arange(4)[None, :, None] + np.arange(4)[None, None, :]  # [nt][g][r]
    pwc = np.zeros((NT, TAIL2_PW_KINDS, 4, 4), np.int32)
    pwc[:, 0] = b2.astype(np.int32)[cho]
    if b["add"][0]:
        m, c1, e = (np.asarray(v, np.int64) for v in rq_pw)
        dead = m == 0
        m2 = np.where(dead, 0, 2 * m - (1 << 32)).astype(np.int32)            # 2 m read as a signed dword
        half = np.where(dead, c1 - 1, np.int64(1) << (e - 1)).astype(np.int32)   # dead: c1 = 1 + 2 q, e = 1 (_strip_requant)
        lo = np.full(N, -(1 << 31), np.int64).astype(np.int32)
        pwc[:, 1] = m2[cho]
        pwc[:, 2] = np.stack([lo[cho][..., 0], half[cho][..., 0], lo[cho][..., 1], half[cho][..., 1]], axis=-1)
        pwc[:, 3] = np.stack([lo[cho][..., 2], half[cho][..., 2], lo[cho][..., 3], half[cho][..., 3]], axis=-1)
        ec = e[cho]
        pwc[:, 4, :, 0] = (ec[..., 0] | (ec[..., 1] << 8) | (ec[..., 2] << 16) | (ec[..., 3] << 24)).astype(np.int32)
    else:
        m, clo, chi, e1 = _rq_hi_consts(rq_pw)
        pwc[:, 1] = m[cho]
        pwc[:, 2] = np.stack([clo[cho][..., 0], chi[cho][..., 0], clo[cho][..., 1], chi[cho][..., 1]], axis=-1)
        pwc[:, 3] = np.stack([clo[cho][..., 2], chi[cho][..., 2], clo[cho][..., 3], chi[cho][..., 3]], axis=-1)
        e1c = e1[cho].astype(np.int64)
        pwc[:, 4, :, 0] = (e1c[..., 0] | (e1c[..., 1] << 8) | (e1c[..., 2] << 16) | (e1c[..., 3] << 24)).astype(np.int32)
    return np.concatenate([dwa.view(np.int32).reshape(-1), dwc.reshape(-1), frag.view(np.int32).reshape(-1), pwc.reshape(-1)]).astype(np.int32)


def tail2_constants(blocks: list[dict], head: dict | None):
    """Constant block and descriptor table of ``i8_tail2_kernel`` (csrc/bn_i8_tail2.hip: the fused tail with the depthwise stage on the
    matrix cores), or ``None`` when a block cannot take its forms (the plan then keeps ``i8_tail_kernel`` only).

    On top of ``tail_constants``' conditions the ADD must have the shape TFLite's converter gives a residual block: the block's own
    value is the input with the LARGER scale (multiplier 2^30, shift 0: its rescale is exactly ``(v - z) << 19``) and the residual's
    zero point is -128 (its rescale ``RoundingDivideByPOT(SRDHM((b + 128) << 20, m), e)`` of a non-negative number is one unsigned
    multiply-add: ``(((b + 128) << 24) m + R 2^24) >> 32 >> (3 + e)`` with ``R = 2^10 + 2^(10 + e)``) — both checked here on all 256 bytes.

    Descriptor words per block (32): H W Cin Cout S OH OW pt pl has_add zp_in dw_lo dw_hi pw_lo pw_hi add_m add_c1 add_e add_lo add_hi
    res_m res_c_lo res_c_hi res_k g_cst 0...; with the ADD the pointwise stage produces its value MINUS its zero point (clamp bounds
    shifted accordingly).  Head words as in ``tail_constants`` except g_w = the classifier as matrix-core fragments, g_bias = its
    constants in lane order, g_mult = g_shift = 0."""
    sections, desc = [], []
    pos = 0

    def put(arr):
        nonlocal pos
        a = np.ascontiguousarray(np.asarray(arr, np.int32).reshape(-1))
        a = np.concatenate([a, np.zeros((-a.size) % 4, np.int32)])
        off = pos
        sections.append(a)
        pos += a.size
        return off

    for i, b in enumerate(blocks):
        C, N = b["C"], b["N"]
        if C not in (32, 64, 128, 256) or N % 64 or b["sh"] != b["sw"] or b["sh"] not in (1, 2) or (b["OH"] * b["OW"]) % 16 or b["OW"] not in (8, 16, 32):
            return None
        wd64, w264 = np.asarray(b["wd"], np.int64), np.asarray(b["w2"], np.int64)
        lo_dw = np.minimum(-128 * wd64, 127 * wd64).sum(axis=(0, 1)) + np.asarray(b["bdw"], np.int64)
        hi_dw = np.maximum(-128 * wd64, 127 * wd64).sum(axis=(0, 1)) + np.asarray(b["bdw"], np.int64)
        lo_pw = np.minimum(-128 * w264, 127 * w264).sum(axis=1) + np.asarray(b["b2"], np.int64)
        hi_pw = np.maximum(-128 * w264, 127 * w264).sum(axis=1) + np.asarray(b["b2"], np.int64)
        if i == 0:  # zero-filled taps: the corrected biases widen the range by at most 128 * sum |w| of the taps left out, already inside lo / hi
            if b["z_in"] != -128:
                return None
        add = b["add"]
        rq_dw = _strip_requant(b["mu"], b["sh_dw"], b["z_dw"], lo_dw, hi_dw)
        rq_pw = _strip_requant(b["mu2"], b["sh2"], 0 if add[0] else b["z_pw"], lo_pw, hi_pw)
        if rq_dw is None or rq_pw is None or b["dw_lo"] < b["z_dw"] or (not add[0] and b["pw_lo"] < b["z_pw"]):
            return None
        if (rq_dw[2] > 32).any() or (not add[0] and (rq_pw[2] > 32).any()):
            return None
        add_m, add_c1, add_e, add_lo, add_hi, res_m, res_c, res_k = 0, 0, 1, 0, 0, 0, 0, 0
        pw_lo, pw_hi = b["pw_lo"], b["pw_hi"]
        if add[0]:
            _, z1, m1, s1, m2, s2, mo, so, zo, amin, amax = (int(v) for v in add)
            if mo < 0 or so >= 0 or -so > STRIP_MAX_SHIFT or amin < zo or z1 != -128 or m1 < 0 or s1 > 0 or -s1 > 16 or m2 != 1 << 30 or s2 != 0:
                return None
            e1 = -s1
            R = (1 << 10) + ((1 << (e1 + 10)) if e1 >= 1 else 0)
            x = np.arange(256, dtype=np.int64)
            want = qz.requantize(x << 20, m1, s1)
            got = ((((x << 24) * m1 + (R << 24)) >> 32) >> (3 + e1))
            own = np.arange(-128, 128, dtype=np.int64) - b["z_pw"]
            if not np.array_equal(want, got) or not np.array_equal(qz.requantize(own << 20, m2, s2), own << 19) or np.abs(want).max() >= 2**29:
                return None
            # the own term v = MBQM(acc) is NOT clamped to int8 in the kernel: where the clamp would act, the sum saturates the output either
            # way (checked on the table of the whole ADD), and v must stay small enough for (v << 19) + f to stay below 2^30
            mq, _, eq = (np.asarray(v, np.int64) for v in rq_pw)
            live = mq != 0
            if (mq[live] == 1 << 30).any():
                return None   # (sign of SRDHM(x, 2^30) differs from the sign of x at x = -1)
            v_lo = qz.requantize(lo_pw, np.asarray(b["mu2"]), np.asarray(b["sh2"]))
            v_hi = qz.requantize(hi_pw, np.asarray(b["mu2"]), np.asarray(b["sh2"]))
            if max(np.abs(v_lo).max(), np.abs(v_hi).max()) >= 1 << 11:   # ((v << 19) + f stays inside int32)
                return None
            full = add_table(add, b["z_pw"])   # [residual byte pattern][own + 128]: columns beyond the int8 clamp of the own value must equal the edge columns
            own_all = np.arange(-2048, 2048, dtype=np.int64)
            sa = qz.requantize((np.arange(256).astype(np.uint8).view(np.int8).astype(np.int64) - z1) << 20, m1, s1)
            unclamped = np.clip(qz.requantize(sa[:, None] + (own_all[None, :] << 19), mo, so) + zo, amin, amax)
            clamped_cols = np.clip(own_all + b["z_pw"], b["pw_lo"], b["pw_hi"]) + 128
            if not np.array_equal(unclamped, full[:, clamped_cols].astype(np.int64)):
                return None
            res_m, res_c, res_k = m1, R << 24, 3 + e1
            add_m, add_e = mo, -so
            add_c1 = (1 << (add_e - 1)) + (zo << add_e)
            add_lo, add_hi = amin, amax
            pw_lo, pw_hi = -2048, 2048   # (not used by the kernel: see above)
        g_cst = put(tail2_layer_section(b, rq_dw, rq_pw, first=(i == 0)))
        row = [b["H"], b["W"], C, N, b["sh"], b["OH"], b["OW"], b["pt"], b["pl"], int(bool(add[0])), b["z_in"], b["dw_lo"], b["dw_hi"],
               pw_lo, pw_hi, add_m, add_c1, add_e, add_lo, add_hi, res_m, res_c & 0xFFFFFFFF, res_c >> 32, res_k, g_cst]
        desc += [int(np.int64(v).astype(np.uint32).view(np.int32)) if v > 0x7FFFFFFF else int(v) for v in row] + [0] * (TAIL2_LAYER_WORDS - len(row))
    if head is None:   # (the chain in front of the tail: its last map goes back to memory)
        return np.concatenate(sections).astype(np.int32), np.asarray(desc, np.int32)
    h = head
    if h["C"] != 256 or h["C"] != blocks[-1]["N"] or h["P"] != blocks[-1]["OH"] * blocks[-1]["OW"] or TAIL_G * h["NC"] > 1024:
        return None
    # FULLY_CONNECTED on the matrix cores: A fragments [class tile][ks][lane][16 bytes] = W[16 ct + m][64 ks + 16 g + j] (classes padded to
    # a multiple of 16 with zero rows), constants [class tile][kind: bias, multiplier, shift][g][4] for class 16 ct + 4 g + r
    fcw = np.asarray(h["fc_w"], np.int8)
    NC, Cfc = h["NC"], h["C"]
    if fcw.shape != (NC, Cfc) or Cfc % 64:
        return None
    nct = (NC + 15) // 16
    wpad = np.zeros((nct * 16, Cfc), np.int8)
    wpad[:NC] = fcw
    lane = np.arange(64)
    m_, g_ = lane & 15, lane >> 4
    fcf = np.zeros((nct, Cfc // 64, 64, 16), np.int8)
    for ct in range(nct):
        for ks in range(Cfc // 64):
            for j in range(16):
                fcf[ct, ks, :, j] = wpad[16 * ct + m_, 64 * ks + 16 * g_ + j]
    cls = 16 * np.arange(nct)[:, None, None] + 4 * np.arange(4)[None, :, None] + np.arange(4)[None, None, :]   # [ct][g][r]
    pad = lambda v: np.concatenate([np.asarray(v, np.int32), np.zeros(nct * 16 - NC, np.int32)])  # noqa: E731
    fcc = np.stack([pad(h["fc_b"])[cls], pad(h["fc_m"])[cls], pad(h["fc_s"])[cls]], axis=1)   # [ct][kind][g][r]
    desc += [h["mean_zp_in"], h["mean_mult"], h["mean_shift"], h["mean_zp_out"], h["fc_zp_out"], h["fc_lo"], h["fc_hi"],
             put(fcf.view(np.int32)), put(fcc), 0, 0,
             put(np.asarray(h["lut"], np.int8).view(np.int32)) if h["lut"] is not None else -1, h["zp_fc"], h["zp_head"], h["P"], h["C"]]
    assert len(desc) == TAIL2_LAYER_WORDS * len(blocks) + TAIL_HEAD_WORDS
    return np.concatenate(sections).astype(np.int32), np.asarray(desc, np.int32)


def lower_i8(model, keep_all: bool = False, fuse: bool = True, softmax_form: str = "fixed", mean_form: str = "int") -> pk.Plan:
    """Build the INT8 plan for a decoded ``TfliteModel``.  ``keep_all`` disables slot reuse; ``fuse=False`` keeps the
    baseline one-kernel-per-operator plan instead of the fused matrix-core blocks.  ``softmax_form`` picks the arithmetic of an int8
    SOFTMAX (attention pooling): ``'fixed'`` = TFLite's reference kernel, ``'lut'`` = its optimized kernel (oracle/int8_graph.py has both).
    ``mean_form`` picks the arithmetic of int8 MEAN: ``'int'`` = the integer form with the count folded into the multiplier (reduce.h),
    ``'float'`` = TFLite's float-arithmetic ``QuantizedMeanOrSum`` (the two differ by one step on ~4 % of the pooled bytes of the shipped
    graph; every MEAN kernel of the device evaluates either, csrc/bn_requant.h: mean_q)."""
    if mean_form not in ("int", "float"):
        raise ValueError("mean_form must be 'int' or 'float'")
    from birdnet_stm32.models._lower_f32 import pick_tile

    g = _Graph(model)
    ops = model.ops
    t = g.t
    _expect(len(model.inputs) == 1 and len(model.outputs) == 1, "single input / single output")
    in_shape = t[model.inputs[0]].shape
    if len(in_shape) == 3:
        # ---- raw frontend of an exported graph (conversion/export.py; reference models/frontend.py:138-164,347-358): QUANTIZE of the
        # waveform -> [PAD] -> RESHAPE (expand_dims) -> CONV_2D 1 x 16 strided VALID (ReLU6) -> magnitude scaling -> TRANSPOSE = ONE operator
        _expect(in_shape[2] == 1, "waveform input must be [B, T, 1]")
        T = int(in_shape[1])
        i = 0
        _expect(ops[i].name == "QUANTIZE" and ops[i].inputs[0] == model.inputs[0], "graph must start with QUANTIZE of the input")
        q_scale, q_zp = g.q(ops[i].outputs[0])
        cur = ops[i].outputs[0]
        i += 1
        pad_left = pad_right = 0
        for _ in range(2):  # PAD and RESHAPE in either order
            if ops[i].name == "PAD" and ops[i].inputs[0] == cur:
                pads = np.asarray(g.const(ops[i].inputs[1])).reshape(-1, 2)
                axis = int(np.argmax(t[ops[i].inputs[0]].shape))
                _expect(all(int(pads[a].sum()) == 0 for a in range(len(pads)) if a != axis) and g.q(ops[i].outputs[0]) == (q_scale, q_zp), "PAD along the time axis only")
                pad_left, pad_right = int(pads[axis][0]), int(pads[axis][1])
                cur = ops[i].outputs[0]
                i += 1
            elif ops[i].name == "RESHAPE" and ops[i].inputs[0] == cur:
                shp = tuple(int(v) for v in t[ops[i].outputs[0]].shape)
                _expect(len(shp) == 4 and shp[1] == 1 and shp[3] == 1, "expand_dims of the waveform to [1, 1, T, 1]")
                cur = ops[i].outputs[0]
                i += 1
        conv = ops[i]
        _expect(conv.name == "CONV_2D" and conv.inputs[0] == cur and tuple(t[cur].shape) == (1, 1, T + pad_left + pad_right, 1), "filterbank CONV_2D on [1, 1, T, 1]")
        wt = t[conv.inputs[1]]
        M = int(wt.shape[0])
        _expect(tuple(wt.shape[1:]) == (1, 16, 1) and conv.options["padding"] == "VALID" and conv.options["stride_h"] == 1, "filterbank must be 1 x 16, VALID")
        stride = int(conv.options["stride_w"])
        W = int(t[conv.outputs[0]].shape[2])
        _expect((T + pad_left + pad_right - 16) // stride + 1 == W and W % 4 == 0, "filterbank output width (a multiple of 4 frames)")
        _expect(g.q(conv.inputs[0]) == (q_scale, q_zp), "filterbank input quantisation")
        s_fb, z_fb = g.q(conv.outputs[0])
        w_fb = wt.data.reshape(M, 16).astype(np.int8)
        bias = g.const(conv.inputs[2]).astype(np.int64) - q_zp * w_fb.astype(np.int64).sum(axis=1)
        mult, shift = qz.channel_multipliers(q_scale, wt.scale, s_fb, M)
        _expect_acc_range(w_fb, bias, 1, "raw filterbank", mult, shift)
        lo, hi = qz.activation_bounds(conv.options["activation"], s_fb, z_fb)
        cur = conv.outputs[0]
        i += 1
        j = i
        while ops[j].name != "TRANSPOSE":
            j += 1
        lut = None
        if j > i:
            lut = _pwl_table(g, ops[i:j], cur, ops[j].inputs[0], M)
        _expect(list(g.const(ops[j].inputs[1])) == [0, 3, 2, 1], "TRANSPOSE [0,3,2,1] after the frontend")
        cur = ops[j].outputs[0]
        i = j + 1
        F = 0
        plan = pk.Plan(pk.DTYPE_I8, pk.INPUT_WAVEFORM, T, 0, W, int(t[model.outputs[0]].shape[-1]), meta={"tflite_ops": len(ops)})
        pb = pk.PlanBuilder(plan)
        tt = [pb.tensor(w_fb, np.int8), pb.tensor(bias, np.int32), pb.tensor(mult, np.int32), pb.tensor(shift, np.int32)]
        if lut is not None:
            tt.append(pb.tensor(lut, np.int8))
        v = pb.value(M * W)
        pb.op(pk.I8_RAWFE, pk.SLOT_INPUT, v, p=[T, W, M, stride, pad_left, q_zp, z_fb, lo, hi, int(lut is not None)], t=tt, f=[q_scale], name=f"t{cur}",
              out_shape=(M, W, 1), out_dtype="int8")
    else:
        _expect(len(in_shape) == 4 and in_shape[3] == 1, "input must be [B, F, W, 1]")
        F, W = int(in_shape[1]), int(in_shape[2])

        i = 0
        _expect(ops[i].name == "QUANTIZE" and ops[i].inputs[0] == model.inputs[0], "graph must start with QUANTIZE of the input")
        q_scale, q_zp = g.q(ops[i].outputs[0])
        cur = ops[i].outputs[0]
        i += 1
        _expect(ops[i].name == "TRANSPOSE" and list(g.const(ops[i].inputs[1])) == [0, 3, 2, 1], "TRANSPOSE [0,3,2,1] after QUANTIZE")
        cur = ops[i].outputs[0]
        i += 1
        if ops[i].name == "STRIDED_SLICE":
            _expect(tuple(t[ops[i].outputs[0]].shape[1:]) == (1, W, F), "frontend STRIDED_SLICE must keep [1, W, F]")
            cur = ops[i].outputs[0]
            i += 1
        fill_value = q_zp
        k_graph = F
        if ops[i].name == "SHAPE":
            j = i
            while ops[j].name != "CONCATENATION":
                _expect(ops[j].name in ("SHAPE", "STRIDED_SLICE", "PACK", "FILL"), f"{ops[j].name} in the channel-padding block")
                if ops[j].name == "FILL":
                    fill_value = int(np.asarray(g.const(ops[j].inputs[1])).reshape(-1)[0])
                j += 1
            cat = ops[j]
            _expect(cat.inputs[0] == cur and cat.options["axis"] in (-1, 3), "CONCATENATION must pad the channel axis")
            _expect(g.q(cat.inputs[0]) == g.q(cat.inputs[1]) == g.q(cat.outputs[0]), "CONCATENATION operands must share quantisation")
            k_graph = int(t[cat.outputs[0]].shape[3])
            cur = cat.outputs[0]
            i = j + 1

        # ---- mel mixer ------------------------------------------------------------------------
        mel = ops[i]
        _expect(mel.name == "CONV_2D" and mel.inputs[0] == cur, "mel mixer CONV_2D")
        wt = t[mel.inputs[1]]
        M = int(wt.shape[0])
        _expect(tuple(wt.shape[1:3]) == (1, 1) and int(wt.shape[3]) == k_graph, "mel mixer must be 1x1 over the padded bins")
        s_in, z_in = g.q(mel.inputs[0])
        _expect((s_in, z_in) == (q_scale, q_zp), "mel mixer input quantisation")
        s_mel, z_mel = g.q(mel.outputs[0])
        Kp = (k_graph + K_ALIGN - 1) // K_ALIGN * K_ALIGN
        w_mel = np.zeros((M, Kp), np.int8)
        w_mel[:, :k_graph] = wt.data.reshape(M, k_graph)
        bias = g.const(mel.inputs[2]).astype(np.int64) - z_in * w_mel.astype(np.int64).sum(axis=1)
        mult, shift = qz.channel_multipliers(s_in, wt.scale, s_mel, M)
        _expect_acc_range(w_mel, bias, 1, "mel mixer", mult, shift)
        lo, hi = qz.activation_bounds(mel.options["activation"], s_mel, z_mel)
        cur = mel.outputs[0]
        i += 1

        # ---- element-wise region up to the TRANSPOSE back ---------------------------------------
        j = i
        while ops[j].name != "TRANSPOSE":
            j += 1
        region = ops[i:j]
        lut = None
        front_out = cur
        maxnorm = None
        if region and region[0].name == "REDUCE_MAX":
            # per-sample max normalisation of current hybrid frontends (reference models/frontend.py:338-342): REDUCE_MAX over the whole map
            # -> ADD epsilon -> DIV by that scalar.  Everything behind the maximum is a function of bytes: the denominator byte per maximum
            # byte (the quantised ADD) and the DIV of every byte by every denominator byte become tables (models/_quant.py: div_table)
            _expect(len(region) >= 3 and region[1].name == "ADD" and region[2].name == "DIV", "REDUCE_MAX must be followed by ADD and DIV")
            rmax, addop, divop = region[:3]
            axes = sorted(int(a) % 4 for a in np.atleast_1d(g.const(rmax.inputs[1])))
            _expect(rmax.inputs[0] == cur and axes == [1, 2, 3] and g.q(rmax.outputs[0]) == g.q(cur), "REDUCE_MAX over the whole map, same quantisation")
            _expect(rmax.outputs[0] in addop.inputs and divop.inputs[0] == cur and divop.inputs[1] == addop.outputs[0], "max normalisation wiring")
            eps_i = [k for k in addop.inputs if k != rmax.outputs[0]]
            _expect(len(eps_i) == 1 and t[eps_i[0]].data is not None and t[eps_i[0]].data.size == 1, "ADD of a scalar constant to the maximum")
            eps_q = int(np.asarray(t[eps_i[0]].data).reshape(-1)[0])
            (s_m, z_m), (s_e, z_e), (s_d, z_d) = g.q(rmax.outputs[0]), g.q(eps_i[0]), g.q(addop.outputs[0])
            mx = np.arange(-128, 128, dtype=np.int64)
            if addop.inputs[0] == rmax.outputs[0]:
                den_tab = qz.AddParams(s_m, z_m, s_e, z_e, s_d, z_d, addop.options["activation"]).apply(mx, np.full(256, eps_q, np.int64))
            else:
                den_tab = qz.AddParams(s_e, z_e, s_m, z_m, s_d, z_d, addop.options["activation"]).apply(np.full(256, eps_q, np.int64), mx)
            s_o, z_o = g.q(divop.outputs[0])
            div_tab = qz.div_table(s_mel, z_mel, s_d, z_d, s_o, z_o, divop.options.get("activation", "none"))
            maxnorm = (den_tab.astype(np.int8), div_tab)
            region = region[3:]
            cur = divop.outputs[0]
            front_out = cur
        if region:
            front_out = ops[j].inputs[0]
            lut = _pwl_table(g, region, cur, front_out, M)
        _expect(list(g.const(ops[j].inputs[1])) == [0, 3, 2, 1], "TRANSPOSE [0,3,2,1] after the frontend")
        cur = ops[j].outputs[0]
        i = j + 1
        if ops[i].name == "STRIDED_SLICE":
            _expect(tuple(t[ops[i].outputs[0]].shape[1:]) == (M, W, 1), "post-frontend STRIDED_SLICE must keep [M, W, 1]")
            cur = ops[i].outputs[0]
            i += 1

        plan = pk.Plan(pk.DTYPE_I8, pk.INPUT_SPECTROGRAM, F * W, F, W, int(t[model.outputs[0]].shape[-1]), meta={"tflite_ops": len(ops)})
        pb = pk.PlanBuilder(plan)
        tens = [pb.tensor(w_mel, np.int8), pb.tensor(bias, np.int32), pb.tensor(mult, np.int32), pb.tensor(shift, np.int32)]
        norm_lut = lut if maxnorm is not None else None  # with the max normalisation the per-channel table sits behind the DIV, not behind the mixer
        if maxnorm is not None:
            lut = None
        if lut is not None:
            tens.append(pb.tensor(lut, np.int8))
        mel_tile = pick_tile(1, W)
        mfma_mel = fuse and mel_tile is not None and M % 16 == 0
        # production plans quantise inside the mel mixer's load (i8_mel_mfma_kernel<QIN>: one pass over the float32 spectrogram, no int8
        # copy of it in HBM); keep_all plans keep QUANTIZE as its own operator so that its tensor can be compared
        quant_in_mel = mfma_mel and not keep_all and M == 64 and Kp % 64 == 0 and W % 64 == 0
        v_q = pk.SLOT_INPUT
        if not quant_in_mel:
            v_q = pb.value(W * Kp)
            pb.op(pk.I8_QUANT, pk.SLOT_INPUT, v_q, p=[F, W, Kp, q_zp, fill_value], f=[q_scale], name=f"t{mel.inputs[0]}",
                  out_shape=(W, Kp), out_dtype="int8")
        v = pb.value(M * W)
        if mfma_mel:
            zero = pb.tensor(np.zeros(4, np.int32), np.int32)
            p = [1, W, Kp, 1, 1, F if quant_in_mel else 0, 1, W, 0, 0, 0, 0, 0, 0, M, z_mel, lo, hi, *([0] * 11), 0, 1, *mel_tile, int(lut is not None),
                 0, int(quant_in_mel), q_zp, fill_value]
            tt = [zero, zero, zero, zero, pb.tensor(pack_i8_fragments(w_mel), np.int8), tens[1], tens[2], tens[3]]
            if lut is not None:
                tt.append(tens[4])
            pb.op(pk.I8_DWPW, v_q, v, p=p, t=tt, f=[q_scale], name=f"t{cur}", out_shape=(M, W, 1), out_dtype="int8")
        else:
            pb.op(pk.I8_MEL, v_q, v, p=[W, Kp, M, z_mel, lo, hi, int(lut is not None)], t=tens, name=f"t{cur}",
                  out_shape=(M, W, 1), out_dtype="int8")
        if maxnorm is not None:
            _expect(W % 4 == 0, "max normalisation kernel: map width must be a multiple of 4")
            pb.plan.ops[-1].name = "mel_mixer"  # (its [W][M] graph tensor is not compared by name: the plan keeps [M][W])
            v_n = pb.value(M * W)
            tt = [pb.tensor(maxnorm[0], np.int8), pb.tensor(maxnorm[1], np.int8)] + ([pb.tensor(norm_lut, np.int8)] if norm_lut is not None else [])
            pb.op(pk.I8_MAXNORM, v, v_n, p=[M, W, int(norm_lut is not None)], t=tt, name=f"t{cur}", out_shape=(M, W, 1), out_dtype="int8")
            v = v_n
    val = {cur: v}
    shape = {cur: (M, W, 1)}
    tail_blocks: list[dict] = []  # fused DW+PW blocks in graph order (candidates for the fused tail kernel)
    tail_head: dict = {}

    def conv_common(op):
        s_i, z_i = g.q(op.inputs[0])
        s_o, z_o = g.q(op.outputs[0])
        wt_ = t[op.inputs[1]]
        n = int(wt_.shape[3] if op.name == "DEPTHWISE_CONV_2D" else wt_.shape[0])
        mu, sh = qz.channel_multipliers(s_i, wt_.scale, s_o, n)
        a_lo, a_hi = qz.activation_bounds(op.options["activation"], s_o, z_o)
        b = g.const(op.inputs[2]).astype(np.int64) if len(op.inputs) > 2 and op.inputs[2] >= 0 else np.zeros(n, np.int64)
        _expect(op.options["padding"] == "SAME" and op.options.get("dilation_w", 1) == 1, "SAME, undilated convolutions")
        return s_i, z_i, s_o, z_o, wt_, n, mu, sh, a_lo, a_hi, b

    # ---- backbone ---------------------------------------------------------------------------
    while i < len(ops):
        op = ops[i]
        src = op.inputs[0]
        if op.name == "CONV_2D" and tuple(t[op.inputs[1]].shape[1:3]) == (3, 3):
            H, Wd, Cin = shape[src]
            _expect(Cin == 1, "3x3 CONV_2D is only lowered for the single-channel stem")
            s_i, z_i, s_o, z_o, wt_, Cout, mu, sh, a_lo, a_hi, b = conv_common(op)
            sh_, sw_ = op.options["stride_h"], op.options["stride_w"]
            OH, pt, _ = same_pad(H, 3, sh_)
            OW, pl, _ = same_pad(Wd, 3, sw_)
            w = np.transpose(wt_.data[:, :, :, 0], (1, 2, 0))  # [3][3][Cout]
            _expect_acc_range(w, b - z_i * w.astype(np.int64).sum(axis=(0, 1)), (0, 1), f"stem conv of operator #{op.index}", mu, sh)
            # front block: stem -> depthwise stride 2 -> pointwise in one kernel
            d_op = ops[i + 1] if i + 1 < len(ops) else None
            p_op = ops[i + 2] if i + 2 < len(ops) else None
            if (fuse and d_op is not None and p_op is not None and d_op.name == "DEPTHWISE_CONV_2D" and p_op.name == "CONV_2D"
                    and g.consumers.get(op.outputs[0], []) == [d_op.index] and g.consumers.get(d_op.outputs[0], []) == [p_op.index]
                    and d_op.inputs[0] == op.outputs[0] and p_op.inputs[0] == d_op.outputs[0]
                    and d_op.options["stride_h"] == 2 and d_op.options["stride_w"] == 2 and (sh_, sw_) == (1, 2) and Cout == 16
                    and tuple(t[p_op.inputs[1]].shape[:3]) == (32, 1, 1)
                    and not (i + 3 < len(ops) and ops[i + 3].name == "ADD" and p_op.outputs[0] in ops[i + 3].inputs)):
                BH, BW = same_pad(OH, 3, 2)[0], same_pad(OW, 3, 2)[0]
                if BH % 8 == 0 and BW % 8 == 0 and H == 2 * BH and Wd == 4 * BW:
                    sd, zd, sdo, zdo, wtd, nd, mud, shd, dlo, dhi, bd = conv_common(d_op)
                    sp, zp_, spo, zpo, wtp, N, mup, shp, plo, phi, bp = conv_common(p_op)
                    bd = bd - zd * wtd.data[0].astype(np.int64).sum(axis=(0, 1))
                    wpw = wtp.data.reshape(N, Cout)
                    bp = bp - zp_ * wpw.astype(np.int64).sum(axis=1)
                    _expect_acc_range(wtd.data[0], bd, (0, 1), f"depthwise conv of operator #{d_op.index}", mud, shd)
                    _expect_acc_range(wpw, bp, 1, f"pointwise conv of operator #{p_op.index}", mup, shp)
                    v = pb.value(BH * BW * N)
                    # (the strip kernels drop the sign term of the rounding shift where a negative result clamps to the zero point anyway)
                    relu_ok = a_lo >= z_o and dlo >= zdo and plo >= zpo
                    cst = front_strip_constants(w, b, mu, sh, z_i, z_o, wtd.data[0], bd, mud, shd, zdo, wpw, bp, mup, shp, zpo) if BW % 16 == 0 and relu_ok else None
                    pb.op(pk.I8_FRONT, val[src], v, p=[H, Wd, Cout, N, BH, BW, z_i, z_o, a_lo, a_hi, zdo, dlo, dhi, zpo, plo, phi, int(cst is not None)],
                          t=[pb.tensor(w, np.int8), pb.tensor(b, np.int32), pb.tensor(mu, np.int32), pb.tensor(sh, np.int32),
                             pb.tensor(wtd.data[0], np.int8), pb.tensor(bd, np.int32), pb.tensor(mud, np.int32), pb.tensor(shd, np.int32),
                             pb.tensor(pack_i8_fragments(wpw), np.int8), pb.tensor(bp, np.int32), pb.tensor(mup, np.int32), pb.tensor(shp, np.int32),
                             pb.tensor(cst, np.int32) if cst is not None else -1],
                          name=f"t{p_op.outputs[0]}", out_shape=(BH, BW, N), out_dtype="int8")
                    val[p_op.outputs[0]], shape[p_op.outputs[0]] = v, (BH, BW, N)
                    i += 3
                    continue
            v = pb.value(OH * OW * Cout)
            pb.op(pk.I8_STEM, val[src], v, p=[H, Wd, Cout, sh_, sw_, 0, OH, OW, pt, pl, z_i, z_o, a_lo, a_hi],
                  t=[pb.tensor(w, np.int8), pb.tensor(b, np.int32), pb.tensor(mu, np.int32), pb.tensor(sh, np.int32)],
                  name=f"t{op.outputs[0]}", out_shape=(OH, OW, Cout), out_dtype="int8")
            val[op.outputs[0]], shape[op.outputs[0]] = v, (OH, OW, Cout)
            i += 1
        elif op.name == "DEPTHWISE_CONV_2D":
            H, Wd, C = shape[src]
            s_i, z_i, s_o, z_o, wt_, n, mu, sh, a_lo, a_hi, b = conv_common(op)
            _expect(tuple(wt_.shape[1:3]) == (3, 3) and n == C and op.options["depth_multiplier"] == 1, "3x3 depthwise, multiplier 1")
            sh_, sw_ = op.options["stride_h"], op.options["stride_w"]
            OH, pt, _ = same_pad(H, 3, sh_)
            OW, pl, _ = same_pad(Wd, 3, sw_)
            nxt = ops[i + 1] if i + 1 < len(ops) else None
            tile = pick_tile(OH, OW)
            cons = g.consumers.get(op.outputs[0], [])
            fusable = (fuse and nxt is not None and nxt.name == "CONV_2D" and cons == [nxt.index] and nxt.inputs[0] == op.outputs[0]
                       and tuple(t[nxt.inputs[1]].shape[1:3]) == (1, 1) and nxt.options["stride_h"] == 1 and nxt.options["stride_w"] == 1
                       and tile is not None and C % 4 == 0 and int(t[nxt.inputs[1]].shape[0]) % 16 == 0)
            if fusable:
                s2, z2, so2, zo2, wt2, Cout, mu2, sh2, lo2, hi2, b2 = conv_common(nxt)
                w2 = wt2.data.reshape(Cout, C)
                b2 = b2 - z2 * w2.astype(np.int64).sum(axis=1)
                bdw = b - z_i * wt_.data[0].astype(np.int64).sum(axis=(0, 1))  # padded taps load zp_in, so the fold is uniform
                _expect_acc_range(w2, b2, 1, f"pointwise conv of operator #{nxt.index}", mu2, sh2)
                _expect_acc_range(wt_.data[0], bdw, (0, 1), f"depthwise conv of operator #{op.index}", mu, sh)
                add_p = [0] * 11
                res_val = pk.SLOT_NONE
                out_t = nxt.outputs[0]
                nn = ops[i + 2] if i + 2 < len(ops) else None
                cons2 = g.consumers.get(nxt.outputs[0], [])
                if nn is not None and nn.name == "ADD" and cons2 == [nn.index] and nxt.outputs[0] in nn.inputs:
                    other = [x for x in nn.inputs if x != nxt.outputs[0]]
                    _expect(len(other) == 1 and other[0] in val and shape[other[0]] == (OH, OW, Cout), "residual ADD operand")
                    s_r, z_r = g.q(other[0])
                    s_a, z_a = g.q(nn.outputs[0])
                    if nn.inputs[0] == other[0]:
                        ap = qz.AddParams(s_r, z_r, so2, zo2, s_a, z_a, nn.options["activation"])
                        add_p = [1, ap.z1, ap.m1, ap.sh1, ap.m2, ap.sh2, ap.mo, ap.sho, ap.zo, ap.amin, ap.amax]
                    else:
                        ap = qz.AddParams(so2, zo2, s_r, z_r, s_a, z_a, nn.options["activation"])
                        add_p = [1, ap.z2, ap.m2, ap.sh2, ap.m1, ap.sh1, ap.mo, ap.sho, ap.zo, ap.amin, ap.amax]
                    res_val = val[other[0]]
                    out_t = nn.outputs[0]
                    i += 1
                v = pb.value(OH * OW * Cout)
                # wide early layers: constant block of the wave-autonomous strip kernel (the residual must be the block input,
                # every per-channel requantisation and the ADD's output one a right shift; tap column 1 never in the padding)
                cst = None
                ow_ = np.arange(OW)
                nw = strip_waves(C, Cout, sh_, OW, bool(add_p[0])) if sh_ == sw_ and (OW != 8 or OH % 2 == 0) else 0
                relu_ok = a_lo >= z_o and (bool(add_p[0]) or lo2 >= zo2)  # see rq_relu in csrc/bn_i8_strip.hip
                if (nw and relu_ok and (not add_p[0] or res_val == val[src])
                        and ((ow_ * sw_ - pl + 1 >= 0) & (ow_ * sw_ - pl + 1 < Wd)).all()):
                    cst = strip_constants(wt_.data[0], bdw, mu, sh, z_o, w2, b2, mu2, sh2, zo2, bool(add_p[0]), nw)
                p = [H, Wd, C, sh_, sw_, 0, OH, OW, pt, pl, z_i, z_o, a_lo, a_hi, Cout, zo2, lo2, hi2, *add_p, 1, 0, *tile, 0, int(cst is not None)]
                tail_blocks.append(dict(op=len(plan.ops), src=val[src], res_is_input=(not add_p[0]) or res_val == val[src], H=H, W=Wd, C=C, N=Cout, sh=sh_, sw=sw_,
                                        OH=OH, OW=OW, pt=pt, pl=pl, z_in=z_i, z_dw=z_o, dw_lo=a_lo, dw_hi=a_hi, z_pw=zo2, pw_lo=lo2, pw_hi=hi2, add=list(add_p),
                                        wd=wt_.data[0], bdw=bdw, mu=mu, sh_dw=sh, w2=w2, b2=b2, mu2=mu2, sh2=sh2, macs=(OH * OW * C * 9, OH * OW * C * Cout)))
                pb.op(pk.I8_DWPW, val[src], v, p=p, in1=res_val,
                      t=[pb.tensor(wt_.data[0], np.int8), pb.tensor(bdw, np.int32), pb.tensor(mu, np.int32), pb.tensor(sh, np.int32),
                         pb.tensor(pack_i8_fragments(w2), np.int8), pb.tensor(b2, np.int32), pb.tensor(mu2, np.int32), pb.tensor(sh2, np.int32),
                         -1, pb.tensor(cst, np.int32) if cst is not None else -1,
                         pb.tensor(add_table(add_p, zo2), np.int8) if cst is not None and add_p[0] else -1],
                      name=f"t{out_t}", out_shape=(OH, OW, Cout), out_dtype="int8")
                val[out_t], shape[out_t] = v, (OH, OW, Cout)
                i += 2
                if fuse and not keep_all:   # (keep_all plans keep every operator's own tensor for the per-tensor tests)
                    _add_mid_op(pb, plan, tail_blocks)
            else:
                _expect_acc_range(wt_.data[0], b - z_i * wt_.data[0].astype(np.int64).sum(axis=(0, 1)), (0, 1), f"depthwise conv of operator #{op.index}", mu, sh)
                v = pb.value(OH * OW * C)
                pb.op(pk.I8_DW, val[src], v, p=[H, Wd, C, sh_, sw_, 0, OH, OW, pt, pl, z_i, z_o, a_lo, a_hi],
                      t=[pb.tensor(wt_.data[0], np.int8), pb.tensor(b, np.int32), pb.tensor(mu, np.int32), pb.tensor(sh, np.int32)],
                      name=f"t{op.outputs[0]}", out_shape=(OH, OW, C), out_dtype="int8")
                val[op.outputs[0]], shape[op.outputs[0]] = v, (OH, OW, C)
                i += 1
        elif op.name == "CONV_2D":
            H, Wd, Cin = shape[src]
            s_i, z_i, s_o, z_o, wt_, Cout, mu, sh, a_lo, a_hi, b = conv_common(op)
            _expect(tuple(wt_.shape[1:3]) == (1, 1) and op.options["stride_h"] == 1 and op.options["stride_w"] == 1, "1x1 stride-1 CONV_2D")
            w = wt_.data.reshape(Cout, Cin)
            b = b - z_i * w.astype(np.int64).sum(axis=1)
            _expect_acc_range(w, b, 1, f"1x1 conv of operator #{op.index}", mu, sh)
            add_p = [0] * 11
            res_val = pk.SLOT_NONE
            out_t = op.outputs[0]
            nxt = ops[i + 1] if i + 1 < len(ops) else None
            cons = g.consumers.get(op.outputs[0], [])
            if nxt is not None and nxt.name == "ADD" and cons == [nxt.index] and op.outputs[0] in nxt.inputs:
                other = [x for x in nxt.inputs if x != op.outputs[0]]
                _expect(len(other) == 1 and other[0] in val and shape[other[0]] == (H, Wd, Cout), "residual ADD operand")
                first_is_res = nxt.inputs[0] == other[0]
                s_r, z_r = g.q(other[0])
                s_a, z_a = g.q(nxt.outputs[0])
                if first_is_res:
                    ap = qz.AddParams(s_r, z_r, s_o, z_o, s_a, z_a, nxt.options["activation"])
                    add_p = [1, ap.z1, ap.m1, ap.sh1, ap.m2, ap.sh2, ap.mo, ap.sho, ap.zo, ap.amin, ap.amax]
                else:  # ADD is commutative in exact integer arithmetic; keep the operand roles
                    ap = qz.AddParams(s_o, z_o, s_r, z_r, s_a, z_a, nxt.options["activation"])
                    add_p = [1, ap.z2, ap.m2, ap.sh2, ap.m1, ap.sh1, ap.mo, ap.sho, ap.zo, ap.amin, ap.amax]
                res_val = val[other[0]]
                out_t = nxt.outputs[0]
                i += 1
            v = pb.value(H * Wd * Cout)
            tile = pick_tile(H, Wd)
            if fuse and tile is not None and Cin % 4 == 0 and Cout % 16 == 0:
                # plain 1x1 convolution (inverted-residual expand / project, embedding conv) on the int8 matrix cores: the fused block kernel
                # without its depthwise stage
                zero = pb.tensor(np.zeros(4, np.int32), np.int32)
                pp = [H, Wd, Cin, 1, 1, 0, H, Wd, 0, 0, 0, 0, 0, 0, Cout, z_o, a_lo, a_hi, *add_p, 0, 0, *tile, 0, 0]
                pb.op(pk.I8_DWPW, val[src], v, p=pp, in1=res_val,
                      t=[zero, zero, zero, zero, pb.tensor(pack_i8_fragments(w), np.int8), pb.tensor(b, np.int32), pb.tensor(mu, np.int32), pb.tensor(sh, np.int32),
                         -1, -1, pb.tensor(add_table(add_p, z_o), np.int8) if add_p[0] else -1],  # the whole ADD as a 64 KB table (i8_pw_wave / i8_pw_lds kernels)
                      name=f"t{out_t}", out_shape=(H, Wd, Cout), out_dtype="int8")
            else:
                pb.op(pk.I8_PW, val[src], v, p=[H * Wd, Cin, Cout, z_o, a_lo, a_hi, *add_p], in1=res_val,
                      t=[pb.tensor(w, np.int8), pb.tensor(b, np.int32), pb.tensor(mu, np.int32), pb.tensor(sh, np.int32)],
                      name=f"t{out_t}", out_shape=(H, Wd, Cout), out_dtype="int8")
            val[out_t], shape[out_t] = v, (H, Wd, Cout)
            i += 1
        elif op.name == "RESHAPE" and len(shape[src]) == 3 and i + 6 < len(ops) and [o.name for o in ops[i + 1 : i + 7]] == ["FULLY_CONNECTED", "RESHAPE", "SOFTMAX", "RESHAPE", "MUL", "SUM"]:
            # attention pooling (reference models/blocks.py:136-159) as the exporter writes it: RESHAPE [P, C] -> FULLY_CONNECTED C -> 1 per
            # position -> RESHAPE -> SOFTMAX over the positions -> RESHAPE -> MUL (broadcast over the channels) -> SUM over the positions:
            # ONE operator (i8_attnpool_kernel); the softmax is table-driven (models/_quant.py: softmax_tables)
            H, Wd, C = shape[src]
            P = H * Wd
            fc, r2, sm, r3, mul, sm_sum = ops[i + 1 : i + 7]
            flat = op.outputs[0]
            _expect(fc.inputs[0] == flat and r2.inputs[0] == fc.outputs[0] and sm.inputs[0] == r2.outputs[0] and r3.inputs[0] == sm.outputs[0]
                    and set(mul.inputs) == {flat, r3.outputs[0]} and sm_sum.inputs[0] == mul.outputs[0], "attention pooling operator chain")
            # the kernel keeps the map, the scores and the softmax bytes in LDS (P C + 2 P + 16 bytes of dynamic LDS beside 16 static ones) and sums
            # up to P exponentials of at most 2^19 in an int32: P <= 4096 (TFLite's kAccumulationIntegerBits = 12 has the same ceiling)
            _expect(g.consumers.get(flat, []) == sorted([fc.index, mul.index]) and C % 4 == 0 and P <= 4096 and P * C + 2 * P + 32 <= 64 * 1024,
                    "attention pooling geometry")
            _expect([int(a) % 3 for a in np.atleast_1d(g.const(sm_sum.inputs[1]))] == [1] and not sm_sum.options.get("keep_dims"), "SUM over the positions")
            wt_ = t[fc.inputs[1]]
            _expect(tuple(wt_.shape) == (1, C) and bool(fc.options.get("keep_num_dims")) and fc.options["activation"] == "none", "score layer C -> 1")
            s_x, z_x = g.q(src)
            _expect(g.q(flat) == (s_x, z_x), "RESHAPE keeps the quantisation")
            s_s, z_s = g.q(fc.outputs[0])
            mu, sh = qz.channel_multipliers(s_x, wt_.scale, s_s, 1)
            w = wt_.data.astype(np.int64).reshape(C)
            b = (g.const(fc.inputs[2]).astype(np.int64).reshape(-1)[0] if len(fc.inputs) > 2 and fc.inputs[2] >= 0 else 0) - z_x * int(w.sum())
            _expect_acc_range(wt_.data.reshape(1, C), np.asarray([b], np.int64), 1, f"attention score operator #{fc.index}", mu, sh)
            s_a, z_a = g.q(sm.outputs[0])
            _expect((round(1.0 / s_a), z_a) == (256, -128) and g.q(r2.outputs[0]) == (s_s, z_s) and g.q(r3.outputs[0]) == (s_a, z_a), "softmax quantisation")
            beta = float(sm.options.get("beta", 1.0))
            s_m, z_m = g.q(mul.outputs[0])
            m_mu, m_sh = qz.quantize_multiplier(float(np.float32(s_x)) * float(np.float32(s_a)) / float(np.float32(s_m)))
            m_lo, m_hi = qz.activation_bounds(mul.options["activation"], s_m, z_m)
            s_o, z_o = g.q(sm_sum.outputs[0])
            o_mu, o_sh = qz.quantize_multiplier(float(np.float32(s_m)) / float(np.float32(s_o)))
            form = {"fixed": 0, "lut": 1}[softmax_form]
            v = pb.value(C)
            pb.op(pk.I8_ATTNPOOL, val[src], v, p=[P, C, int(b), int(mu[0]), int(sh[0]), z_s, form, z_x, z_a, m_mu, m_sh, z_m, m_lo, m_hi, o_mu, o_sh, z_o],
                  t=[pb.tensor(w.astype(np.int8), np.int8), pb.tensor(qz.softmax_tables(s_s, beta, softmax_form), np.int32)],
                  name=f"t{sm_sum.outputs[0]}", out_shape=(C,), out_dtype="int8")
            val[sm_sum.outputs[0]], shape[sm_sum.outputs[0]] = v, (C,)
            i += 7
        elif op.name == "MEAN":
            H, Wd, C = shape[src]
            axes = sorted(int(a) % 4 for a in np.atleast_1d(g.const(op.inputs[1])))
            _expect(axes == [1, 2], "MEAN over the spatial axes")
            s_i, z_i = g.q(src)
            s_o, z_o = g.q(op.outputs[0])
            mu, sh = qz.mean_multiplier(s_i, s_o, H * Wd) if mean_form == "int" else qz.mean_float_params(s_i, s_o)
            v = pb.value(C)
            tail_head.update(mean_op=len(plan.ops), mean_src=val[src], P=H * Wd, C=C, mean_zp_in=z_i, mean_mult=mu, mean_shift=sh, mean_zp_out=z_o)
            pb.op(pk.I8_MEAN, val[src], v, p=[H * Wd, C, z_i, mu, sh, z_o], name=f"t{op.outputs[0]}", out_shape=(C,), out_dtype="int8")
            val[op.outputs[0]], shape[op.outputs[0]] = v, (C,)
            i += 1
        elif op.name == "FULLY_CONNECTED":
            # classifier ([C] -> classes) or a squeeze-excite Dense on the pooled vector ([.., C] -> [.., C']); a LOGISTIC that is the
            # layer's only consumer is folded into it as a 256-entry table (exact: the int8 LOGISTIC is a table in TFLite too)
            Cin = int(shape[src][-1])
            _expect(int(np.prod(shape[src])) == Cin, "FULLY_CONNECTED input must be a vector per chunk")
            s_i, z_i = g.q(src)
            s_o, z_o = g.q(op.outputs[0])
            wt_ = t[op.inputs[1]]
            Cout = int(wt_.shape[0])
            _expect(int(wt_.shape[1]) == Cin, "FULLY_CONNECTED weight shape")
            mu, sh = qz.channel_multipliers(s_i, wt_.scale, s_o, Cout)
            a_lo, a_hi = qz.activation_bounds(op.options["activation"], s_o, z_o)
            b = g.const(op.inputs[2]).astype(np.int64) if len(op.inputs) > 2 and op.inputs[2] >= 0 else np.zeros(Cout, np.int64)
            b = b - z_i * wt_.data.astype(np.int64).sum(axis=1)
            _expect_acc_range(wt_.data, b, 1, f"fully connected operator #{op.index}", mu, sh)
            Kp = (Cin + 3) // 4 * 4
            w_pad = np.zeros((Cout, Kp), np.int8)
            w_pad[:, :Cin] = wt_.data
            nxt = ops[i + 1] if i + 1 < len(ops) else None
            fold_lut = (nxt is not None and nxt.name == "LOGISTIC" and nxt.inputs[0] == op.outputs[0] and g.consumers.get(op.outputs[0], []) == [nxt.index]
                        and i + 2 < len(ops) and ops[i + 2].name != "DEQUANTIZE")
            tt = [pb.tensor(w_pad, np.int8), pb.tensor(b, np.int32), pb.tensor(mu, np.int32), pb.tensor(sh, np.int32)]
            out_t = op.outputs[0]
            if fold_lut:
                s_h, z_h = g.q(nxt.outputs[0])
                tt.append(pb.tensor(qz.logistic_table(s_o, z_o, s_h, z_h), np.int8))
                out_t = nxt.outputs[0]
            v = pb.value(Cout)
            if not fold_lut:
                tail_head.update(fc_op=len(plan.ops), fc_cin=Cin, NC=Cout, fc_zp_out=z_o, fc_lo=a_lo, fc_hi=a_hi, fc_w=wt_.data, fc_b=b, fc_m=mu, fc_s=sh)
            pb.op(pk.I8_FC, val[src], v, p=[Cin, Cout, z_o, a_lo, a_hi, int(fold_lut)], t=tt, name=f"t{out_t}", out_shape=(Cout,), out_dtype="int8")
            val[out_t], shape[out_t] = v, (Cout,)
            i += 2 if fold_lut else 1
        elif op.name == "MUL":
            # squeeze-excite scale: map [H, W, C] times a per-chunk gate [C] (either operand order)
            a_t, b_t = op.inputs
            if len(shape[a_t]) != 3:
                a_t, b_t = b_t, a_t
            _expect(len(shape[a_t]) == 3 and int(np.prod(shape[b_t])) == shape[a_t][2] and shape[a_t][2] % 4 == 0, "MUL must scale a map by a per-channel vector")
            H, Wd, C = shape[a_t]
            s1, z1 = g.q(a_t)
            s2, z2 = g.q(b_t)
            so, zo = g.q(op.outputs[0])
            mu, sh = qz.quantize_multiplier(float(np.float32(s1)) * float(np.float32(s2)) / float(np.float32(so)))
            lo_, hi_ = qz.activation_bounds(op.options["activation"], so, zo)
            v = pb.value(H * Wd * C)
            pb.op(pk.I8_SCALE, val[a_t], v, in1=val[b_t], p=[H * Wd, C, z1, z2, mu, sh, zo, lo_, hi_], name=f"t{op.outputs[0]}", out_shape=(H, Wd, C), out_dtype="int8")
            val[op.outputs[0]], shape[op.outputs[0]] = v, (H, Wd, C)
            i += 1
        elif op.name in ("LOGISTIC", "DEQUANTIZE"):
            # head: [LOGISTIC ->] DEQUANTIZE [-> float32 SOFTMAX] ends the graph
            fc_out = src
            _expect(len(shape[fc_out]) == 1, "the head must follow the classifier")
            Cout = shape[fc_out][0]
            s_fc, z_fc = g.q(fc_out)
            lut_t, s_h, z_h, has = -1, s_fc, z_fc, 0
            if op.name == "LOGISTIC":
                s_h, z_h = g.q(op.outputs[0])
                lut_t, has = pb.tensor(qz.logistic_table(s_fc, z_fc, s_h, z_h), np.int8), 1
                last = op.outputs[0]
                i += 1
            else:
                last = fc_out
            _expect(i < len(ops) and ops[i].name == "DEQUANTIZE" and ops[i].inputs[0] == last, "DEQUANTIZE must end the graph")
            softmax, beta = 0, 1.0
            if i + 1 < len(ops) and ops[i + 1].name == "SOFTMAX":
                _expect(not has and ops[i + 1].inputs[0] == ops[i].outputs[0], "float32 SOFTMAX must read the dequantised classifier output")
                softmax, beta = 1, float(ops[i + 1].options.get("beta", 1.0))
                i += 1
            if not softmax:
                tail_head.update(head_op=len(plan.ops), lut=qz.logistic_table(s_fc, z_fc, s_h, z_h) if has else None, zp_fc=z_fc, zp_head=z_h, s_fc=s_fc, s_head=s_h)
            pb.op(pk.I8_HEAD, val[fc_out], pk.SLOT_SCORES, p=[Cout, z_fc, z_h, has, softmax], f=[s_fc, s_h, beta], t=[lut_t], name=f"t{ops[i].outputs[0]}",
                  out_shape=(Cout,))
            i += 1
            _expect(i == len(ops), "operators after the head")
        else:
            _expect(False, f"operator #{op.index} {op.name} in the backbone")
    if fuse and not keep_all:
        _add_tail_op(pb, plan, tail_blocks, tail_head)
        _tag_scale_pairs(pb)
        _tag_se_gates(pb)
        _tag_pwdw_pairs(pb)
    return pb.finalize(reuse=not keep_all)


def _tag_se_gates(pb: pk.PlanBuilder) -> None:
    """MEAN -> FULLY_CONNECTED (ReLU) -> FULLY_CONNECTED (+ LOGISTIC table): the gate of a squeeze-excite block (reference
    models/blocks.py:27-46).  When each of the three feeds only the next, the library runs them as one kernel per chunk."""
    ops = pb.plan.ops
    for i in range(len(ops) - 2):
        a, f1, f2 = ops[i], ops[i + 1], ops[i + 2]
        if a.kind != pk.I8_MEAN or f1.kind != pk.I8_FC or f2.kind != pk.I8_FC or f1.in0 != a.out or f2.in0 != f1.out:
            continue
        if f1.p[0] != a.p[1] or f2.p[0] != f1.p[1] or f1.p[1] > 256 or a.p[1] > 1024 or a.p[1] % 4:
            continue
        others = [k for k, o in enumerate(ops) if (k != i + 1 and a.out in (o.in0, o.in1)) or (k != i + 2 and f1.out in (o.in0, o.in1))]
        if others or a.out < 0 or f1.out < 0 or f2.out < 0:
            continue
        a.p[pk.TAIL_TAG] = pk.SEGATE_HEAD
        f1.p[pk.TAIL_TAG] = f2.p[pk.TAIL_TAG] = pk.SEGATE_COVERED
        pb._extra_uses.append((i + 2, a.in0))  # the pooled map is read while the gate is written


def _tag_pwdw_pairs(pb: pk.PlanBuilder) -> None:
    """Expand CONV_2D 1x1 followed by the DEPTHWISE_CONV_2D 3x3 of an inverted-residual block (reference models/blocks.py:88-110): when the
    depthwise stage is the only reader of the expanded map the pair is tagged and the library may run it as one kernel (``i8_pwdw_kernel``)
    that never writes the expanded map.  The fused kernel reads the block input while it writes the depthwise output: no slot sharing."""
    ops = pb.plan.ops
    for i in range(len(ops) - 1):
        e, d = ops[i], ops[i + 1]
        if e.kind != pk.I8_DWPW or d.kind != pk.I8_DW or e.p[29] or e.p[30] or e.p[18] or e.p[34] or e.p[36] or d.in0 != e.out or e.out < 0:
            continue
        if e.p[pk.TAIL_TAG] or d.p[pk.TAIL_TAG] or e.p[pk.OP_PATH] != d.p[pk.OP_PATH]:
            continue
        if d.p[2] != e.p[14] or (d.p[0], d.p[1]) != (e.p[6], e.p[7]) or d.p[10] != e.p[15]:
            continue
        readers = [k for k, o in enumerate(ops) if k != i + 1 and e.out in (o.in0, o.in1)]
        writers = [k for k, o in enumerate(ops) if o.out == e.out]
        if readers or writers != [i]:
            continue
        e.p[pk.TAIL_TAG] = pk.PWDW8_HEAD
        d.p[pk.TAIL_TAG] = pk.PWDW8_COVERED
        if e.in0 >= 0:
            pb._extra_uses.append((i + 1, e.in0))


def _tag_scale_pairs(pb: pk.PlanBuilder) -> None:
    """Squeeze-excite MUL followed by the projection 1x1 convolution (reference models/blocks.py:27-46,104-118): when that
    convolution is the only reader of the scaled map the pair is tagged, and the library applies the gate while the convolution
    loads its input (``i8_pw_wave_kernel`` up to 256 channels, ``i8_pw_lds_kernel`` for 384 and 768) instead of writing the scaled map out and reading it back."""
    ops = pb.plan.ops
    for i in range(len(ops) - 1):
        a, b = ops[i], ops[i + 1]
        if a.kind != pk.I8_SCALE or b.kind != pk.I8_DWPW or b.p[29] or b.p[30] or b.p[34] or b.p[36] or b.in0 != a.out:
            continue
        if b.p[2] != a.p[1] or b.p[0] * b.p[1] != a.p[0] or b.p[2] > 768 or (b.p[3], b.p[4]) != (1, 1):
            continue
        readers = [k for k, o in enumerate(ops) if k != i + 1 and a.out in (o.in0, o.in1)]
        if readers or b.in1 == a.out:
            continue
        a.p[pk.TAIL_TAG] = pk.SCALE_HEAD
        b.p[pk.TAIL_TAG] = pk.SCALE_COVERED
        pb._extra_uses.append((i + 1, a.in0))  # the fused kernel reads the unscaled map and the gate while it writes the convolution's output
        pb._extra_uses.append((i + 1, a.in1))


def _add_mid_op(pb, plan, blocks: list[dict]) -> None:
    """Append the fused operator for the three blocks of stage 2 of the shipped topology (csrc/bn_i8_tail2.hip: i8_mid2_kernel) right behind
    them: a stride-2 block 32 -> 64 channels onto a 16 x 32 map (taps from memory), then two residual blocks of 64 channels whose maps
    stay in LDS (two chunks per workgroup), the last map written back.  The three operators stay in the plan tagged MID_COVERED (skipped
    while the fused operator runs); the fused operator is tagged MID_OP (skipped when option ``i8_mid`` is off or the library refuses its
    LDS plan).  Same constants and descriptor words as the fused tail's second form (``tail2_constants`` without head)."""
    if len(blocks) < 3:
        return
    chain = blocks[-3:]
    a, b, c = chain
    if (a["C"], a["N"], a["sh"], a["H"], a["W"], bool(a["add"][0])) != (32, 64, 2, 32, 64, False):
        return
    for blk in (b, c):
        if (blk["C"], blk["N"], blk["sh"], blk["H"], blk["W"], bool(blk["add"][0])) != (64, 64, 1, 16, 32, True) or not blk["res_is_input"]:
            return
    ops_idx = [blk["op"] for blk in chain]
    if ops_idx != list(range(ops_idx[0], ops_idx[0] + 3)) or ops_idx[-1] != len(plan.ops) - 1:
        return
    for x, y in zip(chain, chain[1:]):
        if y["src"] != plan.ops[x["op"]].out:
            return
    if any(plan.ops[k].p[pk.TAIL_TAG] for k in ops_idx):
        return
    packed = tail2_constants(chain, None)
    if packed is None:
        return
    cst, desc = packed
    for k in ops_idx:
        plan.ops[k].p[pk.TAIL_TAG] = pk.MID_COVERED
    last = plan.ops[ops_idx[-1]]
    pb.op(pk.I8_MID, a["src"], last.out,
          p=[a["H"] * a["W"] * a["C"], sum(blk["macs"][1] for blk in chain), sum(blk["macs"][0] for blk in chain), 0, 0, 3,
             a["H"], a["W"], a["C"], c["OH"] * c["OW"], c["N"], *([0] * (pk.TAIL_TAG - 11)), pk.MID_OP],
          t=[pb.tensor(cst, np.int32), pb.tensor(desc, np.int32)], name=last.name, out_shape=last.out_shape, out_dtype="int8")


def _add_tail_op(pb, plan, blocks: list[dict], head: dict) -> None:
    """Append the fused tail operator (csrc/bn_i8_tail.hip) when the plan ends in the shipped topology's back half: a stride-2 block
    from 64 channels onto a 16-wide map, then blocks of 128 / 256 channels on 16- and 8-wide maps (residual ADDs against the block
    input), MEAN, FULLY_CONNECTED, head — as consecutive plan operators.  The per-block operators stay in the plan tagged p[TAIL_TAG]
    = 1 (skipped while the tail kernel runs them), the tail operator is tagged 2 (skipped when the option ``i8_tail`` is off or the
    library finds that the maps do not fit its LDS plan)."""
    if not blocks or not {"mean_op", "fc_op", "head_op"} <= set(head):
        return
    start = next((i for i, b in enumerate(blocks) if (b["C"], b["sh"], b["OW"]) == (64, 2, 16) and not b["add"][0]), None)
    if start is None:
        return
    chain = blocks[start:]
    ops_idx = [b["op"] for b in chain] + [head["mean_op"], head["fc_op"], head["head_op"]]
    if ops_idx != list(range(ops_idx[0], ops_idx[0] + len(ops_idx))) or ops_idx[-1] != len(plan.ops) - 1 or len(chain) > 8:
        return
    if not all(b["res_is_input"] for b in chain) or head["fc_cin"] != head["C"] or head["mean_src"] != plan.ops[chain[-1]["op"]].out:
        return
    for a, b in zip(chain, chain[1:]):  # each block reads the previous block's output
        if b["src"] != plan.ops[a["op"]].out or (b["H"], b["W"], b["C"]) != (a["OH"], a["OW"], a["N"]):
            return
    packed = tail_constants(chain, head)
    if packed is None:
        return
    cst, desc = packed
    packed2 = tail2_constants(chain, head)  # the same blocks for i8_tail2_kernel (depthwise stage on the matrix cores), when they take its forms
    first = chain[0]
    dw_macs = sum(b["macs"][0] for b in chain)
    pw_macs = sum(b["macs"][1] for b in chain)
    for k in ops_idx:
        plan.ops[k].p[pk.TAIL_TAG] = pk.TAIL_COVERED
    pb.op(pk.I8_TAIL, first["src"], pk.SLOT_SCORES,
          p=[first["H"] * first["W"] * first["C"], pw_macs, dw_macs, head["P"] * head["C"] + head["C"] * head["NC"], head["NC"], len(chain),
             first["H"], first["W"], first["C"], head["P"], head["C"], *([0] * (pk.TAIL_TAG - 11)), pk.TAIL_OP],
          t=[pb.tensor(cst, np.int32), pb.tensor(desc, np.int32)] + ([pb.tensor(packed2[0], np.int32), pb.tensor(packed2[1], np.int32)] if packed2 is not None else []),
          f=[head["s_fc"], head["s_head"]], name="tail", out_shape=(head["NC"],))
