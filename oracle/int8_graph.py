"""Integer-exact numpy interpreter for the reference's INT8 `.tflite` graphs.

ORACLE — test infrastructure only (see oracle/__init__.py).  **Parity unpinned** against a
real ``tf.lite.Interpreter`` (TensorFlow cannot be imported here): the arithmetic below
restates the published TFLite *reference kernels* (tensorflow 2.19.0, requirements.txt:2,
not vendored in the reference) and is anchored on the reference's call sites
(birdnet_stm32/models/runners.py:51-95 — ``TFLiteRunner``: set_tensor / invoke /
get_tensor with float32 I/O; birdnet_stm32/conversion/quantize.py:131-152 — full-integer
per-channel PTQ with float I/O).  The graph itself (every scale, zero point, stride,
padding and fused activation) is read from the flatbuffer, nothing is hard-coded.

Conventions restated:
* QUANTIZE   q = clamp(round_half_away(x / s) + zp)           (float32 divide)
* CONV / DW / FC   acc32 = sum((x - zp_x) * w) + bias ; per-channel
      M = double(s_x) * double(s_w[c]) / double(s_y)  ->  QuantizeMultiplier (Q31 mantissa, exponent)
      y = clamp(MultiplyByQuantizedMultiplier(acc, M0, shift) + zp_y, act_min, act_max)
* MultiplyByQuantizedMultiplier = RoundingDivideByPOT(SaturatingRoundingDoublingHighMul(acc << left, M0), right)
* ADD   left_shift 20, twice_max = 2 max(s1, s2), inputs rescaled by s_i / twice_max, sum rescaled by
      twice_max / (2^20 s_y), broadcasting allowed
* MEAN  int32 sum - N zp_x ; multiplier (s_x / s_y) with the 1/N folded in as in reduce.h
* LOGISTIC  256-entry float32 LUT ; DEQUANTIZE (q - zp) * s
* MUL   zo + MBQM((q1 - z1)(q2 - z2), s1 s2 / so), broadcasting ; SOFTMAX only in float32 behind DEQUANTIZE
* REDUCE_MAX  max of the int8 values (input and output share scale and zero point, as the converter guarantees)
* DIV   (reference kernel div.h, int8): x1 = q1 - z1, x2 = q2 - z2 (signs flipped so that x2 > 0); 1 / x2 by gemmlowp's fixed-point
      Newton-Raphson (GetReciprocal: x2 normalised to [1, 2), three iterations from 48/17 - 32/17 d, Q0.31 result + exponent);
      quotient = SRDHM(x1 << headroom(x1), reciprocal), then MBQM with s1 / (s2 so) and the collected shifts, + zo, clamp.
      A zero divisor (undefined in TFLite: debug assertion) is taken as 1 here — a silent chunk divides 0 by it
* TRANSPOSE / STRIDED_SLICE / SHAPE / PACK / FILL / CONCATENATION / RESHAPE are exact data movement; PAD fills with the zero point
      (real 0.0).  Convolutions: SAME (TensorFlow's asymmetric padding) or VALID (the raw frontend's 1 x 16 filterbank).
"""

from __future__ import annotations

import math

import numpy as np

INT32_MIN = -(1 << 31)
INT32_MAX = (1 << 31) - 1


# ----------------------------------------------------------------------------- fixed point
def round_half_away(x):
    """C ``round()``: halves away from zero (numpy's ``rint`` rounds halves to even)."""
    x = np.asarray(x)
    t = np.trunc(x)
    return np.where(np.abs(x - t) >= 0.5, t + np.sign(x), t).astype(x.dtype)


def quantize_multiplier(real: float) -> tuple[int, int]:
    """``QuantizeMultiplier``: real = M0 * 2^(shift-31) with M0 in [2^30, 2^31)."""
    if real == 0.0:
        return 0, 0
    frac, shift = math.frexp(real)
    q = int(math.floor(abs(frac) * (1 << 31) + 0.5)) * (1 if frac >= 0 else -1)
    if q == (1 << 31):
        q //= 2
        shift += 1
    if shift < -31:
        return 0, 0
    if shift > 30:
        return INT32_MAX, 30
    return q, shift


def srdhm(a, b):
    """``SaturatingRoundingDoublingHighMul`` on int64 arrays holding int32 values."""
    a = np.asarray(a, dtype=np.int64)
    b = np.asarray(b, dtype=np.int64)
    ab = a * b
    nudge = np.where(ab >= 0, 1 << 30, 1 - (1 << 30))
    v = ab + nudge
    res = np.sign(v) * (np.abs(v) >> 31)  # C++ integer division truncates toward zero
    overflow = (a == INT32_MIN) & (b == INT32_MIN)
    return np.where(overflow, INT32_MAX, res)


def rounding_divide_by_pot(x, exponent):
    """``RoundingDivideByPOT``: arithmetic shift with round-half-away-from-zero."""
    x = np.asarray(x, dtype=np.int64)
    exponent = np.asarray(exponent, dtype=np.int64)
    mask = (np.int64(1) << exponent) - 1
    remainder = x & mask
    threshold = (mask >> 1) + (x < 0)
    return (x >> exponent) + (remainder > threshold)


def mbqm(x, multiplier, shift):
    """``MultiplyByQuantizedMultiplier(x, M0, shift)`` (shift > 0 = left shift)."""
    shift = np.asarray(shift, dtype=np.int64)
    left = np.maximum(shift, 0)
    right = np.maximum(-shift, 0)
    return rounding_divide_by_pot(srdhm(np.asarray(x, dtype=np.int64) << left, multiplier), right)


def _sat_mul_pot(x, exponent: int):
    """gemmlowp ``SaturatingRoundingMultiplyByPOT<exponent>`` for exponent > 0: x * 2^exponent, saturated to int32."""
    x = np.asarray(x, dtype=np.int64)
    return np.clip(x << exponent, INT32_MIN, INT32_MAX)


def one_over_one_plus_x_for_x_in_0_1(a):
    """gemmlowp fixedpoint.h ``one_over_one_plus_x_for_x_in_0_1``: a = x in [0, 1) as Q0.31 raw -> 1 / (1 + x) as Q0.31 raw.

    Newton-Raphson division on the half denominator d = (1 + x) / 2 in [1/2, 1): x0 = 48/17 - 32/17 d (Q2.29), three steps
    x <- x + x (1 - d x), result x / 2.  Products of FixedPoint<.., A> and FixedPoint<.., B> are SRDHM of the raw values with A + B
    integer bits; Rescale multiplies by a saturating power of two."""
    a = np.asarray(a, dtype=np.int64)
    s = a + INT32_MAX                                   # RoundingHalfSum(a, F0::One()), One() = 2^31 - 1
    half_den = (s + np.where(s >= 0, 1, -1)) // 2       # truncating: s >= 0 here
    x = 1515870810 + srdhm(half_den, -1010580540)       # Q2.29: 48/17 + d * (-32/17)
    one_q2 = 1 << 29
    for _ in range(3):
        hdx = srdhm(half_den, x)                        # Q2.29
        x = x + _sat_mul_pot(srdhm(x, one_q2 - hdx), 2)  # x (1 - d x) is Q4.27 -> Rescale<2>
    return _sat_mul_pot(x, 1)                           # ExactMulByPot<-1> (Q1.30 view of the same raw) -> Rescale<0>


def count_leading_zeros32(x):
    x = np.asarray(x, dtype=np.int64) & 0xFFFFFFFF
    n = np.zeros(x.shape, np.int64)
    for sh in (16, 8, 4, 2, 1):
        big = x >> sh
        take = big != 0
        x = np.where(take, big, x)
        n = n + np.where(take, sh, 0)
    return np.where(x == 0, 32, 31 - n)


def count_leading_sign_bits32(x):
    """TFLite ``CountLeadingSignBits``: redundant sign bits of an int32 (31 for 0)."""
    x = np.asarray(x, dtype=np.int64)
    neg = x < 0
    mag = np.where(neg, 2 * (-x) - 1, x)
    return np.where(x == INT32_MIN, 0, np.where(neg, count_leading_zeros32(mag), count_leading_zeros32(mag) - 1))


def get_reciprocal(x):
    """TFLite common.h ``GetReciprocal(x, 31, &num_bits_over_unit)`` for x > 0: (Q0.31 reciprocal of x normalised to [1, 2), exponent)."""
    x = np.asarray(x, dtype=np.int64)
    lz = count_leading_zeros32(x)
    shifted = ((x << lz) & 0xFFFFFFFF) - (1 << 31)
    return one_over_one_plus_x_for_x_in_0_1(shifted), 31 - lz


def exp_on_interval_between_negative_one_quarter_and_0_excl(a):
    """gemmlowp fixedpoint.h, FixedPoint<int32, 0>: exp(a) for a in [-1/4, 0) as Q0.31 raw — a fourth-order Taylor polynomial around -1/8."""
    a = np.asarray(a, dtype=np.int64)
    constant_term, one_third = 1895147668, 715827883        # exp(-1/8), 1/3
    x = a + (1 << 28)                                       # + 1/8
    x2 = srdhm(x, x)
    x3 = srdhm(x2, x)
    x4 = srdhm(x2, x2)
    x4_over_4 = rounding_divide_by_pot(x4, 2)
    poly = rounding_divide_by_pot(srdhm(x4_over_4 + x3, one_third) + x2, 1)
    return constant_term + srdhm(constant_term, x + poly)     # exp(-1/8) (1 + x + x^2/2 + x^3/6 + x^4/24)


def exp_on_negative_values(a, integer_bits: int):
    """gemmlowp ``exp_on_negative_values`` for FixedPoint<int32, integer_bits> (a <= 0) -> Q0.31 raw: the argument modulo 1/4 through the
    polynomial above, the multiples of 1/4 through a barrel shifter of constants exp(-2^k), k = -2 .. integer_bits - 1."""
    a = np.asarray(a, dtype=np.int64)
    frac = 31 - integer_bits
    quarter = 1 << (frac - 2)
    a_mod = (a & (quarter - 1)) - quarter
    result = exp_on_interval_between_negative_one_quarter_and_0_excl(a_mod << integer_bits)   # Rescale<0>: exact (|a_mod| < 2^(frac - 2))
    remainder = a_mod - a
    for exponent, mult in ((-2, 1672461947), (-1, 1302514674), (0, 790015084), (1, 290630308), (2, 39332535), (3, 720401), (4, 242)):
        if exponent >= integer_bits:
            break
        hit = (remainder & (1 << (frac + exponent))) != 0
        result = np.where(hit, srdhm(result, mult), result)
    return np.where(a == 0, INT32_MAX, result)


def exp_on_negative_values_q5(a):
    """... for FixedPoint<int32, 5> (the int8 SOFTMAX's scaled differences)."""
    return exp_on_negative_values(a, 5)


def logistic_fixed_q4(a):
    """gemmlowp ``logistic`` for FixedPoint<int32, 4> raw -> Q0.31 raw: 1 / (1 + exp(-|a|)) through ``exp_on_negative_values`` and the
    Newton-Raphson reciprocal, mirrored for negative arguments (One() - result), exactly 1/2 at zero."""
    a = np.asarray(a, dtype=np.int64)
    pos = one_over_one_plus_x_for_x_in_0_1(exp_on_negative_values(-np.abs(a), 4))
    return np.where(a == 0, 1 << 30, np.where(a > 0, pos, INT32_MAX - pos))


def logistic_int8_fixed(x, s_in: float, zp_in: int):
    """int8 LOGISTIC as TFLite's fixed-point kernels compute it (reference_integer_ops/logistic.h, registered as
    ``Register_LOGISTIC_FIXED_POINT_OPT`` and the form TFLite-Micro runs; output 1 / 256, -128): the input rescaled to Q4.27 by a
    quantised multiplier of ``s_in 2^27``, saturation outside the input radius, gemmlowp's ``logistic``, RoundingDivideByPOT by 23.
    The interpreter's DEFAULT int8 LOGISTIC is the float32 table of ``Int8Interpreter.logistic_lut``.  Restated from the published
    source, **parity unpinned**."""
    x = np.asarray(x, dtype=np.int64) - int(zp_in)
    mult, left = quantize_multiplier(float(np.float32(s_in)) * float(1 << (31 - 4)))
    if left < 0:
        raise ValueError("LOGISTIC input multiplier below one")
    radius = int(np.floor(1.0 * ((1 << 4) - 1) * (1 << (31 - 4)) / (1 << left)))
    inside = (x > -radius) & (x < radius)
    q4 = mbqm(np.where(inside, x, 0), mult, left)
    y = rounding_divide_by_pot(logistic_fixed_q4(q4), 31 - 8) - 128
    y = np.clip(y, -128, 127)
    return np.where(x <= -radius, -128, np.where(x >= radius, 127, y)).astype(np.int8)


def softmax_fixed_params(s_in: float, beta: float) -> tuple[int, int, int]:
    """TFLite ``PreprocessSoftmaxScaling`` + ``CalculateInputRadius`` for 5 integer bits: (input multiplier, left shift, diff_min)."""
    real = min(float(beta) * float(np.float32(s_in)) * (1 << (31 - 5)), (1 << 31) - 1.0)
    mult, shift = quantize_multiplier(real)
    if shift < 0:
        raise ValueError("softmax input multiplier below one")
    radius = int(np.floor(1.0 * ((1 << 5) - 1) * (1 << (31 - 5)) / (1 << shift)))
    return mult, shift, -radius


def softmax_int8_fixed(x, s_in: float, beta: float = 1.0):
    """int8 SOFTMAX over the last axis as TFLite's REFERENCE kernel computes it (softmax.h: gemmlowp fixed point; output scale 1 / 256,
    zero point -128).  Restated from the published source, **parity unpinned**."""
    x = np.asarray(x, dtype=np.int64)
    mult, shift, diff_min = softmax_fixed_params(s_in, beta)
    diff = x - x.max(axis=-1, keepdims=True)
    live = diff >= diff_min
    e = exp_on_negative_values_q5(srdhm(np.where(live, diff, 0) << shift, mult))
    total = np.where(live, rounding_divide_by_pot(e, 12), 0).sum(axis=-1, keepdims=True)       # FixedPoint<int32, 12>
    lz = count_leading_zeros32(total)
    over = 12 - lz
    inv = one_over_one_plus_x_for_x_in_0_1(((total << lz) & 0xFFFFFFFF) - (1 << 31))
    y = rounding_divide_by_pot(srdhm(inv, e), over + 31 - 8) - 128
    return np.where(live, np.clip(y, -128, 127), -128).astype(np.int8)


def softmax_int8_lut(x, s_in: float, beta: float = 1.0, s_out: float = 1.0 / 256.0, zp_out: int = -128):
    """int8 SOFTMAX over the last axis as TFLite's OPTIMIZED kernel computes it (optimized_ops::Softmax with SoftmaxParams::table, a
    float32 table of exp(-scale beta d), d = max - x; the sum runs over the row in index order).  Restated from the published source,
    **parity unpinned**; numpy's float32 exp stands in for libm's expf."""
    x = np.asarray(x, dtype=np.int64)
    table = np.exp((np.float32(-float(np.float32(s_in)) * float(beta)) * np.arange(256, dtype=np.float32)).astype(np.float32)).astype(np.float32)
    d = x.max(axis=-1, keepdims=True) - x
    e = table[d]
    total = np.zeros(e.shape[:-1] + (1,), np.float32)
    for j in range(e.shape[-1]):                      # sequential float32 accumulation, like the kernel's loop
        total = (total + e[..., j : j + 1]).astype(np.float32)
    inv = (np.float32(1.0) / (total * np.float32(s_out)).astype(np.float32)).astype(np.float32)
    y = round_half_away((e * inv).astype(np.float32)).astype(np.int64) + zp_out
    return np.clip(y, -128, 127).astype(np.int8)


def mean_int8_float(x, axes, keep_dims: bool, n: int, s_in: float, zp_in: int, s_out: float, zp_out: int):
    """int8 MEAN as TFLite's float-arithmetic ``QuantizedMeanOrSum`` computes it (reduce.h, ``compute_sum = false``): the int32 sum of the raw
    bytes, then in float32 ``mean = sum / n``, ``scale = s_in / s_out``, ``bias = -zp_in * scale``, ``round(mean * scale + bias) + zp_out``
    (TfLiteRound: half away from zero), clamped to int8.  Restated from the published source, **parity unpinned**."""
    total = np.asarray(x, dtype=np.int64).sum(axis=axes, keepdims=keep_dims)
    scale = np.float32(np.float32(s_in) / np.float32(s_out))
    bias = np.float32(np.float32(-zp_in) * scale)
    mean = (total.astype(np.float32) / np.float32(n)).astype(np.float32)
    y = round_half_away((mean * scale + bias).astype(np.float32)).astype(np.int64) + int(zp_out)
    return np.clip(y, -128, 127).astype(np.int8)


def activation_range(act: str, scale: float, zp: int, qmin=-128, qmax=127) -> tuple[int, int]:
    """``CalculateActivationRangeQuantized`` for int8 outputs."""
    s = np.float32(scale)

    def q(v):
        return int(zp + int(round_half_away(np.float32(v) / s)))

    if act == "relu":
        return max(qmin, q(0.0)), qmax
    if act == "relu6":
        return max(qmin, q(0.0)), min(qmax, q(6.0))
    if act == "relu_n1_to_1":
        return max(qmin, q(-1.0)), min(qmax, q(1.0))
    if act == "none":
        return qmin, qmax
    raise ValueError(act)


def _per_channel_multipliers(s_in: float, w_scales: np.ndarray, s_out: float, n: int):
    scales = np.asarray(w_scales, dtype=np.float32)
    if scales.size == 1:
        scales = np.repeat(scales, n)
    mult = np.zeros(n, np.int64)
    shift = np.zeros(n, np.int64)
    for c in range(n):
        real = float(np.float32(s_in)) * float(scales[c]) / float(np.float32(s_out))
        mult[c], shift[c] = quantize_multiplier(real)
    return mult, shift


def _same(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


# ----------------------------------------------------------------------------- interpreter
class Int8Interpreter:
    """Executes a decoded ``TfliteModel`` on float32 batches, like ``TFLiteRunner.predict``."""

    def __init__(self, model, softmax_form: str = "fixed", mean_form: str = "int", logistic_form: str = "lut"):
        """``softmax_form``: which published form an int8 SOFTMAX takes — ``'fixed'`` TFLite's reference kernel (gemmlowp fixed point),
        ``'lut'`` its optimized kernel (float32 exponent table).  Nothing here can tell which one the reference's interpreter runs.

        ``mean_form``: int8 MEAN — ``'int'`` the integer form with the count folded into the multiplier (reduce.h; the default here and
        what the device runs), ``'float'`` the float-arithmetic ``QuantizedMeanOrSum`` (sum in int32, ``round(sum / n * (s_in / s_out) -
        zp_in s_in / s_out) + zp_out`` in float32).  ``logistic_form``: int8 LOGISTIC — ``'lut'`` the float32 table of the
        builtin kernel (default), ``'fixed'`` gemmlowp fixed point (``logistic_int8_fixed``).  tests/test_oracle_pinning.py counts the
        bytes and top-1 labels that depend on the choice for the shipped checkpoint."""
        if softmax_form not in ("fixed", "lut"):
            raise ValueError("softmax_form must be 'fixed' or 'lut'")
        if mean_form not in ("int", "float"):
            raise ValueError("mean_form must be 'int' or 'float'")
        if logistic_form not in ("lut", "fixed"):
            raise ValueError("logistic_form must be 'lut' or 'fixed'")
        self.softmax_form, self.mean_form, self.logistic_form = softmax_form, mean_form, logistic_form
        self._init(model)

    def _init(self, model):
        self.model = model
        self._prep: dict[int, dict] = {}

    # -- helpers ---------------------------------------------------------------
    def _q(self, ti):
        t = self.model.tensors[ti]
        return float(t.scale[0]), int(t.zero_point[0])

    def _value(self, env, ti):
        if ti in env:
            return env[ti]
        t = self.model.tensors[ti]
        if t.data is None:
            raise KeyError(f"tensor {ti} ({t.name}) has no value")
        return t.data

    # -- ops ------------------------------------------------------------------
    def _conv(self, op, env, depthwise: bool):
        x = self._value(env, op.inputs[0]).astype(np.int64)
        wt = self.model.tensors[op.inputs[1]]
        w = wt.data.astype(np.int64)
        bias = self.model.tensors[op.inputs[2]].data.astype(np.int64) if len(op.inputs) > 2 and op.inputs[2] >= 0 else None
        s_in, zp_in = self._q(op.inputs[0])
        s_out, zp_out = self._q(op.outputs[0])
        o = op.options
        if o["padding"] not in ("SAME", "VALID") or o.get("dilation_w", 1) != 1 or o.get("dilation_h", 1) != 1:
            raise ValueError("only undilated convolutions occur in the reference graphs")
        sh, sw = o["stride_h"], o["stride_w"]
        B, H, W, Cin = x.shape
        if depthwise:
            _, kh, kw, cout = w.shape
            if o.get("depth_multiplier", 1) != 1:
                raise ValueError("depth_multiplier != 1")
        else:
            cout, kh, kw, _ = w.shape
        key = op.index
        if key not in self._prep:
            mult, shift = _per_channel_multipliers(s_in, wt.scale, s_out, cout)
            self._prep[key] = {"mult": mult, "shift": shift, "act": activation_range(o["activation"], s_out, zp_out)}
        p = self._prep[key]
        if o["padding"] == "SAME":
            oh, pt, pb = _same(H, kh, sh)
            ow, pl, pr = _same(W, kw, sw)
        else:  # VALID (the raw frontend's 1 x 16 filterbank, reference models/frontend.py:147-155): no padding, floor((in - k) / s) + 1 outputs
            oh, pt, pb = (H - kh) // sh + 1, 0, 0
            ow, pl, pr = (W - kw) // sw + 1, 0, 0
        xc = x - zp_in  # padded cells contribute (zp - zp) = 0, as the reference kernels skip them
        xp = np.pad(xc, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
        acc = np.zeros((B, oh, ow, cout), dtype=np.int64)
        for i in range(kh):
            for j in range(kw):
                patch = xp[:, i : i + (oh - 1) * sh + 1 : sh, j : j + (ow - 1) * sw + 1 : sw, :]
                if depthwise:
                    acc += patch * w[0, i, j, :]
                else:
                    acc += patch @ w[:, i, j, :].T
        if bias is not None:
            acc += bias
        y = mbqm(acc, p["mult"], p["shift"]) + zp_out
        return np.clip(y, *p["act"]).astype(np.int8)

    def _add(self, op, env):
        a = self._value(env, op.inputs[0]).astype(np.int64)
        b = self._value(env, op.inputs[1]).astype(np.int64)
        s1, z1 = self._q(op.inputs[0])
        s2, z2 = self._q(op.inputs[1])
        so, zo = self._q(op.outputs[0])
        key = op.index
        if key not in self._prep:
            twice_max = 2.0 * max(float(np.float32(s1)), float(np.float32(s2)))
            m1 = quantize_multiplier(float(np.float32(s1)) / twice_max)
            m2 = quantize_multiplier(float(np.float32(s2)) / twice_max)
            mo = quantize_multiplier(twice_max / ((1 << 20) * float(np.float32(so))))
            self._prep[key] = {"m1": m1, "m2": m2, "mo": mo, "act": activation_range(op.options["activation"], so, zo)}
        p = self._prep[key]
        sa = mbqm((a - z1) << 20, *p["m1"])
        sb = mbqm((b - z2) << 20, *p["m2"])
        y = mbqm(sa + sb, *p["mo"]) + zo
        return np.clip(y, *p["act"]).astype(np.int8)

    def _mean(self, op, env):
        x = self._value(env, op.inputs[0]).astype(np.int64)
        axes = tuple(int(a) % x.ndim for a in np.atleast_1d(self._value(env, op.inputs[1])))
        s_in, zp_in = self._q(op.inputs[0])
        s_out, zp_out = self._q(op.outputs[0])
        n = int(np.prod([x.shape[a] for a in axes]))
        if self.mean_form == "float":
            return mean_int8_float(x, axes, bool(op.options.get("keep_dims")), n, s_in, zp_in, s_out, zp_out)
        mult, shift = quantize_multiplier(float(np.float32(s_in)) / float(np.float32(s_out)))
        fold = min(n.bit_length() - 1, 32, 31 + shift)  # 63 - clz(n), clamped as in reduce.h
        mult = int((mult << fold) // n)
        shift -= fold
        total = x.sum(axis=axes, keepdims=bool(op.options.get("keep_dims"))) - zp_in * n
        y = mbqm(total, mult, shift) + zp_out
        return np.clip(y, -128, 127).astype(np.int8)

    def _sum(self, op, env):
        """int8 SUM over an axis: int32 sum of (q - zp_in), one MultiplyByQuantizedMultiplier by s_in / s_out (the integer form of reduce.h,
        as for MEAN without the division by the count)."""
        x = self._value(env, op.inputs[0]).astype(np.int64)
        axes = tuple(int(a) % x.ndim for a in np.atleast_1d(self._value(env, op.inputs[1])))
        s_in, zp_in = self._q(op.inputs[0])
        s_out, zp_out = self._q(op.outputs[0])
        mult, shift = quantize_multiplier(float(np.float32(s_in)) / float(np.float32(s_out)))
        total = (x - zp_in).sum(axis=axes, keepdims=bool(op.options.get("keep_dims")))
        return np.clip(mbqm(total, mult, shift) + zp_out, -128, 127).astype(np.int8)

    def _fully_connected(self, op, env):
        x = self._value(env, op.inputs[0]).astype(np.int64)
        wt = self.model.tensors[op.inputs[1]]
        w = wt.data.astype(np.int64)  # [out, in]
        bias = self.model.tensors[op.inputs[2]].data.astype(np.int64) if len(op.inputs) > 2 and op.inputs[2] >= 0 else 0
        s_in, zp_in = self._q(op.inputs[0])
        s_out, zp_out = self._q(op.outputs[0])
        key = op.index
        if key not in self._prep:
            mult, shift = _per_channel_multipliers(s_in, wt.scale, s_out, w.shape[0])
            self._prep[key] = {"mult": mult, "shift": shift, "act": activation_range(op.options["activation"], s_out, zp_out)}
        p = self._prep[key]
        acc = (x.reshape(-1, w.shape[1]) - zp_in) @ w.T + bias
        y = np.clip(mbqm(acc, p["mult"], p["shift"]) + zp_out, *p["act"]).astype(np.int8)
        if op.options.get("keep_num_dims"):  # [..., in] -> [..., out] (squeeze-excite Dense layers on [B, 1, 1, C])
            y = y.reshape(*x.shape[:-1], w.shape[0])
        return y

    def _mul(self, op, env):
        """int8 MUL (reference kernel mul.h): zo + MultiplyByQuantizedMultiplier((q1 - z1) * (q2 - z2), s1 * s2 / so), broadcasting."""
        a = self._value(env, op.inputs[0]).astype(np.int64)
        b = self._value(env, op.inputs[1]).astype(np.int64)
        s1, z1 = self._q(op.inputs[0])
        s2, z2 = self._q(op.inputs[1])
        so, zo = self._q(op.outputs[0])
        key = op.index
        if key not in self._prep:
            real = float(np.float32(s1)) * float(np.float32(s2)) / float(np.float32(so))
            self._prep[key] = {"m": quantize_multiplier(real), "act": activation_range(op.options["activation"], so, zo)}
        p = self._prep[key]
        y = mbqm((a - z1) * (b - z2), *p["m"]) + zo
        return np.clip(y, *p["act"]).astype(np.int8)

    def _reduce_max(self, op, env):
        x = self._value(env, op.inputs[0])
        axes = tuple(int(a) % x.ndim for a in np.atleast_1d(self._value(env, op.inputs[1])))
        if self._q(op.inputs[0]) != self._q(op.outputs[0]):
            raise ValueError("REDUCE_MAX with differing input / output quantisation")
        return x.max(axis=axes, keepdims=bool(op.options.get("keep_dims")))

    def div_params(self, op):
        s1, z1 = self._q(op.inputs[0])
        s2, z2 = self._q(op.inputs[1])
        so, zo = self._q(op.outputs[0])
        real = float(np.float32(s1)) / (float(np.float32(s2)) * float(np.float32(so)))
        return z1, z2, zo, quantize_multiplier(real), activation_range(op.options.get("activation", "none"), so, zo)

    def _div(self, op, env):
        """int8 DIV (reference kernel div.h: DivElementwise / BroadcastDivSlow), broadcasting."""
        a = self._value(env, op.inputs[0]).astype(np.int64)
        b = self._value(env, op.inputs[1]).astype(np.int64)
        z1, z2, zo, (mo, so), act = self.div_params(op)
        x1, x2 = np.broadcast_arrays(a - z1, b - z2)
        neg = x2 < 0
        x1 = np.where(neg, -x1, x1)
        x2 = np.where(neg, -x2, x2)
        x2 = np.where(x2 == 0, 1, x2)  # undefined in TFLite (see the header)
        inv, recip_shift = get_reciprocal(x2)
        headroom = count_leading_sign_bits32(x1)
        unscaled = srdhm(x1 << headroom, inv)
        y = mbqm(unscaled, mo, so - recip_shift - headroom) + zo
        return np.clip(y, *act).astype(np.int8)

    def logistic_lut(self, op) -> np.ndarray:
        """int8->int8 table indexed by ``q + 128`` (LUTPopulate in float32)."""
        s_in, zp_in = self._q(op.inputs[0])
        s_out, zp_out = self._q(op.outputs[0])
        q = np.arange(-128, 128, dtype=np.int32)
        deq = np.float32(s_in) * (q - zp_in).astype(np.float32)
        sig = (np.float32(1.0) / (np.float32(1.0) + np.exp(-deq, dtype=np.float32))).astype(np.float32)
        resc = round_half_away((sig / np.float32(s_out)).astype(np.float32))
        return np.clip(resc.astype(np.int64) + zp_out, -128, 127).astype(np.int8)

    def _strided_slice(self, op, env):
        x = self._value(env, op.inputs[0])
        begin = np.asarray(self._value(env, op.inputs[1])).astype(np.int64)
        end = np.asarray(self._value(env, op.inputs[2])).astype(np.int64)
        strides = np.asarray(self._value(env, op.inputs[3])).astype(np.int64)
        o = op.options
        if o["ellipsis_mask"] or o["new_axis_mask"]:
            raise ValueError("ellipsis/new-axis masks unsupported")
        sl = []
        for d in range(len(begin)):
            if o["shrink_axis_mask"] >> d & 1:
                sl.append(int(begin[d]))
                continue
            b = None if o["begin_mask"] >> d & 1 else int(begin[d])
            e = None if o["end_mask"] >> d & 1 else int(end[d])
            sl.append(slice(b, e, int(strides[d])))
        return x[tuple(sl)]

    # -- graph walk ---------------------------------------------------------------
    def invoke(self, x: np.ndarray, return_all: bool = False, resume: tuple[dict, int] | None = None):
        """float32 batch in -> float32 ``[B, C]`` out; optionally every intermediate tensor.  ``resume = (env, first_op)`` continues from the
        tensors of an earlier run at operator ``first_op`` (tests re-run the graph's tail under another form of MEAN / LOGISTIC)."""
        m = self.model
        env: dict[int, np.ndarray] = {m.inputs[0]: np.asarray(x, dtype=np.float32)} if resume is None else dict(resume[0])
        for op in m.ops[resume[1] if resume else 0:]:
            n = op.name
            if n == "QUANTIZE":
                s, zp = self._q(op.outputs[0])
                xin = self._value(env, op.inputs[0]).astype(np.float32)
                q = round_half_away((xin / np.float32(s)).astype(np.float32)).astype(np.int64) + zp
                y = np.clip(q, -128, 127).astype(np.int8)
            elif n == "DEQUANTIZE":
                s, zp = self._q(op.inputs[0])
                y = ((self._value(env, op.inputs[0]).astype(np.int32) - zp).astype(np.float32) * np.float32(s)).astype(np.float32)
            elif n == "TRANSPOSE":
                y = np.transpose(self._value(env, op.inputs[0]), [int(v) for v in self._value(env, op.inputs[1])])
            elif n == "STRIDED_SLICE":
                y = self._strided_slice(op, env)
            elif n == "SHAPE":
                y = np.asarray(self._value(env, op.inputs[0]).shape, dtype=np.int32)
            elif n == "PACK":
                y = np.stack([np.asarray(self._value(env, i)) for i in op.inputs], axis=op.options["axis"]).astype(np.int32)
            elif n == "FILL":
                dims = [int(v) for v in self._value(env, op.inputs[0])]
                val = self._value(env, op.inputs[1])
                y = np.full(dims, np.asarray(val).reshape(-1)[0], dtype=np.asarray(val).dtype)
            elif n == "CONCATENATION":
                parts = [self._value(env, i) for i in op.inputs]
                qs = {self._q(i) for i in op.inputs} | {self._q(op.outputs[0])}
                if len(qs) != 1:
                    raise ValueError("CONCATENATION with differing quantisation needs requantisation")
                y = np.concatenate(parts, axis=op.options["axis"])
            elif n == "RESHAPE":
                xin = self._value(env, op.inputs[0])  # (the graph states batch 1; the batch dimension follows the input here)
                y = xin.reshape([xin.shape[0]] + [int(v) for v in self._value(env, op.inputs[1])][1:])
            elif n == "PAD":  # int8 PAD fills with the zero point (real 0.0): TFLite reference_ops::Pad with pad_value = output zero point
                s_, zp = self._q(op.outputs[0])
                if self._q(op.inputs[0]) != (s_, zp):
                    raise ValueError("PAD input and output must share quantisation")
                pads = [(int(a), int(b)) for a, b in np.asarray(self._value(env, op.inputs[1])).reshape(-1, 2)]
                y = np.pad(self._value(env, op.inputs[0]), pads, constant_values=zp)
            elif n == "CONV_2D":
                y = self._conv(op, env, depthwise=False)
            elif n == "DEPTHWISE_CONV_2D":
                y = self._conv(op, env, depthwise=True)
            elif n == "ADD":
                y = self._add(op, env)
            elif n == "MEAN":
                y = self._mean(op, env)
            elif n == "FULLY_CONNECTED":
                y = self._fully_connected(op, env)
            elif n == "LOGISTIC":
                if self.logistic_form == "fixed":
                    s_i, z_i = self._q(op.inputs[0])
                    s_o, z_o = self._q(op.outputs[0])
                    if (round(1.0 / s_o), z_o) != (256, -128):
                        raise ValueError("fixed-point int8 LOGISTIC writes 1/256, -128")
                    y = logistic_int8_fixed(self._value(env, op.inputs[0]), s_i, z_i)
                else:
                    lut = self.logistic_lut(op)
                    y = lut[self._value(env, op.inputs[0]).astype(np.int32) + 128]
            elif n == "MUL":
                y = self._mul(op, env)
            elif n == "REDUCE_MAX":
                y = self._reduce_max(op, env)
            elif n == "DIV":
                y = self._div(op, env)
            elif n == "SOFTMAX":
                xin = self._value(env, op.inputs[0])
                beta = float(op.options.get("beta", 1.0))
                if xin.dtype == np.float32:  # float32 softmax behind DEQUANTIZE (the classifier head this build's exporter writes): exp(beta (x - max)) / sum
                    z = (xin - xin.max(axis=-1, keepdims=True)) * np.float32(beta)
                    e = np.exp(z, dtype=np.float32)
                    y = (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)
                else:  # int8 (attention pooling): one of the two published kernels
                    s_i, _ = self._q(op.inputs[0])
                    s_o, z_o = self._q(op.outputs[0])
                    if (round(1.0 / s_o), z_o) != (256, -128):
                        raise ValueError("int8 SOFTMAX output must be 1/256, -128")
                    y = softmax_int8_fixed(xin, s_i, beta) if self.softmax_form == "fixed" else softmax_int8_lut(xin, s_i, beta, s_o, z_o)
            elif n == "SUM":
                y = self._sum(op, env)
            else:
                raise ValueError(f"operator {n} not handled by the oracle")
            env[op.outputs[0]] = y
        out = env[m.outputs[0]]
        return (out, env) if return_all else out
