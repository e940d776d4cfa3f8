"""Host-side fixed-point helpers used while lowering an INT8 `.tflite` graph.

At model-load time the lowering pass has to do what ``tf.lite.Interpreter.allocate_tensors``
does in the reference (reference: birdnet_stm32/models/runners.py:62-68): turn every
``(input_scale * weight_scale[c] / output_scale)`` into a Q31 multiplier and an exponent,
compute fused-activation clamp ranges, and — specific to this implementation — collapse the
frontend's element-wise PWL sub-graph (1x1 depthwise convolutions and ADDs on int8 tensors,
SURVEY.md Appendix B ops #9-#19) into one 256-entry table per channel by evaluating those
operators once for every possible int8 input.

The arithmetic follows the public TFLite conventions (gemmlowp-style
``SaturatingRoundingDoublingHighMul`` + ``RoundingDivideByPOT``); the device kernels in
``csrc/bn_i8.hip`` implement the same functions for the per-sample work.
"""

from __future__ import annotations

import math

import numpy as np

_I32_MIN, _I32_MAX = -(2**31), 2**31 - 1


def quantize_multiplier(real: float) -> tuple[int, int]:
    """Split ``real`` into a Q31 mantissa in [2^30, 2^31) and a power-of-two exponent."""
    if real == 0.0:
        return 0, 0
    mant, exp = math.frexp(real)
    q = int(math.floor(abs(mant) * (1 << 31) + 0.5))  # halves away from zero; exact in double (< 2^32)
    if mant < 0:
        q = -q
    if abs(q) == 1 << 31:
        q //= 2
        exp += 1
    if exp < -31:
        return 0, 0
    if exp > 30:
        return _I32_MAX, 30
    return q, exp


def channel_multipliers(s_in: float, w_scales, s_out: float, n: int) -> tuple[np.ndarray, np.ndarray]:
    """Per-output-channel (multiplier, shift) int32 arrays; a single weight scale is broadcast."""
    ws = np.asarray(w_scales, dtype=np.float32).reshape(-1)
    if ws.size == 1:
        ws = np.repeat(ws, n)
    if ws.size != n:
        raise ValueError(f"{ws.size} weight scales for {n} channels")
    mult = np.empty(n, np.int32)
    shift = np.empty(n, np.int32)
    for c in range(n):
        m, e = quantize_multiplier(float(np.float32(s_in)) * float(ws[c]) / float(np.float32(s_out)))
        mult[c], shift[c] = m, e
    return mult, shift


def _c_round(x: float) -> int:
    """C ``round``: nearest, halves away from zero."""
    t = math.trunc(x)
    if abs(x - t) >= 0.5:
        t += 1 if x > 0 else -1
    return int(t)


def activation_bounds(act: str, scale: float, zero_point: int) -> tuple[int, int]:
    """int8 clamp range of a fused activation (``CalculateActivationRangeQuantized``)."""
    s = np.float32(scale)

    def q(v: float) -> int:
        return zero_point + _c_round(float(np.float32(v) / s))

    if act == "none":
        return -128, 127
    if act == "relu":
        return max(-128, q(0.0)), 127
    if act == "relu6":
        return max(-128, q(0.0)), min(127, q(6.0))
    if act == "relu_n1_to_1":
        return max(-128, q(-1.0)), min(127, q(1.0))
    raise NotImplementedError(f"fused activation {act!r}")


# -- vectorised integer ops for building tables ---------------------------------------------
def _high_mul(a: np.ndarray, b) -> np.ndarray:
    a = a.astype(np.int64)
    prod = a * np.asarray(b, dtype=np.int64)
    nudged = prod + np.where(prod >= 0, 1 << 30, 1 - (1 << 30))
    out = np.where(nudged >= 0, nudged >> 31, -((-nudged) >> 31))  # truncating division by 2^31
    return np.where((a == _I32_MIN) & (np.asarray(b) == _I32_MIN), _I32_MAX, out)


def _round_shift(x: np.ndarray, exponent) -> np.ndarray:
    e = np.asarray(exponent, dtype=np.int64)
    mask = (np.int64(1) << e) - 1
    rem = x & mask
    return (x >> e) + (rem > ((mask >> 1) + (x < 0)))


def requantize(acc: np.ndarray, mult, shift) -> np.ndarray:
    """``MultiplyByQuantizedMultiplier`` applied element-wise (int64 holding int32 values)."""
    sh = np.asarray(shift, dtype=np.int64)
    left, right = np.maximum(sh, 0), np.maximum(-sh, 0)
    return _round_shift(_high_mul(acc.astype(np.int64) << left, mult), right)


class AddParams:
    """Multipliers of one TFLite int8 ADD (left shift 20)."""

    def __init__(self, s1: float, z1: int, s2: float, z2: int, so: float, zo: int, act: str):
        s1, s2, so = float(np.float32(s1)), float(np.float32(s2)), float(np.float32(so))
        twice_max = 2.0 * max(s1, s2)
        self.z1, self.z2, self.zo = int(z1), int(z2), int(zo)
        self.m1, self.sh1 = quantize_multiplier(s1 / twice_max)
        self.m2, self.sh2 = quantize_multiplier(s2 / twice_max)
        self.mo, self.sho = quantize_multiplier(twice_max / ((1 << 20) * so))
        self.amin, self.amax = activation_bounds(act, so, zo)

    def apply(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        sa = requantize((a.astype(np.int64) - self.z1) << 20, self.m1, self.sh1)
        sb = requantize((b.astype(np.int64) - self.z2) << 20, self.m2, self.sh2)
        return np.clip(requantize(sa + sb, self.mo, self.sho) + self.zo, self.amin, self.amax)


def mean_multiplier(s_in: float, s_out: float, n: int) -> tuple[int, int]:
    """Multiplier of the int8 MEAN with 1/n folded in the way TFLite's reduce kernel folds it."""
    mult, shift = quantize_multiplier(float(np.float32(s_in)) / float(np.float32(s_out)))
    fold = min(int(n).bit_length() - 1, 32, 31 + shift)
    return int((mult << fold) // n), shift - fold


MEAN_FLOAT_FORM = 0x7FFFFF00  # csrc/bn_requant.h: kMeanFloatForm — `shift` value that selects the float-arithmetic MEAN on the device


def mean_float_params(s_in: float, s_out: float) -> tuple[int, int]:
    """(float32 bits of s_in / s_out, MEAN_FLOAT_FORM): the (mult, shift) pair that makes the device's MEAN evaluate TFLite's float-arithmetic
    ``QuantizedMeanOrSum`` — round(sum / n * scale - zp_in * scale) + zp_out in float32 — instead of the integer form of ``mean_multiplier``."""
    scale = np.float32(np.float32(s_in) / np.float32(s_out))
    return int(np.frombuffer(scale.tobytes(), np.int32)[0]), MEAN_FLOAT_FORM


def logistic_table(s_in: float, z_in: int, s_out: float, z_out: int) -> np.ndarray:
    """int8 -> int8 sigmoid table indexed by ``q + 128`` (float32 evaluation, like LUTPopulate)."""
    q = np.arange(-128, 128, dtype=np.int32)
    x = np.float32(s_in) * (q - z_in).astype(np.float32)
    y = (np.float32(1.0) / (np.float32(1.0) + np.exp(-x).astype(np.float32))).astype(np.float32)
    r = (y / np.float32(s_out)).astype(np.float32)
    t = np.trunc(r)
    r = np.where(np.abs(r - t) >= 0.5, t + np.sign(r), t)
    return np.clip(r.astype(np.int64) + z_out, -128, 127).astype(np.int8)


# -- int8 DIV as a table (the per-sample max normalisation of current hybrid frontends) --------------------------------------
def _clz32(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.int64) & 0xFFFFFFFF
    n = np.full(x.shape, 32, np.int64)
    for bit in range(32):
        n = np.where((x >> bit) & 1 == 1, 31 - bit, n)
    return n


def _reciprocal_q31(x: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """1 / x for int32 x > 0 the way TFLite's ``GetReciprocal(x, 31, ...)`` computes it: x is normalised to 1 + f with f in [0, 1)
    (Q0.31), ``1 / (1 + f)`` comes from gemmlowp's Newton-Raphson division (start 48/17 - 32/17 d on the half denominator d, three
    iterations, Q2.29), the result is a Q0.31 mantissa and the number of bits x lies above 1."""
    x = np.asarray(x, np.int64)
    lz = _clz32(x)
    f = ((x << lz) & 0xFFFFFFFF) - (1 << 31)
    half_den = (f + _I32_MAX + 1) // 2                     # RoundingHalfSum(f, One()) with One() = 2^31 - 1; the sum is >= 0
    est = 1515870810 + _high_mul(half_den, -1010580540)    # 48/17 - 32/17 d
    for _ in range(3):
        err = (1 << 29) - _high_mul(half_den, est)         # 1 - d x   (Q2.29)
        est = est + np.clip(_high_mul(est, err) << 2, _I32_MIN, _I32_MAX)
    return np.clip(est << 1, _I32_MIN, _I32_MAX), 31 - lz


def div_table(s1: float, z1: int, s2: float, z2: int, so: float, zo: int, act: str = "none") -> np.ndarray:
    """The TFLite int8 DIV (kernels/internal/reference/div.h) for every pair of bytes: int8 ``[256][256]``, row = divisor byte + 128,
    column = dividend byte + 128.  A zero divisor (undefined in TFLite) divides by 1."""
    mo, sho = quantize_multiplier(float(np.float32(s1)) / (float(np.float32(s2)) * float(np.float32(so))))
    amin, amax = activation_bounds(act, so, zo)
    x1 = (np.arange(-128, 128, dtype=np.int64) - int(z1))[None, :].repeat(256, axis=0)
    x2 = (np.arange(-128, 128, dtype=np.int64) - int(z2))[:, None].repeat(256, axis=1)
    neg = x2 < 0
    x1 = np.where(neg, -x1, x1)
    x2 = np.where(neg, -x2, x2)
    x2 = np.where(x2 == 0, 1, x2)
    inv, bits = _reciprocal_q31(x2)
    mag = np.where(x1 < 0, 2 * (-x1) - 1, x1)
    headroom = np.where(x1 < 0, _clz32(mag), _clz32(mag) - 1)          # CountLeadingSignBits
    quotient = _high_mul(x1 << headroom, inv)
    y = requantize(quotient, mo, sho - bits - headroom) + int(zo)
    return np.clip(y, amin, amax).astype(np.int8)


# -- int8 SOFTMAX as tables (attention pooling; reference models/blocks.py:136-159) -------------------------------------------------
def _exp_q5_to_q31(a: np.ndarray) -> np.ndarray:
    """gemmlowp ``exp_on_negative_values`` for FixedPoint<int32, 5>: a <= 0 as Q5.26 raw -> exp(a) as Q0.31 raw (fixedpoint.h: the
    argument modulo 1/4 through a fourth-order polynomial around -1/8, the multiples of 1/4 through constants exp(-2^k), k = -2 .. 4)."""
    a = np.asarray(a, np.int64)
    quarter = 1 << 24
    a_mod = (a & (quarter - 1)) - quarter
    x = (a_mod << 5) + (1 << 28)
    x2 = _high_mul(x, x)
    x3, x4 = _high_mul(x2, x), _high_mul(x2, x2)
    poly = _round_shift(_high_mul(_round_shift(x4, 2) + x3, 715827883) + x2, 1)
    result = 1895147668 + _high_mul(np.full(a.shape, 1895147668, np.int64), x + poly)
    rem = a_mod - a
    for bit, mult in ((24, 1672461947), (25, 1302514674), (26, 790015084), (27, 290630308), (28, 39332535), (29, 720401), (30, 242)):
        result = np.where((rem >> bit) & 1 == 1, _high_mul(result, mult), result)
    return np.where(a == 0, _I32_MAX, result)


def softmax_tables(s_in: float, beta: float, form: str) -> np.ndarray:
    """What the device needs of an int8 SOFTMAX (output 1 / 256, -128), indexed by d = max - x in 0 .. 255.

    ``form='fixed'`` (TFLite's reference kernel): int32 ``[2][256]`` — exp(-beta s d) as Q0.31 raw (-1: below diff_min, the output is
    -128 and the term does not enter the sum) and the same value rescaled to the accumulator's Q12.19.  ``form='lut'`` (TFLite's optimized
    kernel): float32 ``[256]`` exp(-beta s d), viewed as int32 bits."""
    d = np.arange(256, dtype=np.int64)
    if form == "lut":
        return np.exp((np.float32(-float(np.float32(s_in)) * float(beta)) * d.astype(np.float32)).astype(np.float32)).astype(np.float32).view(np.int32)
    if form != "fixed":
        raise ValueError("softmax form must be 'fixed' or 'lut'")
    real = min(float(beta) * float(np.float32(s_in)) * (1 << 26), (1 << 31) - 1.0)
    mult, shift = quantize_multiplier(real)
    if shift < 0:
        raise NotImplementedError("softmax input multiplier below one")
    diff_min = -int(np.floor(31.0 * (1 << 26) / (1 << shift)))
    live = -d >= diff_min
    e = _exp_q5_to_q31(_high_mul(np.where(live, -d, 0) << shift, mult))
    return np.stack([np.where(live, e, -1), np.where(live, _round_shift(e, 12), 0)]).astype(np.int32)
