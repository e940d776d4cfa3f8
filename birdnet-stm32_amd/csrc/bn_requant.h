// bn_requant.h — gemmlowp/TFLite fixed-point requantisation on the vector ALU, shared by the INT8 kernels.
//
//   MultiplyByQuantizedMultiplier(x, M0, shift) = RoundingDivideByPOT(SaturatingRoundingDoublingHighMul(x << left, M0), right)
//
// The fast forms below are algebraically identical to the reference definitions for M0 >= 0 whenever |x| * M0 / 2^31 + 2^(e-1) + 1
// stays below 2^31 (e = the right shift) — in particular for |x| < 2^30.  The lowering pass proves that bound per channel from the
// weights and the folded bias before a plan is built (models/_lower_i8.py: _expect_acc_range; (q - zp) << 20 < 2^28 in ADD):
//   SRDHM: trunc((x*M0 + nudge) / 2^31), nudge = 2^30 for x*M0 >= 0 and 1 - 2^30 otherwise, equals the ARITHMETIC shift
//          (x*M0 + 2^30) >> 31 for both signs (for negatives trunc(y/2^31) = floor((y + 2^31 - 1)/2^31) and the nudges differ
//          by exactly 2^31 - 1), i.e. one 32x32+64 -> 64 multiply-add (v_mad_i64_i32) and a funnel shift;
//   RoundingDivideByPOT(v, e): (v >> e) + ((v & mask) > (mask >> 1) + (v < 0)) equals (v + half + (v < 0 ? -1 : 0)) >> e
//          with half = 2^(e-1) (and v itself for e = 0).
// Left shifts (shift > 0) and negative multipliers take the literal reference path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn {

__device__ __forceinline__ int32_t srdhm_ref(int32_t a, int32_t b) {
    const bool overflow = (a == b) && (a == INT32_MIN);
    const int64_t ab = (int64_t)a * (int64_t)b;
    const int64_t nudge = ab >= 0 ? (1ll << 30) : (1ll - (1ll << 30));
    const int32_t r = (int32_t)((ab + nudge) / (1ll << 31));
    return overflow ? INT32_MAX : r;
}
__device__ __forceinline__ int32_t rdivpot_ref(int32_t x, int exponent) {
    const int32_t mask = (int32_t)((1u << exponent) - 1u);
    const int32_t remainder = x & mask;
    const int32_t threshold = (mask >> 1) + (x < 0 ? 1 : 0);
    return (x >> exponent) + (remainder > threshold ? 1 : 0);
}
__device__ __forceinline__ int32_t mbqm_ref(int32_t x, int32_t mult, int shift) {
    const int left = shift > 0 ? shift : 0;
    const int right = shift > 0 ? 0 : -shift;
    return rdivpot_ref(srdhm_ref(x * (1 << left), mult), right);
}

__device__ __forceinline__ int32_t srdhm_pos(int32_t x, int32_t m) {  // m >= 0
    // (x*m + 2^30) >> 31 on the 64-bit product: one v_mad_i64_i32 and one v_alignbit_b32
    return (int32_t)(((int64_t)x * (int64_t)m + (1ll << 30)) >> 31);
}
__device__ __forceinline__ int32_t rdivpot_fast(int32_t v, int e) {  // e >= 0
    const int32_t half = e ? (1 << (e - 1)) : 0;  // e up to 31 occurs (dead channels with vanishing scales)
    const int32_t adj = e ? (v >> 31) : 0;
    return (v + half + adj) >> e;
}
__device__ __forceinline__ int32_t mbqm(int32_t x, int32_t mult, int shift) {
    if (shift <= 0 && mult >= 0) return rdivpot_fast(srdhm_pos(x, mult), -shift);
    return mbqm_ref(x, mult, shift);
}

// Every multiplier >= 0 and every shift < 0 (checked on the host over all channels of an operator at load time): the
// workgroup-uniform flag `all_right` selects the branch-free form, with no per-element sign/zero tests on the exponent.
__device__ __forceinline__ int32_t mbqm_right(int32_t x, int32_t mult, int shift) {
    const int e = -shift;  // >= 1
    const int32_t v = srdhm_pos(x, mult);
    return (v + (1 << (e - 1)) + (v >> 31)) >> e;
}
__device__ __forceinline__ int32_t mbqm_u(int32_t x, int32_t mult, int shift, bool all_right) {
    return all_right ? mbqm_right(x, mult, shift) : mbqm(x, mult, shift);
}

__device__ __forceinline__ int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// clamp as ONE instruction; lo <= hi.  (The compiler cannot prove lo <= hi for run-time bounds and emits compare + select + min.)
__device__ __forceinline__ int med3i(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}
// the low bytes of four values as one dword (three v_perm_b32)
__device__ __forceinline__ int pack4(const int (&q)[4]) {
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)q[1], (uint32_t)q[0], 0x0c0c0400u), hi = __builtin_amdgcn_perm((uint32_t)q[3], (uint32_t)q[2], 0x0c0c0400u);
    return (int)__builtin_amdgcn_perm(hi, lo, 0x05040100u);
}


// int8 MEAN of `raw_sum` = sum of the P raw bytes of a channel.  Two published forms (oracle/int8_graph.py has both behind mean_form):
//   integer (default): MultiplyByQuantizedMultiplier(raw_sum - zp_in P, mult, shift) with 1 / P folded into the multiplier (reduce.h);
//   float (shift == kMeanFloatForm, mult = the float32 bits of s_in / s_out): TFLite's float-arithmetic QuantizedMeanOrSum —
//       round(raw_sum / P * scale + (-zp_in * scale)) in float32, every operation rounded on its own (no fused multiply-add).
// The lowering pass picks the form (models/_lower_i8.py: lower_i8(mean_form=...)); ~4 % of the pooled bytes of the shipped graph differ by
// one step between the two (tests/test_oracle_pinning.py counts them), so a deployment that must match a given interpreter picks its form.
constexpr int32_t kMeanFloatForm = 0x7fffff00;
__device__ __forceinline__ int32_t mean_q(int32_t raw_sum, int32_t P, int32_t zp_in, int32_t mult, int32_t shift, int32_t zp_out) {
#pragma clang fp contract(off)
    if (shift == kMeanFloatForm) {
        const float scale = __int_as_float(mult);
        const float bias = (float)(-zp_in) * scale;
        const float mean = (float)raw_sum / (float)P;
        const float t = mean * scale;
        const float r = roundf(t + bias) + (float)zp_out;
        return (int32_t)fminf(fmaxf(r, -128.0f), 127.0f);
    }
    return clampi(mbqm(raw_sum - zp_in * P, mult, shift) + zp_out, -128, 127);
}

}  // namespace bn
