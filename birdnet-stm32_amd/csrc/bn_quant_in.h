// bn_quant_in.h — the arithmetic of the INT8 graph's input QUANTIZE and of numpy's |complex64|, shared by the kernels that must
// agree on it bit for bit: i8_mel_mfma_kernel<QIN> (bn_i8_fused.hip) and the exactness pass of the audio path (bn_stft_exact.hip).
//
// Reference: birdnet_stm32/audio/spectrogram.py:12-21 ((S - min) / (max - min + 1e-10), float32), :106-115 (np.abs of the
// complex64 STFT); TFLite QUANTIZE (op #0 of the shipped graph): q = clamp(round(x / scale) + zp), halves away from zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn {

// Exact float32 division by a constant whose correctly rounded reciprocal y = RN(1 / b) is known: q0 = RN(a y),
// r = fma(-b, q0, a) (exact residual), q = fma(r, y, q0) is RN(a / b) (Markstein's correction step; no underflow or overflow
// occurs for the operands here: a / b lies in [0, 256]).  Three instructions instead of the ~10 of the IEEE division sequence;
// tests compare it with true division on 10^8 operand pairs and the fused kernel with the separate QUANTIZE bit for bit.
__device__ __forceinline__ float div_by_const(float a, float b, float y) {
    const float q0 = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
}

// roundf(v) + zp clamped to int8.  With zp = -128 (every quantised spectrogram input: the value range starts at 0) negative v
// lands on -128 whichever way a tie goes, and for v >= 0 round-half-away-from-zero is floor(v + 0.5) — one v_cvt_rpi_i32_f32
// instead of the seven instructions of roundf + cvt.  Other zero points take the general form.
__device__ __forceinline__ int quantise_i8_zp128(float v) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return min(max(r, 0), 255) - 128;  // v_med3_i32
}
__device__ __forceinline__ int quantise_i8_any(float v, int zp) {
    const int r = (int32_t)roundf(v) + zp;
    return r < -128 ? -128 : (r > 127 ? 127 : r);
}
__device__ __forceinline__ int quantise_i8(float v, int zp) { return zp == -128 ? quantise_i8_zp128(v) : quantise_i8_any(v, zp); }

// The normalise + quantise chain of one spectrogram value, as every consumer evaluates it.
struct QuantIn {
    float mn, rng, y_rng, scale, y_scale;
    int zp;
    bool renorm;
    __device__ __forceinline__ void set(const float* minmax_of_chunk, float scale_, int zp_) {
        renorm = minmax_of_chunk != nullptr;
        mn = 0.0f;
        rng = 1.0f;
        y_rng = 1.0f;
        if (renorm) {
            mn = minmax_of_chunk[0];
            rng = (float)((double)(minmax_of_chunk[1] - mn) + 1e-10);
            y_rng = (float)(1.0 / (double)rng);
        }
        scale = scale_;
        y_scale = (float)(1.0 / (double)scale_);
        zp = zp_;
    }
    __device__ __forceinline__ float value(float x) const {  // the float handed to the rounding step
        if (renorm) x = div_by_const(x - mn, rng, y_rng);
        return div_by_const(x, scale, y_scale);
    }
    __device__ __forceinline__ int q(float x) const { return quantise_i8(value(x), zp); }
};

// numpy's np.abs on complex64 (x86 builds with FMA3, numpy >= 1.25: loops_unary_complex, `simd_cabsf`): with larger / smaller of
// (|re|, |im|), ratio = smaller / larger, the result is  larger * sqrt(fma(ratio, ratio, 1)),  every step rounded to float32.
// It is NOT the correctly rounded magnitude (35 % of random inputs differ from (float)sqrt((double)re^2 + im^2) by one ulp);
// oracle/stft.py restates it and tests/test_oracle_pinning.py pins that restatement to the installed numpy.
__device__ __forceinline__ float numpy_cabsf(float re, float im) {
    re = fabsf(re);
    im = fabsf(im);
    const float larger = fmaxf(re, im), smaller = fminf(im, re);
    if (larger == 0.0f) return 0.0f;
    const float ratio = __fdiv_rn(smaller, larger);  // (compiles to the IEEE division sequence)
    // correctly rounded float32 square root through float64: v_sqrt_f32 alone is 1 ulp, and the float64 root of a 24-bit
    // number is never within 2^-51 (relative) of a float32 rounding boundary
    const float h = (float)sqrt((double)__builtin_fmaf(ratio, ratio, 1.0f));
    return __fmul_rn(h, larger);
}

// Bound on |S' - S|: S' = the float32 FFT's magnitude (stft512_mag_kernel), S = the reference value (float64 window
// product and FFT, complex64 store, numpy_cabsf).  For an element of frame t
//     eps(S') = u (kA ||x_t||_2 + kB max_k S'_tk + kC S'),   u = 2^-24,  x_t = the frame's 512 samples.
// The three terms follow the three ways the float32 transform errs: rounding noise that adds up like a random walk over the
// frame (measured rms 2.1 u ||x||_2 for every signal family); the systematic errors of the rebuilt twiddles / window, which
// scale with the largest partial sums and show at the small bins of tonal frames as well (up to 26 u ||x||_2 next to a peak of
// 8..16 ||x||_2); and the few ulp of the element itself (square, add, root; the reference's own complex64 and |.| roundings).
// The constants are 4 x what tools/stft_error_stats.py needs to cover the largest error it finds over 16 signal families
// (tones on and between bins, two tones, chirps, AM, harmonics, noise, impulses, DC, square, clipped, onsets, near-DC and
// near-Nyquist tones; 2.5e7 elements: largest |S' - S| / eps = 0.25; tools/guard_margin.py on the device, 20 families with random parameters,
// 2.2e11 elements: 0.343 — profiles/r03_guard_margin.md).  It is an EMPIRICAL bound with that margin, not a
// worst-case one: the worst case over all rounding patterns (every one of ~65 roundings per path aligned over 512 samples)
// is ~1000 u ||x||_2 and would put 5e-3 of all elements in doubt instead of 6e-4.  bn_set_option("stft_exact", 1) computes
// every bin in float64 for callers who want no bound at all.
constexpr float kGuardU = 5.9604644775390625e-8f * 1.001f;  // (0.1 % on top for the float32 evaluation of the bound itself)
constexpr float kGuardL2 = 48.0f * kGuardU, kGuardPeak = 8.0f * kGuardU, kGuardRel = 14.0f * kGuardU;
// The PROVEN frame term (option stft_guard = 1; docs/exactness.md "A worst-case bound" derives it from the kernel's operation sequence: window
// 2.5 u ||x||, eight addition levels u each, three twiddle levels (2 u per complex multiply-add + the twiddle's own error: u, 43 u for the
// powers built by square-and-multiply, u), the split pass; element <= norm): |X'_k - X_k| <= 1148 u ||x_t||_2, taken as 1200 u for the
// second-order terms.  No peak term: the norm-wise argument already covers every element.
constexpr float kGuardL2Proven = 1200.0f * kGuardU;
// ends of the interval [S' - eps(S'), S' + eps(S')] given the frame's part eps_f = kGuardL2 ||x|| + kGuardPeak peak (monotone in S')
__device__ __forceinline__ float guard_hi(float s, float eps_f) { return __builtin_fmaf(s, kGuardRel, s) + eps_f; }
__device__ __forceinline__ float guard_lo(float s, float eps_f) { return __builtin_fmaf(-s, kGuardRel, s) - eps_f; }

// From the bound on S to the band around a rounding boundary: the quantiser's argument v = value(S') moves by at most
// eps(S') / (range * scale) when S' moves by eps(S'); both evaluations of value() carry three roundings of relative size u each
// (6 u v together) and the test's own v + 0.5 one more (u (v + 1)).  With v <= S' / (range * scale) the v-proportional part rides on the
// S'-proportional term of the bound (kBandRel), what is left is a constant of a few u (kQuantSlack).
constexpr float kBandRel = kGuardRel + 8.0f * kGuardU;
constexpr float kQuantSlack = 4.0f * kGuardU;
// The guarded mixer (i8_mel_mfma_kernel<QIN, 1>) evaluates v - 128 = fma(S' - min, RN(1 / (range scale)), -128) instead of the two divisions:
// against the reference's chain on the same S' that is u v for the rounded reciprocal (the chain's own two division roundings are
// already in the 6 u v above) and one rounding of a number of magnitude <= 128, the test's v - 128 + 0.5 another: 257 u, taken as 320 u
// (1.9e-5 of a quantisation step: 8 % more elements in doubt than with 4 u).
constexpr float kQuantSlackFolded = 320.0f * kGuardU;

}  // namespace bn
