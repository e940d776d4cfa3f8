/* oracle_i8.c — plain-C restatement of the TFLite int8 reference kernels the shipped graph uses, for the INT8 CPU baseline
 * and as a second opinion on the numpy interpreter (oracle/int8_graph.py).  TEST INFRASTRUCTURE ONLY: nothing under
 * birdnet-stm32_amd/ links or loads this file (see oracle/__init__.py).
 *
 * Follows the public TFLite reference implementations (tensorflow 2.19, not vendored by the reference repository; call sites
 * birdnet_stm32/models/runners.py:51-95): MultiplyByQuantizedMultiplier = RoundingDivideByPOT(SaturatingRoundingDoublingHighMul(
 * x << left, M0), right) (kernels/internal/common.h), CONV_2D / DEPTHWISE_CONV_2D per-channel (reference_integer_ops/conv.h,
 * depthwise_conv.h: padded taps are skipped, i.e. contribute (zp - zp) = 0), ADD with left shift 20 (reference_integer_ops/add.h),
 * MEAN with the folded multiplier (reduce.h), FULLY_CONNECTED per-channel.  Parallelism: OpenMP over chunks / rows. */
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oi_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline int32_t srdhm(int32_t a, int32_t b) {
    if (a == b && a == INT32_MIN) return INT32_MAX;
    const int64_t ab = (int64_t)a * (int64_t)b;
    const int64_t nudge = ab >= 0 ? (1ll << 30) : (1ll - (1ll << 30));
    return (int32_t)((ab + nudge) / (1ll << 31));
}
static inline int32_t rdivpot(int32_t x, int e) {
    const int32_t mask = (int32_t)((1u << e) - 1u);
    const int32_t rem = x & mask;
    const int32_t thr = (mask >> 1) + (x < 0 ? 1 : 0);
    return (x >> e) + (rem > thr ? 1 : 0);
}
static inline int32_t mbqm(int32_t x, int32_t mult, int shift) {
    const int left = shift > 0 ? shift : 0, right = shift > 0 ? 0 : -shift;
    return rdivpot(srdhm(x * (1 << left), mult), right);
}
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* x [B][H][W][Cin], w [Cout][kh][kw][Cin], y [B][OH][OW][Cout] */
void oi_conv(const int8_t* x, int8_t* y, int B, int H, int W, int Cin, int kh, int kw, int Cout, int sh, int sw, int OH, int OW,
             int pt, int pl, const int8_t* w, const int32_t* bias, int zp_in, int zp_out, const int32_t* mult, const int32_t* shift,
             int amin, int amax) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int oh = 0; oh < OH; ++oh)
            for (int ow = 0; ow < OW; ++ow) {
                int8_t* yo = y + (((size_t)b * OH + oh) * OW + ow) * Cout;
                for (int n = 0; n < Cout; ++n) {
                    int32_t acc = bias ? bias[n] : 0;
                    for (int i = 0; i < kh; ++i) {
                        const int ih = oh * sh - pt + i;
                        if (ih < 0 || ih >= H) continue;
                        for (int j = 0; j < kw; ++j) {
                            const int iw = ow * sw - pl + j;
                            if (iw < 0 || iw >= W) continue;
                            const int8_t* xi = x + (((size_t)b * H + ih) * W + iw) * Cin;
                            const int8_t* wi = w + (((size_t)n * kh + i) * kw + j) * Cin;
                            int32_t s = 0;
                            for (int c = 0; c < Cin; ++c) s += ((int32_t)xi[c] - zp_in) * (int32_t)wi[c];
                            acc += s;
                        }
                    }
                    yo[n] = (int8_t)clampi(mbqm(acc, mult[n], shift[n]) + zp_out, amin, amax);
                }
            }
}

/* x [B][H][W][C], w [kh][kw][C], y [B][OH][OW][C] */
void oi_dwconv(const int8_t* x, int8_t* y, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int OH, int OW, int pt, int pl,
               const int8_t* w, const int32_t* bias, int zp_in, int zp_out, const int32_t* mult, const int32_t* shift, int amin,
               int amax) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int oh = 0; oh < OH; ++oh)
            for (int ow = 0; ow < OW; ++ow) {
                int8_t* yo = y + (((size_t)b * OH + oh) * OW + ow) * C;
                for (int c = 0; c < C; ++c) {
                    int32_t acc = bias ? bias[c] : 0;
                    for (int i = 0; i < kh; ++i) {
                        const int ih = oh * sh - pt + i;
                        if (ih < 0 || ih >= H) continue;
                        for (int j = 0; j < kw; ++j) {
                            const int iw = ow * sw - pl + j;
                            if (iw < 0 || iw >= W) continue;
                            acc += ((int32_t)x[(((size_t)b * H + ih) * W + iw) * C + c] - zp_in) * (int32_t)w[((size_t)i * kw + j) * C + c];
                        }
                    }
                    yo[c] = (int8_t)clampi(mbqm(acc, mult[c], shift[c]) + zp_out, amin, amax);
                }
            }
}

/* element-wise ADD; b is broadcast with period nb (nb == n: same shape) */
void oi_add(const int8_t* a, const int8_t* b, int8_t* y, long n, long nb, int z1, int m1, int s1, int z2, int m2, int s2, int mo, int so,
            int zo, int amin, int amax) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        const int32_t sa = mbqm(((int32_t)a[i] - z1) * (1 << 20), m1, s1);
        const int32_t sb = mbqm(((int32_t)b[i % nb] - z2) * (1 << 20), m2, s2);
        y[i] = (int8_t)clampi(mbqm(sa + sb, mo, so) + zo, amin, amax);
    }
}

/* MEAN over the P positions of x [B][P][C] */
void oi_mean(const int8_t* x, int8_t* y, int B, int P, int C, int zp_in, int mult, int shift, int zp_out) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            int32_t s = 0;
            for (int p = 0; p < P; ++p) s += x[((size_t)b * P + p) * C + c];
            s -= zp_in * P;
            y[(size_t)b * C + c] = (int8_t)clampi(mbqm(s, mult, shift) + zp_out, -128, 127);
        }
}

/* x [B][Cin], w [Cout][Cin] */
void oi_fc(const int8_t* x, int8_t* y, int B, int Cin, int Cout, const int8_t* w, const int32_t* bias, int zp_in, int zp_out,
           const int32_t* mult, const int32_t* shift, int amin, int amax) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < Cout; ++n) {
            int32_t acc = bias ? bias[n] : 0;
            for (int k = 0; k < Cin; ++k) acc += ((int32_t)x[(size_t)b * Cin + k] - zp_in) * (int32_t)w[(size_t)n * Cin + k];
            y[(size_t)b * Cout + n] = (int8_t)clampi(mbqm(acc, mult[n], shift[n]) + zp_out, amin, amax);
        }
}
