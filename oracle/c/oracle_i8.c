/* oracle_i8.c — plain-C restatement of the TFLite int8 reference kernels the shipped graph uses, for the INT8 CPU baseline
 * and as a second opinion on the numpy interpreter (oracle/int8_graph.py).  TEST INFRASTRUCTURE ONLY: nothing under
 * birdnet-stm32_amd/ links or loads this file (see oracle/__init__.py).
 *
 * Follows the public TFLite reference implementations (tensorflow 2.19, not vendored by the reference repository; call sites
 * birdnet_stm32/models/runners.py:51-95): MultiplyByQuantizedMultiplier = RoundingDivideByPOT(SaturatingRoundingDoublingHighMul(
 * x << left, M0), right) (kernels/internal/common.h), CONV_2D / DEPTHWISE_CONV_2D per-channel (reference_integer_ops/conv.h,
 * depthwise_conv.h: padded taps are skipped, i.e. contribute (zp - zp) = 0), ADD with left shift 20 (reference_integer_ops/add.h),
 * MEAN with the folded multiplier (reduce.h), FULLY_CONNECTED per-channel.  Parallelism: OpenMP over chunks / rows.
 *
 * Two builds of this one file (oracle/Makefile): the portable one (-march=x86-64-v3: the scalar loops below, what the CPU tests load) and
 * `make native` (-march=native on the box the baseline is timed on).  With AVX-512 + VNNI the native build takes the vector paths:
 * 1x1 convolutions as u8 x s8 dot products (vpdpbusd: (x ^ 0x80) is x + 128 as an unsigned byte, the 128 and the input zero point are
 * folded into the bias as (128 + zp) * sum(w)), sixteen output channels per accumulator register, weights re-packed [K/4][N/16][16][4];
 * everything else sixteen channels per register in int32; MultiplyByQuantizedMultiplier on sixteen lanes with the literal definitions
 * (64-bit products, sign-dependent nudge, truncating division, remainder / threshold rounding shift).  Same integers either way:
 * tests/test_oracle_pinning.py holds BOTH builds against oracle/int8_graph.py on every tensor of the shipped graph. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__AVX512F__) && defined(__AVX512BW__) && defined(__AVX512VNNI__)
#include <immintrin.h>
#define OI_VEC 1
#else
#define OI_VEC 0
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

int oi_vectorised(void) { return OI_VEC; }  /* 1: this build takes the AVX-512 / VNNI paths */

int oi_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* threads of the following parallel regions (oracle/cport.py sets the process's CPU SHARE — affinity mask and cgroup quota — not the host's CPU count) */
void oi_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
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


#if OI_VEC
/* MultiplyByQuantizedMultiplier on sixteen int32 lanes, the definitions above lane by lane */
static inline __m512i srdhm_half(__m512i p) { /* eight 64-bit products -> (p + nudge) / 2^31, C truncation */
    const __mmask8 neg = _mm512_cmplt_epi64_mask(p, _mm512_setzero_si512());
    const __m512i t = _mm512_add_epi64(p, _mm512_mask_blend_epi64(neg, _mm512_set1_epi64(1ll << 30), _mm512_set1_epi64(1ll - (1ll << 30))));
    const __mmask8 tneg = _mm512_cmplt_epi64_mask(t, _mm512_setzero_si512());
    return _mm512_srai_epi64(_mm512_mask_add_epi64(t, tneg, t, _mm512_set1_epi64((1ll << 31) - 1)), 31);
}
static inline __m512i mbqm16(__m512i x, __m512i mult, __m512i shift) {
    const __m512i zero = _mm512_setzero_si512(), one = _mm512_set1_epi32(1);
    const __m512i left = _mm512_max_epi32(shift, zero), right = _mm512_max_epi32(_mm512_sub_epi32(zero, shift), zero);
    const __m512i xl = _mm512_sllv_epi32(x, left); /* x * (1 << left), int32 wrap-around */
    const __m512i re = srdhm_half(_mm512_mul_epi32(xl, mult));
    const __m512i ro = srdhm_half(_mm512_mul_epi32(_mm512_srli_epi64(xl, 32), _mm512_srli_epi64(mult, 32)));
    __m512i v = _mm512_mask_blend_epi32(0xAAAA, re, _mm512_slli_epi64(ro, 32));
    const __mmask16 sat = _mm512_cmpeq_epi32_mask(xl, _mm512_set1_epi32(INT32_MIN)) & _mm512_cmpeq_epi32_mask(mult, _mm512_set1_epi32(INT32_MIN));
    v = _mm512_mask_blend_epi32(sat, v, _mm512_set1_epi32(INT32_MAX));
    const __m512i mask = _mm512_sub_epi32(_mm512_sllv_epi32(one, right), one);
    const __m512i rem = _mm512_and_si512(v, mask);
    const __m512i thr = _mm512_add_epi32(_mm512_srli_epi32(mask, 1), _mm512_srli_epi32(v, 31));
    return _mm512_mask_add_epi32(_mm512_srav_epi32(v, right), _mm512_cmpgt_epi32_mask(rem, thr), _mm512_srav_epi32(v, right), one);
}
static inline void store16_i8(int8_t* dst, __m512i v, int zp, int lo, int hi, __mmask16 live) {
    v = _mm512_min_epi32(_mm512_max_epi32(_mm512_add_epi32(v, _mm512_set1_epi32(zp)), _mm512_set1_epi32(lo)), _mm512_set1_epi32(hi));
    _mm_mask_storeu_epi8(dst, live, _mm512_cvtepi32_epi8(v));
}
static inline __mmask16 live16(int n0, int N) { return N - n0 >= 16 ? (__mmask16)0xFFFF : (__mmask16)((1u << (N - n0)) - 1u); }

/* 1x1 convolution, stride 1: P positions x Cin -> Cout */
static void conv1x1_vnni(const int8_t* x, int8_t* y, long P, int Cin, int Cout, const int8_t* w, const int32_t* bias, int zp_in, int zp_out,
                         const int32_t* mult, const int32_t* shift, int amin, int amax) {
    const int K4 = (Cin + 3) / 4, N16 = (Cout + 15) / 16;
    int8_t* wp = (int8_t*)aligned_alloc(64, (size_t)K4 * N16 * 64);
    int32_t* cst = (int32_t*)aligned_alloc(64, (size_t)N16 * 16 * 3 * 4);
    memset(wp, 0, (size_t)K4 * N16 * 64);
    for (int n = 0; n < N16 * 16; ++n) {
        int32_t ws = 0;
        if (n < Cout)
            for (int c = 0; c < Cin; ++c) {
                wp[((size_t)(c / 4) * N16 + n / 16) * 64 + (n % 16) * 4 + c % 4] = w[(size_t)n * Cin + c];
                ws += w[(size_t)n * Cin + c];
            }
        cst[n] = n < Cout ? (bias ? bias[n] : 0) - (128 + zp_in) * ws : 0;
        cst[N16 * 16 + n] = n < Cout ? mult[n] : 0;
        cst[2 * N16 * 16 + n] = n < Cout ? shift[n] : 0;
    }
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p) {
        uint32_t xu[K4];
        uint8_t tmp[4 * K4];
        memset(tmp, 0, sizeof tmp);
        memcpy(tmp, x + p * Cin, Cin);
        memcpy(xu, tmp, sizeof tmp);
        for (int k = 0; k < K4; ++k) xu[k] ^= 0x80808080u; /* (padding bytes meet zero weights) */
        for (int nb = 0; nb < N16; ++nb) {
            __m512i acc = _mm512_load_si512(cst + 16 * nb);
            for (int k = 0; k < K4; ++k)
                acc = _mm512_dpbusd_epi32(acc, _mm512_set1_epi32((int)xu[k]), _mm512_load_si512(wp + ((size_t)k * N16 + nb) * 64));
            store16_i8(y + p * Cout + 16 * nb, mbqm16(acc, _mm512_load_si512(cst + N16 * 16 + 16 * nb), _mm512_load_si512(cst + 2 * N16 * 16 + 16 * nb)),
                       zp_out, amin, amax, live16(16 * nb, Cout));
        }
    }
    free(wp);
    free(cst);
}
#endif

/* x [B][H][W][Cin], w [Cout][kh][kw][Cin], y [B][OH][OW][Cout] */
void oi_conv(const int8_t* x, int8_t* y, int B, int H, int W, int Cin, int kh, int kw, int Cout, int sh, int sw, int OH, int OW,
             int pt, int pl, const int8_t* w, const int32_t* bias, int zp_in, int zp_out, const int32_t* mult, const int32_t* shift,
             int amin, int amax) {
#if OI_VEC
    if (kh == 1 && kw == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0 && OH == H && OW == W) {
        conv1x1_vnni(x, y, (long)B * H * W, Cin, Cout, w, bias, zp_in, zp_out, mult, shift, amin, amax);
        return;
    }
    {   /* any other convolution (the 3x3 stem): sixteen output channels per register, one broadcast input value per multiply-add */
        const int N16 = (Cout + 15) / 16, taps = kh * kw * Cin;
        int32_t* wp = (int32_t*)aligned_alloc(64, (size_t)taps * N16 * 64);
        int32_t* cst = (int32_t*)aligned_alloc(64, (size_t)N16 * 16 * 3 * 4);
        for (int n = 0; n < N16 * 16; ++n) {
            for (int t = 0; t < taps; ++t) wp[((size_t)t * N16 + n / 16) * 16 + n % 16] = n < Cout ? w[(size_t)n * taps + t] : 0;
            cst[n] = n < Cout && bias ? bias[n] : 0;
            cst[N16 * 16 + n] = n < Cout ? mult[n] : 0;
            cst[2 * N16 * 16 + n] = n < Cout ? shift[n] : 0;
        }
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int oh = 0; oh < OH; ++oh)
                for (int ow = 0; ow < OW; ++ow)
                    for (int nb = 0; nb < N16; ++nb) {
                        __m512i acc = _mm512_load_si512(cst + 16 * nb);
                        for (int i = 0; i < kh; ++i) {
                            const int ih = oh * sh - pt + i;
                            if (ih < 0 || ih >= H) continue;
                            for (int j = 0; j < kw; ++j) {
                                const int iw = ow * sw - pl + j;
                                if (iw < 0 || iw >= W) continue;
                                const int8_t* xi = x + (((size_t)b * H + ih) * W + iw) * Cin;
                                for (int c = 0; c < Cin; ++c)
                                    acc = _mm512_add_epi32(acc, _mm512_mullo_epi32(_mm512_set1_epi32((int32_t)xi[c] - zp_in),
                                                                                   _mm512_load_si512(wp + ((size_t)((i * kw + j) * Cin + c) * N16 + nb) * 16)));
                            }
                        }
                        store16_i8(y + (((size_t)b * OH + oh) * OW + ow) * Cout + 16 * nb,
                                   mbqm16(acc, _mm512_load_si512(cst + N16 * 16 + 16 * nb), _mm512_load_si512(cst + 2 * N16 * 16 + 16 * nb)), zp_out, amin, amax,
                                   live16(16 * nb, Cout));
                    }
        free(wp);
        free(cst);
        return;
    }
#endif
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
#if OI_VEC
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int oh = 0; oh < OH; ++oh)
            for (int ow = 0; ow < OW; ++ow)
                for (int c0 = 0; c0 < C; c0 += 16) {
                    const __mmask16 live = live16(c0, C);
                    __m512i acc = bias ? _mm512_maskz_loadu_epi32(live, bias + c0) : _mm512_setzero_si512();
                    for (int i = 0; i < kh; ++i) {
                        const int ih = oh * sh - pt + i;
                        if (ih < 0 || ih >= H) continue;
                        for (int j = 0; j < kw; ++j) {
                            const int iw = ow * sw - pl + j;
                            if (iw < 0 || iw >= W) continue;
                            const __m512i xv = _mm512_sub_epi32(_mm512_cvtepi8_epi32(_mm_maskz_loadu_epi8(live, x + (((size_t)b * H + ih) * W + iw) * C + c0)),
                                                                _mm512_set1_epi32(zp_in));
                            const __m512i wv = _mm512_cvtepi8_epi32(_mm_maskz_loadu_epi8(live, w + ((size_t)i * kw + j) * C + c0));
                            acc = _mm512_add_epi32(acc, _mm512_mullo_epi32(xv, wv));
                        }
                    }
                    store16_i8(y + (((size_t)b * OH + oh) * OW + ow) * C + c0,
                               mbqm16(acc, _mm512_maskz_loadu_epi32(live, mult + c0), _mm512_maskz_loadu_epi32(live, shift + c0)), zp_out, amin, amax, live);
                }
    return;
#endif
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
#if OI_VEC
    if (nb == n) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; i += 16) {
            const __mmask16 live = n - i >= 16 ? (__mmask16)0xFFFF : (__mmask16)((1u << (n - i)) - 1u);
            const __m512i av = _mm512_slli_epi32(_mm512_sub_epi32(_mm512_cvtepi8_epi32(_mm_maskz_loadu_epi8(live, a + i)), _mm512_set1_epi32(z1)), 20);
            const __m512i bv = _mm512_slli_epi32(_mm512_sub_epi32(_mm512_cvtepi8_epi32(_mm_maskz_loadu_epi8(live, b + i)), _mm512_set1_epi32(z2)), 20);
            const __m512i sa = mbqm16(av, _mm512_set1_epi32(m1), _mm512_set1_epi32(s1)), sb = mbqm16(bv, _mm512_set1_epi32(m2), _mm512_set1_epi32(s2));
            store16_i8(y + i, mbqm16(_mm512_add_epi32(sa, sb), _mm512_set1_epi32(mo), _mm512_set1_epi32(so)), zo, amin, amax, live);
        }
        return;
    }
    if (nb % 16 == 0 && n % nb == 0) {   /* b broadcast with a period of whole vectors (a per-channel constant): the same sixteen-lane form */
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; i += 16) {
            const __m512i av = _mm512_slli_epi32(_mm512_sub_epi32(_mm512_cvtepi8_epi32(_mm_loadu_si128((const __m128i*)(a + i))), _mm512_set1_epi32(z1)), 20);
            const __m512i bv = _mm512_slli_epi32(_mm512_sub_epi32(_mm512_cvtepi8_epi32(_mm_loadu_si128((const __m128i*)(b + i % nb))), _mm512_set1_epi32(z2)), 20);
            const __m512i sa = mbqm16(av, _mm512_set1_epi32(m1), _mm512_set1_epi32(s1)), sb = mbqm16(bv, _mm512_set1_epi32(m2), _mm512_set1_epi32(s2));
            store16_i8(y + i, mbqm16(_mm512_add_epi32(sa, sb), _mm512_set1_epi32(mo), _mm512_set1_epi32(so)), zo, amin, amax, (__mmask16)0xFFFF);
        }
        return;
    }
#endif
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

/* ---- the whole graph per chunk (CPU baseline: no interpreter between the operators) -------------------------------------------------------
 * oracle/cport.py: CpuInt8Program traces the numpy interpreter ONCE (batch 1) and writes the graph down as a flat program: one record per
 * operator with its shapes, quantisation parameters and constants; every pure data-movement operator (TRANSPOSE, STRIDED_SLICE,
 * CONCATENATION with a constant, RESHAPE) as ONE gather map.  oi_program_run walks the program for every chunk, the chunks dealt to the
 * OpenMP threads (a thread's activations — ~1.5 MB for the shipped graph — stay in its cache), the kernels above called with batch 1
 * (their own parallel regions collapse to the calling thread inside this one).  1x1 convolutions take weights re-packed ONCE
 * (oi_pack_1x1).  Same integers as the interpreter: tests/test_oracle_pinning.py compares every output and, through the records' tensor
 * ids, every intermediate tensor. */
enum { OI_QUANT = 1, OI_GATHER = 2, OI_CONV = 3, OI_DWCONV = 4, OI_ADD = 5, OI_MEAN = 6, OI_FC = 7, OI_LUT = 8, OI_DEQUANT = 9 };
typedef struct {
    int32_t kind, in0, in1, out;
    int64_t n;           /* elements of the output */
    int32_t p[24];
    float f[4];
    const void* ptr[6];
} oi_op;

#if OI_VEC
/* weights [Cout][Cin] -> [K/4][N/16][16][4] + (bias - (128 + zp_in) sum w, mult, shift) per padded channel; returns bytes written to wp / cst */
void oi_pack_1x1(const int8_t* w, const int32_t* bias, const int32_t* mult, const int32_t* shift, int Cin, int Cout, int zp_in, int8_t* wp, int32_t* cst) {
    const int K4 = (Cin + 3) / 4, N16 = (Cout + 15) / 16;
    memset(wp, 0, (size_t)K4 * N16 * 64);
    for (int n = 0; n < N16 * 16; ++n) {
        int32_t ws = 0;
        if (n < Cout)
            for (int c = 0; c < Cin; ++c) {
                wp[((size_t)(c / 4) * N16 + n / 16) * 64 + (n % 16) * 4 + c % 4] = w[(size_t)n * Cin + c];
                ws += w[(size_t)n * Cin + c];
            }
        cst[n] = n < Cout ? (bias ? bias[n] : 0) - (128 + zp_in) * ws : 0;
        cst[N16 * 16 + n] = n < Cout ? mult[n] : 0;
        cst[2 * N16 * 16 + n] = n < Cout ? shift[n] : 0;
    }
}
static void conv1x1_packed(const int8_t* x, int8_t* y, long P, int Cin, int Cout, const int8_t* wp, const int32_t* cst, int zp_out, int amin, int amax) {
    const int K4 = (Cin + 3) / 4, N16 = (Cout + 15) / 16;
    for (long p = 0; p < P; ++p) {
        uint32_t xu[K4];
        uint8_t tmp[4 * K4];
        memset(tmp, 0, sizeof tmp);
        memcpy(tmp, x + p * Cin, Cin);
        memcpy(xu, tmp, sizeof tmp);
        for (int k = 0; k < K4; ++k) xu[k] ^= 0x80808080u;
        for (int nb = 0; nb < N16; ++nb) {
            __m512i acc = _mm512_load_si512(cst + 16 * nb);
            for (int k = 0; k < K4; ++k)
                acc = _mm512_dpbusd_epi32(acc, _mm512_set1_epi32((int)xu[k]), _mm512_load_si512(wp + ((size_t)k * N16 + nb) * 64));
            store16_i8(y + p * Cout + 16 * nb, mbqm16(acc, _mm512_load_si512(cst + N16 * 16 + 16 * nb), _mm512_load_si512(cst + 2 * N16 * 16 + 16 * nb)),
                       zp_out, amin, amax, live16(16 * nb, Cout));
        }
    }
}
#else
void oi_pack_1x1(const int8_t* w, const int32_t* bias, const int32_t* mult, const int32_t* shift, int Cin, int Cout, int zp_in, int8_t* wp, int32_t* cst) {
    (void)w; (void)bias; (void)mult; (void)shift; (void)Cin; (void)Cout; (void)zp_in; (void)wp; (void)cst;   /* (the portable build convolves from the plain weights) */
}
#endif

static void run_one(const oi_op* ops, int n_ops, const float* xin, float* yout, int8_t* arena, const int64_t* off) {
    for (int i = 0; i < n_ops; ++i) {
        const oi_op* o = &ops[i];
        const int32_t* p = o->p;
        int8_t* y = arena + off[o->out];
        const int8_t* a = o->in0 >= 0 ? arena + off[o->in0] : NULL;
        switch (o->kind) {
            case OI_QUANT: {   /* q = clamp(round_half_away(x / s) + zp): float32 division, like the interpreter */
                const float s = o->f[0];
                int64_t k0 = 0;
#if OI_VEC
                /* sixteen at a time, the same operations: IEEE division, truncation, the step away from zero where the fraction reaches a half;
                 * the float is clamped to +-512 before the conversion (the scalar form clamps the int64 afterwards: same byte) — a quarter of the
                 * per-chunk time went here in scalar code */
                for (; k0 + 16 <= o->n; k0 += 16) {
                    const __m512 v = _mm512_div_ps(_mm512_loadu_ps(xin + k0), _mm512_set1_ps(s));
                    __m512 r = _mm512_roundscale_ps(v, _MM_FROUND_TO_ZERO | _MM_FROUND_NO_EXC);
                    const __m512 d = _mm512_sub_ps(v, r);
                    r = _mm512_mask_add_ps(r, _mm512_cmp_ps_mask(d, _mm512_set1_ps(0.5f), _CMP_GE_OQ), r, _mm512_set1_ps(1.0f));
                    r = _mm512_mask_sub_ps(r, _mm512_cmp_ps_mask(d, _mm512_set1_ps(-0.5f), _CMP_LE_OQ), r, _mm512_set1_ps(1.0f));
                    r = _mm512_min_ps(_mm512_max_ps(r, _mm512_set1_ps(-512.0f)), _mm512_set1_ps(512.0f));
                    __m512i q = _mm512_add_epi32(_mm512_cvtps_epi32(r), _mm512_set1_epi32(p[0]));
                    q = _mm512_min_epi32(_mm512_max_epi32(q, _mm512_set1_epi32(-128)), _mm512_set1_epi32(127));
                    _mm_storeu_si128((__m128i*)(y + k0), _mm512_cvtepi32_epi8(q));
                }
#endif
                for (int64_t k = k0; k < o->n; ++k) {
                    const float v = xin[k] / s;
                    float r = (float)(int64_t)v;                                   /* trunc; |v - trunc| >= 0.5 -> one step away from zero (C round()) */
                    const float d = v - r;
                    if (d >= 0.5f) r += 1.0f; else if (d <= -0.5f) r -= 1.0f;
                    int64_t q = (int64_t)r + p[0];
                    y[k] = (int8_t)(q < -128 ? -128 : (q > 127 ? 127 : q));
                }
                break;
            }
            case OI_GATHER: {  /* out[k] = idx[k] >= 0 ? in[idx[k]] : fill[k] */
                const int32_t* idx = (const int32_t*)o->ptr[0];
                const int8_t* fill = (const int8_t*)o->ptr[1];
                for (int64_t k = 0; k < o->n; ++k) y[k] = idx[k] >= 0 ? a[idx[k]] : fill[k];
                break;
            }
            case OI_CONV:  /* p: H W Cin kh kw Cout sh sw OH OW pt pl zp_in zp_out amin amax packed */
#if OI_VEC
                if (p[16]) {
                    conv1x1_packed(a, y, (long)p[0] * p[1], p[2], p[5], (const int8_t*)o->ptr[4], (const int32_t*)o->ptr[5], p[13], p[14], p[15]);
                    break;
                }
#endif
                oi_conv(a, y, 1, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], (const int8_t*)o->ptr[0], (const int32_t*)o->ptr[1], p[12], p[13],
                        (const int32_t*)o->ptr[2], (const int32_t*)o->ptr[3], p[14], p[15]);
                break;
            case OI_DWCONV:  /* p: H W C kh kw - sh sw OH OW pt pl zp_in zp_out amin amax */
                oi_dwconv(a, y, 1, p[0], p[1], p[2], p[3], p[4], p[6], p[7], p[8], p[9], p[10], p[11], (const int8_t*)o->ptr[0], (const int32_t*)o->ptr[1], p[12], p[13],
                          (const int32_t*)o->ptr[2], (const int32_t*)o->ptr[3], p[14], p[15]);
                break;
            case OI_ADD: {   /* p: nb z1 m1 s1 z2 m2 s2 mo so zo amin amax const_b */
                const int8_t* b = p[12] ? (const int8_t*)o->ptr[0] : arena + off[o->in1];
                oi_add(a, b, y, (long)o->n, (long)p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]);
                break;
            }
            case OI_MEAN:  /* p: P C zp_in mult shift zp_out */
                oi_mean(a, y, 1, p[0], p[1], p[2], p[3], p[4], p[5]);
                break;
            case OI_FC:    /* p: Cin Cout zp_in zp_out amin amax */
                oi_fc(a, y, 1, p[0], p[1], (const int8_t*)o->ptr[0], (const int32_t*)o->ptr[1], p[2], p[3], (const int32_t*)o->ptr[2], (const int32_t*)o->ptr[3], p[4], p[5]);
                break;
            case OI_LUT: {
                const int8_t* lut = (const int8_t*)o->ptr[0];
                for (int64_t k = 0; k < o->n; ++k) y[k] = lut[(int)a[k] + 128];
                break;
            }
            case OI_DEQUANT:
                for (int64_t k = 0; k < o->n; ++k) yout[k] = (float)((int32_t)a[k] - p[0]) * o->f[0];
                break;
            default: break;
        }
    }
}

/* x [B][in_elems] float32 -> out [B][out_elems] float32; off[t] = byte offset of tensor t in a thread's arena of arena_bytes.
 * keep (or NULL): [B][arena_bytes] — every chunk's arena copied out (the per-tensor comparison of the tests). */
int oi_program_run(const oi_op* ops, int n_ops, const float* x, int B, int64_t in_elems, float* out, int64_t out_elems, const int64_t* off,
                   int64_t arena_bytes, int8_t* keep) {
    int failed = 0;
#pragma omp parallel
    {
        /* a thread's arena lives as long as the thread (OpenMP keeps its pool between calls): a fresh 1.2 MB allocation per call is 300 page
         * faults per thread under one address-space lock — with 128 threads that, not the arithmetic, set the rate */
        static _Thread_local int8_t* arena = NULL;
        static _Thread_local int64_t arena_cap = 0;
        if (arena_cap < arena_bytes) {
            free(arena);
            arena = (int8_t*)aligned_alloc(64, (size_t)((arena_bytes + 63) & ~63LL));
            arena_cap = arena ? arena_bytes : 0;
            if (arena) memset(arena, 0, (size_t)arena_bytes);
        }
        if (!arena) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            if (!arena) continue;
            run_one(ops, n_ops, x + (size_t)b * in_elems, out + (size_t)b * out_elems, arena, off);
            if (keep) memcpy(keep + (size_t)b * arena_bytes, arena, (size_t)arena_bytes);
        }
    }
    return failed ? -1 : 0;
}
int oi_op_bytes(void) { return (int)sizeof(oi_op); }
