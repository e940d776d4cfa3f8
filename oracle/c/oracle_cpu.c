/* oracle_cpu.c — plain-C restatement of the float hot path, for the CPU baseline and as a second
 * opinion on the numpy oracle.  TEST INFRASTRUCTURE ONLY: nothing under birdnet-stm32_amd/ links or
 * loads this file (see oracle/__init__.py).
 *
 * Follows the reference's evaluate path:
 *   oc_stft_norm    birdnet_stm32/audio/spectrogram.py:12-21,61,106-115,133,149 (librosa-style framing:
 *                   centre zero pad, periodic Hann, float64 FFT, |.| in float32, min-max normalise)
 *   oc_mel_pwl      birdnet_stm32/models/frontend.py:299-345 + magnitude.py:179-192 (hybrid, norm optional)
 *   oc_conv3x3_c1 / oc_dw3x3 / oc_pw   birdnet_stm32/models/dscnn.py:28-84,198-202 with BatchNorm folded by
 *                   the caller (oracle/cport.py), TensorFlow SAME padding, ReLU6
 *   oc_gap_dense    birdnet_stm32/models/dscnn.py:256-261
 * Parallelism: OpenMP over chunks (each chunk is independent in the reference too).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NFFT 512
#define NBIN 257

int oc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* threads of the following parallel regions (oracle/cport.py sets the process's CPU SHARE — affinity mask and cgroup quota — not the host's CPU count) */
void oc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* in-place radix-2 DIT complex FFT, double precision, n = 512 */
static void fft512(double* re, double* im, const double* cs, const double* sn, const int* rev) {
    for (int i = 0; i < NFFT; ++i) {
        int j = rev[i];
        if (j > i) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= NFFT; len <<= 1) {
        int half = len >> 1, step = NFFT / len;
        for (int s = 0; s < NFFT; s += len)
            for (int k = 0; k < half; ++k) {
                double wr = cs[k * step], wi = -sn[k * step];
                int a = s + k, b = a + half;
                double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

void oc_stft_norm(const float* audio, int B, int T, int hop, int W, float* out) {
    static double win[NFFT], cs[NFFT / 2], sn[NFFT / 2];
    static int rev[NFFT];
    for (int i = 0; i < NFFT; ++i) {
        win[i] = 0.5 - 0.5 * cos(2.0 * M_PI * i / NFFT);
        int r = 0;
        for (int b = 0; b < 9; ++b) r |= ((i >> b) & 1) << (8 - b);
        rev[i] = r;
    }
    for (int i = 0; i < NFFT / 2; ++i) { cs[i] = cos(2.0 * M_PI * i / NFFT); sn[i] = sin(2.0 * M_PI * i / NFFT); }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float* x = audio + (size_t)b * T;
        float* S = out + (size_t)b * NBIN * W;
        double re[NFFT], im[NFFT];
        float mn = INFINITY, mx = -INFINITY;
        for (int t = 0; t < W; ++t) {
            long s0 = (long)t * hop - NFFT / 2;
            for (int i = 0; i < NFFT; ++i) {
                long g = s0 + i;
                re[i] = (g >= 0 && g < T) ? (double)x[g] * win[i] : 0.0;
                im[i] = 0.0;
            }
            fft512(re, im, cs, sn, rev);
            for (int k = 0; k < NBIN; ++k) {
                float fr = (float)re[k], fi = (float)im[k]; /* complex64 storage */
                float m = hypotf(fr, fi);
                S[(size_t)k * W + t] = m;
                if (m < mn) mn = m;
                if (m > mx) mx = m;
            }
        }
        float rng = (float)((double)(mx - mn) + 1e-10);
        for (size_t i = 0; i < (size_t)NBIN * W; ++i) S[i] = (S[i] - mn) / rng;
    }
}

/* spec [B][F][W] -> frontend out [B][M][W]; mel [Fp][M]; pwl rows k0,k1..3,w1..3,b1..3 ([10][M]) or NULL */
void oc_mel_pwl(const float* spec, int B, int F, int W, int M, const float* mel, const float* pwl, int norm, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float* S = spec + (size_t)b * F * W;
        float* y = out + (size_t)b * M * W;
        float peak = 0.f;
        for (int m = 0; m < M; ++m)
            for (int t = 0; t < W; ++t) {
                float acc = 0.f;
                for (int f = 0; f < F; ++f) acc += S[(size_t)f * W + t] * mel[(size_t)f * M + m];
                acc = acc > 0.f ? acc : 0.f;
                y[(size_t)m * W + t] = acc;
                if (acc > peak) peak = acc;
            }
        for (int m = 0; m < M; ++m)
            for (int t = 0; t < W; ++t) {
                float v = y[(size_t)m * W + t];
                if (norm) v = v / (peak + 1e-6f);
                if (pwl) {
                    float o = pwl[m] * v;
                    for (int i = 0; i < 3; ++i) {
                        float z = pwl[(4 + i) * M + m] * v + pwl[(7 + i) * M + m];
                        o += pwl[(1 + i) * M + m] * (z > 0.f ? z : 0.f);
                    }
                    v = o;
                }
                y[(size_t)m * W + t] = v;
            }
    }
}

static inline float act6(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    return v;
}

/* x [B][H][W] -> y [B][OH][OW][C]; w [3][3][C] */
void oc_conv3x3_c1(const float* x, float* y, int B, int H, int W, int C, int sh, int sw, int OH, int OW, int pt, int pl,
                   const float* w, const float* bias, int act) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        for (int oh = 0; oh < OH; ++oh)
            for (int ow = 0; ow < OW; ++ow) {
                float* o = y + (((size_t)b * OH + oh) * OW + ow) * C;
                for (int c = 0; c < C; ++c) o[c] = bias[c];
                for (int i = 0; i < 3; ++i) {
                    int ih = oh * sh + i - pt;
                    if (ih < 0 || ih >= H) continue;
                    for (int j = 0; j < 3; ++j) {
                        int iw = ow * sw + j - pl;
                        if (iw < 0 || iw >= W) continue;
                        float v = x[((size_t)b * H + ih) * W + iw];
                        const float* k = w + (i * 3 + j) * C;
                        for (int c = 0; c < C; ++c) o[c] += v * k[c];
                    }
                }
                for (int c = 0; c < C; ++c) o[c] = act6(o[c], act);
            }
}

void oc_dw3x3(const float* x, float* y, int B, int H, int W, int C, int sh, int sw, int OH, int OW, int pt, int pl,
              const float* w, const float* bias, int act) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        for (int oh = 0; oh < OH; ++oh)
            for (int ow = 0; ow < OW; ++ow) {
                float* o = y + (((size_t)b * OH + oh) * OW + ow) * C;
                for (int c = 0; c < C; ++c) o[c] = bias[c];
                for (int i = 0; i < 3; ++i) {
                    int ih = oh * sh + i - pt;
                    if (ih < 0 || ih >= H) continue;
                    for (int j = 0; j < 3; ++j) {
                        int iw = ow * sw + j - pl;
                        if (iw < 0 || iw >= W) continue;
                        const float* v = x + (((size_t)b * H + ih) * W + iw) * C;
                        const float* k = w + (i * 3 + j) * C;
                        for (int c = 0; c < C; ++c) o[c] += v[c] * k[c];
                    }
                }
                for (int c = 0; c < C; ++c) o[c] = act6(o[c], act);
            }
}

/* x [R][Cin] -> y [R][Cout], w [Cin][Cout], optional residual */
void oc_pw(const float* x, const float* res, float* y, long R, int Cin, int Cout, const float* w, const float* bias, int act) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        float* o = y + r * Cout;
        const float* xi = x + r * Cin;
        for (int n = 0; n < Cout; ++n) o[n] = bias[n];
        for (int k = 0; k < Cin; ++k) {
            float v = xi[k];
            const float* wk = w + (size_t)k * Cout;
            for (int n = 0; n < Cout; ++n) o[n] += v * wk[n];
        }
        if (res) for (int n = 0; n < Cout; ++n) o[n] += res[r * Cout + n];
        for (int n = 0; n < Cout; ++n) o[n] = act6(o[n], act);
    }
}

/* x [B][P][C] -> logits/scores [B][N]; w [C][N]; act 1 = sigmoid, 2 = softmax */
void oc_gap_dense(const float* x, int B, int P, int C, int N, const float* w, const float* bias, int act, float* logits, float* scores) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float g[1024];
        for (int c = 0; c < C; ++c) g[c] = 0.f;
        for (int p = 0; p < P; ++p)
            for (int c = 0; c < C; ++c) g[c] += x[((size_t)b * P + p) * C + c];
        for (int c = 0; c < C; ++c) g[c] /= (float)P;
        float* z = logits + (size_t)b * N;
        for (int n = 0; n < N; ++n) z[n] = bias[n];
        for (int c = 0; c < C; ++c)
            for (int n = 0; n < N; ++n) z[n] += g[c] * w[(size_t)c * N + n];
        float* s = scores + (size_t)b * N;
        if (act == 2) {
            float mx = z[0], den = 0.f;
            for (int n = 1; n < N; ++n) if (z[n] > mx) mx = z[n];
            for (int n = 0; n < N; ++n) { s[n] = expf(z[n] - mx); den += s[n]; }
            for (int n = 0; n < N; ++n) s[n] /= den;
        } else {
            for (int n = 0; n < N; ++n) s[n] = act == 1 ? 1.f / (1.f + expf(-z[n])) : z[n];
        }
    }
}
