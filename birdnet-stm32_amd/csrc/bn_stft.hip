// bn_stft.hip — batched 512-point STFT magnitude for gfx950.
//
// Replaces the reference's per-chunk numpy/librosa call
//   np.abs(librosa.stft(y, n_fft=512, hop_length=len(y)//W, win_length=512, window='hann'))[:, :W]
// (reference: birdnet_stm32/audio/spectrogram.py:61,106-115,133) and its min-max
// normalisation (:12-21,149).  Framing: centre zero padding of 256 samples, periodic Hann,
// frame t = samples [t*hop-256, t*hop+256).
//
// One 512-point real FFT = one 256-point complex FFT on z[m] = x[2m] + i x[2m+1] followed by
// the split post-pass.  The 256-point FFT is two radix-16 passes: 16 lanes own one frame, each
// lane runs a 16-point FFT entirely in registers, the 16x16 transpose between the passes and the
// k <-> 256-k pairing of the post-pass go through LDS.  A 256-thread workgroup therefore
// transforms 16 consecutive frames of one chunk; magnitudes are staged in LDS and written as
// frequency-major rows so that the global stores are 64-byte runs.
#include <cstdlib>

#include "bn_kernels.h"

namespace bn {

namespace {

constexpr int kFT = 16;    // frames per workgroup
constexpr int kFS = 272;   // complex elements reserved per frame in the exchange buffer

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// forward 4-point DFT, natural order in and out
__device__ __forceinline__ void fft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 s02 = cadd(x0, x2), d02 = csub(x0, x2);
    const float2 s13 = cadd(x1, x3), d13 = csub(x1, x3);
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = make_float2(d02.x + d13.y, d02.y - d13.x);  // d02 - i d13
    x3 = make_float2(d02.x - d13.y, d02.y + d13.x);  // d02 + i d13
}

// forward 16-point DFT in registers: n = 4p+q, k = r+4s
__device__ __forceinline__ void fft16(float2 (&a)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
    for (int q = 0; q < 4; ++q) fft4(a[q], a[4 + q], a[8 + q], a[12 + q]);
    // a[4r+q] *= W16^(q r)
    a[4 * 1 + 1] = cmul(a[4 * 1 + 1], make_float2(c1, -s1));
    a[4 * 1 + 2] = cmul(a[4 * 1 + 2], make_float2(h, -h));
    a[4 * 1 + 3] = cmul(a[4 * 1 + 3], make_float2(s1, -c1));
    a[4 * 2 + 1] = cmul(a[4 * 2 + 1], make_float2(h, -h));
    a[4 * 2 + 2] = make_float2(a[4 * 2 + 2].y, -a[4 * 2 + 2].x);  // * (-i)
    a[4 * 2 + 3] = cmul(a[4 * 2 + 3], make_float2(-h, -h));
    a[4 * 3 + 1] = cmul(a[4 * 3 + 1], make_float2(s1, -c1));
    a[4 * 3 + 2] = cmul(a[4 * 3 + 2], make_float2(-h, -h));
    a[4 * 3 + 3] = cmul(a[4 * 3 + 3], make_float2(-c1, s1));
#pragma unroll
    for (int r = 0; r < 4; ++r) fft4(a[4 * r], a[4 * r + 1], a[4 * r + 2], a[4 * r + 3]);
    // X[r+4s] sits in a[4r+s]: transpose the register names
    float2 b[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[r + 4 * s] = a[4 * r + s];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = b[i];
}

__global__ void minmax_init_kernel(float* minmax, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        minmax[2 * i] = __uint_as_float(0x7f800000u);  // +inf
        minmax[2 * i + 1] = 0.0f;
    }
}

// With MEL_OUT the band-sparse mel mixer is applied while the magnitudes of the tile are still in LDS and the kernel
// writes [B][M][W] un-normalised mel energies (64 KB/chunk) instead of the [B][257][W] spectrogram (263 KB/chunk).
// min/max of the magnitudes still go to `minmax`: the reference's min-max normalisation commutes with the linear mixer,
//   mel((S - mn)/rng)[m] = (mel(S)[m] - mn * sum_f w[f][m]) / rng,   and is applied by the consumer.
struct MelOut {
    const float* wvals;  // band-sparse mixer values
    const int* bands;    // [3][M] start, len, offset
    float* out;          // [B][M][W]
    int M;
};

template <bool MEL_OUT>
__global__ __launch_bounds__(256) void stft512_mag_kernel(StftTables tb, const float* __restrict__ audio, int T, int hop,
                                                          int W, float* __restrict__ spec, float* minmax, MelOut mel) {
    __shared__ float2 xch[kFT][kFS];
    __shared__ float mag[257][kFT + 1];
    __shared__ float red_min[4], red_max[4];

    const int b = blockIdx.y;
    const int t0 = blockIdx.x * kFT;
    const int f = threadIdx.x >> 4;
    const int j = threadIdx.x & 15;
    const int t = t0 + f;
    const float* x = audio + (size_t)b * T;
    const long start = (long)t * hop - 256;

    // pass 1: lane j owns z[16 n1 + j], n1 = 0..15
    float2 a[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int i0 = 2 * (16 * n1 + j);
        // unconditional loads from clamped addresses (all 32 in flight at once), zeroed afterwards when out of range
        const long g0 = start + i0, g1 = g0 + 1;
        const long c0 = g0 < 0 ? 0 : (g0 >= T ? T - 1 : g0), c1 = g1 < 0 ? 0 : (g1 >= T ? T - 1 : g1);
        float v0 = x[c0], v1 = x[c1];
        v0 = (g0 == c0) ? v0 : 0.0f;
        v1 = (g1 == c1) ? v1 : 0.0f;
        a[n1] = make_float2(v0 * tb.window[i0], v1 * tb.window[i0 + 1]);
    }
    fft16(a);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) xch[f][k1 * 17 + j] = cmul(a[k1], tb.tw256[j * k1]);
    __syncthreads();

    // pass 2: lane j is now k1; gathers over n2
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) a[n2] = xch[f][j * 17 + n2];
    fft16(a);  // a[k2] = Z[j + 16 k2]
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) xch[f][j + 16 * k2] = a[k2];
    __syncthreads();

    // split post-pass: X[k] = E - i W512^k O, E = (Z[k] + conj Z[256-k])/2, O = (Z[k] - conj Z[256-k])/2
    float lmin = __uint_as_float(0x7f800000u), lmax = 0.0f;
    const bool live = t < W;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        const int k = j + 16 * k2;
        const float2 z = a[k2];
        const float2 p = xch[f][(256 - k) & 255];
        const float er = 0.5f * (z.x + p.x), ei = 0.5f * (z.y - p.y);
        const float orr = 0.5f * (z.x - p.x), oi = 0.5f * (z.y + p.y);
        const float2 w = tb.tw512[k];  // (cos, -sin)
        const float c = w.x, sn = -w.y;
        const float re = er - sn * orr + c * oi;
        const float im = ei - c * orr - sn * oi;
        const float m = sqrtf(re * re + im * im);
        mag[k][f] = m;
        if (live) {
            lmin = fminf(lmin, m);
            lmax = fmaxf(lmax, m);
        }
    }
    if (j == 0) {
        const float m = fabsf(a[0].x - a[0].y);  // Nyquist bin
        mag[256][f] = m;
        if (live) {
            lmin = fminf(lmin, m);
            lmax = fmaxf(lmax, m);
        }
    }
    __syncthreads();

    if (!MEL_OUT) {
        // frequency-major rows, 16 consecutive frames each
        float* out = spec + (size_t)b * 257 * W;
        for (int idx = threadIdx.x; idx < 257 * kFT; idx += 256) {
            const int k = idx / kFT, ff = idx % kFT;
            if (t0 + ff < W) out[(size_t)k * W + t0 + ff] = mag[k][ff];
        }
    } else {
        // thread = (frame ff, mel bins m0, m0+16, ...): narrow low bands and wide high bands mix in every thread
        float* out = mel.out + (size_t)b * mel.M * W;
        const int ff = threadIdx.x & 15;
        for (int m = threadIdx.x >> 4; m < mel.M; m += 16) {
            const int s0 = mel.bands[m], len = mel.bands[mel.M + m], off = mel.bands[2 * mel.M + m];
            float acc = 0.0f;
            for (int i = 0; i < len; ++i) acc = fmaf(mag[s0 + i][ff], mel.wvals[off + i], acc);
            if (t0 + ff < W) out[(size_t)m * W + t0 + ff] = acc;
        }
    }

    // per-chunk min / max (magnitudes are >= 0, so the unsigned bit patterns order like the floats)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lmin = fminf(lmin, __shfl_xor(lmin, off));
        lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red_min[wave] = lmin;
        red_max[wave] = lmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mn = fminf(fminf(red_min[0], red_min[1]), fminf(red_min[2], red_min[3]));
        const float mx = fmaxf(fmaxf(red_max[0], red_max[1]), fmaxf(red_max[2], red_max[3]));
        atomicMin(reinterpret_cast<unsigned int*>(minmax + 2 * b), __float_as_uint(mn));
        atomicMax(reinterpret_cast<unsigned int*>(minmax + 2 * b + 1), __float_as_uint(mx));
    }
}

// (S - min) / (max - min + 1e-10): float32 subtraction and division, the 1e-10 added in double
// and rounded back, which is what numpy does with a float32 array and a Python float.
__global__ void spec_normalize_kernel(float* spec, const float* __restrict__ minmax, int per_chunk) {
    const int b = blockIdx.y;
    const float mn = minmax[2 * b], mx = minmax[2 * b + 1];
    const float rng = (float)((double)(mx - mn) + 1e-10);
    float* p = spec + (size_t)b * per_chunk;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_chunk; i += gridDim.x * blockDim.x)
        p[i] = (p[i] - mn) / rng;
}

}  // namespace

void launch_minmax_init(float* minmax, int B, hipStream_t s) {
    hipLaunchKernelGGL(minmax_init_kernel, dim3((B + 255) / 256), dim3(256), 0, s, minmax, B);
}

void launch_stft512(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, float* minmax,
                    hipStream_t s) {
    hipLaunchKernelGGL((stft512_mag_kernel<false>), dim3((W + kFT - 1) / kFT, B), dim3(256), 0, s, tb, audio, T, hop, W, spec,
                       minmax, MelOut{});
}

bool launch_stft512_mel(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* mel_out, int M,
                        const float* wvals, const int* bands, float* minmax, hipStream_t s) {
    hipLaunchKernelGGL((stft512_mag_kernel<true>), dim3((W + kFT - 1) / kFT, B), dim3(256), 0, s, tb, audio, T, hop, W, nullptr,
                       minmax, MelOut{wvals, bands, mel_out, M});
    return true;
}

void launch_spec_normalize(float* spec, const float* minmax, int B, int per_chunk, hipStream_t s) {
    hipLaunchKernelGGL(spec_normalize_kernel, dim3(32, B), dim3(256), 0, s, spec, minmax, per_chunk);
}

}  // namespace bn
