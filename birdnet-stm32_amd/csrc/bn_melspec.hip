// bn_melspec.hip — host-side spectrogram modes of the precomputed frontends on the GPU (gfx950).
//
// reference: birdnet_stm32/audio/spectrogram.py:63-149 get_spectrogram_from_audio(mode = 'mel' | 'log_mel' | 'mfcc',
// mag_scale = 'none' | 'pwl' | 'pcen' | 'db'), whose arithmetic is librosa 0.11's melspectrogram / amplitude_to_db /
// power_to_db / pcen / mfcc followed by the module's min-max normalise.
//
// The STFT kernel (bn_stft.hip, MEL_OUT) leaves un-normalised mel energies [B][M][W] (magnitude or power); this file
// holds the per-chunk finishing pass: one workgroup per chunk keeps the whole [M][W] map (64 x 256 floats = 64 KB) in
// LDS, so every min/max reduction and rescaling of a mode happens without touching HBM again.
#include <hip/hip_runtime.h>

#include "../../include/birdnet_hip.h"
#include "bn_kernels.h"

// numpy rounds every product and sum; a contracted a*b-c can turn an exactly constant map (silence) into noise that
// the min-max normalisation then stretches to [0, 1]
#pragma clang fp contract(off)

namespace bn {
namespace {

struct MinMax {
    float mn, mx;
};

// min and max of tile[0..n) over the workgroup (256 threads); every thread gets the result
__device__ MinMax block_minmax(const float* tile, int n, float* red /* [8] */) {
    float mn = __uint_as_float(0x7f800000u), mx = -__uint_as_float(0x7f800000u);
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = tile[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __syncthreads();  // `red` may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = mn;
        red[4 + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    return {fminf(fminf(red[0], red[1]), fminf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]))};
}

// (S - min) / (max - min + 1e-10), the module's normalize() (reference spectrogram.py:12-21)
__device__ __forceinline__ float norm_range(const MinMax& r) { return (float)((double)(r.mx - r.mn) + 1e-10); }

// 10 log10(max(amin, x)) - 10 log10(max(amin, ref)), floored at (its maximum - 80 dB): librosa.power_to_db(ref, top_db=80).
// The maximum of the first term is reached at x = max(tile), so the floor is known before the pass.
__device__ void power_to_db(float* tile, int n, float amin, bool square, float* red) {
    const MinMax r = block_minmax(tile, n, red);
    const float ref = square ? r.mx * r.mx : r.mx;
    const float ref_db = 10.0f * log10f(fmaxf(amin, ref));
    const float top = 10.0f * log10f(fmaxf(amin, ref)) - ref_db - 80.0f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = square ? tile[i] * tile[i] : tile[i];
        tile[i] = fmaxf(10.0f * log10f(fmaxf(amin, v)) - ref_db, top);
    }
    __syncthreads();
}

struct FinishArgs {
    const float* mel;  // [B][M][W] un-normalised mel energies
    float* out;        // [B][n_out][W]
    const float* dct;  // [n_mfcc][M] (mode mfcc)
    int M, W, Wout, mode, mag, n_mfcc;  // W frames in `mel`, the first Wout of them kept (mfcc sees all frames before the cut)
    double pcen_b;     // smoothing coefficient of librosa.pcen for this (sr, hop)
};

__global__ __launch_bounds__(256) void melspec_finish_kernel(FinishArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[8];
    const int n = a.M * a.W;
    float* tile = lds;
    float* aux = lds + n;  // pcen: smoothed energies; mfcc: cepstral coefficients
    const float* src = a.mel + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        tile[i] = v.x;
        tile[i + 1] = v.y;
        tile[i + 2] = v.z;
        tile[i + 3] = v.w;
    }
    __syncthreads();

    float* res = tile;
    int n_res = n;
    if (a.mode == BN_SPEC_LOGMEL) {  // np.log1p(S)
        for (int i = threadIdx.x; i < n; i += 256) tile[i] = log1pf(tile[i]);
    } else if (a.mode == BN_SPEC_MFCC) {  // power_to_db(power mel, ref=max) -> orthonormal DCT-II over the mel axis, first n_mfcc rows
        power_to_db(tile, n, 1e-10f, false, red);
        n_res = a.n_mfcc * a.Wout;
        for (int i = threadIdx.x; i < n_res; i += 256) {
            const int k = i / a.Wout, t = i - k * a.Wout;
            float acc = 0.0f;
            for (int m = 0; m < a.M; ++m) acc = fmaf(a.dct[k * a.M + m], tile[m * a.W + t], acc);
            aux[i] = acc;
        }
        res = aux;
    } else if (a.mag == BN_MAG_PWL) {  // pre-normalise, three-knee piecewise-linear compression
        const MinMax r = block_minmax(tile, n, red);
        const float rng = norm_range(r);
        for (int i = threadIdx.x; i < n; i += 256) {
            const float x = (tile[i] - r.mn) / rng;
            tile[i] = 0.40f * x + 0.25f * fmaxf(x - 0.10f, 0.0f) + 0.15f * fmaxf(x - 0.35f, 0.0f) + 0.08f * fmaxf(x - 0.65f, 0.0f);
        }
    } else if (a.mag == BN_MAG_DB) {  // amplitude_to_db(S, ref=max) = power_to_db(S^2, ref=max^2, amin=1e-10)
        power_to_db(tile, n, 1e-10f, true, red);
    } else if (a.mag == BN_MAG_PCEN) {
        // librosa.pcen(S * 2^31, sr, hop): first-order smoother along time (scipy.signal.lfilter([b], [1, b-1]) started
        // from lfilter_zi, i.e. state 1 - b), then (bias^power) expm1(power log1p(S smooth / bias)) with
        // smooth = exp(-gain (log eps + log1p(M / eps))); gain .98, bias 2, power .5, eps 1e-6, in double like the host.
        const double b = a.pcen_b;
        if (threadIdx.x < a.M) {
            const float* row = tile + threadIdx.x * a.W;
            float* sm = aux + threadIdx.x * a.W;
            double z = 1.0 - b;
            for (int t = 0; t < a.W; ++t) {
                const double y = b * ((double)row[t] * 2147483648.0) + z;
                z = (1.0 - b) * y;
                sm[t] = (float)y;
            }
        }
        __syncthreads();
        const double log_eps = log(1e-6), sqrt2 = sqrt(2.0);
        for (int i = threadIdx.x; i < n; i += 256) {
            const double s = (double)tile[i] * 2147483648.0;
            const double smooth = exp(-0.98 * (log_eps + log1p((double)aux[i] / 1e-6)));
            tile[i] = (float)(sqrt2 * expm1(0.5 * log1p(s * smooth / 2.0)));
        }
    }
    __syncthreads();
    const MinMax r = block_minmax(res, n_res, red);
    const float rng = norm_range(r);
    float* dst = a.out + (size_t)blockIdx.x * n_res;
    for (int i = threadIdx.x; i < n_res; i += 256) dst[i] = (res[i] - r.mn) / rng;
}

}  // namespace

size_t melspec_finish_lds_bytes(int M, int W, int Wout, int mode, int mag, int n_mfcc) {
    size_t floats = (size_t)M * W;
    if (mode == BN_SPEC_MFCC) floats += (size_t)n_mfcc * Wout;
    else if (mode == BN_SPEC_MEL && mag == BN_MAG_PCEN) floats += (size_t)M * W;
    return floats * sizeof(float);
}

bool launch_melspec_finish(const float* mel, float* out, const float* dct, int B, int M, int W, int Wout, int mode, int mag,
                           int n_mfcc, double pcen_b, hipStream_t s) {
    const size_t smem = melspec_finish_lds_bytes(M, W, Wout, mode, mag, n_mfcc);
    if (smem + 64 > 64 * 1024 &&  // the static reduction scratch counts against the same limit
        !ensure_dynamic_lds((const void*)melspec_finish_kernel, 160 * 1024 - 64))
        return false;
    FinishArgs a{mel, out, dct, M, W, Wout, mode, mag, n_mfcc, pcen_b};
    hipLaunchKernelGGL(melspec_finish_kernel, dim3(B), dim3(256), smem, s, a);
    return true;
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_melspec() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&melspec_finish_kernel));
}

}  // namespace bn
