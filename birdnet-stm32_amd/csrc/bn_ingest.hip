// bn_ingest.hip — audio ingest in front of the hot path, and score pooling behind it (gfx950).
//
//   ingest_resample_kernel : interleaved PCM window(s) -> channel mean -> polyphase FIR resampling -> mono float32,
//                            plus the absolute peak of every window (atomic max on the float's bit pattern)
//   ingest_chunks_kernel   : peak-normalised fixed-length chunks gathered from the resampled windows
//   pool_scores_kernel     : per-file mean / max / log-mean-exp over the file's rows of the score matrix
//
// The arithmetic follows numpy / scipy operation by operation (reference: birdnet_stm32/audio/io.py:14-30,
// :63-130, :133-174 and evaluation/pooling.py:6-47): float32 multiply then float32 add per tap, oldest input
// sample first (scipy upfirdn's loop order), channels summed in numpy's order, true division by the peak.
// All three are streaming kernels bound by HBM; the resampler keeps its input tile and the polyphase filter in LDS.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "bn_kernels.h"

// numpy and scipy round after every multiply and after every add; a fused multiply-add would differ in the last bit
#pragma clang fp contract(off)

namespace bn {
namespace {


// individually rounded float32 operations (defined under the pragma above, so they never fuse)
__device__ __forceinline__ float f_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float f_div(float a, float b) { return a / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FMT>
__device__ __forceinline__ float pcm_sample(const void* p, long i) {
    if (FMT == 0) return (float)((const short*)p)[i] * (1.0f / 32768.0f);
    if (FMT == 1) {
        const unsigned char* b = (const unsigned char*)p + 3 * i;
        const int v = (int)(b[0] | (b[1] << 8) | ((unsigned)b[2] << 16));
        return (float)((v << 8) >> 8) * (1.0f / 8388608.0f);
    }
    if (FMT == 2) return (float)((const int*)p)[i] * (1.0f / 2147483648.0f);
    return ((const float*)p)[i];
}

// numpy's float32 mean over the channel axis: left to right below 8 channels, the unrolled pairwise tree for 8.
template <int FMT>
__device__ __forceinline__ float mono_frame(const void* pcm, long frame, int ch) {
    if (ch == 1) return pcm_sample<FMT>(pcm, frame);
    if (ch == 2) {
        float a, b;
        if (FMT == 0) {
            const unsigned v = ((const unsigned*)pcm)[frame];
            a = (float)(short)(v & 0xffff) * (1.0f / 32768.0f);
            b = (float)(short)(v >> 16) * (1.0f / 32768.0f);
        } else {
            a = pcm_sample<FMT>(pcm, 2 * frame);
            b = pcm_sample<FMT>(pcm, 2 * frame + 1);
        }
        return f_div(f_add(a, b), 2.0f);
    }
    const long base = frame * ch;
    float acc;
    if (ch < 8) {
        acc = pcm_sample<FMT>(pcm, base);
        for (int c = 1; c < ch; ++c) acc = f_add(acc, pcm_sample<FMT>(pcm, base + c));
    } else {  // ch == 8
        float r[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] = pcm_sample<FMT>(pcm, base + c);
        acc = f_add(f_add(f_add(r[0], r[1]), f_add(r[2], r[3])),
                        f_add(f_add(r[4], r[5]), f_add(r[6], r[7])));
    }
    return f_div(acc, (float)ch);
}

struct __attribute__((packed, aligned(4))) U32x4 {
    unsigned v[4];
};
struct __attribute__((packed, aligned(2))) U32x2 {
    unsigned v[2];
};

// Four consecutive mono samples starting at absolute frame `frame`, from one wide load (the hardware takes dword-aligned
// 128-bit global loads).  Only for the layouts quad_layout() names; the caller guarantees all four frames exist.
template <int FMT, int CH>
__device__ __forceinline__ void mono_quad_wide(const void* __restrict__ pcm, long frame, float out[4]) {
    if (FMT == 0 && CH == 2) {
        const U32x4 q = *reinterpret_cast<const U32x4*>((const unsigned*)pcm + frame);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float l = (float)(short)(q.v[e] & 0xffff) * (1.0f / 32768.0f);
            const float r = (float)(short)(q.v[e] >> 16) * (1.0f / 32768.0f);
            out[e] = f_mul(f_add(l, r), 0.5f);  // x / 2 and x * 0.5 round identically
        }
    } else if (FMT == 0 && CH == 1) {
        const U32x2 q = *reinterpret_cast<const U32x2*>((const short*)pcm + frame);
        out[0] = (float)(short)(q.v[0] & 0xffff) * (1.0f / 32768.0f);
        out[1] = (float)(short)(q.v[0] >> 16) * (1.0f / 32768.0f);
        out[2] = (float)(short)(q.v[1] & 0xffff) * (1.0f / 32768.0f);
        out[3] = (float)(short)(q.v[1] >> 16) * (1.0f / 32768.0f);
    } else if (FMT == 3 && CH == 1) {
        const U32x4 q = *reinterpret_cast<const U32x4*>((const float*)pcm + frame);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = __uint_as_float(q.v[e]);
    } else {  // FMT == 3 && CH == 2
        const U32x4 q0 = *reinterpret_cast<const U32x4*>((const float*)pcm + 2 * frame);
        const U32x4 q1 = *reinterpret_cast<const U32x4*>((const float*)pcm + 2 * frame + 4);
        out[0] = f_mul(f_add(__uint_as_float(q0.v[0]), __uint_as_float(q0.v[1])), 0.5f);
        out[1] = f_mul(f_add(__uint_as_float(q0.v[2]), __uint_as_float(q0.v[3])), 0.5f);
        out[2] = f_mul(f_add(__uint_as_float(q1.v[0]), __uint_as_float(q1.v[1])), 0.5f);
        out[3] = f_mul(f_add(__uint_as_float(q1.v[2]), __uint_as_float(q1.v[3])), 0.5f);
    }
}

template <int FMT>
__device__ __forceinline__ bool quad_layout(int ch) {
    return (FMT == 0 || FMT == 3) && ch <= 2;
}

// Four consecutive mono samples of a window starting at frame k (window-relative; may hang over either end, where the
// window reads as zeros).
template <int FMT>
__device__ __forceinline__ void mono_quad(const void* __restrict__ pcm, long in0, long k, long n_in, int ch, float out[4]) {
    if (quad_layout<FMT>(ch) && k >= 0 && k + 3 < n_in) {
        if (ch == 2) mono_quad_wide<FMT, 2>(pcm, in0 + k, out);
        else mono_quad_wide<FMT, 1>(pcm, in0 + k, out);
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = (k + e >= 0 && k + e < n_in) ? mono_frame<FMT>(pcm, in0 + k + e, ch) : 0.0f;
}

// Input tile [k_lo, k_lo + tile) of a window -> LDS as mono float32.  A tile that lies inside the window in a wide-load
// layout is filled by a branch-free loop whose loads the compiler batches (many bytes in flight per thread); a tile
// touching a window end, or any other layout, goes quad by quad with bounds checks.
template <int FMT, int CH>
__device__ __forceinline__ void fill_tile_wide(const void* __restrict__ pcm, long frame0, int tile, float* xs, int tid) {
#pragma unroll 4
    for (int i0 = 4 * tid; i0 < tile; i0 += 1024) {
        float v[4];
        mono_quad_wide<FMT, CH>(pcm, frame0 + i0, v);
        *reinterpret_cast<f32x4*>(xs + i0) = (f32x4){v[0], v[1], v[2], v[3]};
    }
}

template <int FMT>
__device__ __forceinline__ void fill_tile(const void* __restrict__ pcm, long in0, long k_lo, int tile, long n_in, int ch, float* xs,
                                          int tid) {
    const bool interior = k_lo >= 0 && k_lo + ((tile + 3) & ~3) <= n_in;
    if (interior && quad_layout<FMT>(ch)) {
        if (ch == 2) fill_tile_wide<(FMT == 0 || FMT == 3) ? FMT : 0, 2>(pcm, in0 + k_lo, tile, xs, tid);
        else fill_tile_wide<(FMT == 0 || FMT == 3) ? FMT : 0, 1>(pcm, in0 + k_lo, tile, xs, tid);
        return;
    }
    for (int i0 = 4 * tid; i0 < tile; i0 += 1024) {
        float v[4];
        mono_quad<FMT>(pcm, in0, k_lo + i0, n_in, ch, v);
        *reinterpret_cast<f32x4*>(xs + i0) = (f32x4){v[0], v[1], v[2], v[3]};
    }
}

// Workgroup maximum -> partial[window][block].  (Atomic maxima on the windows' peaks were the bottleneck: a few
// cache lines took every workgroup's read-modify-write, ~20 ns each in L2.)  ingest_peak_kernel folds the partials.
__device__ __forceinline__ void store_block_peak(float* partial, float vmax, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    __shared__ float wmax[4];
    if ((tid & 63) == 0) wmax[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 0) *partial = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

struct ResampleArgs {
    const void* __restrict__ pcm;
    const long* __restrict__ in_off;
    const long* __restrict__ out_off;
    const float* __restrict__ taps;  // [up][hpp], phase-major, oldest input sample first
    float* __restrict__ mono;
    float* __restrict__ partial;  // [n_windows][gridDim.x] block maxima
    int ch, up, down, hpp, n_pre_remove;
    int blk;  // outputs per workgroup (a multiple of 1024)
};

template <int FMT>
__global__ __launch_bounds__(256) void ingest_resample_kernel(ResampleArgs a) {
    extern __shared__ float lds[];
    const int f = blockIdx.y, tid = threadIdx.x;
    const long in0 = a.in_off[f], n_in = a.in_off[f + 1] - in0;
    const long out0 = a.out_off[f], n_out = a.out_off[f + 1] - out0;
    const long n0 = (long)blockIdx.x * a.blk;
    if (n0 >= n_out) return;
    const int cnt = n_out - n0 < a.blk ? (int)(n_out - n0) : a.blk;
    float vmax = 0.0f;
    if (a.hpp == 0) {  // same rate: decode + channel mean only; four loads in flight per thread
        const bool vec_store = ((out0 + n0) & 3) == 0;
        for (int j0 = 4 * tid; j0 < cnt; j0 += 1024) {
            float v[4];
            mono_quad<FMT>(a.pcm, in0, n0 + j0, n_in, a.ch, v);
            float* dst = a.mono + out0 + n0 + j0;
            if (vec_store && j0 + 3 < cnt) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j0 + e < cnt) dst[e] = v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j0 + e < cnt) vmax = fmaxf(vmax, fabsf(v[e]));
        }
    } else {
        const int up = a.up, down = a.down, hpp = a.hpp;
        float* ht = lds;
        float* xs = lds + ((up * hpp + 3) & ~3);  // 16-byte aligned for the 128-bit tile stores
        for (int i = tid; i < up * hpp; i += 256) ht[i] = a.taps[i];
        // input samples the outputs [n0, n0 + cnt) touch: newest = floor((n + pre) down / up), hpp samples back from there
        const unsigned t_first = (unsigned)(n0 + a.n_pre_remove) * (unsigned)down;
        const unsigned t_last = (unsigned)(n0 + cnt - 1 + a.n_pre_remove) * (unsigned)down;
        const long k_lo = (long)(t_first / (unsigned)up) - (hpp - 1);
        const int tile = (int)((long)(t_last / (unsigned)up) - k_lo) + 1;
        fill_tile<FMT>(a.pcm, in0, k_lo, tile, n_in, a.ch, xs, tid);
        __syncthreads();
        for (int j = tid; j < cnt; j += 256) {
            const unsigned t = t_first + (unsigned)j * (unsigned)down;
            const unsigned kmax = t / (unsigned)up;
            const unsigned phase = t - kmax * (unsigned)up;
            const float* hp = ht + phase * hpp;
            const float* xp = xs + ((long)kmax - (hpp - 1) - k_lo);
            float acc = 0.0f;
            for (int q = 0; q < hpp; ++q) acc = f_add(acc, f_mul(xp[q], hp[q]));
            a.mono[out0 + n0 + j] = acc;
            vmax = fmaxf(vmax, fabsf(acc));
        }
    }
    store_block_peak(a.partial + (size_t)f * gridDim.x + blockIdx.x, vmax, tid);
}

// Integer decimation (up == 1: 48 kHz or 96 kHz -> 24 kHz).  Every output uses the same coefficients, so they are read
// through the scalar cache; a thread produces four consecutive outputs from one register window of the input tile
// (HPP + 3 DOWN samples fetched with 128-bit LDS reads instead of 4 HPP scalar ones).  Same operations, same order.
template <int FMT, int DOWN>
__global__ __launch_bounds__(256) void ingest_decimate_kernel(ResampleArgs a, const float* __restrict__ taps, float* __restrict__ mono) {
    constexpr int HPP = 21 * DOWN + 1;  // 2 * 10 DOWN + 1 coefficients behind DOWN centring zeros
    constexpr int NX = (HPP + 3 * DOWN + 3) & ~3;
    extern __shared__ float lds[];
    const int f = blockIdx.y, tid = threadIdx.x;
    const long in0 = a.in_off[f], n_in = a.in_off[f + 1] - in0;
    const long out0 = a.out_off[f], n_out = a.out_off[f + 1] - out0;
    const long n0 = (long)blockIdx.x * a.blk;
    if (n0 >= n_out) return;
    const int cnt = n_out - n0 < a.blk ? (int)(n_out - n0) : a.blk;
    const long k_lo = (n0 + a.n_pre_remove) * DOWN - (HPP - 1);
    const int tile = (cnt - 1) * DOWN + HPP;
    fill_tile<FMT>(a.pcm, in0, k_lo, tile, n_in, a.ch, lds, tid);
    __syncthreads();
    float vmax = 0.0f;
    const bool vec_store = ((out0 + n0) & 3) == 0;
    for (int j0 = 4 * tid; j0 < cnt; j0 += 1024) {
        float x[NX];
        const f32x4* src = reinterpret_cast<const f32x4*>(lds + j0 * DOWN);  // 16-byte aligned: j0 is a multiple of 4
#pragma unroll
        for (int i = 0; i < NX / 4; ++i) {
            const f32x4 v = src[i];
            x[4 * i] = v[0];
            x[4 * i + 1] = v[1];
            x[4 * i + 2] = v[2];
            x[4 * i + 3] = v[3];
        }
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < HPP + 3 * DOWN; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = i - DOWN * r;
                if (q >= 0 && q < HPP) acc[r] = f_add(acc[r], f_mul(x[i], taps[q]));
            }
        float* dst = mono + out0 + n0 + j0;
        if (vec_store && j0 + 3 < cnt) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j0 + r < cnt) dst[r] = acc[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (j0 + r < cnt) vmax = fmaxf(vmax, fabsf(acc[r]));
    }
    store_block_peak(a.partial + (size_t)f * gridDim.x + blockIdx.x, vmax, tid);
}

// Rational ratios with up <= 256 and one of the filter lengths scipy designs for the common rate pairs (HPP taps per phase).
// Outputs j and j + up share their filter phase, so a thread that walks outputs at stride `up` (its input pointer advancing
// by `down`) keeps its HPP coefficients in registers: the tap loop reads only the input tile from LDS — half the LDS traffic
// of the generic kernel, which is what bounds it.  Lanes of a wave still cover consecutive outputs (coalesced stores).
template <int FMT, int HPP>
__global__ __launch_bounds__(256) void ingest_resample_phase_kernel(ResampleArgs a, const float* __restrict__ taps, float* __restrict__ mono) {
    extern __shared__ float lds[];
    const int f = blockIdx.y, tid = threadIdx.x;
    const long in0 = a.in_off[f], n_in = a.in_off[f + 1] - in0;
    const long out0 = a.out_off[f], n_out = a.out_off[f + 1] - out0;
    const long n0 = (long)blockIdx.x * a.blk;
    if (n0 >= n_out) return;
    const int cnt = n_out - n0 < a.blk ? (int)(n_out - n0) : a.blk;
    const int up = a.up, down = a.down;
    const unsigned t_first = (unsigned)(n0 + a.n_pre_remove) * (unsigned)down;
    const unsigned t_last = (unsigned)(n0 + cnt - 1 + a.n_pre_remove) * (unsigned)down;
    const long k_lo = (long)(t_first / (unsigned)up) - (HPP - 1);
    const int tile = (int)((long)(t_last / (unsigned)up) - k_lo) + 1;
    fill_tile<FMT>(a.pcm, in0, k_lo, tile, n_in, a.ch, lds, tid);
    const int streams = 256 / up;           // output streams per phase slot that fit the workgroup
    const int slot = tid % up, stream = tid / up;
    const bool active = stream < streams;
    const int j0 = slot + stream * up;
    const unsigned t0 = t_first + (unsigned)j0 * (unsigned)down;
    const unsigned kmax0 = t0 / (unsigned)up;
    const unsigned phase = t0 - kmax0 * (unsigned)up;
    float h[HPP];
#pragma unroll
    for (int q = 0; q < HPP; ++q) h[q] = taps[phase * HPP + q];
    __syncthreads();
    float vmax = 0.0f;
    if (active) {
        const float* xp = lds + ((long)kmax0 - (HPP - 1) - k_lo);
        const int xstep = streams * down, jstep = streams * up;
        for (int j = j0; j < cnt; j += jstep, xp += xstep) {
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < HPP; ++q) acc = f_add(acc, f_mul(xp[q], h[q]));
            mono[out0 + n0 + j] = acc;
            vmax = fmaxf(vmax, fabsf(acc));
        }
    }
    store_block_peak(a.partial + (size_t)f * gridDim.x + blockIdx.x, vmax, tid);
}

__global__ __launch_bounds__(256) void ingest_peak_kernel(const float* __restrict__ partial, const long* __restrict__ out_off,
                                                          int stride, int blk, float* __restrict__ peak) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const long n_out = out_off[f + 1] - out_off[f];
    const int nblk = (int)((n_out + blk - 1) / blk);  // blocks past the window's end returned without writing
    float m = 0.0f;
    for (int i = tid; i < nblk; i += 256) m = fmaxf(m, partial[(size_t)f * stride + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wmax[4];
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) peak[f] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

__global__ __launch_bounds__(256) void ingest_chunks_kernel(const float* __restrict__ mono, const float* __restrict__ peak,
                                                            const long* __restrict__ src, const int* __restrict__ valid,
                                                            const int* __restrict__ file, int T, float* __restrict__ out) {
    const int c = blockIdx.y;
    const long s = src[c];
    const int v = valid[c];
    const float p = peak[file[c]];
    float* dst = out + (size_t)c * T;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t0 >= T) return;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = t0 + e;
        x[e] = t < v ? mono[s + t] : 0.0f;
        if (p > 0.0f) x[e] = f_div(x[e], p);
    }
    if ((T & 3) == 0) {
        *reinterpret_cast<float4*>(dst + t0) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t0 + e < T) dst[t0 + e] = x[e];
    }
}

// Per-chunk peak normalisation of the raw frontend's input: x / (max|x| + eps) in float32 (reference:
// evaluation/metrics.py:62-69, conversion/quantize.py:96-98).  One workgroup per chunk, two passes over 288 KB (the second
// one hits L2).
__global__ __launch_bounds__(256) void chunk_peaknorm_kernel(const float* x, float* y, int T, float eps) {  // x and y may be the same buffer
    const float* src = x + (size_t)blockIdx.x * T;
    float* dst = y + (size_t)blockIdx.x * T;
    // 16-byte accesses, four in flight per thread, when the chunks are 16-byte aligned (T % 4 == 0: every real chunk length); the scalar
    // loop paid one memory round trip per element and thread (186 us per 1024 chunks of 48000 samples, waves 86 % parked)
    const bool vec = (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    const int T4 = vec ? T >> 2 : 0;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float m = 0.0f;
    for (int i0 = threadIdx.x; i0 < T4; i0 += 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + 256 * u < T4 ? s4[i0 + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    for (int i = 4 * T4 + threadIdx.x; i < T; i += 256) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    const float denom = f_add(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])), eps);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i0 = threadIdx.x; i0 < T4; i0 += 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < T4) v[u] = s4[i0 + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < T4) d4[i0 + 256 * u] = make_float4(f_div(v[u].x, denom), f_div(v[u].y, denom), f_div(v[u].z, denom), f_div(v[u].w, denom));
    }
    for (int i = 4 * T4 + threadIdx.x; i < T; i += 256) dst[i] = f_div(src[i], denom);
}

// One thread per (file, class): rows of one file are read in order, so the float32 sums match numpy's axis-0 reduction.
__global__ __launch_bounds__(256) void pool_scores_kernel(const float* __restrict__ scores, const long* __restrict__ seg,
                                                          int F, int C, int method, float beta, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= F * C) return;
    const int f = idx / C, c = idx - f * C;
    const long r0 = seg[f], r1 = seg[f + 1];
    const float* col = scores + c;
    float res = 0.0f;
    if (r1 > r0) {
        if (method == 0) {
            float acc = col[r0 * C];
            for (long r = r0 + 1; r < r1; ++r) acc = f_add(acc, col[r * C]);
            res = f_div(acc, (float)(r1 - r0));
        } else if (method == 1) {
            float m = col[r0 * C];
            for (long r = r0 + 1; r < r1; ++r) m = fmaxf(m, col[r * C]);
            res = m;
        } else {
            float m = f_mul(beta, col[r0 * C]);
            for (long r = r0 + 1; r < r1; ++r) m = fmaxf(m, f_mul(beta, col[r * C]));
            float acc = 0.0f;
            for (long r = r0; r < r1; ++r) {
                const float e = expf(f_mul(beta, col[r * C]) - m);
                acc = r == r0 ? e : f_add(acc, e);
            }
            const float mean = f_div(acc, (float)(r1 - r0));
            // the reference adds the double 1e-12 to a float32 array: numpy keeps float32, where 1e-12 still counts
            res = f_div(f_add(m, logf(f_add(mean, 1e-12f))), beta);
        }
    }
    out[idx] = res;
}

}  // namespace

size_t ingest_resample_lds_bytes(int up, int down, int hpp, int blk) {
    if (hpp == 0) return 0;
    return ((size_t)up * hpp + ((size_t)blk * down) / up + hpp + 12) * sizeof(float);
}

// up == 1 ratios with a register-window kernel (the filter scipy designs for them has 21 down + 1 entries)
static bool decimate_case(int up, int down, int hpp) { return up == 1 && (down == 2 || down == 4) && hpp == 21 * down + 1; }

// outputs per workgroup: as many as keep the input tile + filter within ~40 KB of LDS (4 workgroups per CU)
int ingest_resample_block(int up, int down, int hpp) {
    if (hpp == 0) return 1024;  // same rate: one quad per thread, parallelism comes from the grid
    if (g_opt.ingest_blk > 0) return g_opt.ingest_blk;
    for (int blk = 4096; blk > 1024; blk >>= 1)
        if (ingest_resample_lds_bytes(up, down, hpp, blk) <= 40 * 1024) return blk;
    return 1024;
}

static void launch_resample_kernels(const ResampleArgs& a, int fmt, int n_files, long max_out, const float* taps, float* mono,
                                    hipStream_t s) {
    const int blk = a.blk, up = a.up, down = a.down, hpp = a.hpp;
    const size_t smem = ingest_resample_lds_bytes(up, down, hpp, blk);
    const dim3 grid((unsigned)((max_out + blk - 1) / blk), (unsigned)n_files);
    if (decimate_case(up, down, hpp)) {
#define BN_DECIMATE(F)                                                                                   \
    if (down == 2) hipLaunchKernelGGL((ingest_decimate_kernel<F, 2>), grid, dim3(256), smem, s, a, taps, mono);    \
    else hipLaunchKernelGGL((ingest_decimate_kernel<F, 4>), grid, dim3(256), smem, s, a, taps, mono)
        switch (fmt) {
            case 0: BN_DECIMATE(0); break;
            case 1: BN_DECIMATE(1); break;
            case 2: BN_DECIMATE(2); break;
            default: BN_DECIMATE(3); break;
        }
#undef BN_DECIMATE
        return;
    }
    if (up > 1 && up <= 256 && (hpp == 21 || hpp == 29 || hpp == 39) && !g_opt.ingest_generic) {
#define BN_PHASE(F)                                                                                                   \
    if (hpp == 21) hipLaunchKernelGGL((ingest_resample_phase_kernel<F, 21>), grid, dim3(256), smem, s, a, taps, mono);     \
    else if (hpp == 29) hipLaunchKernelGGL((ingest_resample_phase_kernel<F, 29>), grid, dim3(256), smem, s, a, taps, mono); \
    else hipLaunchKernelGGL((ingest_resample_phase_kernel<F, 39>), grid, dim3(256), smem, s, a, taps, mono)
        switch (fmt) {
            case 0: BN_PHASE(0); break;
            case 1: BN_PHASE(1); break;
            case 2: BN_PHASE(2); break;
            default: BN_PHASE(3); break;
        }
#undef BN_PHASE
        return;
    }
    if (smem > 64 * 1024) {  // unusual ratios (96 kHz -> 22.05 kHz = 147/640, 11.025 -> 32 kHz = 1280/441): the whole polyphase filter needs up to ~110 KB
        const int f = fmt < 0 || fmt > 3 ? 3 : fmt;
        const void* kernels[4] = {reinterpret_cast<const void*>(ingest_resample_kernel<0>), reinterpret_cast<const void*>(ingest_resample_kernel<1>),
                                  reinterpret_cast<const void*>(ingest_resample_kernel<2>), reinterpret_cast<const void*>(ingest_resample_kernel<3>)};
        // exactly what this launch needs (the kernel's static LDS comes on top; asking for the CU's whole 160 KB is refused); a refusal
        // shows as the error of the launch below
        (void)ensure_dynamic_lds(kernels[f], smem);
    }
    switch (fmt) {
        case 0: hipLaunchKernelGGL(ingest_resample_kernel<0>, grid, dim3(256), smem, s, a); break;
        case 1: hipLaunchKernelGGL(ingest_resample_kernel<1>, grid, dim3(256), smem, s, a); break;
        case 2: hipLaunchKernelGGL(ingest_resample_kernel<2>, grid, dim3(256), smem, s, a); break;
        default: hipLaunchKernelGGL(ingest_resample_kernel<3>, grid, dim3(256), smem, s, a); break;
    }
}

size_t ingest_partial_elems(int n_files, long max_out, int up, int down, int hpp) {
    const int blk = ingest_resample_block(up, down, hpp);
    return (size_t)n_files * (size_t)((max_out + blk - 1) / blk);
}

void launch_ingest_resample(const void* pcm, int fmt, int ch, const long* in_off, const long* out_off, int n_files,
                            long max_out, const float* taps, int up, int down, int hpp, int n_pre_remove, float* mono,
                            float* partial, float* peak, hipStream_t s) {
    const int blk = ingest_resample_block(up, down, hpp);
    ResampleArgs a{pcm, in_off, out_off, taps, mono, partial, ch, up, down, hpp, n_pre_remove, blk};
    launch_resample_kernels(a, fmt, n_files, max_out, taps, mono, s);
    hipLaunchKernelGGL(ingest_peak_kernel, dim3(n_files), dim3(256), 0, s, partial, out_off, (int)((max_out + blk - 1) / blk), blk, peak);
}

void launch_ingest_chunks(const float* mono, const float* peak, const long* src, const int* valid, const int* file,
                          int n_chunks, int T, float* out, hipStream_t s) {
    const dim3 grid((unsigned)((T + 1023) / 1024), (unsigned)n_chunks);
    hipLaunchKernelGGL(ingest_chunks_kernel, grid, dim3(256), 0, s, mono, peak, src, valid, file, T, out);
}

void launch_chunk_peaknorm(const float* x, float* y, int B, int T, float eps, hipStream_t s) {
    hipLaunchKernelGGL(chunk_peaknorm_kernel, dim3(B), dim3(256), 0, s, x, y, T, eps);
}

void launch_pool_scores(const float* scores, const long* seg, int F, int C, int method, float beta, float* out,
                        hipStream_t s) {
    const int total = F * C;
    hipLaunchKernelGGL(pool_scores_kernel, dim3((total + 255) / 256), dim3(256), 0, s, scores, seg, F, C, method, beta, out);
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_ingest() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&ingest_chunks_kernel));
}

}  // namespace bn
