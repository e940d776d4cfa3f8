// bn_f32.hip — float32 DS-CNN kernels for gfx950 (baseline generation: one thread per small
// output vector, float4 channel vectors, weights served from L1/L2).
//
// Graph semantics follow the reference's Keras layers with BatchNorm folded into the
// preceding convolution by the packer:
//   hybrid frontend  birdnet_stm32/models/frontend.py:299-345, magnitude.py:166-192
//   stem / ds block  birdnet_stm32/models/dscnn.py:28-84,198-202
//   inverted residual + squeeze-excite  birdnet_stm32/models/blocks.py:27-133
//   head             birdnet_stm32/models/dscnn.py:248-261
// Layout: NHWC per chunk, C innermost.  TensorFlow SAME padding (pad_top/pad_left come from the
// packer; the odd cell is after).
#include "bn_kernels.h"

namespace bn {
namespace {

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}

// channel-wise magnitude scaling (reference: magnitude.py:166-192); magp is [NP][M]
__device__ __forceinline__ float mag_scale(float y, int m, int M, const float* __restrict__ magp, int mag) {
    if (mag == 1) {  // pwl: k0 y + sum_i k_i relu(w_i y + b_i); rows: k0, k1..3, w1..3, b1..3
        float out = y * magp[m];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            out += magp[(1 + i) * M + m] * fmaxf(magp[(4 + i) * M + m] * y + magp[(7 + i) * M + m], 0.0f);
        return out;
    }
    if (mag == 2) {  // pcen-like: rows agc, k1, sw, sb, k2
        const float y0 = fmaxf(y - magp[m] * y, 0.0f);
        const float b1 = magp[M + m] * y0;
        const float b2 = magp[4 * M + m] * fmaxf(magp[2 * M + m] * y0 + magp[3 * M + m], 0.0f);
        return fmaxf(b1 + b2, 0.0f);
    }
    if (mag == 3) return 10.0f * logf(fmaxf(y, 1e-6f)) / logf(10.0f);
    return y;
}

__global__ void u32_fill_kernel(uint32_t* p, uint32_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// spec [F][W] -> relu(mel) [M][W].  Each mel row is a band [start, start+len) of frequency bins
// (the Slaney triangles touch a few bins each; a dense mixer is just len = F).
// One thread per frame t, blockIdx.y = mel bin, blockIdx.z = chunk.
__global__ void f32_mel_kernel(const float* __restrict__ spec, const float* __restrict__ minmax, float* __restrict__ out,
                               float* smax, int F, int W, int M, const float* __restrict__ wvals,
                               const int* __restrict__ bands, const float* __restrict__ magp, int mag, int norm) {
    const int b = blockIdx.z, m = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int start = bands[m], len = bands[M + m], off = bands[2 * M + m];
    const float* S = spec + (size_t)b * F * W;
    float mn = 0.0f, rng = 1.0f;
    const bool renorm = minmax != nullptr;
    if (renorm) {
        mn = minmax[2 * b];
        rng = (float)((double)(minmax[2 * b + 1] - mn) + 1e-10);
    }
    float acc = 0.0f;
    if (t < W) {
        for (int i = 0; i < len; ++i) {
            float v = S[(size_t)(start + i) * W + t];
            if (renorm) v = (v - mn) / rng;
            acc = fmaf(v, wvals[off + i], acc);
        }
    }
    acc = fmaxf(acc, 0.0f);
    if (!norm) {
        if (t < W) out[((size_t)b * M + m) * W + t] = mag_scale(acc, m, M, magp, mag);
        return;
    }
    if (t < W) out[((size_t)b * M + m) * W + t] = acc;
    float v = (t < W) ? acc : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(smax + b), __float_as_uint(v));
}

// per-sample max normalisation y / (max + 1e-6) followed by magnitude scaling, in place on [M][W]
__global__ void f32_mag_kernel(float* x, const float* __restrict__ smax, int M, int W, const float* __restrict__ magp,
                               int mag) {
    const int b = blockIdx.y;
    const float inv_d = smax[b] + 1e-6f;
    float* p = x + (size_t)b * M * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M * W; i += gridDim.x * blockDim.x) {
        const int m = i / W;
        p[i] = mag_scale(p[i] / inv_d, m, M, magp, mag);
    }
}

// frontend finalisation for the fused STFT+mel path: un-normalised mel energies [M][W] -> frontend output.
//   y = relu((mel - mn * wsum[m]) / rng)   (= the mixer applied to the min-max normalised spectrogram)
//   [y /= max(y) + 1e-6]  then magnitude scaling.  One block per chunk, two passes over the 64 KB tile when norm is on.
__global__ __launch_bounds__(256) void f32_melfin_kernel(const float* __restrict__ melraw, const float* __restrict__ minmax,
                                                         float* __restrict__ out, int M, int W, const float* __restrict__ wsum,
                                                         const float* __restrict__ magp, int mag, int norm) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float mn = minmax[2 * b];
    const float rng = (float)((double)(minmax[2 * b + 1] - mn) + 1e-10);
    const float* src = melraw + (size_t)b * M * W;
    float* dst = out + (size_t)b * M * W;
    float peak = 0.0f;
    if (norm) {
        for (int i = threadIdx.x; i < M * W; i += 256) peak = fmaxf(peak, fmaxf((src[i] - mn * wsum[i / W]) / rng, 0.0f));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) peak = fmaxf(peak, __shfl_xor(peak, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = peak;
        __syncthreads();
        peak = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) + 1e-6f;
    }
    // (16-byte accesses, four in flight per thread, when rows are whole float4s — same arithmetic per element; element by element every
    // iteration paid a memory round trip: 64 of them per thread and chunk)
    const int n = M * W;
    const int n4 = (W & 3) == 0 ? n >> 2 : 0;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < n4) v[u] = s4[i0 + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i >= n4) break;
            const int m = (4 * i) / W;
            const float off = mn * wsum[m];
            float y[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = fmaxf((y[e] - off) / rng, 0.0f);
                if (norm) t = t / peak;
                y[e] = mag_scale(t, m, M, magp, mag);
            }
            d4[i] = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) {
        const int m = i / W;
        float y = fmaxf((src[i] - mn * wsum[m]) / rng, 0.0f);
        if (norm) y = y / peak;
        dst[i] = mag_scale(y, m, M, magp, mag);
    }
}

// raw-waveform frontend: symmetric zero pad, VALID strided 1x16 convolution (BatchNorm folded), ReLU6, magnitude scaling,
// output transposed to [M][W] (reference: birdnet_stm32/models/frontend.py:138-164,347-358).  A thread owns ONE frame: it loads its 16
// samples once (64 contiguous bytes; the kernel touches 16 of every `stride` samples) and walks over all M filters, whose taps are
// wave-uniform LDS broadcasts ([M][16], transposed while staging); every store of a wave is one run of 64 consecutive frames of a
// filter row.  (First version: thread = (frame, 4 filters) — every sample was loaded by M / 4 threads in sixteen 4-byte pieces at a
// stride of `stride` samples between lanes: 0.24 ms per 1024 chunks at 0.8 TB/s; this form: see DESIGN.md.)  Same summation order.
__global__ __launch_bounds__(256) void f32_rawfe_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int W, int M, int stride,
                                                        int pad_left, const float* __restrict__ fb, const float* __restrict__ bias,
                                                        const float* __restrict__ magp, int mag) {
    extern __shared__ __attribute__((aligned(16))) float fbt[];  // [M][16] taps, then [M] bias
    for (int i = threadIdx.x; i < 16 * M; i += 256) fbt[(i % M) * 16 + i / M] = fb[i];  // fb is [16][M]
    for (int i = threadIdx.x; i < M; i += 256) fbt[16 * M + i] = bias[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= W) return;
    const float* xin = x + (size_t)b * T;
    const int s0 = t * stride - pad_left;
    // range-checked raw buffer loads over this chunk: samples of the symmetric zero pad (before / behind the waveform) read as 0, no branch
    // around any of the sixteen loads
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, T * 4, 0x00020000);
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (s0 + k) * 4, 0, 0));
    float* o = out + (size_t)b * M * W + t;
    for (int m = 0; m < M; ++m) {
        const float4* wr = reinterpret_cast<const float4*>(fbt + 16 * m);
        float acc = fbt[16 * M + m];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 w4 = wr[k4];
            acc = fmaf(v[4 * k4 + 0], w4.x, acc);
            acc = fmaf(v[4 * k4 + 1], w4.y, acc);
            acc = fmaf(v[4 * k4 + 2], w4.z, acc);
            acc = fmaf(v[4 * k4 + 3], w4.w, acc);
        }
        o[(size_t)m * W] = mag_scale(fminf(fmaxf(acc, 0.0f), 6.0f), m, M, magp, mag);
    }
}

// stem: [H][W] (one channel) -> [OH][OW][Cout], 3x3.  One thread = 4 output channels of one pixel.
__global__ void f32_stem_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Cout, int sh,
                                int sw, int act, int OH, int OW, int pt, int pl, const float* __restrict__ w,
                                const float* __restrict__ bias, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c4 = Cout >> 2;
    const int cg = (int)(gid % c4);
    long r = gid / c4;
    const int ow = (int)(r % OW);
    r /= OW;
    const int oh = (int)(r % OH);
    const long b = r / OH;
    const float* xin = x + b * H * W;
    float4 acc = *reinterpret_cast<const float4*>(bias + 4 * cg);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ih = oh * sh + i - pt;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int iw = ow * sw + j - pl;
            if (iw < 0 || iw >= W) continue;
            const float v = xin[ih * W + iw];
            const float4 k = *reinterpret_cast<const float4*>(w + (i * 3 + j) * Cout + 4 * cg);
            acc.x = fmaf(v, k.x, acc.x);
            acc.y = fmaf(v, k.y, acc.y);
            acc.z = fmaf(v, k.z, acc.z);
            acc.w = fmaf(v, k.w, acc.w);
        }
    }
    acc.x = apply_act(acc.x, act);
    acc.y = apply_act(acc.y, act);
    acc.z = apply_act(acc.z, act);
    acc.w = apply_act(acc.w, act);
    *reinterpret_cast<float4*>(y + ((b * OH + oh) * OW + ow) * Cout + 4 * cg) = acc;
}

// depthwise 3x3: [H][W][C] -> [OH][OW][C].  One thread = 4 channels of one output pixel.
__global__ void f32_dw_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int sh, int sw,
                              int act, int OH, int OW, int pt, int pl, const float* __restrict__ w,
                              const float* __restrict__ bias, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c4 = C >> 2;
    const int cg = (int)(gid % c4);
    long r = gid / c4;
    const int ow = (int)(r % OW);
    r /= OW;
    const int oh = (int)(r % OH);
    const long b = r / OH;
    const float* xin = x + b * H * W * C;
    float4 acc = *reinterpret_cast<const float4*>(bias + 4 * cg);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ih = oh * sh + i - pt;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int iw = ow * sw + j - pl;
            if (iw < 0 || iw >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(xin + ((long)ih * W + iw) * C + 4 * cg);
            const float4 k = *reinterpret_cast<const float4*>(w + (i * 3 + j) * C + 4 * cg);
            acc.x = fmaf(v.x, k.x, acc.x);
            acc.y = fmaf(v.y, k.y, acc.y);
            acc.z = fmaf(v.z, k.z, acc.z);
            acc.w = fmaf(v.w, k.w, acc.w);
        }
    }
    acc.x = apply_act(acc.x, act);
    acc.y = apply_act(acc.y, act);
    acc.z = apply_act(acc.z, act);
    acc.w = apply_act(acc.w, act);
    *reinterpret_cast<float4*>(y + ((b * OH + oh) * OW + ow) * C + 4 * cg) = acc;
}

// pointwise 1x1: [P][Cin] -> [P][Cout] (+gate on the input channels, +residual, activation).
// One thread = 4 output channels of one position; rows = B*P.
__global__ void f32_pw_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gate,
                              float* __restrict__ y, int P, int Cin, int Cout, int act, const float* __restrict__ w,
                              const float* __restrict__ bias, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int n4 = Cout >> 2;
    const int ng = (int)(gid % n4);
    const long row = gid / n4;
    const float* xr = x + row * Cin;
    const float* g = gate ? gate + (row / P) * Cin : nullptr;
    float4 acc = *reinterpret_cast<const float4*>(bias + 4 * ng);
    for (int k = 0; k < Cin; k += 4) {
        float4 v = *reinterpret_cast<const float4*>(xr + k);
        if (g) {
            const float4 gg = *reinterpret_cast<const float4*>(g + k);
            v.x *= gg.x;
            v.y *= gg.y;
            v.z *= gg.z;
            v.w *= gg.w;
        }
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 kk = *reinterpret_cast<const float4*>(w + (long)(k + e) * Cout + 4 * ng);
            acc.x = fmaf(vv[e], kk.x, acc.x);
            acc.y = fmaf(vv[e], kk.y, acc.y);
            acc.z = fmaf(vv[e], kk.z, acc.z);
            acc.w = fmaf(vv[e], kk.w, acc.w);
        }
    }
    if (res) {
        const float4 rr = *reinterpret_cast<const float4*>(res + row * Cout + 4 * ng);
        acc.x += rr.x;
        acc.y += rr.y;
        acc.z += rr.z;
        acc.w += rr.w;
    }
    acc.x = apply_act(acc.x, act);
    acc.y = apply_act(acc.y, act);
    acc.z = apply_act(acc.z, act);
    acc.w = apply_act(acc.w, act);
    *reinterpret_cast<float4*>(y + row * Cout + 4 * ng) = acc;
}

// global average pool [P][C] -> [C]; blockIdx.x = chunk, thread = channel
__global__ void f32_gap_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float* p = x + (size_t)b * P * C + c;
        float s = 0.0f;
        for (int i = 0; i < P; ++i) s += p[(size_t)i * C];
        y[(size_t)b * C + c] = s / (float)P;
    }
}

// squeeze-excite gate: mean over positions -> Dense(Cr, relu) -> Dense(C, sigmoid); one 256-thread block per chunk.
// The pooling is the whole cost (the block reads its chunk's [P][C] activation once): all 256 threads take part — thread
// (g, cq) adds the float4 channel quad cq of positions g, g + G, ... (G = 256 / (C/4) position groups, eight loads in
// flight), the G partial sums are then added in group order.  (The first version used one thread per channel: 48 of 256.)
__global__ __launch_bounds__(256) void f32_segate_kernel(const float* __restrict__ x, float* __restrict__ gate, int P, int C, int Cr,
                                                         const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ part_in,
                                                         int R) {
    extern __shared__ float sm[];  // [C] means, [Cr] hidden, [G][C] partial sums
    float* mean = sm;
    float* hid = sm + C;
    float* part = hid + Cr;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Cq = C >> 2;
    if (part_in) {  // R partial sums per chunk from the kernel that wrote the map (f32_pwdw_kernel): added in row-block order
        for (int c = tid; c < C; c += 256) {
            float s = 0.0f;
            for (int r = 0; r < R; ++r) s += part_in[((size_t)b * R + r) * C + c];
            mean[c] = s / (float)P;
        }
    } else if ((C & 3) == 0 && Cq <= 256) {
        const int G = 256 / Cq;
        const int g = tid / Cq, cq = tid - g * Cq;
        if (g < G) {
            const float4* p = reinterpret_cast<const float4*>(x + (size_t)b * P * C) + cq;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int i = g; i < P; i += G) {
                const float4 v = p[(size_t)i * Cq];
                s.x += v.x;
                s.y += v.y;
                s.z += v.z;
                s.w += v.w;
            }
            *reinterpret_cast<float4*>(part + g * C + 4 * cq) = s;
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float s = 0.0f;
            for (int k = 0; k < G; ++k) s += part[k * C + c];
            mean[c] = s / (float)P;
        }
    } else {
        for (int c = tid; c < C; c += 256) {
            const float* p = x + (size_t)b * P * C + c;
            float s = 0.0f;
            for (int i = 0; i < P; ++i) s += p[(size_t)i * C];
            mean[c] = s / (float)P;
        }
    }
    __syncthreads();
    // (the two dense layers: sixteen weight loads in flight per thread, summed in the same order — one at a time the C = 256 .. 768 terms of a
    // hidden unit were as many dependent cache round trips: 0.035 of the kernel's 0.04 ms behind a fused block)
    for (int r = tid; r < Cr; r += 256) {
        float s = 0.0f;
        int c = 0;
        for (; c + 16 <= C; c += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w1[(c + u) * Cr + r];
#pragma unroll
            for (int u = 0; u < 16; ++u) s = fmaf(mean[c + u], wv[u], s);
        }
        for (; c < C; ++c) s = fmaf(mean[c], w1[c * Cr + r], s);
        hid[r] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.0f;
        int r = 0;
        for (; r + 16 <= Cr; r += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w2[(r + u) * C + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) s = fmaf(hid[r + u], wv[u], s);
        }
        for (; r < Cr; ++r) s = fmaf(hid[r], w2[r * C + c], s);
        gate[(size_t)b * C + c] = 1.0f / (1.0f + expf(-s));
    }
}

__global__ void f32_scale_kernel(const float* __restrict__ x, const float* __restrict__ gate, float* __restrict__ y,
                                 int P, int C, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const long b = gid / ((long)P * C);
    y[gid] = x[gid] * gate[b * C + c];
}

// attention pooling over positions (reference: blocks.py:136-159); one block per chunk
__global__ void f32_attnpool_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int C,
                                    const float* __restrict__ score) {
    extern __shared__ float sm[];  // [P] attention weights
    const int b = blockIdx.x;
    const float* xb = x + (size_t)b * P * C;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s = fmaf(xb[(size_t)p * C + c], score[c], s);
        sm[p] = s;
    }
    __syncthreads();
    float mx = -3.4e38f;
    for (int p = 0; p < P; ++p) mx = fmaxf(mx, sm[p]);
    float den = 0.0f;
    for (int p = 0; p < P; ++p) den += expf(sm[p] - mx);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.0f;
        for (int p = 0; p < P; ++p) s = fmaf(xb[(size_t)p * C + c], expf(sm[p] - mx) / den, s);
        y[(size_t)b * C + c] = s;
    }
}

// classifier head: Dense + sigmoid / softmax; one block per chunk
__global__ void f32_dense_kernel(const float* __restrict__ x, float* __restrict__ scores, float* __restrict__ logits,
                                 int Cin, int Cout, int act, const float* __restrict__ w,
                                 const float* __restrict__ bias) {
    extern __shared__ float sm[];  // [Cin] input, [Cout] logits
    float* xin = sm;
    float* z = sm + Cin;
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < Cin; k += blockDim.x) xin[k] = x[(size_t)b * Cin + k];
    __syncthreads();
    for (int n = threadIdx.x; n < Cout; n += blockDim.x) {
        float s = bias ? bias[n] : 0.0f;
        for (int k = 0; k < Cin; ++k) s = fmaf(xin[k], w[(size_t)k * Cout + n], s);
        z[n] = s;
        if (logits) logits[(size_t)b * Cout + n] = s;
    }
    __syncthreads();
    if (act == 2) {
        float mx = -3.4e38f;
        for (int n = 0; n < Cout; ++n) mx = fmaxf(mx, z[n]);
        float den = 0.0f;
        for (int n = 0; n < Cout; ++n) den += expf(z[n] - mx);
        for (int n = threadIdx.x; n < Cout; n += blockDim.x) scores[(size_t)b * Cout + n] = expf(z[n] - mx) / den;
    } else {
        for (int n = threadIdx.x; n < Cout; n += blockDim.x) {
            const float v = z[n];
            scores[(size_t)b * Cout + n] = act == 1 ? 1.0f / (1.0f + expf(-v)) : v;
        }
    }
}

// global average pool + Dense (+ sigmoid / softmax): one 256-thread block per kGdChunks chunks, so that the dense weight
// matrix (100 KB for 256 x 100: three times the activations it multiplies) is fetched from L2 once per four chunks.
// Threads first average their channel over the P positions, then 4 groups of 64 lanes each take a quarter of the
// contraction for up to 64 x 4 outputs and the partial sums meet in LDS (reference: birdnet_stm32/models/dscnn.py:256-261).
// Per chunk the operations and their order are those of the separate GAP and Dense kernels.
constexpr int kGdChunks = 4;
__global__ __launch_bounds__(256) void f32_gap_dense_kernel(const float* __restrict__ x, float* __restrict__ scores,
                                                            float* __restrict__ logits, int B, int P, int Cin, int Cout, int act,
                                                            const float* __restrict__ w, const float* __restrict__ bias) {
    extern __shared__ float sm[];  // [NC][Cin] pooled, [NC][4][Cout] partial sums, [NC][Cout] logits
    float* pooled = sm;
    float* part = sm + kGdChunks * Cin;
    float* z = part + kGdChunks * 4 * Cout;
    const int b0 = blockIdx.x * kGdChunks, tid = threadIdx.x;
    const int nc = B - b0 < kGdChunks ? B - b0 : kGdChunks;
    for (int item = tid; item < nc * Cin; item += 256) {
        const int cb = item / Cin, c = item - cb * Cin;
        const float* p = x + (size_t)(b0 + cb) * P * Cin + c;
        float s = 0.0f;
#pragma unroll 16
        for (int i = 0; i < P; ++i) s += p[(size_t)i * Cin];  // summed in position order; sixteen loads in flight
        pooled[cb * Cin + c] = s / (float)P;
    }
    for (int item = nc * Cin + tid; item < kGdChunks * Cin; item += 256) pooled[item] = 0.0f;  // chunks past the batch end
    __syncthreads();
    const int g = tid >> 6, lane = tid & 63;
    const int k0 = g * ((Cin + 3) / 4), k1 = min(Cin, k0 + (Cin + 3) / 4);
    for (int n = lane; n < Cout; n += 64) {
        float s[kGdChunks];
#pragma unroll
        for (int cb = 0; cb < kGdChunks; ++cb) s[cb] = 0.0f;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float wk = w[(size_t)k * Cout + n];
#pragma unroll
            for (int cb = 0; cb < kGdChunks; ++cb) s[cb] = fmaf(pooled[cb * Cin + k], wk, s[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < kGdChunks; ++cb) part[(cb * 4 + g) * Cout + n] = s[cb];
    }
    __syncthreads();
    for (int item = tid; item < nc * Cout; item += 256) {
        const int cb = item / Cout, n = item - cb * Cout;
        const float* pp = part + cb * 4 * Cout;
        // summed in contraction order, bias first, like the separate Dense kernel
        const float v = (((((bias ? bias[n] : 0.0f) + pp[n]) + pp[Cout + n]) + pp[2 * Cout + n]) + pp[3 * Cout + n]);
        z[cb * Cout + n] = v;
        if (logits) logits[(size_t)(b0 + cb) * Cout + n] = v;
    }
    __syncthreads();
    for (int item = tid; item < nc * Cout; item += 256) {
        const int cb = item / Cout, n = item - cb * Cout;
        const float* zz = z + cb * Cout;
        float out;
        if (act == 2) {
            float mx = -3.4e38f;
            for (int j = 0; j < Cout; ++j) mx = fmaxf(mx, zz[j]);
            float den = 0.0f;
            for (int j = 0; j < Cout; ++j) den += expf(zz[j] - mx);
            out = expf(zz[n] - mx) / den;
        } else {
            out = act == 1 ? 1.0f / (1.0f + expf(-zz[n])) : zz[n];
        }
        scores[(size_t)(b0 + cb) * Cout + n] = out;
    }
}

inline dim3 grid1d(long total, int block) { return dim3((unsigned)((total + block - 1) / block)); }

}  // namespace

void launch_u32_fill(uint32_t* p, uint32_t v, int n, hipStream_t s) {
    hipLaunchKernelGGL(u32_fill_kernel, grid1d(n, 256), dim3(256), 0, s, p, v, n);
}

void launch_f32_mel(const float* spec, const float* minmax, float* out, float* smax, int B, int F, int W, int M,
                    const float* wvals, const int* bands, const float* magp, int mag, int norm, hipStream_t s) {
    hipLaunchKernelGGL(f32_mel_kernel, dim3((W + 255) / 256, M, B), dim3(256), 0, s, spec, minmax, out, smax, F, W, M,
                       wvals, bands, magp, mag, norm);
}

void launch_f32_mag(float* x, const float* smax, int B, int M, int W, const float* magp, int mag, hipStream_t s) {
    hipLaunchKernelGGL(f32_mag_kernel, dim3(16, B), dim3(256), 0, s, x, smax, M, W, magp, mag);
}

void launch_f32_rawfe(const float* x, float* out, int B, int T, int W, int M, int stride, int pad_left, const float* fb,
                      const float* bias, const float* magp, int mag, hipStream_t s) {
    hipLaunchKernelGGL(f32_rawfe_kernel, dim3((W + 255) / 256, B), dim3(256), (size_t)17 * M * sizeof(float), s, x, out, T, W, M, stride, pad_left, fb,
                       bias, magp, mag);
}

void launch_f32_melfin(const float* melraw, const float* minmax, float* out, int B, int M, int W, const float* wsum,
                       const float* magp, int mag, int norm, hipStream_t s) {
    hipLaunchKernelGGL(f32_melfin_kernel, dim3(B), dim3(256), 0, s, melraw, minmax, out, M, W, wsum, magp, mag, norm);
}

void launch_f32_stem(const float* x, float* y, int B, int H, int W, int Cout, int sh, int sw, int act, int OH, int OW,
                     int pt, int pl, const float* w, const float* bias, hipStream_t s) {
    const long total = (long)B * OH * OW * (Cout / 4);
    hipLaunchKernelGGL(f32_stem_kernel, grid1d(total, 256), dim3(256), 0, s, x, y, H, W, Cout, sh, sw, act, OH, OW, pt,
                       pl, w, bias, total);
}

bool launch_f32_dw(const float* x, float* y, int B, int H, int W, int C, int sh, int sw, int act, int OH, int OW,
                   int pt, int pl, const float* w, const float* bias, float* gap_part, hipStream_t s) {
    if (launch_f32_dw_stream(x, y, B, H, W, C, sh, sw, act, OH, OW, pt, pl, w, bias, gap_part, s)) return true;
    const long total = (long)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(f32_dw_kernel, grid1d(total, 256), dim3(256), 0, s, x, y, H, W, C, sh, sw, act, OH, OW, pt, pl,
                       w, bias, total);
    return false;
}

void launch_f32_pw(const float* x, const float* res, const float* gate, float* y, int B, int P, int Cin, int Cout,
                   int act, const float* w, const float* bias, hipStream_t s) {
    const long total = (long)B * P * (Cout / 4);
    hipLaunchKernelGGL(f32_pw_kernel, grid1d(total, 256), dim3(256), 0, s, x, res, gate, y, P, Cin, Cout, act, w, bias,
                       total);
}

void launch_f32_segate(const float* x, float* gate, int B, int P, int C, int Cr, const float* w1, const float* w2, const float* part, int R,
                       hipStream_t s) {
    const int groups = (C & 3) == 0 && C / 4 <= 256 ? 256 / (C / 4) : 0;
    hipLaunchKernelGGL(f32_segate_kernel, dim3(B), dim3(256), (C + Cr + (size_t)groups * C) * sizeof(float), s, x, gate, P, C, Cr, w1, w2, part, R);
}

void launch_f32_scale(const float* x, const float* gate, float* y, int B, int P, int C, hipStream_t s) {
    const long total = (long)B * P * C;
    hipLaunchKernelGGL(f32_scale_kernel, grid1d(total, 256), dim3(256), 0, s, x, gate, y, P, C, total);
}

void launch_f32_gap(const float* x, float* y, int B, int P, int C, hipStream_t s) {
    hipLaunchKernelGGL(f32_gap_kernel, dim3(B), dim3(256), 0, s, x, y, P, C);
}

void launch_f32_dense(const float* x, float* scores, float* logits, int B, int Cin, int Cout, int act, const float* w,
                      const float* bias, hipStream_t s) {
    hipLaunchKernelGGL(f32_dense_kernel, dim3(B), dim3(128), (Cin + Cout) * sizeof(float), s, x, scores, logits, Cin,
                       Cout, act, w, bias);
}

void launch_f32_gap_dense(const float* x, float* scores, float* logits, int B, int P, int Cin, int Cout, int act, const float* w,
                          const float* bias, hipStream_t s) {
    hipLaunchKernelGGL(f32_gap_dense_kernel, dim3((B + kGdChunks - 1) / kGdChunks), dim3(256),
                       kGdChunks * (Cin + 5 * Cout) * sizeof(float), s, x, scores, logits, B, P, Cin, Cout, act, w, bias);
}

void launch_f32_attnpool(const float* x, float* y, int B, int P, int C, const float* score, hipStream_t s) {
    hipLaunchKernelGGL(f32_attnpool_kernel, dim3(B), dim3(256), P * sizeof(float), s, x, y, P, C, score);
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_f32() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&u32_fill_kernel));
}

}  // namespace bn
