// bn_i8.hip — bit-faithful INT8 DS-CNN kernels for gfx950 (baseline generation: dot4 on the
// vector ALU; the MFMA pointwise kernels build on the same requantisation helpers).
//
// Integer semantics are those of the TFLite reference kernels that execute the reference's
// shipped birdnet_stm32n6_100.tflite through tf.lite.Interpreter
// (reference: birdnet_stm32/models/runners.py:51-95; graph: SURVEY.md Appendix B):
//   acc32 = sum((x - zp_x) * w) + bias
//   y = clamp(MultiplyByQuantizedMultiplier(acc32, M0[c], shift[c]) + zp_y, act_min, act_max)
// with MultiplyByQuantizedMultiplier = RoundingDivideByPOT(SaturatingRoundingDoublingHighMul(.)).
// Where a layer has no spatial padding (1x1 convs, mel mixer, FC) the packer folds -zp_x*sum(w)
// into the bias, which is the same int32 arithmetic re-associated (exact).
#include "bn_kernels.h"
#include "bn_requant.h"

namespace bn {
namespace {





__device__ __forceinline__ int32_t dot4(int32_t a, int32_t b, int32_t c) {
#if __has_builtin(__builtin_amdgcn_sdot4)
    return __builtin_amdgcn_sdot4(a, b, c, false);
#else
    c += (int32_t)(int8_t)(a) * (int32_t)(int8_t)(b);
    c += (int32_t)(int8_t)(a >> 8) * (int32_t)(int8_t)(b >> 8);
    c += (int32_t)(int8_t)(a >> 16) * (int32_t)(int8_t)(b >> 16);
    c += (int32_t)(int8_t)(a >> 24) * (int32_t)(int8_t)(b >> 24);
    return c;
#endif
}

// TFLite int8 ADD (left_shift = 20); a = first ADD input, b = second
struct AddQ {
    int z1, m1, s1, z2, m2, s2, mo, so, zo, amin, amax;
};
__device__ __forceinline__ int32_t add_q(int32_t a, int32_t b, const AddQ& q) {
    const int32_t sa = mbqm((a - q.z1) * (1 << 20), q.m1, q.s1);
    const int32_t sb = mbqm((b - q.z2) * (1 << 20), q.m2, q.s2);
    return clampi(mbqm(sa + sb, q.mo, q.so) + q.zo, q.amin, q.amax);
}

// QUANTIZE + TRANSPOSE + zero-pad: spec f32 [F][W] -> int8 [W][Kp].  64x64 tile through LDS so that both sides move whole
// dwords: float4 loads along t (four frequency rows per wave-instruction), dword stores along f (four frames per
// wave-instruction) — byte-wide global stores were the bottleneck of the first version.
__device__ __forceinline__ int quant_one(float v, bool renorm, float mn, float rng, float scale, int zp) {
    if (renorm) v = (v - mn) / rng;
    return clampi((int32_t)roundf(v / scale) + zp, -128, 127);  // roundf: halves away from zero
}

__global__ __launch_bounds__(256) void i8_quant_kernel(const float* __restrict__ spec, const float* __restrict__ minmax,
                                                       int8_t* __restrict__ out, int F, int W, int Kp, int zp,
                                                       int fill, float scale) {
    __shared__ __attribute__((aligned(4))) int8_t tile[64][68];  // [t][f], row stride 17 dwords
    const int b = blockIdx.z;
    const int f0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const float* S = spec + (size_t)b * F * W;
    float mn = 0.0f, rng = 1.0f;
    const bool renorm = minmax != nullptr;
    if (renorm) {
        mn = minmax[2 * b];
        rng = (float)((double)(minmax[2 * b + 1] - mn) + 1e-10);
    }
    const int c4 = threadIdx.x & 15, r4 = threadIdx.x >> 4;
    const bool vec = (W & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r4 + 16 * i, f = f0 + r, t = t0 + 4 * c4;
        int q[4] = {fill, fill, fill, fill};  // padded frequency columns: the graph's FILL constant (the zero point = real 0.0)
        if (f < F) {
            if (vec && t + 3 < W) {
                const float4 v = *reinterpret_cast<const float4*>(S + (size_t)f * W + t);
                q[0] = quant_one(v.x, renorm, mn, rng, scale, zp);
                q[1] = quant_one(v.y, renorm, mn, rng, scale, zp);
                q[2] = quant_one(v.z, renorm, mn, rng, scale, zp);
                q[3] = quant_one(v.w, renorm, mn, rng, scale, zp);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + e < W) q[e] = quant_one(S[(size_t)f * W + t + e], renorm, mn, rng, scale, zp);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[4 * c4 + e][r] = (int8_t)q[e];
    }
    __syncthreads();
    const bool dword_rows = (Kp & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r4 + 16 * i, t = t0 + r, f = f0 + 4 * c4;
        if (t >= W || f >= Kp) continue;
        int8_t* dst = out + ((size_t)b * W + t) * Kp + f;
        if (dword_rows && f + 3 < Kp) {
            *reinterpret_cast<int*>(dst) = *reinterpret_cast<const int*>(&tile[r][4 * c4]);
        } else {
            for (int e = 0; e < 4 && f + e < Kp; ++e) dst[e] = tile[r][4 * c4 + e];
        }
    }
}

// Raw frontend of an exported INT8 graph (reference models/frontend.py:138-164,347-358): QUANTIZE of the waveform -> [PAD] ->
// CONV_2D 1 x 16, stride `stride`, VALID (BatchNorm folded, ReLU6 clamp) -> per-channel table of the magnitude scaling behind it ->
// output transposed to [M][W].  A workgroup = 256 consecutive frames of one chunk; a thread owns four consecutive frames (one dword
// of every output row) and quantises their 4 x 16 samples once; wave v walks the filters v, v + 4, ... (filter constants are
// wave-uniform: scalar loads).  Samples outside the waveform are the zero point (the PAD's fill; the bias carries -zp sum(w)).
__global__ __launch_bounds__(256) void i8_rawfe_kernel(const float* __restrict__ x, int8_t* __restrict__ y, int T, int W, int M, int stride,
                                                       int pad_left, float q_scale, int q_zp, int zp_out, int amin, int amax,
                                                       const int8_t* __restrict__ w, const int32_t* __restrict__ bias,
                                                       const int32_t* __restrict__ mult, const int32_t* __restrict__ shift,
                                                       const int8_t* __restrict__ lut) {
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 256 + 4 * (threadIdx.x & 63);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t0 >= W) return;
    const float* xb = x + (size_t)b * T;
    int32_t win[4][4];  // [frame][4 samples per dword]
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const long s0 = (long)(t0 + f) * stride - pad_left;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            int32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long si = s0 + 4 * d + e;
                const int q = (si >= 0 && si < T && t0 + f < W) ? quant_one(xb[si], false, 0.f, 1.f, q_scale, q_zp) : q_zp;
                packed |= (q & 0xff) << (8 * e);
            }
            win[f][d] = packed;
        }
    }
    for (int m = wave; m < M; m += 4) {
        const int32_t* wr = reinterpret_cast<const int32_t*>(w + (size_t)m * 16);
        const int32_t w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
        const int32_t bm = bias[m], mm = mult[m], sm = shift[m];
        int32_t packed = 0;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            int32_t acc = dot4(win[f][0], w0, bm);
            acc = dot4(win[f][1], w1, acc);
            acc = dot4(win[f][2], w2, acc);
            acc = dot4(win[f][3], w3, acc);
            int32_t q = clampi(mbqm(acc, mm, sm) + zp_out, amin, amax);
            if (lut) q = lut[m * 256 + q + 128];
            packed |= (q & 0xff) << (8 * f);
        }
        int8_t* dst = y + ((size_t)b * M + m) * W + t0;
        if (t0 + 3 < W) {
            *reinterpret_cast<int32_t*>(dst) = packed;
        } else {
            for (int f = 0; f < 4 && t0 + f < W; ++f) dst[f] = (int8_t)(packed >> (8 * f));
        }
    }
}

// mel mixer (CONV_2D 1x1 over Kp frequency columns) + ReLU clamp + the PWL chain as a per-channel
// 256-entry table; output transposed to [M][W].  One thread = one (t, m); lanes run along t.
__global__ void i8_mel_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, int W, int Kp, int M, int zp_out,
                              int amin, int amax, const int8_t* __restrict__ w, const int32_t* __restrict__ bias,
                              const int32_t* __restrict__ mult, const int32_t* __restrict__ shift,
                              const int8_t* __restrict__ lut) {
    const int b = blockIdx.z, m = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W) return;
    const int32_t* xr = reinterpret_cast<const int32_t*>(x + ((size_t)b * W + t) * Kp);
    const int32_t* wr = reinterpret_cast<const int32_t*>(w + (size_t)m * Kp);
    int32_t acc = bias[m];
    for (int k = 0; k < Kp / 4; ++k) acc = dot4(xr[k], wr[k], acc);
    int32_t q = clampi(mbqm(acc, mult[m], shift[m]) + zp_out, amin, amax);
    if (lut) q = lut[m * 256 + q + 128];
    y[((size_t)b * M + m) * W + t] = (int8_t)q;
}

// stem 3x3 on a single input channel: [H][W] -> [OH][OW][Cout]; one thread = 4 output channels
__global__ void i8_stem_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, I8ConvGeom g,
                               const int8_t* __restrict__ w, const int32_t* __restrict__ bias,
                               const int32_t* __restrict__ mult, const int32_t* __restrict__ shift, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c4 = g.C >> 2;
    const int cg = (int)(gid % c4);
    long r = gid / c4;
    const int ow = (int)(r % g.OW);
    r /= g.OW;
    const int oh = (int)(r % g.OH);
    const long b = r / g.OH;
    const int8_t* xin = x + b * g.H * g.W;
    int32_t acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = bias[4 * cg + e];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ih = oh * g.sh + i - g.pt;
        if (ih < 0 || ih >= g.H) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int iw = ow * g.sw + j - g.pl;
            if (iw < 0 || iw >= g.W) continue;
            const int32_t v = (int32_t)xin[ih * g.W + iw] - g.zp_in;
            const int32_t k4 = *reinterpret_cast<const int32_t*>(w + (i * 3 + j) * g.C + 4 * cg);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v * (int32_t)(int8_t)(k4 >> (8 * e));
        }
    }
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * cg + e;
        const int32_t q = clampi(mbqm(acc[e], mult[c], shift[c]) + g.zp_out, g.amin, g.amax);
        packed |= ((uint32_t)(uint8_t)(int8_t)q) << (8 * e);
    }
    *reinterpret_cast<uint32_t*>(y + ((b * g.OH + oh) * g.OW + ow) * g.C + 4 * cg) = packed;
}

// depthwise 3x3: [H][W][C] -> [OH][OW][C]; one thread = 4 channels of one output pixel
__global__ void i8_dw_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, I8ConvGeom g,
                             const int8_t* __restrict__ w, const int32_t* __restrict__ bias,
                             const int32_t* __restrict__ mult, const int32_t* __restrict__ shift, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c4 = g.C >> 2;
    const int cg = (int)(gid % c4);
    long r = gid / c4;
    const int ow = (int)(r % g.OW);
    r /= g.OW;
    const int oh = (int)(r % g.OH);
    const long b = r / g.OH;
    const int8_t* xin = x + b * g.H * g.W * g.C;
    int32_t acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = bias[4 * cg + e];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ih = oh * g.sh + i - g.pt;
        if (ih < 0 || ih >= g.H) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int iw = ow * g.sw + j - g.pl;
            if (iw < 0 || iw >= g.W) continue;
            const int32_t v4 = *reinterpret_cast<const int32_t*>(xin + ((long)ih * g.W + iw) * g.C + 4 * cg);
            const int32_t k4 = *reinterpret_cast<const int32_t*>(w + (i * 3 + j) * g.C + 4 * cg);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[e] += ((int32_t)(int8_t)(v4 >> (8 * e)) - g.zp_in) * (int32_t)(int8_t)(k4 >> (8 * e));
        }
    }
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * cg + e;
        const int32_t q = clampi(mbqm(acc[e], mult[c], shift[c]) + g.zp_out, g.amin, g.amax);
        packed |= ((uint32_t)(uint8_t)(int8_t)q) << (8 * e);
    }
    *reinterpret_cast<uint32_t*>(y + ((b * g.OH + oh) * g.OW + ow) * g.C + 4 * cg) = packed;
}

// pointwise 1x1 (+ TFLite ADD with the residual): [P][Cin] -> [P][Cout]; one thread = 4 output channels
__global__ void i8_pw_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ res, int8_t* __restrict__ y,
                             int Cin, int Cout, int zp_out, int amin, int amax, I8AddParams add,
                             const int8_t* __restrict__ w, const int32_t* __restrict__ bias,
                             const int32_t* __restrict__ mult, const int32_t* __restrict__ shift, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int n4 = Cout >> 2;
    const int ng = (int)(gid % n4);
    const long row = gid / n4;
    const int32_t* xr = reinterpret_cast<const int32_t*>(x + row * Cin);
    int32_t acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = bias[4 * ng + e];
    for (int k = 0; k < Cin / 4; ++k) {
        const int32_t v = xr[k];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc[e] = dot4(v, reinterpret_cast<const int32_t*>(w + (size_t)(4 * ng + e) * Cin)[k], acc[e]);
    }
    AddQ aq{add.z1, add.m1, add.s1, zp_out, add.m2, add.s2, add.mo, add.so, add.zo, add.amin, add.amax};
    uint32_t r4 = 0;
    if (add.enabled) r4 = *reinterpret_cast<const uint32_t*>(res + row * Cout + 4 * ng);
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * ng + e;
        int32_t q = clampi(mbqm(acc[e], mult[c], shift[c]) + zp_out, amin, amax);
        if (add.enabled) q = add_q((int32_t)(int8_t)(r4 >> (8 * e)), q, aq);
        packed |= ((uint32_t)(uint8_t)(int8_t)q) << (8 * e);
    }
    *reinterpret_cast<uint32_t*>(y + row * Cout + 4 * ng) = packed;
}

// MEAN over positions: int32 sum - P*zp, then the folded multiplier; blockIdx.x = chunk.  Thread (slice, channel quad): the 256 threads
// split the positions into 256 / (C / 4) slices (coalesced dword loads, four channels per thread), the slices meet in LDS.
__global__ __launch_bounds__(256) void i8_mean_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, int P, int C, int zp_in, int mult,
                                                      int shift, int zp_out) {
    __shared__ int part[256][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cq = C >> 2;
    if ((C & 3) || cq > 256) {  // odd channel counts: one thread per channel, all positions
        for (int c = tid; c < C; c += blockDim.x) {
            const int8_t* p = x + (size_t)b * P * C + c;
            int32_t s = 0;
            for (int i = 0; i < P; ++i) s += p[(size_t)i * C];
            y[(size_t)b * C + c] = (int8_t)mean_q(s, P, zp_in, mult, shift, zp_out);
        }
        return;
    }
    const int slices = 256 / cq;             // >= 1
    const int q = tid % cq, sl = tid / cq;
    int acc[4] = {0, 0, 0, 0};
    if (sl < slices) {
        // range-checked raw buffer loads over this chunk's map: a position beyond P lies behind the buffer and reads as 0 (adds nothing) — no
        // branch around the loads, so the eight really are in flight together (one workgroup per chunk has to cover the latency itself)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(x) + (size_t)b * P * C, 0, P * C, 0x00020000);
        for (int i0 = sl; i0 < P; i0 += 8 * slices) {
            int32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, ((i0 + u * slices) * cq + q) * 4, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += (int32_t)(int8_t)(v[u] >> (8 * e));
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[tid][e] = acc[e];
    __syncthreads();
    if (tid < cq) {
        int s4[4] = {0, 0, 0, 0};
        for (int k = 0; k < slices; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] += part[k * cq + tid][e];
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int32_t qv = mean_q(s4[e], P, zp_in, mult, shift, zp_out);
            packed |= ((uint32_t)(uint8_t)(int8_t)qv) << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(y + (size_t)b * C + 4 * tid) = packed;
    }
}

// Squeeze-excite gate in one kernel per chunk: MEAN over the positions (as i8_mean_kernel) -> FULLY_CONNECTED C -> R (fused ReLU
// clamp) -> FULLY_CONNECTED R -> C with the LOGISTIC table: three launches of ~10 us each (the two dense layers are a few
// hundred multiply-adds) become one; the pooled vector and the hidden vector stay in LDS.
struct SeGate8Args {
    const int8_t* x; int8_t* y;   // [B][P][C] -> gate [B][C]
    int P, C, zp_in, mean_mult, mean_shift, mean_zp;
    int R, Kp1, zo1, amin1, amax1;
    const int8_t* w1; const int32_t* b1; const int32_t* m1; const int32_t* s1; const int8_t* lut1;
    int Kp2, zo2, amin2, amax2;
    const int8_t* w2; const int32_t* b2; const int32_t* m2; const int32_t* s2; const int8_t* lut2;
    int32_t* sums;  // [B][C] channel sums of x taken by the kernel that wrote it (read and zeroed here), or null
};

// dot product of an int8 vector in LDS with a weight row in memory, `nd` dwords: sixteen weight dwords requested before the first is
// used (integer sums: any order gives the same result); one dword per iteration was one cache round trip per four channels
__device__ __forceinline__ int32_t row_dot(const int32_t* __restrict__ v, const int32_t* __restrict__ wr, int nd, int32_t acc) {
    int k = 0;
    for (; k + 16 <= nd; k += 16) {
        int32_t w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = wr[k + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = dot4(v[k + u], w[u], acc);
    }
    for (; k < nd; ++k) acc = dot4(v[k], wr[k], acc);
    return acc;
}

__global__ __launch_bounds__(256) void i8_segate_kernel(SeGate8Args a) {
    __shared__ int part[256][4];
    extern __shared__ int32_t se_vec[];  // pooled vector (Kp1 bytes) then hidden vector (Kp2 bytes), as dwords
    const int b = blockIdx.x, tid = threadIdx.x;
    const int C = a.C, P = a.P, cq = C >> 2;
    int8_t* pooled = reinterpret_cast<int8_t*>(se_vec);
    int8_t* hidden = pooled + a.Kp1;
    for (int i = tid; i < (a.Kp1 + a.Kp2) / 4; i += 256) se_vec[i] = 0;
    __syncthreads();
    if (a.sums) {
        // the depthwise kernel in front added the map up while it stored it (i8_dw_stream_kernel): only the requantisation is left
        int32_t* srow = a.sums + (size_t)b * C;
        for (int q4 = tid; q4 < cq; q4 += 256) {
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s4 = srow[4 * q4 + e];
                srow[4 * q4 + e] = 0;  // ready for the next stage that pools into this buffer
                const int32_t qv = mean_q(s4, P, a.zp_in, a.mean_mult, a.mean_shift, a.mean_zp);
                packed |= ((uint32_t)(uint8_t)(int8_t)qv) << (8 * e);
            }
            se_vec[q4] = (int32_t)packed;
        }
        __syncthreads();
    } else {
        for (int q0 = 0; q0 < cq; q0 += 256) {  // channel quads beyond 256 go round again
            const int nq = cq - q0 < 256 ? cq - q0 : 256;
            const int slices = 256 / nq;
            const int q = tid % nq, sl = tid / nq;
            int acc[4] = {0, 0, 0, 0};
            if (sl < slices) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x) + (size_t)b * P * C, 0, P * C, 0x00020000);
                for (int i0 = sl; i0 < P; i0 += 8 * slices) {
                    int32_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, ((i0 + u * slices) * cq + q0 + q) * 4, 0, 0);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] += (int32_t)(int8_t)(v[u] >> (8 * e));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) part[tid][e] = acc[e];
            __syncthreads();
            if (tid < nq) {
                uint32_t packed = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int s4 = 0;
                    for (int k = 0; k < slices; ++k) s4 += part[k * nq + tid][e];
                    const int32_t qv = mean_q(s4, P, a.zp_in, a.mean_mult, a.mean_shift, a.mean_zp);
                    packed |= ((uint32_t)(uint8_t)(int8_t)qv) << (8 * e);
                }
                se_vec[q0 + tid] = (int32_t)packed;
            }
            __syncthreads();
        }
    }
    for (int n = tid; n < a.R; n += 256) {
        const int32_t* wr = reinterpret_cast<const int32_t*>(a.w1 + (size_t)n * a.Kp1);
        int32_t acc = a.b1[n];
        acc = row_dot(se_vec, wr, a.Kp1 / 4, acc);
        int32_t qv = clampi(mbqm(acc, a.m1[n], a.s1[n]) + a.zo1, a.amin1, a.amax1);
        if (a.lut1) qv = a.lut1[qv + 128];
        hidden[n] = (int8_t)qv;
    }
    __syncthreads();
    const int32_t* hv = se_vec + a.Kp1 / 4;
    for (int n = tid; n < C; n += 256) {
        const int32_t* wr = reinterpret_cast<const int32_t*>(a.w2 + (size_t)n * a.Kp2);
        int32_t acc = a.b2[n];
        acc = row_dot(hv, wr, a.Kp2 / 4, acc);
        int32_t qv = clampi(mbqm(acc, a.m2[n], a.s2[n]) + a.zo2, a.amin2, a.amax2);
        if (a.lut2) qv = a.lut2[qv + 128];
        a.y[(size_t)b * C + n] = (int8_t)qv;
    }
}

// FULLY_CONNECTED: a workgroup takes kFcChunks chunks, so a weight row is fetched once per kFcChunks chunks (the 25.6 KB matrix
// was read from L2 once per chunk: 0.048 -> 0.02 ms per 4096 chunks); the activations of the chunks sit in LDS.
constexpr int kFcChunks = 8;
// Cin = bytes per input row as the producer wrote them; Kp = Cin rounded up to a multiple of 4 = row length of the (zero-padded)
// weight matrix.  lut (optional): the int8 LOGISTIC behind the layer (squeeze-excite gates), applied to the clamped result.
__global__ __launch_bounds__(128) void i8_fc_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, int B, int Cin, int Kp, int Cout,
                                                    int zp_out, int amin, int amax, const int8_t* __restrict__ w,
                                                    const int32_t* __restrict__ bias, const int32_t* __restrict__ mult,
                                                    const int32_t* __restrict__ shift, const int8_t* __restrict__ lut) {
    extern __shared__ int32_t fc_x[];  // [kFcChunks][Kp / 4]
    const int b0 = blockIdx.x * kFcChunks;
    const int nb = B - b0 < kFcChunks ? B - b0 : kFcChunks;
    const int kq = Kp / 4;
    if (Cin == Kp) {
        const int32_t* xr = reinterpret_cast<const int32_t*>(x + (size_t)b0 * Cin);
        for (int i = threadIdx.x; i < nb * kq; i += blockDim.x) fc_x[i] = xr[i];
        for (int i = nb * kq + threadIdx.x; i < kFcChunks * kq; i += blockDim.x) fc_x[i] = 0;
    } else {  // rows that are not a whole number of dwords: byte copies, zero padding up to Kp (the padded weights are zero there)
        int8_t* dst = reinterpret_cast<int8_t*>(fc_x);
        for (int i = threadIdx.x; i < kFcChunks * Kp; i += blockDim.x) {
            const int c = i / Kp, k = i - c * Kp;
            dst[i] = (c < nb && k < Cin) ? x[(size_t)(b0 + c) * Cin + k] : (int8_t)0;
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < Cout; n += blockDim.x) {
        const int32_t* wr = reinterpret_cast<const int32_t*>(w + (size_t)n * Kp);
        int32_t acc[kFcChunks];
#pragma unroll
        for (int c = 0; c < kFcChunks; ++c) acc[c] = bias[n];
        for (int k = 0; k < kq; ++k) {
            const int32_t wk = wr[k];
#pragma unroll
            for (int c = 0; c < kFcChunks; ++c) acc[c] = dot4(fc_x[c * kq + k], wk, acc[c]);
        }
        const int32_t m = mult[n], sh = shift[n];
#pragma unroll
        for (int c = 0; c < kFcChunks; ++c)
            if (c < nb) {
                int32_t q = clampi(mbqm(acc[c], m, sh) + zp_out, amin, amax);
                if (lut) q = lut[q + 128];
                y[(size_t)(b0 + c) * Cout + n] = (int8_t)q;
            }
    }
}

// MUL of a map with a per-chunk, per-channel gate (squeeze-excite): y = clamp(zo + MBQM((x - zx)(g - zg), m, shift)); one thread
// per four channels of one position
__global__ void i8_scale_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ gate, int8_t* __restrict__ y, int P, int C,
                                int zx, int zg, int mult, int shift, int zo, int amin, int amax, long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int cq = C / 4;
    const long row = gid / cq;             // chunk * P + position
    const int c4 = (int)(gid - row * cq);
    const long b = row / P;
    const uint32_t xv = *reinterpret_cast<const uint32_t*>(x + row * C + 4 * c4);
    const uint32_t gv = *reinterpret_cast<const uint32_t*>(gate + b * C + 4 * c4);
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int32_t a = (int32_t)(int8_t)(xv >> (8 * e)) - zx, g = (int32_t)(int8_t)(gv >> (8 * e)) - zg;
        const int32_t q = clampi(mbqm(a * g, mult, shift) + zo, amin, amax);
        packed |= ((uint32_t)(uint8_t)(int8_t)q) << (8 * e);
    }
    *reinterpret_cast<uint32_t*>(y + row * C + 4 * c4) = packed;
}

// Per-sample max normalisation of current hybrid frontends in INT8 (reference models/frontend.py:338-342; TFLite writes REDUCE_MAX ->
// ADD 1e-6 -> DIV): one workgroup per chunk.  The only data-dependent quantity is the map's maximum byte; the quantised ADD of the
// epsilon is a 256-entry table of it, and the int8 DIV by the resulting scalar is one row of a [256][256] byte table (both built by
// the packer from the operators' parameters, models/_quant.py: div_table) — the kernel does no fixed-point division at all.
__global__ __launch_bounds__(256) void i8_maxnorm_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ y, int C, int W,
                                                         const int8_t* __restrict__ den_tab, const int8_t* __restrict__ div_tab,
                                                         const int8_t* __restrict__ lut) {
    __shared__ int red[4];
    __shared__ uint8_t row[256];
    const int tid = threadIdx.x;
    const int n4 = C * W / 4;  // dwords per chunk
    const uint32_t* src = reinterpret_cast<const uint32_t*>(x + (size_t)blockIdx.x * C * W);
    uint32_t* dst = reinterpret_cast<uint32_t*>(y + (size_t)blockIdx.x * C * W);
    int mx = -128;
    for (int i = tid; i < n4; i += 256) {
        const uint32_t v = src[i];
        mx = max(max(mx, (int)(int8_t)v), max((int)(int8_t)(v >> 8), max((int)(int8_t)(v >> 16), (int)(int8_t)(v >> 24))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = max(max(red[0], red[1]), max(red[2], red[3]));
    const int den = den_tab[mx + 128];
    row[tid] = (uint8_t)div_tab[(den + 128) * 256 + tid];
    __syncthreads();
    for (int i = tid; i < n4; i += 256) {
        const uint32_t v = src[i];
        const int c = (4 * i) / W;  // W is a multiple of 4: the four bytes share the channel
        uint32_t o = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t q = row[((v >> (8 * e)) & 0xff) ^ 0x80];  // byte + 128
            if (lut) q = (uint8_t)lut[c * 256 + ((q & 0xff) ^ 0x80)];
            o |= (q & 0xff) << (8 * e);
        }
        dst[i] = o;
    }
}

// Attention pooling of an exported INT8 graph (reference models/blocks.py:136-159; conversion/export.py writes it as RESHAPE ->
// FULLY_CONNECTED C -> 1 per position -> SOFTMAX over the positions -> MUL -> SUM): one workgroup per chunk, the [P][C] map in LDS.
//   scores s[p] = requantised dot product of the position's channels with the score vector        (thread per position, v_dot4)
//   a[p]        = int8 SOFTMAX of s over the positions, output 1 / 256, -128 — table-driven, in either of TFLite's two published forms:
//                 form 0 (reference kernel, gemmlowp fixed point): table[0][d] = exp(-beta s d) as Q0.31, table[1][d] the same in the
//                 accumulator's Q12.19, d = max - s; reciprocal of the sum by gemmlowp's Newton-Raphson division (three steps on the
//                 half denominator), output RoundingDivideByPOT(SRDHM(1 / sum, exp), bits + 31 - 8) - 128;
//                 form 1 (optimized kernel, float32 table): sum in float32 in position order, prob = exp / (sum / 256) rounded half away.
//   out[c]      = requantised sum over p of the int8 MUL (x[p][c] - zx) (a[p] + 128)                (thread per channel)
// Integer steps use the literal gemmlowp definitions (srdhm_ref / rdivpot_ref): exactness over speed, the operator is ~1e5 operations per chunk.
struct AttnPool8Args {
    const int8_t* x;      // [B][P][C]
    int8_t* y;            // [B][C]
    const int8_t* w;      // [C] score vector
    const int32_t* table; // softmax tables (form 0: int32 [2][256]; form 1: float32 [256])
    int P, C, fc_bias, fc_mult, fc_shift, fc_zo, form, zx, za, mul_mult, mul_shift, mul_zo, mul_lo, mul_hi, sum_mult, sum_shift, sum_zo;
};

__device__ __forceinline__ int32_t sat_shl(int32_t x, int e) {  // gemmlowp SaturatingRoundingMultiplyByPOT<e>, e > 0
    const long v = (long)x << e;
    return v > 2147483647L ? 2147483647 : (v < -2147483648L ? (int32_t)-2147483648L : (int32_t)v);
}
// 1 / (1 + f) for f in [0, 1) as Q0.31 (gemmlowp one_over_one_plus_x_for_x_in_0_1)
__device__ __forceinline__ int32_t one_over_one_plus(int32_t f) {
    const int32_t half_den = (int32_t)(((long)f + 2147483647L + 1) / 2);  // RoundingHalfSum(f, One()); the sum is >= 0
    int32_t est = 1515870810 + srdhm_ref(half_den, -1010580540);          // 48/17 - 32/17 d  (Q2.29)
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int32_t err = (1 << 29) - srdhm_ref(half_den, est);
        est = est + sat_shl(srdhm_ref(est, err), 2);
    }
    return sat_shl(est, 1);
}

__global__ __launch_bounds__(256) void i8_attnpool_kernel(AttnPool8Args a) {
    extern __shared__ __attribute__((aligned(16))) int8_t apx[];  // [P][C] map, then P score bytes, P attention bytes
    __shared__ int sh_max, sh_over, sh_inv;
    __shared__ float sh_finv;
    const int tid = threadIdx.x, P = a.P, C = a.C;
    const int8_t* src = a.x + (size_t)blockIdx.x * P * C;
    int8_t* sc = apx + (size_t)P * C;
    int8_t* at = sc + P;
    for (int i = tid; i < P * C / 4; i += 256) reinterpret_cast<int*>(apx)[i] = reinterpret_cast<const int*>(src)[i];
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        int acc = a.fc_bias;  // bias - zx * sum(w) folded by the packer
        for (int k = 0; k < C / 4; ++k) acc = __builtin_amdgcn_sdot4(reinterpret_cast<const int*>(apx + (size_t)p * C)[k], reinterpret_cast<const int*>(a.w)[k], acc, false);
        sc[p] = (int8_t)clampi(mbqm_ref(acc, a.fc_mult, a.fc_shift) + a.fc_zo, -128, 127);
    }
    __syncthreads();
    if (tid == 0) {
        int mx = -128;
        for (int p = 0; p < P; ++p) mx = max(mx, (int)sc[p]);
        sh_max = mx;
        if (a.form == 0) {
            int total = 0;
            for (int p = 0; p < P; ++p) total += a.table[256 + mx - sc[p]];
            const int lz = __clz(total);               // total >= 2^19: the maximum contributes exp(0) = 1.0
            sh_over = 12 - lz;
            sh_inv = one_over_one_plus((int32_t)(((uint32_t)total << lz) - 0x80000000u));
        } else {
            float total = 0.0f;
            for (int p = 0; p < P; ++p) total = __fadd_rn(total, __int_as_float(a.table[mx - sc[p]]));
            sh_finv = __fdiv_rn(1.0f, __fmul_rn(total, 1.0f / 256.0f));
        }
    }
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        const int d = sh_max - sc[p];
        int q;
        if (a.form == 0) {
            const int e = a.table[d];
            q = e < 0 ? -128 : clampi(rdivpot_ref(srdhm_ref(sh_inv, e), sh_over + 31 - 8) - 128, -128, 127);
        } else {
            q = clampi((int)roundf(__fmul_rn(__int_as_float(a.table[d]), sh_finv)) - 128, -128, 127);
        }
        at[p] = (int8_t)q;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        int total = 0;
        for (int p = 0; p < P; ++p) {
            const int v = clampi(mbqm_ref(((int)apx[(size_t)p * C + c] - a.zx) * ((int)at[p] - a.za), a.mul_mult, a.mul_shift) + a.mul_zo, a.mul_lo, a.mul_hi);
            total += v - a.mul_zo;
        }
        a.y[(size_t)blockIdx.x * C + c] = (int8_t)clampi(mbqm_ref(total, a.sum_mult, a.sum_shift) + a.sum_zo, -128, 127);
    }
}

// DEQUANTIZE -> float32 SOFTMAX over the classes of a chunk: one wave per chunk (scores), logits = the dequantised input
__global__ __launch_bounds__(64) void i8_head_softmax_kernel(const int8_t* __restrict__ x, float* __restrict__ scores, float* __restrict__ logits,
                                                             int C, int zp_fc, float s_fc, float beta) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int8_t* row = x + (size_t)b * C;
    float mx = -3.4e38f;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, (float)(row[c] - zp_fc) * s_fc);
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
    for (int c = lane; c < C; c += 64) sum += expf(((float)(row[c] - zp_fc) * s_fc - mx) * beta);
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    for (int c = lane; c < C; c += 64) {
        const float v = (float)(row[c] - zp_fc) * s_fc;
        if (logits) logits[(size_t)b * C + c] = v;
        scores[(size_t)b * C + c] = expf((v - mx) * beta) / sum;
    }
}

// LOGISTIC (table) + DEQUANTIZE -> float32 scores; dequantised FC output -> float32 logits
__global__ void i8_head_kernel(const int8_t* __restrict__ x, float* __restrict__ scores, float* __restrict__ logits,
                               int C, int zp_fc, int zp_out, float s_fc, float s_out, const int8_t* __restrict__ lut,
                               long total) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int32_t q = x[gid];
    if (logits) logits[gid] = (float)(q - zp_fc) * s_fc;
    if (lut) {
        const int32_t o = lut[q + 128];
        scores[gid] = (float)(o - zp_out) * s_out;
    } else {
        scores[gid] = (float)(q - zp_fc) * s_fc;
    }
}

inline dim3 grid1d(long total, int block) { return dim3((unsigned)((total + block - 1) / block)); }

}  // namespace

void launch_i8_quant(const float* spec, const float* minmax, int8_t* out, int B, int F, int W, int Kp, int zp,
                     int fill, float scale, hipStream_t s) {
    hipLaunchKernelGGL(i8_quant_kernel, dim3((W + 63) / 64, (Kp + 63) / 64, B), dim3(256), 0, s, spec, minmax, out, F,
                       W, Kp, zp, fill, scale);
}

void launch_i8_mel(const int8_t* x, int8_t* y, int B, int W, int Kp, int M, int zp_out, int amin, int amax,
                   const int8_t* w, const int32_t* bias, const int32_t* mult, const int32_t* shift, const int8_t* lut,
                   hipStream_t s) {
    hipLaunchKernelGGL(i8_mel_kernel, dim3((W + 255) / 256, M, B), dim3(256), 0, s, x, y, W, Kp, M, zp_out, amin, amax,
                       w, bias, mult, shift, lut);
}

void launch_i8_stem(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias,
                    const int32_t* mult, const int32_t* shift, hipStream_t s) {
    const long total = (long)B * g.OH * g.OW * (g.C / 4);
    hipLaunchKernelGGL(i8_stem_kernel, grid1d(total, 256), dim3(256), 0, s, x, y, g, w, bias, mult, shift, total);
}

void launch_i8_dw(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias,
                  const int32_t* mult, const int32_t* shift, hipStream_t s) {
    const long total = (long)B * g.OH * g.OW * (g.C / 4);
    hipLaunchKernelGGL(i8_dw_kernel, grid1d(total, 256), dim3(256), 0, s, x, y, g, w, bias, mult, shift, total);
}

void launch_i8_pw(const int8_t* x, const int8_t* res, int8_t* y, int B, int P, int Cin, int Cout, int zp_out, int amin,
                  int amax, const I8AddParams& add, const int8_t* w, const int32_t* bias, const int32_t* mult,
                  const int32_t* shift, hipStream_t s) {
    const long total = (long)B * P * (Cout / 4);
    hipLaunchKernelGGL(i8_pw_kernel, grid1d(total, 256), dim3(256), 0, s, x, res, y, Cin, Cout, zp_out, amin, amax, add,
                       w, bias, mult, shift, total);
}

void launch_i8_mean(const int8_t* x, int8_t* y, int B, int P, int C, int zp_in, int mult, int shift, int zp_out,
                    hipStream_t s) {
    hipLaunchKernelGGL(i8_mean_kernel, dim3(B), dim3(256), 0, s, x, y, P, C, zp_in, mult, shift, zp_out);
}

void launch_i8_fc(const int8_t* x, int8_t* y, int B, int Cin, int Cout, int zp_out, int amin, int amax, const int8_t* w,
                  const int32_t* bias, const int32_t* mult, const int32_t* shift, const int8_t* lut, hipStream_t s) {
    const int Kp = (Cin + 3) & ~3;
    hipLaunchKernelGGL(i8_fc_kernel, dim3((B + kFcChunks - 1) / kFcChunks), dim3(128), (size_t)kFcChunks * Kp, s, x, y, B, Cin, Kp, Cout, zp_out,
                       amin, amax, w, bias, mult, shift, lut);
}

void launch_i8_segate(const int8_t* x, int8_t* y, int B, int P, int C, int zp_in, int mean_mult, int mean_shift, int mean_zp, int R, int zo1, int amin1,
                      int amax1, const int8_t* w1, const int32_t* b1, const int32_t* m1, const int32_t* s1, const int8_t* lut1, int zo2, int amin2,
                      int amax2, const int8_t* w2, const int32_t* b2, const int32_t* m2, const int32_t* s2, const int8_t* lut2, hipStream_t s, int32_t* sums) {
    const int Kp1 = (C + 3) & ~3, Kp2 = (R + 3) & ~3;
    SeGate8Args a{x, y, P, C, zp_in, mean_mult, mean_shift, mean_zp, R, Kp1, zo1, amin1, amax1, w1, b1, m1, s1, lut1, Kp2, zo2, amin2, amax2, w2, b2, m2, s2, lut2, sums};
    hipLaunchKernelGGL(i8_segate_kernel, dim3(B), dim3(256), (size_t)(Kp1 + Kp2), s, a);
}

void launch_i8_scale(const int8_t* x, const int8_t* gate, int8_t* y, int B, int P, int C, int zx, int zg, int mult, int shift, int zo,
                     int amin, int amax, hipStream_t s) {
    const long total = (long)B * P * (C / 4);
    hipLaunchKernelGGL(i8_scale_kernel, grid1d(total, 256), dim3(256), 0, s, x, gate, y, P, C, zx, zg, mult, shift, zo, amin, amax, total);
}

void launch_i8_rawfe(const float* x, int8_t* y, int B, int T, int W, int M, int stride, int pad_left, float q_scale, int q_zp, int zp_out, int amin,
                     int amax, const int8_t* w, const int32_t* bias, const int32_t* mult, const int32_t* shift, const int8_t* lut, hipStream_t s) {
    hipLaunchKernelGGL(i8_rawfe_kernel, dim3((W + 255) / 256, B), dim3(256), 0, s, x, y, T, W, M, stride, pad_left, q_scale, q_zp, zp_out, amin, amax, w,
                       bias, mult, shift, lut);
}

void launch_i8_maxnorm(const int8_t* x, int8_t* y, int B, int C, int W, const int8_t* den_tab, const int8_t* div_tab, const int8_t* lut, hipStream_t s) {
    hipLaunchKernelGGL(i8_maxnorm_kernel, dim3(B), dim3(256), 0, s, x, y, C, W, den_tab, div_tab, lut);
}

bool launch_i8_attnpool(const int8_t* x, int8_t* y, int B, const int* p, const int8_t* w, const int32_t* table, hipStream_t s) {
    AttnPool8Args a{x, y, w, table, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15], p[16]};
    const size_t smem = (size_t)a.P * a.C + 2 * (size_t)a.P + 16;
    if (a.C % 4 || smem > 64 * 1024) return false;
    hipLaunchKernelGGL(i8_attnpool_kernel, dim3(B), dim3(256), smem, s, a);
    return true;
}

void launch_i8_head_softmax(const int8_t* x, float* scores, float* logits, int B, int C, int zp_fc, float s_fc, float beta, hipStream_t s) {
    hipLaunchKernelGGL(i8_head_softmax_kernel, dim3(B), dim3(64), 0, s, x, scores, logits, C, zp_fc, s_fc, beta);
}

void launch_i8_head(const int8_t* x, float* scores, float* logits, int B, int C, int zp_fc, int zp_out, float s_fc,
                    float s_out, const int8_t* lut, hipStream_t s) {
    const long total = (long)B * C;
    hipLaunchKernelGGL(i8_head_kernel, grid1d(total, 256), dim3(256), 0, s, x, scores, logits, C, zp_fc, zp_out, s_fc,
                       s_out, lut, total);
}

// Test hook (bn_debug_requant): the requantisation forms of bn_requant.h and of the strip kernels, one element per thread.
// mode 0: mbqm (what the generic kernels call), 1: mbqm_ref (literal gemmlowp definitions), 2: mbqm_right (branch-free right-shift form),
// 3: the strip kernels' folded form ((v + c1 + (v >> 31)) >> e, c1 = 2^(e-1) + (zp << e)) minus zp
__global__ void debug_requant_kernel(const int32_t* x, const int32_t* mult, const int32_t* shift, int n, int mode, int zp, int32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = x[i], m = mult[i];
    const int sh = shift[i];
    int32_t r;
    if (mode == 0) r = mbqm(v, m, sh);
    else if (mode == 1) r = mbqm_ref(v, m, sh);
    else if (mode == 2) r = mbqm_right(v, m, sh);
    else {
        const int e = -sh;
        const int32_t c1 = (1 << (e - 1)) + (zp << e);
        const int32_t hv = srdhm_pos(v, m);
        r = ((hv + c1 + (hv >> 31)) >> e) - zp;
    }
    out[i] = r;
}

void launch_debug_requant(const int32_t* x, const int32_t* mult, const int32_t* shift, int n, int mode, int zp, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(debug_requant_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, mult, shift, n, mode, zp, out);
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_i8() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&i8_quant_kernel));
}

}  // namespace bn
