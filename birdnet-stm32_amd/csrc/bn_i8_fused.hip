// bn_i8_fused.hip — bit-faithful INT8 depthwise-separable block as ONE kernel on gfx950:
//
//   [DEPTHWISE_CONV_2D 3x3 (int32 acc, per-channel requantisation, fused ReLU6 clamp)] -> int8 LDS tile
//   -> CONV_2D 1x1 on the int8 matrix cores (v_mfma_i32_16x16x64_i8) -> per-channel requantisation
//   [-> TFLite ADD with the residual] -> int8
//
// The same kernel without the depthwise stage is the frontend's mel mixer (CONV_2D 1x1 over the padded
// frequency axis + ReLU clamp + per-channel PWL table, output transposed to [M][W]).
//
// Integer semantics are the TFLite reference kernels' (see bn_i8.hip and oracle/int8_graph.py); the matrix
// cores only change the order in which exact int32 products are added, so results are bit-identical to the
// baseline kernels.  Zero points: the depthwise stage loads the input zero point for padded taps and the
// packer folds -zp*sum(w) into both biases.
//
// One 256-thread workgroup owns 64 output positions and a slice of the output channels:
//   phase 1: depthwise outputs [64][Cin] int8 (4 channels per work item) -> LDS
//   phase 2: A fragments = 16 consecutive channel bytes per lane (one ds_read_b128), B fragments from the
//            weights the packer stored in fragment order (1 KiB contiguous per wave-instruction)
//   epilogue: int32 accumulators -> LDS -> requantise 4 channels per thread -> packed dword stores.
#include <type_traits>
#include <stdlib.h>

#include "bn_kernels.h"
#include "bn_requant.h"
#include "bn_quant_in.h"
#include "bn_exact_dft.h"

namespace bn {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int32_t sx8(int32_t v, int byte) { return (int32_t)(int8_t)(v >> (8 * byte)); }

// One multiply-accumulate of channel e of a packed 4-channel dword without unpacking: v_dot4_i32_i8 against a weight dword
// whose other three bytes are zero (the reduction over the four bytes then has a single non-zero term).  Exact int32.
__device__ __forceinline__ int32_t lane_byte(int32_t w, int e) { return w & (0xff << (8 * e)); }
__device__ __forceinline__ int32_t dot4(int32_t a, int32_t b, int32_t c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// 4 x 4 byte transpose: rows r0..r3 hold four channels of one tap each; c[e] gets channel e of the four taps.
// v_perm_b32 picks bytes out of the 8-byte value {s0 (bytes 4-7), s1 (bytes 0-3)}.
__device__ __forceinline__ void transpose4x4(int r0, int r1, int r2, int r3, int c[4]) {
    const uint32_t a_lo = __builtin_amdgcn_perm((uint32_t)r1, (uint32_t)r0, 0x05010400u);  // r0.0 r1.0 r0.1 r1.1
    const uint32_t a_hi = __builtin_amdgcn_perm((uint32_t)r1, (uint32_t)r0, 0x07030602u);  // r0.2 r1.2 r0.3 r1.3
    const uint32_t b_lo = __builtin_amdgcn_perm((uint32_t)r3, (uint32_t)r2, 0x05010400u);
    const uint32_t b_hi = __builtin_amdgcn_perm((uint32_t)r3, (uint32_t)r2, 0x07030602u);
    c[0] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u);  // a_lo.01 b_lo.01 = r0.0 r1.0 r2.0 r3.0
    c[1] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
    c[2] = (int)__builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
    c[3] = (int)__builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u);
}

// Depthwise 3x3 weights of one channel quad, transposed once per thread: per channel the bytes of taps 0-3 and 4-7, and
// tap 8 left in its own lane of a dword.  dw9() then needs 16 byte-permutes + 12 dot4 for 36 multiply-accumulates.
struct DwTaps {
    int w03[4], w47[4], w8[4];
};
__device__ __forceinline__ DwTaps load_dw_taps(const int8_t* w, int stride) {  // tap t at w + t * stride (4 bytes = 4 channels)
    int r[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) r[t] = *reinterpret_cast<const int*>(w + t * stride);
    DwTaps d;
    transpose4x4(r[0], r[1], r[2], r[3], d.w03);
    transpose4x4(r[4], r[5], r[6], r[7], d.w47);
#pragma unroll
    for (int e = 0; e < 4; ++e) d.w8[e] = lane_byte(r[8], e);
    return d;
}
__device__ __forceinline__ void dw9(const int v9[9], const DwTaps& d, int acc[4]) {
    int c[4];
    transpose4x4(v9[0], v9[1], v9[2], v9[3], c);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = dot4(c[e], d.w03[e], acc[e]);
    transpose4x4(v9[4], v9[5], v9[6], v9[7], c);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = dot4(c[e], d.w47[e], dot4(v9[8], d.w8[e], acc[e]));
}

struct PosInfo8 {
    int in_base;   // byte offset of tap (0,0) in x
    int out_base;  // byte offset of channel 0 in y / res (transposed output: offset of (chunk, n = 0, t)), -1 = skip
    int mask;      // valid taps
    int pad;
};

constexpr int kKC8 = 512;  // contraction channels staged in LDS at a time

template <int RG, int CT, bool HAS_DW, bool TRANSPOSED>
__global__ __launch_bounds__(256) void i8_dwpw_kernel(DwPw8Args a) {
    extern __shared__ __attribute__((aligned(16))) int lds_raw[];
    __shared__ PosInfo8 pos[64];
    v4i* lds16 = reinterpret_cast<v4i*>(lds_raw);  // activation tile [64][kcp/16 + 1] x 16 bytes
    const int K = a.Cin, N = a.Cout;
    const int tid = threadIdx.x;
    const bool rq = (a.rq_right & 1) != 0;  // uniform: every requantisation of this operator is a pure right shift

    if (tid < 64) {
        const int tiles_x = a.OW / a.TW, tiles_y = a.OH / a.TH;
        int bid = xcd_tile(blockIdx.x, gridDim.x);
        const int tx0 = (bid % tiles_x) * a.TW;
        bid /= tiles_x;
        const int ty0 = (bid % tiles_y) * a.TH;
        const int chunk0 = (bid / tiles_y) * a.NB;
        const int tile_hw = a.TH * a.TW;
        const int nb = tid / tile_hw, rr = tid - nb * tile_hw;
        const int oh = ty0 + rr / a.TW, ow = tx0 + rr % a.TW;
        const int chunk = chunk0 + nb;
        PosInfo8 pi;
        pi.pad = 0;
        if (TRANSPOSED)
            pi.out_base = chunk < a.B ? chunk * N * (a.OH * a.OW) + oh * a.OW + ow : -1;
        else
            pi.out_base = chunk < a.B ? ((chunk * a.OH + oh) * a.OW + ow) * N : -1;
        const int ih0 = oh * a.sh - a.pt, iw0 = ow * a.sw - a.pl;
        pi.in_base = ((chunk * a.H + ih0) * a.W + iw0) * K;
        int mask = 0;
        if (chunk < a.B) {
            if (HAS_DW) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (ih0 + i >= 0 && ih0 + i < a.H && iw0 + j >= 0 && iw0 + j < a.W) mask |= 1 << (i * 3 + j);
            } else {
                mask = 1;
            }
        }
        pi.mask = mask;
        pos[tid] = pi;
    }
    // TFLite ADD rescales both int8 inputs before summing: two of its three requantisations depend only on ONE byte
    // (the residual value, the block's own output), so they are tabulated once per workgroup — 2 LDS reads instead of ~24
    // vector instructions per output element; the third (on the 32-bit sum) stays arithmetic.
    __shared__ int add_lut[2][256];
    if (!TRANSPOSED && a.add.enabled) {
        const int v = (int)(int8_t)tid;  // entry index = the byte pattern
        add_lut[0][tid] = mbqm((v - a.add.z1) * (1 << 20), a.add.m1, a.add.s1);
        add_lut[1][tid] = mbqm((v - a.pw_zp_out) * (1 << 20), a.add.m2, a.add.s2);
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    constexpr int WM = 4 / RG;
    const int wm = wave % WM, wn = wave / WM;
    const int row0 = wm * RG * 16;
    const int ct0 = blockIdx.y * (RG * CT) + wn * CT;
    const int n_ct = N >> 4;

    v4i acc[RG][CT];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[g][c] = (v4i){0, 0, 0, 0};
    const v4i* wp = reinterpret_cast<const v4i*>(a.pw_w);  // [Kp/64][N/16][64 lanes] x 16 bytes

    const int zp4 = (a.dw_zp_in & 0xff) * 0x01010101;  // four copies of the input zero point (padded taps)
    for (int k0 = 0; k0 < K; k0 += kKC8) {
        const int kc = (K - k0) < kKC8 ? (K - k0) : kKC8;
        const int kcp = (kc + 63) & ~63;  // padded to whole MFMA steps; the pad columns meet zero weights
        const int kq = kc >> 2;           // dwords (4 channels) per position
        const int S16 = (kcp >> 4) + 1;   // row stride in 16-byte units
        if (k0) __syncthreads();

        // ---- phase 1 -------------------------------------------------------------------------------------
        const bool fixed_cq = (256 % kq) == 0;
        const int cq_fixed = tid % kq;
        DwTaps taps;
        int bias4[4], mult4[4], shift4[4];
        auto load_consts = [&](int cq) {
            taps = load_dw_taps(a.dw_w + k0 + 4 * cq, K);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bias4[e] = a.dw_b[k0 + 4 * cq + e];
                mult4[e] = a.dw_mult[k0 + 4 * cq + e];
                shift4[e] = a.dw_shift[k0 + 4 * cq + e];
            }
        };
        if (HAS_DW && fixed_cq) load_consts(cq_fixed);
        int* lds32 = lds_raw;
        const bool copy16 = !HAS_DW && (K & 15) == 0 && (kc & 15) == 0;
        if (copy16) {
            // plain 1x1 convolution: the tile is a copy.  16 bytes per item, four items of a thread requested before the first is stored (one
            // dword per item, one memory round trip each: sixteen round trips per 256-channel slice and thread)
            const int k16 = kc >> 4;
            v4i* lds16 = reinterpret_cast<v4i*>(lds_raw);
            constexpr int NPF = 4;
            for (int item0 = tid; item0 < 64 * k16; item0 += 256 * NPF) {
                v4i v[NPF];
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const int item = item0 + 256 * u;
                    v[u] = (v4i){0, 0, 0, 0};
                    if (item < 64 * k16) {
                        const int p = item / k16, c16 = item - p * k16;
                        if (pos[p].mask) v[u] = *reinterpret_cast<const v4i*>(a.x + (long)pos[p].in_base + k0 + 16 * c16);
                    }
                }
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const int item = item0 + 256 * u;
                    if (item >= 64 * k16) break;
                    const int p = item / k16, c16 = item - p * k16;
                    lds16[p * S16 + c16] = v[u];
                }
            }
        }
        for (int item = tid; !copy16 && item < 64 * kq; item += 256) {
            const int p = item / kq;
            const int cq = fixed_cq ? cq_fixed : item - p * kq;
            const PosInfo8 pi = pos[p];
            int packed = 0;
            if (HAS_DW) {
                if (!fixed_cq) load_consts(cq);
                const int8_t* xin = a.x + (long)pi.in_base + k0 + 4 * cq;
                int v9[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int t = i * 3 + j;
                        v9[t] = (pi.mask >> t) & 1 ? *reinterpret_cast<const int*>(xin + (i * a.W + j) * K) : zp4;
                    }
                int s4[4] = {bias4[0], bias4[1], bias4[2], bias4[3]};  // bias already holds -zp_in * sum of the nine weights
                dw9(v9, taps, s4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qv = clampi(mbqm_u(s4[e], mult4[e], shift4[e], rq) + a.dw_zp_out, a.dw_amin, a.dw_amax);
                    packed |= (qv & 0xff) << (8 * e);
                }
            } else if (pi.mask) {
                packed = *reinterpret_cast<const int*>(a.x + (long)pi.in_base + k0 + 4 * cq);
            }
            lds32[p * S16 * 4 + cq] = packed;
        }
        __syncthreads();

        // ---- phase 2 -------------------------------------------------------------------------------------
        const int ksteps = kcp >> 6, s0 = k0 >> 6;
        v4i bf[CT], bnext[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) bf[c] = wp[((size_t)s0 * n_ct + ct0 + c) * 64 + lane];
        for (int s = 0; s < ksteps; ++s) {
            if (s + 1 < ksteps) {
#pragma unroll
                for (int c = 0; c < CT; ++c) bnext[c] = wp[((size_t)(s0 + s + 1) * n_ct + ct0 + c) * 64 + lane];
            }
            v4i af[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) af[g] = lds16[(row0 + 16 * g + r) * S16 + 4 * s + q];
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[g][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[g], bf[c], acc[g][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c) bf[c] = bnext[c];
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    constexpr int NS = RG * CT * 16;
    __syncthreads();
    const int n_base = blockIdx.y * NS;
    if (!TRANSPOSED) {
        constexpr int SO = NS + 4;  // int32 row stride of the [64][NS] accumulator tile
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) lds_raw[(row0 + 16 * g + 4 * q + reg) * SO + (wn * CT + c) * 16 + r] = acc[g][c][reg];
        __syncthreads();
        constexpr int Q4 = NS / 4;
        for (int item = tid; item < 64 * Q4; item += 256) {
            const int p = item / Q4, c4 = item - p * Q4;
            const int ob = pos[p].out_base;
            if (ob < 0) continue;
            const v4i v = *reinterpret_cast<const v4i*>(lds_raw + p * SO + 4 * c4);
            const int n0 = n_base + 4 * c4;
            const v4i b = *reinterpret_cast<const v4i*>(a.pw_b + n0);
            const v4i m = *reinterpret_cast<const v4i*>(a.pw_mult + n0);
            const v4i sh = *reinterpret_cast<const v4i*>(a.pw_shift + n0);
            int rv = 0;
            if (a.add.enabled) rv = *reinterpret_cast<const int*>(a.res + (long)ob + n0);
            int packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int qv = clampi(mbqm_u(v[e] + b[e], m[e], sh[e], rq) + a.pw_zp_out, a.pw_amin, a.pw_amax);
                if (a.add.enabled) {
                    const int sa = add_lut[0][(rv >> (8 * e)) & 0xff];
                    const int sb = add_lut[1][qv & 0xff];
                    qv = clampi(mbqm(sa + sb, a.add.mo, a.add.so) + a.add.zo, a.add.amin, a.add.amax);
                }
                packed |= (qv & 0xff) << (8 * e);
            }
            *reinterpret_cast<int*>(a.y + (long)ob + n0) = packed;
        }
    } else {
        // transposed output y[chunk][n][t]: stage as [NS][64 + 4] so that a thread packs 4 consecutive positions
        constexpr int ST = 64 + 4;
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < CT; ++c)
                *reinterpret_cast<v4i*>(lds_raw + ((wn * CT + c) * 16 + r) * ST + row0 + 16 * g + 4 * q) = acc[g][c];
        __syncthreads();
        const int hw = a.OH * a.OW;
        for (int item = tid; item < NS * 16; item += 256) {
            const int n = item >> 4, t4 = item & 15;  // 4 consecutive positions of output channel n
            const int ob = pos[4 * t4].out_base;      // tiles of a transposed launch are 64 consecutive positions of one chunk
            if (ob < 0) continue;
            const v4i v = *reinterpret_cast<const v4i*>(lds_raw + n * ST + 4 * t4);
            const int nn = n_base + n;
            const int b = a.pw_b[nn], m = a.pw_mult[nn], sh = a.pw_shift[nn];
            int packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int qv = clampi(mbqm_u(v[e] + b, m, sh, rq) + a.pw_zp_out, a.pw_amin, a.pw_amax);
                if (a.lut) qv = a.lut[nn * 256 + qv + 128];
                packed |= (qv & 0xff) << (8 * e);
            }
            *reinterpret_cast<int*>(a.y + (long)ob + (long)nn * hw) = packed;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// INT8 front block: frontend output [64][256] int8 -> CONV_2D 3x3 stem (stride 1x2, ReLU6) -> DEPTHWISE_CONV_2D 3x3
// stride 2 (ReLU6) -> CONV_2D 1x1 (ReLU6) in one kernel; the int8 stem activation (128 KB per chunk) stays in LDS.
// Padding follows TFLite: cells outside the stem's input contribute (zp - zp) = 0, cells outside the stem map hold the
// stem output zero point for the depthwise stage.  Bit-identical to the three separate kernels.
struct Front8Args {
    const int8_t* fe;        // [B][H0][W0]
    int8_t* y;               // [B][OH][OW][N]
    const int8_t* stem_w;    // [3][3][C]
    const int32_t* stem_b;   // [C]
    const int32_t* stem_mult;
    const int32_t* stem_shift;
    const int8_t* dw_w;      // [3][3][C]
    const int32_t* dw_b;     // [C], zero point folded
    const int32_t* dw_mult;
    const int32_t* dw_shift;
    const int8_t* pw_w;      // fragment order
    const int32_t* pw_b;     // zero point folded
    const int32_t* pw_mult;
    const int32_t* pw_shift;
    int B, H0, W0, SH, SW, N, OH, OW;
    int stem_zp_in, stem_zp_out, stem_amin, stem_amax, dw_zp_out, dw_amin, dw_amax, pw_zp_out, pw_amin, pw_amax;
    int rq_right;
};

__global__ __launch_bounds__(256) void i8_front_kernel(Front8Args a) {
    constexpr int TS = 17, FH = 19, FW = 35, C = 16, NS = 32;
    __shared__ int8_t fe_t[FH][FW + 1];
    __shared__ __attribute__((aligned(16))) int stem_t[TS * TS][C / 4];       // 4 channels per dword
    __shared__ __attribute__((aligned(16))) int tile[64 * (NS + 4)];          // A tile [64][64 + 16 bytes], later int32 accumulators
    const int tid = threadIdx.x;
    const bool rq = (a.rq_right & 1) != 0;
    int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int tiles_x = a.OW / 8, tiles_y = a.OH / 8;
    const int tx0 = (bid % tiles_x) * 8;
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * 8;
    const int chunk = bid / tiles_y;

    const int r_base = 2 * ty0 - 1, c_base = 4 * tx0;
    const int8_t* fe = a.fe + (size_t)chunk * a.H0 * a.W0;
    for (int i = tid; i < FH * FW; i += 256) {
        const int rr = i / FW, cc = i - rr * FW;
        const int gr = r_base + rr, gc = c_base + cc;
        fe_t[rr][cc] = (gr >= 0 && gr < a.H0 && gc >= 0 && gc < a.W0) ? fe[gr * a.W0 + gc] : (int8_t)a.stem_zp_in;
    }
    __syncthreads();

    {   // stem patch: thread = (stem position, channel quad)
        const int cq = tid & 3;
        int b4[4], m4[4], s4[4];
        int wt[4][3];  // per channel: taps 0-3, 4-7 and 8 as bytes of three dwords (the window is a 9-byte vector of ONE input channel)
#pragma unroll
        for (int e = 0; e < 4; ++e) wt[e][0] = wt[e][1] = wt[e][2] = 0;
        int wsum[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int w = *reinterpret_cast<const int*>(a.stem_w + t * C + 4 * cq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wt[e][t >> 2] |= ((w >> (8 * e)) & 0xff) << (8 * (t & 3));
                wsum[e] += sx8(w, e);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b4[e] = a.stem_b[4 * cq + e] - a.stem_zp_in * wsum[e];  // sum (x - zp) w = sum x w - zp sum w
            m4[e] = a.stem_mult[4 * cq + e];
            s4[e] = a.stem_shift[4 * cq + e];
        }
        const int zpo4 = (a.stem_zp_out & 0xff) * 0x01010101;
        for (int sp = tid >> 2; sp < TS * TS; sp += 64) {
            const int sr = sp / TS, sc = sp - sr * TS;
            int tb[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) tb[i * 3 + j] = (uint8_t)fe_t[sr + i][2 * sc + j];
            const int x0 = tb[0] | (tb[1] << 8) | (tb[2] << 16) | (tb[3] << 24);
            const int x1 = tb[4] | (tb[5] << 8) | (tb[6] << 16) | (tb[7] << 24);
            const int x2 = tb[8];
            int acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = dot4(x2, wt[e][2], dot4(x1, wt[e][1], dot4(x0, wt[e][0], b4[e])));
            int packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                packed |= (clampi(mbqm_u(acc[e], m4[e], s4[e], rq) + a.stem_zp_out, a.stem_amin, a.stem_amax) & 0xff) << (8 * e);
            const bool inside = (2 * ty0 + sr) < a.SH && (2 * tx0 + sc) < a.SW;
            stem_t[sp][cq] = inside ? packed : zpo4;
        }
    }
    __syncthreads();

    {   // depthwise stride 2: 64 positions x 4 channel quads = one item per thread
        const int cq = tid & 3, p = tid >> 2;
        const int py = p >> 3, px = p & 7;
        int acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = a.dw_b[4 * cq + e];
        const DwTaps taps = load_dw_taps(a.dw_w + 4 * cq, C);
        int v9[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) v9[i * 3 + j] = stem_t[(2 * py + i) * TS + 2 * px + j][cq];
        dw9(v9, taps, acc);
        int packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            packed |= (clampi(mbqm_u(acc[e], a.dw_mult[4 * cq + e], a.dw_shift[4 * cq + e], rq) + a.dw_zp_out, a.dw_amin, a.dw_amax) & 0xff) << (8 * e);
        tile[p * 20 + cq] = packed;  // row stride 80 bytes = 64 (one MFMA k-step) + 16; columns 16..63 meet zero weights
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 2 waves: 32 rows x 16 columns each
    const int row0 = wm * 32;
    const v4i* lds16 = reinterpret_cast<const v4i*>(tile);
    const v4i* wp = reinterpret_cast<const v4i*>(a.pw_w);
    v4i acc[2];
    const v4i bf = wp[(size_t)wn * 64 + lane];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        acc[g] = (v4i){0, 0, 0, 0};
        v4i af = lds16[(row0 + 16 * g + r) * 5 + q];
        if (q) af = (v4i){0, 0, 0, 0};  // only the first 16 of the 64 contraction bytes are real channels
        acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[g], 0, 0, 0);
    }
    constexpr int SO = NS + 4;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tile[(row0 + 16 * g + 4 * q + reg) * SO + wn * 16 + r] = acc[g][reg];
    __syncthreads();
    for (int item = tid; item < 64 * 8; item += 256) {
        const int p = item >> 3, c4 = item & 7;
        const int oh = ty0 + (p >> 3), ow = tx0 + (p & 7);
        const v4i v = *reinterpret_cast<const v4i*>(tile + p * SO + 4 * c4);
        const v4i b = *reinterpret_cast<const v4i*>(a.pw_b + 4 * c4);
        const v4i m = *reinterpret_cast<const v4i*>(a.pw_mult + 4 * c4);
        const v4i sh = *reinterpret_cast<const v4i*>(a.pw_shift + 4 * c4);
        int packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            packed |= (clampi(mbqm_u(v[e] + b[e], m[e], sh[e], rq) + a.pw_zp_out, a.pw_amin, a.pw_amax) & 0xff) << (8 * e);
        *reinterpret_cast<int*>(a.y + (((size_t)chunk * a.OH + oh) * a.OW + ow) * a.N + 4 * c4) = packed;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The frontend's mel mixer as its own kernel: CONV_2D 1x1 over the padded frequency axis (K = Kp, a multiple of 64) to 64 mel
// bins + ReLU clamp + per-channel PWL table, output transposed to [M][W].  Same results as i8_dwpw_kernel<.., TRANSPOSED>; the
// generic kernel spends its time on position tables and item loops, here the 64 x Kp activation tile of a workgroup is ONE
// contiguous 20 KB run of the [W][Kp] input (plain 16-byte copies into LDS), each wave owns 16 mel bins, and a lane requantises
// four consecutive frames of one bin (one dword store into the transposed output).
// QIN: QUANTIZE fused into the load — the input is the float32 spectrogram [B][qF][W] (frequency-major like the reference's
// array); a 64 x 64 block is read as float4 along the frames, normalised with the chunk's min / max (audio path), quantised
// with the same roundf(v / scale) + zp as i8_quant_kernel and written into the activation tile transposed.
// MODE (tile-major audio path only, bn_stft_exact.hip): 1 = also list every element whose byte could differ from the reference's
// within the STFT's error bound (the quantiser is monotone, so the test is the distance of its argument to the next rounding
// boundary: 4 instructions per element); 2 = run only the (chunk, block) pairs of the work list (the blocks whose bytes the
// float64 pass changed), nothing is listed.
constexpr int kMelFlagCap = 1022;  // flagged elements a workgroup keeps in LDS before it hands them to the chunk's list
// BN_TAIL_STAMPS (the measurement build lib/libbirdnet_hip_stamps.so, tools/mel_stamps.py; never defined in the production library): the waves of
// kMelStampWg workgroups from the MIDDLE of the guarded mixer's grid record when they started, had quantised their block, left the barrier behind the
// tile, finished the float64 settle, the matrix phase and the epilogue (s_memrealtime, 10 ns ticks).
#ifdef BN_TAIL_STAMPS
__device__ long long* g_mel_stamps = nullptr;   // [kMelStampWg][4 waves][8]
constexpr int kMelStampWg = 1024;
#define BN_MSTAMP(i) do { if ((MODE == 1 || MODE == 3) && mstamp) mst[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BN_MSTAMP(i) do {} while (0)
#endif

// MODE 3 = MODE 1 + the audit (option stft_audit): elements that are NOT in doubt but lie within kAuditBands bounds of a rounding boundary are listed too
// (bit 31 of the entry), re-evaluated with the flagged ones, and only COMPARED: a kept byte that differs from the exact one is a violation of the bound.
constexpr float kAuditBands = 4.0f;
template <bool QIN, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 3 : 4, 8))) void i8_mel_mfma_kernel(DwPw8Args a) {
    constexpr bool GUARDED = MODE == 1 || MODE == 3, AUDIT = MODE == 3;
#ifdef BN_TAIL_STAMPS
    long long mst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int mslot = (int)blockIdx.x - (int)gridDim.x / 2;
    const bool mstamp = GUARDED && g_mel_stamps && mslot >= 0 && mslot < kMelStampWg;
#endif
    BN_MSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) int lds_raw[];
    __shared__ int flag_n, flags[GUARDED ? kMelFlagCap : 1];
    __shared__ int audit_n, audit_bad, undecided;
    __shared__ std::conditional_t<GUARDED, ExactTabsW, int> xtabs_s;  // float64 twiddles + window for the elements this workgroup re-evaluates itself
    v4i* lds16 = reinterpret_cast<v4i*>(lds_raw);
    const int Kp = a.Cin, W = a.W, M = a.Cout;
    const int S16 = (Kp >> 4) + 1;  // row stride in 16-byte units (one unit of padding: conflict-free 16-byte reads along rows)
    const int tid = threadIdx.x;
    const int tiles_x = W >> 6;
    const int n_items = MODE == 2 ? *a.qguard.n_work : 1;
    for (int item = MODE == 2 ? blockIdx.x : 0; item < n_items; item += MODE == 2 ? gridDim.x : 1) {
    const int bid = MODE == 2 ? a.qguard.work[item] : xcd_tile(blockIdx.x, gridDim.x);
    const int chunk = bid / tiles_x, t0 = (bid - chunk * tiles_x) << 6;
    if (GUARDED && tid == 0) {
        flag_n = 0;
        audit_n = audit_bad = undecided = 0;
    }
    if (MODE != 0) __syncthreads();  // (MODE 2: the previous item's tile has been consumed)
    if constexpr (!QIN) {
        const v4i* src = reinterpret_cast<const v4i*>(a.x + ((size_t)chunk * W + t0) * Kp);
        const int per_row = Kp >> 4;
        for (int i = tid; i < 64 * per_row; i += 256) {
            const int row = i / per_row, c = i - row * per_row;
            lds16[row * S16 + c] = src[i];
        }
    } else {
        int8_t* tile = reinterpret_cast<int8_t*>(lds_raw);
        const int stride = S16 * 16;
        const float* S = a.qx + (size_t)chunk * a.qF * W + t0;
        QuantIn qi;
        qi.set(a.qminmax ? a.qminmax + 2 * chunk : nullptr, a.qscale, a.qzp);
        const bool renorm = qi.renorm;
        if (a.qtiled) {
            // tile-major spectrogram [W/16][qF][16]: wave wv reads ITS 16-frame block as 1 KB runs (16 frequency rows x 64 bytes);
            // lane = (row fr within the group of 16, frame quad ft).  Measured alternatives, all slower or equal:
            // dword loads per frame with one ds_write_b32 per four frequencies (0.288 ms), batches of five loads at eight waves
            // per SIMD (0.273 ms) against this form's 0.262 ms — the kernel sits at ~0.55 of the HBM peak on its 1.08 GB read.
            const int lane = tid & 63, wv = tid >> 6;
            const int ft = lane & 3, fr = lane >> 2;
            const float* Sb = a.qx + (size_t)chunk * a.qF * W + (size_t)(t0 / 16 + wv) * a.qF * 16 + 4 * ft;
            // half-width of the band around a rounding boundary inside which the reference's byte may differ (bn_quant_in.h)
            float dband[4] = {0.f, 0.f, 0.f, 0.f}, aband[4] = {0.f, 0.f, 0.f, 0.f}, crel = 0.f, inv_step = 0.f;
            if (GUARDED) {
                inv_step = (float)(1.0 / ((double)qi.rng * (double)qi.scale));
                crel = kBandRel * (qi.y_rng * qi.y_scale * 1.000001f);
                const float4 e = *reinterpret_cast<const float4*>(a.qguard.eps + (size_t)chunk * W + t0 + 16 * wv + 4 * ft);
                const float dsc = qi.y_rng * qi.y_scale * 1.000001f;
                // The chunk's minimum may be known as an interval only (stft_minmax_exact_kernel: qi.mn is its upper end, mn_lo the lower one).  Moving
                // the minimum by d moves a quantiser argument by at most d / (range scale) through the difference S - min and by as much again through
                // the range (v <= 255): the band grows by 2 d / (range scale), evaluated with the smaller range of the upper end (docs/exactness.md).
                const float mlo = a.qguard.mn_lo[chunk];
                const float mterm = mlo >= 0.0f ? 2.0f * (qi.mn - mlo) * dsc : 0.0f;
                // A bound of 0 = the frame's values are exact (zeros, or a chunk recomputed as a whole in float64).  Its band is NOT empty: the
                // kept bytes come from the folded multiply-add, whose own error (kQuantSlackFolded + the S'-proportional term) can cross a
                // rounding boundary — those elements are listed like any other and get the exact chain (found by tools/exact_soak.py: with
                // "nothing to list" for exact frames 304 of 245 812 whole-float64 chunks ended with different scores).
                const float slack = kQuantSlackFolded * a.qguard.slack_scale + mterm;
                dband[0] = 0.5f - (e.x * dsc + slack);
                dband[1] = 0.5f - (e.y * dsc + slack);
                dband[2] = 0.5f - (e.z * dsc + slack);
                dband[3] = 0.5f - (e.w * dsc + slack);
                if (AUDIT) {   // the audit's band: kAuditBands times as wide
                    const float as = a.qguard.audit_scale * dsc;
                    aband[0] = 0.5f - (kAuditBands * (e.x * as + kQuantSlackFolded) + mterm);
                    aband[1] = 0.5f - (kAuditBands * (e.y * as + kQuantSlackFolded) + mterm);
                    aband[2] = 0.5f - (kAuditBands * (e.z * as + kQuantSlackFolded) + mterm);
                    aband[3] = 0.5f - (kAuditBands * (e.w * as + kQuantSlackFolded) + mterm);
                }
            }
            // renormalisation and the zero-point fast path are wave-uniform: picked once, outside the per-element code
            int8_t* trow[4];  // the lane's four frames of the tile, at its frequency row
#pragma unroll
            for (int k = 0; k < 4; ++k) trow[k] = tile + (16 * wv + 4 * ft + k) * stride + fr;
            auto run = [&](auto RN, auto FAST) {
                constexpr int kBatch = 20;  // Kp <= 320 (257 bins padded to 320): every load of the wave's block is issued before the first use
                for (int f0 = 0; f0 < Kp; f0 += 16 * kBatch) {
                    float4 v[kBatch];
#pragma unroll
                    for (int i = 0; i < kBatch; ++i) {
                        const int f = f0 + 16 * i + fr;
                        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (f < a.qF) v[i] = *reinterpret_cast<const float4*>(Sb + (size_t)f * 16);
                    }
#pragma unroll
                    for (int i = 0; i < kBatch; ++i) {
                        const int f = f0 + 16 * i + fr;
                        if (f0 + 16 * i >= Kp) break;
                        const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                        if constexpr (GUARDED && RN.value && FAST.value) {
                            // Guarded form: every byte that is not provably the reference's is re-evaluated from the exact S below, so the
                            // bytes kept here only have to be right OUTSIDE the band — one multiply-add by RN(1 / (range scale)) with the zero
                            // point folded in stands for the two divisions (its error, u v + 128 u, is part of the band: bn_quant_in.h), and
                            // the four tests of a load share one branch.  Rows past the last frequency get the FILL byte behind the loop.
                            float s[4], sa[4] = {-1.f, -1.f, -1.f, -1.f};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float x = __builtin_fmaf(e[k] - qi.mn, inv_step, -128.0f);
                                const float xr = __builtin_rintf(x);  // (nearest-even instead of half-away: they differ on ties only, and a tie is in doubt)
                                trow[k][f0 + 16 * i] = (int8_t)min(max((int)xr, -128), 127);
                                s[k] = __builtin_fmaf(e[k], crel, fabsf(x - xr)) - dband[k];  // |x - round(x)| >= 1/2 - band: within the band of a boundary
                                if (AUDIT) sa[k] = __builtin_fmaf(e[k], kAuditBands * crel, fabsf(x - xr)) - aband[k];
                            }
                            if (fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])) >= 0.0f || (AUDIT && fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3])) >= 0.0f)) {
                                if (f < a.qF) {
#pragma unroll
                                    for (int k = 0; k < 4; ++k)
                                        if (s[k] >= 0.0f || (AUDIT && sa[k] >= 0.0f)) {
                                            const int sl = atomicAdd(&flag_n, 1);
                                            // (a near miss of the audit carries bit 31: re-evaluated and compared, never replaced)
                                            if (sl < kMelFlagCap) flags[sl] = ((t0 + 16 * wv + 4 * ft + k) << 16) | f | (s[k] >= 0.0f ? 0 : (int)0x80000000u);
                                        }
                                }
                            }
                            continue;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float x = e[k];
                            if (RN.value) x = div_by_const(x - qi.mn, qi.rng, qi.y_rng);
                            x = div_by_const(x, qi.scale, qi.y_scale);
                            const int q = FAST.value ? quantise_i8_zp128(x) : quantise_i8_any(x, a.qzp);
                            tile[(16 * wv + 4 * ft + k) * stride + f] = (int8_t)(f < a.qF ? q : a.qfill);
                            if (GUARDED) {
                                // in doubt: distance of the quantiser's argument to the next rounding boundary <= what eps(S') = eps_f + kGuardRel S' moves it
                                const float tt = x + 0.5f;
                                if (__builtin_fmaf(e[k], crel, fabsf(__builtin_amdgcn_fractf(tt) - 0.5f)) >= dband[k] && f < a.qF) {
                                    const int sl = atomicAdd(&flag_n, 1);
                                    if (sl < kMelFlagCap) flags[sl] = ((t0 + 16 * wv + 4 * ft + k) << 16) | f;
                                }
                            }
                        }
                    }
                }
            };
            using T = std::true_type;
            using F = std::false_type;
            if constexpr (GUARDED) {
                run(T{}, T{});  // the guarded form exists for renormalised input with zero point -128 only (bn_api.hip: guard_form_ok)
                for (int f = fr + ((a.qF - fr + 15) & ~15); f < Kp; f += 16) {  // padded frequency rows: the graph's FILL constant (same lane, same
#pragma unroll                                                                 // addresses as the stores above: LDS keeps a wave's order)
                    for (int k = 0; k < 4; ++k) trow[k][f - fr] = (int8_t)a.qfill;
                }
            } else if (a.qzp == -128) {
                if (renorm) run(T{}, T{}); else run(F{}, T{});
            } else {
                if (renorm) run(T{}, F{}); else run(F{}, F{});
            }
        } else {
        const int c4 = tid & 15, r4 = tid >> 4;
        for (int f0 = 0; f0 < Kp; f0 += 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = f0 + r4 + 16 * i;
                int q[4] = {a.qfill, a.qfill, a.qfill, a.qfill};  // padded frequency columns: the graph's FILL constant
                if (f < a.qF) {
                    const float4 v = *reinterpret_cast<const float4*>(S + (size_t)f * W + 4 * c4);
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[k] = qi.q(e[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) tile[(4 * c4 + k) * stride + f] = (int8_t)q[k];
            }
        }
        }
    }
    BN_MSTAMP(1);
    __syncthreads();
    BN_MSTAMP(2);
    if constexpr (GUARDED) {
        // The elements in doubt are settled HERE, before the tile is multiplied: a 16-lane row per element re-evaluates it the reference's way
        // (float64 DFT over the frame's 512 samples, complex64, numpy's |.|: bn_exact_dft.h), the byte in the tile and the value in the
        // spectrogram are replaced.  About 11 elements per workgroup (6.5e-4 of its 16 448): one round of 16.  (As a separate kernel over
        // per-chunk lists this cost 0.09 ms per 4096 chunks for the pass itself plus a second run of the mixer over every block with a changed
        // byte; here the sample reads overlap the other workgroups' spectrogram reads.)  A workgroup that flags more than it can keep — no
        // sane audio does — puts the chunk on the give-up list: stft512_f64_list_kernel then takes the whole chunk in float64 and mode 2 redoes its blocks.
        const int nf = flag_n;
        const int keep = min(max(a.qguard.flag_cap, 0), kMelFlagCap);
        if (tid == 0 && nf) {
            atomicAdd(a.qguard.count + chunk, nf > keep ? a.qguard.cap + 1 : nf);  // (statistics; beyond `cap`: given up)
            if (nf > keep) {
                // more in doubt than this workgroup keeps (no sane audio): the chunk goes to stft512_f64_list_kernel as a whole and ALL its blocks
                // through the mixer again (the first of the chunk's workgroups to get here lists it)
                const int all = tiles_x >= 32 ? -1 : (1 << tiles_x) - 1;
                if (atomicOr(a.qguard.dirty + chunk, all) == 0) {
                    a.qguard.hard[a.qguard.hard_cap + atomicAdd(a.qguard.n_hard + 1, 1)] = chunk;
                    for (int i = 0; i < tiles_x; ++i) a.qguard.work[atomicAdd(a.qguard.n_work, 1)] = chunk * tiles_x + i;
                }
            }
        }
        if (nf > 0 && nf <= keep) {
            // (twiddles and window staged into LDS, 12 KB behind one barrier: gathered straight from the table in L1 / L2 the kernel took 0.41 instead of 0.36 ms)
            ExactTabsW& xt = reinterpret_cast<ExactTabsW&>(xtabs_s);
            stage_tabs(xt, a.qguard.tabs);
            __syncthreads();
            int8_t* tile = reinterpret_cast<int8_t*>(lds_raw);
            const int stride = S16 * 16;
            QuantIn qi;
            qi.set(a.qminmax + 2 * chunk, a.qscale, a.qzp);
            // interval minimum: the reference's chain v = RN(RN(RN(S - min) / range) / scale), range = RN(RN(max - min) + 1e-10), is made of float32
            // operations that are each MONOTONE in their operands.  For every float32 minimum inside [lo, hi] the difference lies between the
            // differences at the two ends and so does the range, hence the quotient lies between the quotients of the four (difference, range)
            // corner pairs, and so on down the chain: an element whose byte is the same for the smallest and the largest corner value has that byte
            // whatever the minimum is — no rounding margin involved.  One that has not makes the workgroup hand the chunk over like one with too
            // many elements in doubt (stft512_f64_list_kernel finds the exact minimum, mode 2 redoes the chunk's blocks).
            const float mlo = a.qguard.mn_lo[chunk];
            const bool mint = mlo >= 0.0f;
            QuantIn qil = qi;
            if (mint) {
                const float ends[2] = {mlo, a.qminmax[2 * chunk + 1]};
                qil.set(ends, a.qscale, a.qzp);
            }
            auto decide = [&](float ex, int& q) {
                if (!mint) {
                    q = qi.q(ex);
                    return true;
                }
                const float th = ex - qi.mn, tl = ex - qil.mn;   // (qi: the interval's upper end = the smaller difference and range)
                const float c0 = div_by_const(div_by_const(th, qil.rng, qil.y_rng), qi.scale, qi.y_scale);
                const float c1 = div_by_const(div_by_const(tl, qi.rng, qi.y_rng), qi.scale, qi.y_scale);
                const float c2 = div_by_const(div_by_const(th, qi.rng, qi.y_rng), qi.scale, qi.y_scale);
                const float c3 = div_by_const(div_by_const(tl, qil.rng, qil.y_rng), qi.scale, qi.y_scale);
                q = quantise_i8(fminf(fminf(c0, c1), fminf(c2, c3)), a.qzp);
                return q == quantise_i8(fmaxf(fmaxf(c0, c1), fmaxf(c2, c3)), a.qzp);
            };
            const float* x = a.qguard.audio + (size_t)chunk * a.qguard.T;
            float* Sc = const_cast<float*>(a.qx) + (size_t)chunk * a.qF * W;
            const LdsWindow lw{xt.hann};
            const int grp = tid >> 4, gl = tid & 15;
            for (int i0 = 0; i0 < nf; i0 += 16) {
                const bool act = i0 + grp < nf;
                const int e = act ? flags[i0 + grp] : 0;
                const bool near_miss = AUDIT && e < 0;
                const int t = (e & 0x7fffffff) >> 16, f = e & 0xffff;
                const float ex = exact_mag_row(xt, lw, x, a.qguard.T, a.qguard.hop, t, f);
                if (act && gl == 0) {
                    int qex;
                    const bool decided = decide(ex, qex);
                    if (near_miss) {   // not in doubt by the bound: the kept byte must already be the exact one
                        if (decided) {
                            atomicAdd(&audit_n, 1);
                            if (tile[(t - t0) * stride + f] != (int8_t)qex) atomicAdd(&audit_bad, 1);
                        }
                    } else {
                        if (!decided) undecided = 1;   // (a benign race: every writer stores 1)
                        tile[(t - t0) * stride + f] = (int8_t)qex;
                        Sc[(size_t)(t / 16) * a.qF * 16 + (size_t)f * 16 + (t % 16)] = ex;  // (tile-major; keeps bn_debug_input_bytes' view consistent)
                    }
                }
            }
            __syncthreads();
            if (tid == 0 && undecided) {   // an element whose byte depends on where in its interval the minimum lies: the chunk as a whole in float64
                atomicAdd(a.qguard.count + chunk, a.qguard.cap + 1);
                const int all = tiles_x >= 32 ? -1 : (1 << tiles_x) - 1;
                if (atomicOr(a.qguard.dirty + chunk, all) == 0) {
                    a.qguard.hard[a.qguard.hard_cap + atomicAdd(a.qguard.n_hard + 1, 1)] = chunk;
                    for (int i = 0; i < tiles_x; ++i) a.qguard.work[atomicAdd(a.qguard.n_work, 1)] = chunk * tiles_x + i;
                }
            }
            if (AUDIT && tid == 0 && a.qguard.audit && audit_n) {
                atomicAdd(a.qguard.audit, audit_n);
                if (audit_bad) atomicAdd(a.qguard.audit + 1, audit_bad);
            }
        }
    }
    BN_MSTAMP(3);
    const int lane = tid & 63, wv = tid >> 6;  // wave wv: mel bins 16 wv .. 16 wv + 15
    const int r = lane & 15, q = lane >> 4;
    const v4i* wp = reinterpret_cast<const v4i*>(a.pw_w);  // [Kp/64][M/16][64 lanes] x 16 bytes
    const int n_ct = M >> 4;
    v4i acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (v4i){0, 0, 0, 0};
    const int ksteps = Kp >> 6;
    // every k-step's B fragment (the wave's 16 mel bins) and the per-bin constants requested TOGETHER up front: fetched one step ahead inside the loop they
    // were five dependent L2 round trips per workgroup (four matrix instructions per k-step hide ~64 of a round trip's ~700 cycles).  Not before the
    // barrier above: 23 more live registers across the float64 settle spilled.
    constexpr int kPreK = 5;  // k-steps kept in registers (Kp <= 320: the 257-bin spectrogram padded to 320)
    v4i bpre[kPreK];
#pragma unroll
    for (int s = 0; s < kPreK; ++s) bpre[s] = wp[((size_t)(s < ksteps ? s : 0) * n_ct + wv) * 64 + lane];
    const int cpre[3] = {a.pw_b[16 * wv + r], a.pw_mult[16 * wv + r], a.pw_shift[16 * wv + r]};
    if (ksteps <= kPreK) {
#pragma unroll
        for (int s = 0; s < kPreK; ++s) {
            if (s < ksteps) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16[(16 * g + r) * S16 + 4 * s + q], bpre[s], acc[g], 0, 0, 0);
            }
        }
    } else {
        v4i bf = wp[((size_t)0 * n_ct + wv) * 64 + lane];
        for (int s = 0; s < ksteps; ++s) {
            v4i bnext = bf;
            if (s + 1 < ksteps) bnext = wp[((size_t)(s + 1) * n_ct + wv) * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(lds16[(16 * g + r) * S16 + 4 * s + q], bf, acc[g], 0, 0, 0);
            bf = bnext;
        }
    }
#ifdef BN_TAIL_STAMPS
    asm volatile("" :: "v"(acc[0]), "v"(acc[3]) : "memory");  // (the stamp sits behind the matrix results)
#endif
    BN_MSTAMP(4);
    // lane (n = r, q): accumulator register reg of row group g = frame t0 + 16 g + 4 q + reg of mel bin 16 wv + r
    const int mel = 16 * wv + r;
    const int b = cpre[0], m = cpre[1], sh = cpre[2];
    const bool rq = (a.rq_right & 1) != 0;
    int8_t* yrow = a.y + ((size_t)chunk * M + mel) * W + t0 + 4 * q;
    // all sixteen values first, then all sixteen table reads in flight together (with the lookup inside the loop — a branch around a load per value —
    // the gathers went out one round trip at a time: 4.3 of a wave's 19 us, tools/mel_stamps.py)
    int qv[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[g][e] = clampi(mbqm_u(acc[g][e] + b, m, sh, rq) + a.pw_zp_out, a.pw_amin, a.pw_amax);
    if (a.lut) {  // (uniform)
        const int8_t* lrow = a.lut + mel * 256 + 128;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[g][e] = lrow[qv[g][e]];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<int*>(yrow + 16 * g) = pack4(qv[g]);
    BN_MSTAMP(5);
#ifdef BN_TAIL_STAMPS
    if (mstamp && (tid & 63) == 0) {
        long long* o = g_mel_stamps + ((size_t)mslot * 4 + (tid >> 6)) * 8;
        for (int i = 0; i < 6; ++i) o[i] = mst[i];
        o[6] = flag_n;
    }
#endif
    }
}

// Plain 1x1 convolution (the expand / project convolutions of exported inverted-residual graphs, Cin up to 256) without LDS tiles and
// workgroup barriers in the data path: a wave owns 16 consecutive positions.  The activations are the B operand as they lie in memory
// (lane (position r, k quarter q) loads the 16 channel bytes 64 s + 16 q .. of its position with one 16-byte load; bytes past Cin
// meet zero weights), the weight fragments the packer already writes for the tile kernel are the A operand, so the accumulators of
// lane (r, q) are the four CONSECUTIVE output channels 16 ct + 4 q .. of position r: requantise, [ADD], one dword store per tile.
// Per-channel constants are staged in LDS once per workgroup.
// MODE picks the requantisation forms at compile time (the launcher checks the conditions; all three are bit-identical where they apply):
//   0  any multiplier / shift, any gate (runtime-uniform choices);
//   1  linear layers (projections): every multiplier >= 0 and every shift < 0 — mbqm_right; the ADD's output rescale likewise; a gate, if
//      present, has both zero points -128 and a right shift 1..20 (one multiply-add per byte);
//   2  ReLU layers (expansions; no ADD): the clamp starts at the zero point and every shift lies in [-20, -1] — the sign-free
//      clamp(hi32(acc m + C) >> (e - 1)) of i8_pw_lds_kernel, three instructions per output; gate as in 1.
// TAB (projections with a residual ADD, MODE 1): the whole ADD is ONE lookup in the packer's 64 KB table [residual byte][own value + 128] held in LDS
// (a v_perm for the index + a byte read instead of two rescale lookups and a third requantisation: 46 -> ~9 issue cycles per output).  The table
// fills a CU's LDS budget for one workgroup, so that one has sixteen waves (1024 threads).
template <bool ADD, int NCT, int KS, int MODE, bool TAB>  // NCT > 0: the tile / k-step counts are compile-time constants and the weight fragments live in registers
__global__ __launch_bounds__(TAB ? 1024 : 256) void i8_pw_wave_kernel(DwPw8Args a, long n_pos, int lds_w16) {
    constexpr bool AREG = NCT > 0;
    constexpr int CW = MODE == 2 ? 5 : 3;  // v4i per channel quad: bias, multiplier, shift [, addend low, addend high]
    constexpr int NTH = TAB ? 1024 : 256, NWV = NTH / 64;
    extern __shared__ __attribute__((aligned(16))) int lds_raw[];
    const unsigned char* add_tab = reinterpret_cast<const unsigned char*>(lds_raw);  // (TAB) first 64 KB
    v4i* cst = reinterpret_cast<v4i*>(lds_raw + (TAB ? 16384 : 0));  // [Cout / 4][CW]
    __shared__ int add_lut[2][256];
    const int tid = threadIdx.x;
    const int K = a.Cin, N = a.Cout;
    const bool rq = (a.rq_right & 1) != 0;
    if constexpr (TAB) {
        const v4i* tsrc = reinterpret_cast<const v4i*>(a.add_tab);
        v4i* tdst = reinterpret_cast<v4i*>(lds_raw);
        v4i tv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) tv[k] = tsrc[tid + NTH * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) tdst[tid + NTH * k] = tv[k];
    }
    for (int i = tid; i < N / 4; i += NTH) {
        cst[CW * i + 0] = *reinterpret_cast<const v4i*>(a.pw_b + 4 * i);
        cst[CW * i + 1] = *reinterpret_cast<const v4i*>(a.pw_mult + 4 * i);
        v4i ss = *reinterpret_cast<const v4i*>(a.pw_shift + 4 * i);
        if constexpr (MODE == 2) {
            v4i clo, chi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ex = -ss[e];  // 1 .. 20 (checked at load)
                const long long C = (1ll << 30) + (((1ll << (ex - 1)) + (long long)a.pw_zp_out * (1ll << ex)) << 31);
                clo[e] = (int)(unsigned)(C & 0xffffffffll);
                chi[e] = (int)(C >> 32);
                ss[e] = ex - 1;
            }
            cst[CW * i + 3] = clo;
            cst[CW * i + 4] = chi;
        }
        cst[CW * i + 2] = ss;
    }
    if (ADD && !TAB && tid < 256) {
        const int v = (int)(int8_t)tid;
        add_lut[0][tid] = mbqm((v - a.add.z1) * (1 << 20), a.add.m1, a.add.s1);
        add_lut[1][tid] = mbqm((v - a.pw_zp_out) * (1 << 20), a.add.m2, a.add.s2);
    }
    // layers with more fragments than registers: the whole weight matrix (fragment order, up to 48 KB) is staged in LDS once per
    // workgroup when the launcher made room for it — a 16-byte LDS read per tile and k-step instead of a trip to L1 / L2
    v4i* wl = cst + CW * (N / 4);
    // ... in CONSUMPTION order: entry (ct KS + s) 64 + lane is the A fragment lane (row rr, k quarter qq) feeds to tile ct in k-step s — the
    // walk reads them with one running address, no index arithmetic per matrix instruction
    const bool w_in_lds = !AREG && lds_w16 > 0;
    if (w_in_lds) {
        const v4i* wsrc = reinterpret_cast<const v4i*>(a.pw_w);
        const int n_ct_all = N >> 4, cpl0 = N >> 2;
        for (int i = tid; i < lds_w16; i += NTH) {
            const int ln = i & 63, f = i >> 6, s_ = f % KS, ct = f / KS;
            const int rr = ln & 15, qq = ln >> 4;
            const int ch = cpl0 * (rr >> 2) + (rr & 3) + 4 * ct;
            wl[i] = wsrc[((size_t)s_ * n_ct_all + (ch >> 4)) * 64 + qq * 16 + (ch & 15)];
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x), 0, (int)(n_pos * K), 0x00020000);
    constexpr int ksteps = KS;
    const int n_ct = AREG ? NCT : N >> 4;
    const v4i* wp = reinterpret_cast<const v4i*>(a.pw_w);  // [Kp/64][N/16][64 lanes] x 16 bytes
    // a wave walks over groups of 16 positions (the constants above are staged once per workgroup, not once per 64 positions); the
    // activations of the next group are requested before the current one is multiplied
    // (each wave owns a CONTIGUOUS run of groups: it stays inside one chunk for P / 16 groups, so the squeeze-excite gate bytes are loaded once per
    // chunk and wave, not once per group)
    // (32-bit arithmetic throughout: the launcher checked n_pos * Cin and n_pos * Cout < 2^31)
    const int n_groups = (int)(n_pos / 16), n_walkers = (int)gridDim.x * NWV, per_wave = (n_groups + n_walkers - 1) / n_walkers;
    auto fetch = [&](int grp, v4i (&dst)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int koff = 64 * s + 16 * q;
            dst[s] = (v4i){0, 0, 0, 0};
            if (grp < n_groups && s < ksteps && koff < K)
                dst[s] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (grp * 16 + r) * K + koff, 0, 0));
        }
    };
    int grp = ((int)blockIdx.x * NWV + wave) * per_wave;
    const int g_end = grp + per_wave < n_groups ? grp + per_wave : n_groups;
    v4i bfr[4], bnx[4];
    fetch(grp < g_end ? grp : n_groups, bfr);
    // AREG (at most six weight fragments: the wide early layers, 24 -> 48, 48 -> 96, 96 -> 48): the A operands live in registers for the
    // whole walk — the counters showed the waves 61 % of their time waiting on memory with a fragment fetch in front of every tile
    v4i areg[AREG ? NCT * KS : 1];
    if constexpr (AREG) {
        const int cpl0 = N >> 2;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                const int ch = cpl0 * (r >> 2) + (r & 3) + 4 * ct;
                areg[ct * KS + s_] = wp[((size_t)s_ * NCT + (ch >> 4)) * 64 + q * 16 + (ch & 15)];
            }
    }
    const int P = a.OH * a.OW;
    const bool g_right = a.g_mult >= 0 && a.g_shift < 0;  // (uniform) the gate's MUL requantises with a plain right shift: branch-free form
    const bool g_fast = MODE != 0 || (a.gate && a.g_zx == -128 && a.g_zg == -128 && a.g_mult >= 0 && a.g_shift <= -1 && a.g_shift >= -20);
    const int ge = g_fast ? -a.g_shift : 1, gsh = ge - 1;
    const long long gC = (1ll << 30) + (((1ll << (ge - 1)) + (long long)a.g_zo * (1ll << ge)) << 31);
    // the gate bytes and the residual of a group travel with its activations, one group ahead (the walk is bound by load latency: a load at
    // the point of use costs a round trip per group); out-of-range requests (past the last group, k past Cin) read zeros
    const int cpl = N >> 2;                         // channels per lane
    constexpr int RN = AREG ? NCT : 4;              // residual dwords requested ahead: all tiles (register variants) or the first block of four
    const __amdgpu_buffer_rsrc_t rs_g =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.gate ? a.gate : a.x), 0, a.gate ? (int)(n_pos / P * K) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(ADD ? a.res : a.x), 0, ADD ? (int)(n_pos * N) : 0, 0x00020000);
    auto fetch_gate = [&](int chunk, v4i (&dst)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int koff = 64 * s + 16 * q;
            dst[s] = (v4i){0, 0, 0, 0};
            if (s < ksteps) dst[s] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_g, koff < K ? chunk * K + koff : 0x7ffffff0, 0, 0));
        }
    };
    auto fetch_res = [&](int g_, int (&dst)[RN]) {
        const int base = (g_ < n_groups ? g_ * 16 + r : (int)n_pos) * N + cpl * q;
#pragma unroll
        for (int u = 0; u < RN; ++u) {
            dst[u] = 0;
            if (ADD && u < n_ct) dst[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_res, base + 4 * u, 0, 0);
        }
    };
    v4i gfr[4];
    int rcur[RN], rnx[RN];
    // the chunk of the walk's first group (one division per wave), then counted: gpc groups per chunk
    const int gpc = P >> 4;
    int gchunk = a.gate && grp < g_end ? grp / gpc : 0, g_left = 0;  // groups left with the gate bytes in gfr
    if (a.gate && grp < g_end) {
        fetch_gate(gchunk, gfr);
        g_left = (gchunk + 1) * gpc - grp;
    }
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(n_pos * N), 0x00020000);
    fetch_res(grp < g_end ? grp : n_groups, rcur);
    for (; grp < g_end; ++grp) {
    const int nxt = grp + 1 < g_end ? grp + 1 : n_groups;
    fetch(nxt, bnx);
    fetch_res(nxt, rnx);
    const int pos = grp * 16 + r;
    if (a.gate) {
        if (g_left == 0) {  // (wave-uniform)
            fetch_gate(++gchunk, gfr);
            g_left = gpc;
        }
        --g_left;
        // squeeze-excite MUL on the way in: the 16 bytes of every k-step times the chunk's gate bytes, requantised exactly as
        // i8_scale_kernel does (the scaled map is never written); a group of 16 positions lies inside one chunk (P % 16 == 0)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int koff = 64 * s + 16 * q;
            if (s < ksteps && koff < K) {
                const v4i gv = gfr[s];
                v4i o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    int packed = 0;
                    if (MODE != 0 || g_fast) {
                        // both zero points -128: (x + 128)(g + 128) >= 0, no sign term, and the requantisation with the output zero point folded in
                        // is the high dword of one multiply-add (bn_i8_pw.hip) — 5 instead of 12 instructions per byte
                        const unsigned xu = (unsigned)bfr[s][d] ^ 0x80808080u, gu = (unsigned)gv[d] ^ 0x80808080u;
                        int gq[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int p = (int)((xu >> (8 * e)) & 0xff) * (int)((gu >> (8 * e)) & 0xff);
                            gq[e] = med3i((int)(((long long)p * a.g_mult + gC) >> 32) >> gsh, a.g_amin, a.g_amax);
                        }
                        packed = pack4(gq);
                    } else if constexpr (MODE == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int xv = (int)(int8_t)(bfr[s][d] >> (8 * e)) - a.g_zx, g = (int)(int8_t)(gv[d] >> (8 * e)) - a.g_zg;
                            packed |= (clampi(mbqm_u(xv * g, a.g_mult, a.g_shift, g_right) + a.g_zo, a.g_amin, a.g_amax) & 0xff) << (8 * e);
                        }
                    }
                    o[d] = packed;
                }
                bfr[s] = o;
            }
        }
    }
    // Tile ct computes the channels (N / 4) (i >> 2) + 4 ct + (i & 3) in its rows i (the A fragment of row i is fetched from wherever
    // the packer put that channel), so lane (r, q) ends up with the N / 4 CONSECUTIVE channels (N / 4) q .. of its position: the four
    // lanes of a position write one contiguous run of N bytes, in 16-byte pieces where the tile count allows.
    const int y_base = pos * N + cpl * q, res_base = y_base;
    const int ch_r = cpl * (r >> 2) + (r & 3);      // + 4 ct: the channel of this lane's A row
    int rmid[4] = {0, 0, 0, 0};                     // (tile-loop variant) residual dwords of the block after the current one
#pragma unroll AREG ? 2 : 1
    for (int ct0 = 0; ct0 < n_ct; ct0 += 4) {
        int rblk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (AREG) rblk[u] = ct0 + u < NCT ? rcur[ct0 + u < NCT ? ct0 + u : 0] : 0;
            else rblk[u] = ct0 == 0 ? rcur[u] : rmid[u];
        }
        if constexpr (ADD && !AREG) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ct0 + 4 + u < n_ct) rmid[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_res, res_base + 4 * (ct0 + 4 + u), 0, 0);
        }
        int outw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ct = ct0 + u;
            outw[u] = 0;
            if (ct >= n_ct) break;
            const int ch = ch_r + 4 * ct;
            const int cidx = ((cpl * q) >> 2) + ct;  // constants of the lane's four channels (N / 4) q + 4 ct ..
            v4i acc = cst[cidx * CW + 0];
            if constexpr (AREG) {
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(areg[ct * KS + s_], bfr[s_], acc, 0, 0, 0);
            } else {
                if (w_in_lds) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wl[(ct * KS + s) * 64 + lane], bfr[s], acc, 0, 0, 0);
                } else {
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp[((size_t)s * n_ct + (ch >> 4)) * 64 + q * 16 + (ch & 15)], bfr[s], acc, 0, 0, 0);
                }
            }
            const v4i m = cst[cidx * CW + 1], sh = cst[cidx * CW + 2];
            const int rv = rblk[u];
            if constexpr (MODE == 0) {
                int packed = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int qv = clampi(mbqm_u(acc[e], m[e], sh[e], rq) + a.pw_zp_out, a.pw_amin, a.pw_amax);
                    if (ADD) {
                        const int sa = add_lut[0][(rv >> (8 * e)) & 0xff];
                        const int sb = add_lut[1][qv & 0xff];
                        qv = clampi(mbqm(sa + sb, a.add.mo, a.add.so) + a.add.zo, a.add.amin, a.add.amax);
                    }
                    packed |= (qv & 0xff) << (8 * e);
                }
                outw[u] = packed;
            } else {
                int qv[4];
                if constexpr (MODE == 2) {
                    const v4i clo = cst[cidx * CW + 3], chi = cst[cidx * CW + 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const long long C = (long long)(((unsigned long long)(unsigned)chi[e] << 32) | (unsigned)clo[e]);
                        qv[e] = med3i((int)(((long long)acc[e] * m[e] + C) >> 32) >> sh[e], a.pw_amin, a.pw_amax);
                    }
                } else if constexpr (TAB) {
                    // own value + 128 (the table's column), then the ADD as one byte read
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int own = med3i(mbqm_right(acc[e], m[e], sh[e]) + (a.pw_zp_out + 128), a.pw_amin + 128, a.pw_amax + 128);
                        qv[e] = add_tab[__builtin_amdgcn_perm((unsigned)rv, (unsigned)own, 0x0c0c0400u + (e << 8))];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) qv[e] = med3i(mbqm_right(acc[e], m[e], sh[e]) + a.pw_zp_out, a.pw_amin, a.pw_amax);
                }
                if (ADD && !TAB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int sa = add_lut[0][(rv >> (8 * e)) & 0xff];
                        const int sb = add_lut[1][qv[e] & 0xff];
                        qv[e] = med3i(mbqm_right(sa + sb, a.add.mo, a.add.so) + a.add.zo, a.add.amin, a.add.amax);
                    }
                }
                outw[u] = pack4(qv);
            }
            if constexpr (AREG) asm volatile("" : "+v"(outw[u]) :: "memory");  // one tile at a time: the unrolled tiles must not all be in flight (288 registers)
        }
        const int left = n_ct - ct0;
        if (left >= 4 && (cpl & 15) == 0) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(int)))) int, (v4i){outw[0], outw[1], outw[2], outw[3]}), rs_y, y_base + 4 * ct0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < left) __builtin_amdgcn_raw_buffer_store_b32(outw[u], rs_y, y_base + 4 * (ct0 + u), 0, 0);
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) bfr[s] = bnx[s];
#pragma unroll
    for (int u = 0; u < RN; ++u) rcur[u] = rnx[u];
    }
}

bool i8_pw_wave_supported(const DwPw8Args& a) {
    const long n_pos = (long)a.B * a.OH * a.OW;
    if (a.gate && (a.Cin % 16 || ((long)a.OH * a.OW) % 16)) return false;  // gate rows are read 16 bytes at a time, a group stays inside a chunk
    return g_opt.i8_strip && !a.has_dw && !a.transposed && !a.lut && !a.qx && a.sh == 1 && a.sw == 1 && a.H == a.OH && a.W == a.OW && a.Cin <= 256 &&
           a.Cin % 4 == 0 && a.Cout % 16 == 0 && n_pos % 16 == 0 && n_pos * a.Cin < 0x7fff0000L && n_pos * a.Cout < 0x7fff0000L;
}

void launch_i8_pw_wave(const DwPw8Args& a, hipStream_t s) {
    const long n_pos = (long)a.B * a.OH * a.OW;
    const long wanted = (n_pos / 16 + 3) / 4;
    const long cap = (long)256 * ((a.Cout / 16) * ((a.Cin + 63) / 64) > 6 ? 6 : 16);  // persistent; fewer workgroups where each stages the weights
    const unsigned blocks = (unsigned)(wanted < cap ? wanted : cap);
    const int nct = a.Cout / 16, ks = (a.Cin + 63) / 64;
    size_t w_bytes = (size_t)ks * 64 * a.Cout;  // the fragment-ordered weight matrix
    if (w_bytes > 48 * 1024) w_bytes = 0;
    // requantisation forms (see the kernel's MODE): a gate must have the one-multiply-add form for modes 1 and 2
    const bool gate_fast = !a.gate || (a.g_zx == -128 && a.g_zg == -128 && a.g_mult >= 0 && a.g_shift <= -1 && a.g_shift >= -20);
    bool tab = false;  // (set below)
    const bool add_right = !a.add.enabled || (a.add.mo >= 0 && a.add.so < 0);
    int mode = 0;
    if (g_opt.i8_pw_forms && (a.rq_right & 4) && gate_fast && add_right) {
        mode = 1;
        if ((a.rq_right & 2) && a.pw_amin >= a.pw_zp_out && !a.add.enabled) mode = 2;
    }
    const size_t smem = (size_t)a.Cout * (mode == 2 ? 20 : 12);
    // the ADD as a 64 KB table in LDS: one sixteen-wave workgroup per CU
    tab = mode == 1 && a.add.enabled && a.add_tab && g_opt.i8_add_tab;
    const long wanted_t = (n_pos / 16 + 15) / 16;
    const unsigned blocks_t = (unsigned)(wanted_t < 256 ? wanted_t : 256);
#define BN_PWW(ADDV, NCTV, KSV, MODEV) \
    hipLaunchKernelGGL((i8_pw_wave_kernel<ADDV, NCTV, KSV, MODEV, false>), dim3(blocks), dim3(256), smem + (NCTV ? 0 : w_bytes), s, a, n_pos, NCTV ? 0 : (int)(w_bytes / 16))
#define BN_PWWT(NCTV, KSV)                                                                                                                   \
    {                                                                                                                                        \
        const size_t sm = 65536 + smem + (NCTV ? 0 : w_bytes);                                                                               \
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(i8_pw_wave_kernel<true, NCTV, KSV, 1, true>), sm))                              \
            hipLaunchKernelGGL((i8_pw_wave_kernel<true, NCTV, KSV, 1, true>), dim3(blocks_t), dim3(1024), sm, s, a, n_pos, NCTV ? 0 : (int)(w_bytes / 16)); \
        else BN_PWW(true, NCTV, KSV, 1); /* (the runtime refused the LDS size: the two-table form) */                                        \
    }
#define BN_PWW1(NCTV, KSV)                                   \
    {                                                        \
        if (a.add.enabled) {                                 \
            if (tab) BN_PWWT(NCTV, KSV)                      \
            else if (mode == 1) BN_PWW(true, NCTV, KSV, 1);  \
            else BN_PWW(true, NCTV, KSV, 0);                 \
        } else if (mode == 2) BN_PWW(false, NCTV, KSV, 2);   \
        else if (mode == 1) BN_PWW(false, NCTV, KSV, 1);     \
        else BN_PWW(false, NCTV, KSV, 0);                    \
        return;                                              \
    }
#define BN_PWW2(NCTV, KSV) \
    if (nct == NCTV && ks == KSV) BN_PWW1(NCTV, KSV)
    BN_PWW2(3, 1)
    BN_PWW2(6, 1)
    BN_PWW2(3, 2)
    BN_PWW2(2, 1)
    BN_PWW2(4, 1)
#undef BN_PWW2
    // more fragments than registers: tile loop over the LDS copy, k-steps still a compile-time constant (Cin <= 256)
    if (ks == 1) BN_PWW1(0, 1)
    if (ks == 2) BN_PWW1(0, 2)
    if (ks == 3) BN_PWW1(0, 3)
    BN_PWW1(0, 4)
#undef BN_PWW1
#undef BN_PWWT
#undef BN_PWW
}

template <int RG, int CT>
void launch_cfg8(const DwPw8Args& a, hipStream_t s) {
    const int tiles = (a.OH / a.TH) * (a.OW / a.TW) * ((a.B + a.NB - 1) / a.NB);
    const int slices = (a.Cout / 16) / (RG * CT);
    const int kc = a.Cin < kKC8 ? a.Cin : kKC8;
    const int kcp = (kc + 63) & ~63;
    const size_t a_bytes = (size_t)64 * (kcp + 16);
    const size_t o_bytes = a.transposed ? (size_t)(RG * CT * 16) * 68 * 4 : (size_t)64 * (RG * CT * 16 + 4) * 4;
    const size_t smem = a_bytes > o_bytes ? a_bytes : o_bytes;
    if (a.transposed)
        hipLaunchKernelGGL((i8_dwpw_kernel<RG, CT, false, true>), dim3(tiles, slices), dim3(256), smem, s, a);
    else if (a.has_dw)
        hipLaunchKernelGGL((i8_dwpw_kernel<RG, CT, true, false>), dim3(tiles, slices), dim3(256), smem, s, a);
    else
        hipLaunchKernelGGL((i8_dwpw_kernel<RG, CT, false, false>), dim3(tiles, slices), dim3(256), smem, s, a);
}

}  // namespace

bool i8_front_supported(int H0, int W0, int C, int N, int OH, int OW) {
    return C == 16 && N == 32 && OH % 8 == 0 && OW % 8 == 0 && H0 == 2 * OH && W0 == 4 * OW;
}

void launch_i8_front(const I8FrontParams& q, const int8_t* fe, int8_t* y, int B, hipStream_t s) {
    Front8Args a{fe, y, q.stem_w, q.stem_b, q.stem_mult, q.stem_shift, q.dw_w, q.dw_b, q.dw_mult, q.dw_shift, q.pw_w, q.pw_b, q.pw_mult,
                 q.pw_shift, B, q.H0, q.W0, q.H0, q.W0 / 2, q.N, q.OH, q.OW, q.stem_zp_in, q.stem_zp_out, q.stem_amin, q.stem_amax,
                 q.dw_zp_out, q.dw_amin, q.dw_amax, q.pw_zp_out, q.pw_amin, q.pw_amax, q.rq_right};
    hipLaunchKernelGGL(i8_front_kernel, dim3((q.OH / 8) * (q.OW / 8) * B), dim3(256), 0, s, a);
}

bool i8_mel_mfma_supported(const DwPw8Args& a) {
    return a.transposed && !a.has_dw && !a.add.enabled && a.Cout == 64 && a.Cin % 64 == 0 && a.W % 64 == 0 && a.H == 1 && a.OH == 1 && a.OW == a.W &&
           (size_t)64 * (a.Cin + 16) <= 65536;
}

bool i8_dwpw_supported(int Cin, int Cout) { return Cin % 4 == 0 && Cout % 16 == 0 && Cin >= 4; }

bool i8_pw_wave_takes(const DwPw8Args& a) { return i8_pw_lds_supported(a) || i8_pw_wave_supported(a); }

void launch_i8_dwpw(const DwPw8Args& a, hipStream_t s) {
    const bool mel_kernel = !g_opt.i8_mel_generic;
    if (a.qx || (mel_kernel && i8_mel_mfma_supported(a))) {  // (the packer only fuses QUANTIZE for shapes this kernel takes)
        const unsigned nb = (unsigned)(a.B * (a.W / 64));
        const size_t lds = (size_t)64 * (a.Cin + 16);
        if (a.qx && a.qmode == 1 && a.qtiled)
            if (a.qguard.audit) hipLaunchKernelGGL((i8_mel_mfma_kernel<true, 3>), dim3(nb), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((i8_mel_mfma_kernel<true, 1>), dim3(nb), dim3(256), lds, s, a);
        else if (a.qx && a.qmode == 2 && a.qtiled)  // dirty blocks only: a modest grid walks the work list
            hipLaunchKernelGGL((i8_mel_mfma_kernel<true, 2>), dim3(nb < 2048u ? nb : 2048u), dim3(256), lds, s, a);
        else if (a.qx)
            hipLaunchKernelGGL((i8_mel_mfma_kernel<true, 0>), dim3(nb), dim3(256), lds, s, a);
        else
            hipLaunchKernelGGL((i8_mel_mfma_kernel<false, 0>), dim3(nb), dim3(256), lds, s, a);
        return;
    }
    if (i8_pw_lds_supported(a)) return launch_i8_pw_lds(a, s);
    if (i8_pw_wave_supported(a)) return launch_i8_pw_wave(a, s);
    // (a gate is only ever set by the caller after i8_pw_wave_takes() said yes)
    const int ct_total = a.Cout / 16;
    static const int kSlices[] = {16, 12, 8, 6, 4, 3, 2, 1};
    int slice = 1;
    for (int v : kSlices)
        if (ct_total % v == 0) {
            slice = v;
            break;
        }
    switch (slice) {
        case 16: launch_cfg8<4, 4>(a, s); break;
        case 12: launch_cfg8<4, 3>(a, s); break;
        case 8: launch_cfg8<4, 2>(a, s); break;
        case 6: launch_cfg8<2, 3>(a, s); break;
        case 4: launch_cfg8<4, 1>(a, s); break;
        case 3: launch_cfg8<1, 3>(a, s); break;
        case 2: launch_cfg8<2, 1>(a, s); break;
        default: launch_cfg8<1, 1>(a, s); break;
    }
}

// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_i8_fused() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&i8_mel_mfma_kernel<true, 1>));
}

}  // namespace bn

#ifdef BN_TAIL_STAMPS
// debug export of the stamps build only: where the guarded mel mixer writes its stamps ([1024][4][8] int64, zeroed by the caller)
extern "C" __attribute__((visibility("default"))) int bn_debug_mel_stamps(long long* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(bn::g_mel_stamps), &d_buf, sizeof d_buf) == hipSuccess ? 0 : -1;

}
#endif
