// bn_f32_strip.hip — float32 depthwise-separable blocks of stages 1-3 (Cin 32 / 64 / 128, maps at least 16 columns wide) and the
// front block as row-streaming strips, the float32 sibling of bn_i8_strip.hip:
//
//   DW 3x3 (+bias, activation) -> PW 1x1 on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32 FMA chain)
//   [+ residual = the block input] -> activation
//
// These layers move 0.5-0.8 GB per 1024 chunks and do almost no arithmetic; the tile kernels (bn_f32_fused.hip) reach
// 2.4 TB/s on them because every 64-position tile pays nine tap loads per depthwise output, an LDS round trip and two
// barriers before its first store.  Here a workgroup owns a strip of 16 output columns and walks down the rows:
//
//   * NW = Cin/16 waves share the strip, wave w owning input channels 16 w .. 16 w + 15: lane (n, kq) = (column, channel quad)
//     holds ONE float4 per tap, the 3x3 window of its quad lives in 36 registers and a new input row costs three 16-byte
//     buffer loads that serve three output rows (rows are requested two steps ahead; SAME padding is the hardware range
//     check of the buffer descriptor — padded taps read as 0);
//   * the depthwise outputs of a lane are B operands as they stand: MFMA (ks, g) contracts channels 16 ks + 4 kq + g over the
//     four lane groups kq, so the waves only swap their float4 through LDS (one barrier per output row, double-buffered)
//     and each wave multiplies all Cin channels into ITS Cout/NW output channels;
//   * the A operand holds the pointwise weights with rows permuted so that lane (n, q) ends up with output channels
//     (Cout/NW) w + 4 NT q + 4 t + 0..3: consecutive channels of its own column (16/32-byte stores, no LDS transpose), and for
//     Cin = Cout exactly the quad it read — the residual is the centre tap it already holds.
//
// Weights are gathered from the tensors the packer already emits (fragment-ordered pointwise matrix, [3][3][C] depthwise
// taps); no blob change.
#include <stdlib.h>

#include "bn_kernels.h"

namespace bn {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// every activation code (none / ReLU / ReLU6) is one v_med3_f32 with run-time bounds: no per-code copies of the row loop
struct ActBounds { float lo, hi; };
__device__ __forceinline__ ActBounds act_bounds(int act) {
    return {act == 0 ? -__builtin_inff() : 0.0f, act == 2 ? 6.0f : __builtin_inff()};
}
__device__ __forceinline__ v4f act4(v4f v, ActBounds b) {
    return (v4f){__builtin_amdgcn_fmed3f(v.x, b.lo, b.hi), __builtin_amdgcn_fmed3f(v.y, b.lo, b.hi), __builtin_amdgcn_fmed3f(v.z, b.lo, b.hi),
                 __builtin_amdgcn_fmed3f(v.w, b.lo, b.hi)};
}

__device__ __forceinline__ void mfma_acc(v4f& acc, float a, float b) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }

// A 16-byte store must not see its data registers rewritten right behind it.  Measured on MI355X: `buffer_store_dwordx4 v[4:7], ..,
// s6 offen` followed at once by `v_med3_f32 v4, ..` (the next tile's result re-using the registers) stored the NEW v4 for lanes
// 12-15 of every 16 — wrong output columns 12-15 in 1 launch of 50 up to every launch, depending on how busy the memory pipeline
// was.  The compiler knows this hazard only for stores without an SGPR offset.  Keeping the data an in/out operand of a
// two-wait-state no-op placed after the store makes the allocator pick other registers for what follows;
// tools/store_hazard_check.py scans the generated assembly of every kernel file and the Makefile fails the build on a hit.
__device__ __forceinline__ void store16(__amdgpu_buffer_rsrc_t rs, v4f r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, r), rs, voff, soff, 0);
    asm volatile("s_nop 1" : "+v"(r));
}

struct Row4 { v4f t[3]; };  // the three taps (columns j = 0..2) of one input row, one channel quad

template <int NW, int COUT, int S, bool RES>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(((NW == 2 && COUT == 32) || NW == 8) ? 4 : 3))) void f32_strip_kernel(DwPwArgs a) {
    constexpr int CIN = 16 * NW, CWO = COUT / NW, NT = CWO / 16;
    static_assert(NT == 1 || NT == 2, "16 or 32 output channels per wave");
    static_assert(!RES || (CIN == COUT && S == 1), "the residual is the block input");
    __shared__ v4f xchg[2][NW][64];
    __shared__ v4f dw_lds[9][CIN / 4];  // depthwise taps: re-read every row (9 x 16 B per lane) instead of pinning 36 registers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int strips_x = a.OW >> 4;
    const int rblocks = (a.OH + a.TH - 1) / a.TH;
    int wid = xcd_tile(blockIdx.x, gridDim.x);
    const int sx = wid % strips_x;
    wid /= strips_x;
    const int ry = wid % rblocks;
    const int chunk = wid / rblocks;
    const int oh0 = ry * a.TH;
    const int nrows = (a.OH - oh0) < a.TH ? (a.OH - oh0) : a.TH;
    const int ow = sx * 16 + n;
    const int c0 = 16 * w + 4 * kq;  // first input channel of the lane
    const ActBounds dw_bounds = act_bounds(a.dw_act), pw_bounds = act_bounds(a.pw_act);

    for (int i = tid; i < 9 * (CIN / 4); i += 64 * NW) (&dw_lds[0][0])[i] = reinterpret_cast<const v4f*>(a.dw_w)[i];
    __syncthreads();
    const v4f* dw = &dw_lds[0][c0 >> 2];  // tap t of this lane's quad at dw[t * (CIN / 4)]
    const v4f dwb = *reinterpret_cast<const v4f*>(a.dw_b + c0);
    // A operands: pa[t][ks] = W[16 ks + 4 kq + g][ch], g = 0..3, ch = CWO w + 4 NT (m >> 2) + 4 t + (m & 3) for lane (m, kq);
    // the packer's fragment order [K/16][N/16][64][4] holds W[16 j + 4 (l >> 4) + e][16 ct + (l & 15)] at [j][ct][l][e].
    v4f pa[NT][NW], pb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ch = CWO * w + 4 * NT * (n >> 2) + 4 * t + (n & 3);
#pragma unroll
        for (int ks = 0; ks < NW; ++ks)
            pa[t][ks] = reinterpret_cast<const v4f*>(a.pw_w)[(ks * (COUT / 16) + (ch >> 4)) * 64 + kq * 16 + (ch & 15)];
        pb[t] = *reinterpret_cast<const v4f*>(a.pw_b + CWO * w + 4 * NT * kq + 4 * t);
    }

    const int in_chunk_bytes = a.H * a.W * CIN * 4;
    const int row_bytes = a.W * CIN * 4;
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)chunk * a.H * a.W * CIN, 0, in_chunk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * COUT, 0, a.OH * a.OW * COUT * 4, 0x00020000);
    const int iw0 = ow * S - a.pl;
    int voff_in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)  // padding columns: an offset beyond the descriptor's range reads as 0
        voff_in[j] = (iw0 + j >= 0 && iw0 + j < a.W) ? ((iw0 + j) * CIN + c0) * 4 : 0x7fff0000;
    const int voff_out = (ow * COUT + CWO * w + 4 * NT * kq) * 4;
    const int ir0 = oh0 * S - a.pt;
    const int rows_needed = S * (nrows - 1) + 3;

    Row4 raw[2], T[3];
    auto row_ok = [&](int rr) { const int ir = ir0 + rr; return rr < rows_needed && ir >= 0 && ir < a.H; };
    auto issue = [&](int slot, int rr) {
        if (row_ok(rr)) {
            const int soff = (ir0 + rr) * row_bytes;
#pragma unroll
            for (int j = 0; j < 3; ++j) raw[slot].t[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff_in[j], soff, 0));
        }
    };
    auto consume = [&](int slot, int rr, int ti) {
        if (row_ok(rr)) {
            T[ti] = raw[slot];
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) T[ti].t[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto emit = [&](int i0, int i1, int i2, int oh, int step) {
        asm volatile("" ::: "memory");  // keeps the tap reads inside the row loop
        v4f acc = dwb;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc = __builtin_elementwise_fma(T[i0].t[j], dw[(0 + j) * (CIN / 4)], acc);
            acc = __builtin_elementwise_fma(T[i1].t[j], dw[(3 + j) * (CIN / 4)], acc);
            acc = __builtin_elementwise_fma(T[i2].t[j], dw[(6 + j) * (CIN / 4)], acc);
        }
        acc = act4(acc, dw_bounds);
        v4f (*buf)[64] = xchg[step & 1];
        buf[w][lane] = acc;
        // LDS only: the prefetched global loads stay in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        v4f o[NT];
        if constexpr (NW >= 8 && NT == 1) {
            // eight waves per strip: the other waves' fragments two at a time (the eight of them at once are 32 registers: with them the kernel
            // needs ~145 and ONE workgroup of eight waves fits a CU; at <= 128 two fit, and this kernel is bound by the per-row latency chain)
            o[0] = pb[0];
            v4f fc = buf[0][lane], fn = buf[1][lane];
#pragma unroll
            for (int ks = 0; ks < NW; ++ks) {
                const v4f fu = fc;
                fc = fn;
                if (ks + 2 < NW) fn = buf[ks + 2][lane];
#pragma unroll
                for (int g = 0; g < 4; ++g) mfma_acc(o[0], pa[0][ks][g], fu[g]);
            }
        } else {
        v4f f[NW];
#pragma unroll
        for (int ks = 0; ks < NW; ++ks) f[ks] = buf[ks][lane];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            o[t] = pb[t];
#pragma unroll
            for (int ks = 0; ks < NW; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g) mfma_acc(o[t], pa[t][ks][g], f[ks][g]);
        }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            v4f r = o[t];
            if constexpr (RES) r += T[i1].t[1];  // centre tap = the block input at this position, channels 16 w + 4 q + 0..3
            r = act4(r, pw_bounds);
            store16(rs_out, r, voff_out + 16 * t, oh * a.OW * COUT * 4);
        }
    };

    constexpr int P = 3 - S;
    issue(0, 0);
    issue(1, 1);
#pragma unroll
    for (int rr = 0; rr < P; ++rr) {
        consume(rr & 1, rr, rr % 3);
        issue(rr & 1, rr + 2);
    }
    constexpr int U = 6 / S;
    for (int k = 0; k < nrows; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u >= nrows) break;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int rs = P + S * u + s;
                consume(rs & 1, S * k + rs, rs % 3);
                issue(rs & 1, S * k + rs + 2);
            }
            emit((S * u) % 3, (S * u + 1) % 3, (S * u + 2) % 3, oh0 + k + u, k + u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Front block as strips: frontend map [H0][W0] (finalised, or raw mel energies finalised while loading like f32_front_kernel)
// -> 3x3 stem (stride 1x2, 16 channels) -> depthwise 3x3 stride 2 -> pointwise 16 -> 32; same arithmetic as f32_front_kernel up to
// the FMA order.  One wave per strip of 16 output columns, no barrier in the row loop.  The stem runs on the f32 matrix cores:
// the contraction index of MFMA j is the window ROW (lane group kq; kq = 3 carries zero weights), lane (n, kq) streams input
// row (stem row - 1 + kq) and feeds the element fe[row][2 sc + j] of ITS row as the B operand, so three MFMAs (j = 0..2) give
// stem column sc for the four channels 4 q .. 4 q + 3 in lane (n, q) — the quad its depthwise stage needs.  Nine MFMAs per stem
// row cover stem columns 2 ow, 2 ow + 1, 2 ow + 2 (the taps of the stride-2 depthwise window).
// STAGED: the workgroup = the OW / 16 strips of one row block.  It copies the 2 TH + 4 input rows it needs into LDS once —
// whole rows, finalised ONCE per element on the way — and the row loop reads its B operands from LDS.  The per-wave variant
// (each lane streaming its own rows from memory one step ahead) was bound by memory latency, 3150 cycles per wave and row, and
// finalised every element three times.
template <bool STAGED>
__global__ __launch_bounds__(STAGED ? 1024 : 256) __attribute__((amdgpu_waves_per_eu(STAGED ? 4 : 3))) void f32_front_strip_kernel(F32FrontStripArgs a) {  // (the per-wave A/B form spilled 4 registers at four waves)
    extern __shared__ __attribute__((aligned(16))) float fe_tile[];  // STAGED: [2 TH + 4][W0 + 8], rows sr0 - 1 .., zero outside the map
    __shared__ float rowc[64][12];          // per input row: wsum, then the ten magnitude-scaling rows (finalising mode)
    __shared__ v4f dw_lds[9][4];            // depthwise taps [tap][quad]
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const bool fin = a.minmax != nullptr;
    if (fin)
        for (int i = tid; i < a.H0 * 12; i += nthreads) {
            const int rr = i / 12, c = i - rr * 12;
            rowc[rr][c] = c == 0 ? a.wsum[rr] : (c <= 10 ? a.magp[(c - 1) * a.H0 + rr] : 0.0f);
        }
    if (tid < 36) (&dw_lds[0][0])[tid] = reinterpret_cast<const v4f*>(a.dw_w)[tid];
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int strips_x = a.OW >> 4;
    const int rblocks = (a.OH + a.TH - 1) / a.TH;
    int wid = STAGED ? xcd_tile(blockIdx.x, gridDim.x) * strips_x + wave : xcd_tile(blockIdx.x, gridDim.x) * 4 + wave;
    if (!STAGED && wid >= a.B * strips_x * rblocks) return;
    const int sx = wid % strips_x;
    wid /= strips_x;
    const int ry = wid % rblocks;
    const int chunk = wid / rblocks;
    const int oh0 = ry * a.TH;
    const int nrows = (a.OH - oh0) < a.TH ? (a.OH - oh0) : a.TH;
    const int ow = sx * 16 + n;
    const ActBounds st_bounds = act_bounds(a.stem_act), dw_bounds = act_bounds(a.dw_act), pw_bounds = act_bounds(a.pw_act);

    // stem A operands: lane (m, kq) holds w[kq][j][m] for j = 0..2 (0 for kq = 3); bias of channels 4 q .. 4 q + 3 as C operand
    float sa[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) sa[j] = kq < 3 ? a.stem_w[(kq * 3 + j) * 16 + n] : 0.0f;
    const v4f stb = *reinterpret_cast<const v4f*>(a.stem_b + 4 * kq);
    const v4f dwb = *reinterpret_cast<const v4f*>(a.dw_b + 4 * kq);
    v4f pa[2], pb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ch = 8 * (n >> 2) + 4 * t + (n & 3);
        pa[t] = reinterpret_cast<const v4f*>(a.pw_w)[(ch >> 4) * 64 + kq * 16 + (ch & 15)];
        pb[t] = *reinterpret_cast<const v4f*>(a.pw_b + 8 * kq + 4 * t);
    }
    float mn = 0.0f, inv_rng = 1.0f;
    if (fin) {
        mn = a.minmax[2 * chunk];
        inv_rng = 1.0f / (float)((double)(a.minmax[2 * chunk + 1] - mn) + 1e-10);
    }

    const int fe_bytes = a.H0 * a.W0 * 4;
    const __amdgpu_buffer_rsrc_t rs_fe =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fe) + (size_t)chunk * a.H0 * a.W0, 0, fe_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * 32, 0, a.OH * a.OW * 32 * 4, 0x00020000);
    const int voff_out = (ow * 32 + 8 * kq) * 4;
    const bool right_st = 2 * ow + 2 >= a.W0 / 2;  // third stem column lies beyond the stem map: depthwise padding (0)
    const int sr0 = 2 * oh0;
    const int rows_needed = 2 * (nrows - 1) + 3;

    const int tile_w = a.W0 + 8;
    if constexpr (STAGED) {
        const int quads = a.W0 >> 2, nr = 2 * a.TH + 4;
        for (int i = tid; i < nr * (quads + 2); i += nthreads) {
            const int row = i / (quads + 2), c = i - row * (quads + 2);
            const int fr = sr0 - 1 + row;
            v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (c < quads && fr >= 0 && fr < a.H0) {
                v = *reinterpret_cast<const v4f*>(a.fe + ((size_t)chunk * a.H0 + fr) * a.W0 + 4 * c);
                if (fin) {
                    const v4f c0 = *reinterpret_cast<const v4f*>(&rowc[fr][0]), c1 = *reinterpret_cast<const v4f*>(&rowc[fr][4]),
                              c2 = *reinterpret_cast<const v4f*>(&rowc[fr][8]);
                    const float off = mn * c0.x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = fmaxf((v[e] - off) * inv_rng, 0.0f);
                        float r = y;
                        if (a.mag == 1) {
                            r = y * c0.y;
                            r += c0.z * fmaxf(c1.y * y + c2.x, 0.0f);
                            r += c0.w * fmaxf(c1.z * y + c2.y, 0.0f);
                            r += c1.x * fmaxf(c1.w * y + c2.z, 0.0f);
                        } else if (a.mag == 2) {
                            const float y0 = fmaxf(y - c0.y * y, 0.0f);
                            r = fmaxf(c0.z * y0 + c1.y * fmaxf(c0.w * y0 + c1.x, 0.0f), 0.0f);
                        } else if (a.mag == 3) {
                            r = 10.0f * logf(fmaxf(y, 1e-6f)) / logf(10.0f);
                        }
                        v[e] = r;
                    }
                }
            }
            *reinterpret_cast<v4f*>(fe_tile + row * tile_w + 4 * c) = v;
        }
        __syncthreads();
    }
    v4f raw[2][2];
    Row4 T[3];
    auto fe_row = [&](int srel) { return sr0 + srel - 1 + kq; };
    auto issue = [&](int slot, int srel) {
        if (!STAGED && srel < rows_needed) {
            const int fr = fe_row(srel);
            const int base = (fr >= 0 && fr < a.H0) ? (fr * a.W0 + 4 * ow) * 4 : 0x7fff0000;  // padding rows read as 0
            raw[slot][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_fe, base, 0, 0));
            raw[slot][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_fe, base + 16, 0, 0));  // beyond the row end only for ow = OW - 1
        }
    };
    auto stem_row = [&](int slot, int srel, int ti) {
        if (srel < rows_needed && sr0 + srel < a.H0) {
            const int fr = fe_row(srel);
            const bool ok = fr >= 0 && fr < a.H0;
            const bool has_hi = 4 * ow + 4 < a.W0;
            float x[7];
            if constexpr (STAGED) {
                const float* trow = fe_tile + (srel + kq) * tile_w + 4 * ow;  // input row sr - 1 + kq, already finalised / zero-padded
                const v4f lo = *reinterpret_cast<const v4f*>(trow), hi = *reinterpret_cast<const v4f*>(trow + 4);
                x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z;
            } else {
                x[0] = raw[slot][0].x; x[1] = raw[slot][0].y; x[2] = raw[slot][0].z; x[3] = raw[slot][0].w;
                x[4] = raw[slot][1].x; x[5] = raw[slot][1].y; x[6] = raw[slot][1].z;
            }
            if (!STAGED && fin) {
                const int rr = ok ? fr : 0;
                const v4f c0 = *reinterpret_cast<const v4f*>(&rowc[rr][0]), c1 = *reinterpret_cast<const v4f*>(&rowc[rr][4]),
                          c2 = *reinterpret_cast<const v4f*>(&rowc[rr][8]);
                const float off = mn * c0.x;
#pragma unroll
                for (int e = 0; e < 7; ++e) {
                    const float y = fmaxf((x[e] - off) * inv_rng, 0.0f);
                    float v = y;
                    if (a.mag == 1) {  // pwl: rows k0, k1..3, w1..3, b1..3
                        v = y * c0.y;
                        v += c0.z * fmaxf(c1.y * y + c2.x, 0.0f);
                        v += c0.w * fmaxf(c1.z * y + c2.y, 0.0f);
                        v += c1.x * fmaxf(c1.w * y + c2.z, 0.0f);
                    } else if (a.mag == 2) {  // pcen-like: rows agc, k1, sw, sb, k2
                        const float y0 = fmaxf(y - c0.y * y, 0.0f);
                        v = fmaxf(c0.z * y0 + c1.y * fmaxf(c0.w * y0 + c1.x, 0.0f), 0.0f);
                    } else if (a.mag == 3) {
                        v = 10.0f * logf(fmaxf(y, 1e-6f)) / logf(10.0f);
                    }
                    x[e] = ok ? v : 0.0f;  // padding rows stay an exact zero
                }
            }
            if (!STAGED && !has_hi) x[4] = x[5] = x[6] = 0.0f;  // columns beyond the map (the second load wrapped into the next row)
            v4f st[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                st[c] = stb;
#pragma unroll
                for (int j = 0; j < 3; ++j) mfma_acc(st[c], sa[j], x[2 * c + j]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) T[ti].t[c] = act4(st[c], st_bounds);
            if (right_st) T[ti].t[2] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) T[ti].t[c] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto emit = [&](int i0, int i1, int i2, int oh) {
        asm volatile("" ::: "memory");
        const v4f* dw = &dw_lds[0][kq];
        v4f d = dwb;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            d = __builtin_elementwise_fma(T[i0].t[j], dw[(0 + j) * 4], d);
            d = __builtin_elementwise_fma(T[i1].t[j], dw[(3 + j) * 4], d);
            d = __builtin_elementwise_fma(T[i2].t[j], dw[(6 + j) * 4], d);
        }
        d = act4(d, dw_bounds);
        v4f o[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            o[t] = pb[t];
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_acc(o[t], pa[t][g], d[g]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            store16(rs_out, act4(o[t], pw_bounds), voff_out + 16 * t, oh * a.OW * 32 * 4);
        }
    };

    issue(0, 0);
    issue(1, 1);
    stem_row(0, 0, 0);
    issue(0, 2);
    for (int k = 0; k < nrows; k += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (k + u >= nrows) break;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int rs = 1 + 2 * u + s;
                stem_row(rs & 1, 2 * k + rs, rs % 3);
                issue(rs & 1, 2 * k + rs + 2);
            }
            emit((2 * u) % 3, (2 * u + 1) % 3, (2 * u + 2) % 3, oh0 + k + u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Front block + the first residual block (stage1_ds2: depthwise 3x3 stride 1 -> pointwise 32 -> 32 -> + input -> activation) in ONE
// kernel: the 32-channel map between them (256 KB per chunk written and read back: 512 MiB of the 2.9 GiB a 1024-chunk step moves)
// never exists in HBM.  One workgroup per chunk, 3 x OW/16 waves in two roles that meet at ONE barrier per output row:
//   * producers (one wave per strip of 16 columns): the front-strip row loop as above, input map staged in LDS once; the finished
//     row goes into a two-row ring in LDS ([row & 1][column + 1][32 + 4 floats], zero border columns = the SAME padding);
//   * consumers (two waves per strip, 16 channels each — f32_strip_kernel<2, 32, 1, true>): at step t they take row t - 1 from the
//     ring into their register window, run the depthwise stage of row t - 2, swap B fragments through LDS, store row t - 3 (whose
//     MFMAs were issued right behind the previous barrier) and meet the producers at the barrier.
// The residual is the centre tap a consumer lane already holds.  Per SIMD one producer and two consumer waves: the matrix
// instructions of one role run under the vector instructions of the other.
struct F32Front2Args {
    F32FrontStripArgs f;   // f.y unused
    const float* dw_w; const float* dw_b;   // [3][3][32], [32]
    const float* pw_w; const float* pw_b;   // fragment order [2][2][64][4], [32]
    float* y;              // [B][OH][OW][32]
    int dw_act, pw_act;
};

__global__ __launch_bounds__(768) void f32_front2_kernel(F32Front2Args A) {
    const F32FrontStripArgs& a = A.f;
    extern __shared__ __attribute__((aligned(16))) float lds2[];
    __shared__ float rowc[64][12];
    __shared__ v4f dw_lds[9][4];
    __shared__ v4f dw2_lds[9][8];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int strips_x = a.OW >> 4;
    const int tile_w = a.W0 + 8, nr = a.H0 + 4;
    constexpr int RP = 36;                              // ring pitch per column (floats): 32 channels + 4
    float* fe_tile = lds2;                              // [H0 + 4][W0 + 8], rows -1 .., zero outside the map
    float* ring = fe_tile + nr * tile_w;                // [2][OW + 2][RP]
    v4f* xchg = reinterpret_cast<v4f*>(ring + 2 * (a.OW + 2) * RP);  // [strip][2][2][64]
    const bool fin = a.minmax != nullptr;
    const int chunk = xcd_tile(blockIdx.x, gridDim.x);
    if (fin)
        for (int i = tid; i < a.H0 * 12; i += nthreads) {
            const int rr = i / 12, c = i - rr * 12;
            rowc[rr][c] = c == 0 ? a.wsum[rr] : (c <= 10 ? a.magp[(c - 1) * a.H0 + rr] : 0.0f);
        }
    if (tid < 36) (&dw_lds[0][0])[tid] = reinterpret_cast<const v4f*>(a.dw_w)[tid];
    if (tid >= 64 && tid < 64 + 72) (&dw2_lds[0][0])[tid - 64] = reinterpret_cast<const v4f*>(A.dw_w)[tid - 64];
    for (int i = tid; i < 4 * RP; i += nthreads) {      // the ring's border columns stay zero: SAME padding of the second block
        const int rrow = i / (2 * RP), rest = i - rrow * 2 * RP;
        ring[(rrow * (a.OW + 2) + (rest < RP ? 0 : a.OW + 1)) * RP + (rest % RP)] = 0.0f;
    }
    __syncthreads();
    float mn = 0.0f, inv_rng = 1.0f;
    if (fin) {
        mn = a.minmax[2 * chunk];
        inv_rng = 1.0f / (float)((double)(a.minmax[2 * chunk + 1] - mn) + 1e-10);
    }
    {   // the whole input map, finalised once per element on the way (same arithmetic as f32_front_strip_kernel<true>); the loads of a
        // thread are all in flight before the first is used (six per thread for the 64 x 256 map: one HBM round trip, not six)
        const int quads = a.W0 >> 2, total = nr * (quads + 2);
        constexpr int kInFlight = 6;
        for (int i0 = tid; i0 < total; i0 += kInFlight * nthreads) {
            v4f vin[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                const int i = i0 + u * nthreads;
                const int row = i / (quads + 2), c = i - row * (quads + 2), fr = row - 1;
                vin[u] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                if (i < total && c < quads && fr >= 0 && fr < a.H0)
                    vin[u] = *reinterpret_cast<const v4f*>(a.fe + ((size_t)chunk * a.H0 + fr) * a.W0 + 4 * c);
            }
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                const int i = i0 + u * nthreads;
                if (i >= total) break;
                const int row = i / (quads + 2), c = i - row * (quads + 2), fr = row - 1;
                v4f v = vin[u];
                if (fin && c < quads && fr >= 0 && fr < a.H0) {
                    const v4f c0 = *reinterpret_cast<const v4f*>(&rowc[fr][0]), c1 = *reinterpret_cast<const v4f*>(&rowc[fr][4]),
                              c2 = *reinterpret_cast<const v4f*>(&rowc[fr][8]);
                    const float off = mn * c0.x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = fmaxf((v[e] - off) * inv_rng, 0.0f);
                        float r = y;
                        if (a.mag == 1) {
                            r = y * c0.y;
                            r += c0.z * fmaxf(c1.y * y + c2.x, 0.0f);
                            r += c0.w * fmaxf(c1.z * y + c2.y, 0.0f);
                            r += c1.x * fmaxf(c1.w * y + c2.z, 0.0f);
                        } else if (a.mag == 2) {
                            const float y0 = fmaxf(y - c0.y * y, 0.0f);
                            r = fmaxf(c0.z * y0 + c1.y * fmaxf(c0.w * y0 + c1.x, 0.0f), 0.0f);
                        } else if (a.mag == 3) {
                            r = 10.0f * logf(fmaxf(y, 1e-6f)) / logf(10.0f);
                        }
                        v[e] = r;
                    }
                }
                *reinterpret_cast<v4f*>(fe_tile + row * tile_w + 4 * c) = v;
            }
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int OH = a.OH;
    auto step_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (wave < strips_x) {
        // ---------------------------------------------------------------- producer: front block rows -> ring
        const int ow = wave * 16 + n;
        const ActBounds st_bounds = act_bounds(a.stem_act), dw_bounds = act_bounds(a.dw_act), pw_bounds = act_bounds(a.pw_act);
        float sa[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) sa[j] = kq < 3 ? a.stem_w[(kq * 3 + j) * 16 + n] : 0.0f;
        const v4f stb = *reinterpret_cast<const v4f*>(a.stem_b + 4 * kq);
        const v4f dwb = *reinterpret_cast<const v4f*>(a.dw_b + 4 * kq);
        v4f pa[2], pb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = 8 * (n >> 2) + 4 * t + (n & 3);
            pa[t] = reinterpret_cast<const v4f*>(a.pw_w)[(ch >> 4) * 64 + kq * 16 + (ch & 15)];
            pb[t] = *reinterpret_cast<const v4f*>(a.pw_b + 8 * kq + 4 * t);
        }
        const bool right_st = 2 * ow + 2 >= a.W0 / 2;
        const int rows_needed = 2 * (OH - 1) + 3;
        Row4 T[3];
        auto stem_row = [&](int srel, int ti) {
            if (srel < rows_needed && srel < a.H0) {
                const float* trow = fe_tile + (srel + kq) * tile_w + 4 * ow;  // input row srel - 1 + kq
                const v4f lo = *reinterpret_cast<const v4f*>(trow), hi = *reinterpret_cast<const v4f*>(trow + 4);
                const float x[7] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z};
                v4f st[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    st[c] = stb;
#pragma unroll
                    for (int j = 0; j < 3; ++j) mfma_acc(st[c], sa[j], x[2 * c + j]);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) T[ti].t[c] = act4(st[c], st_bounds);
                if (right_st) T[ti].t[2] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) T[ti].t[c] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            }
        };
        auto emit = [&](int i0, int i1, int i2, int oh) {
            asm volatile("" ::: "memory");
            const v4f* dw = &dw_lds[0][kq];
            v4f d = dwb;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                d = __builtin_elementwise_fma(T[i0].t[j], dw[(0 + j) * 4], d);
                d = __builtin_elementwise_fma(T[i1].t[j], dw[(3 + j) * 4], d);
                d = __builtin_elementwise_fma(T[i2].t[j], dw[(6 + j) * 4], d);
            }
            d = act4(d, dw_bounds);
            float* dst = ring + (((oh & 1) * (a.OW + 2)) + ow + 1) * RP + 8 * kq;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v4f o = pb[t];
#pragma unroll
                for (int g = 0; g < 4; ++g) mfma_acc(o, pa[t][g], d[g]);
                *reinterpret_cast<v4f*>(dst + 4 * t) = act4(o, pw_bounds);
            }
            step_barrier();
        };
        // (queueing the next row's stem matrix instructions before this row's ring write and barrier changed nothing: 0.198 vs 0.190 ms)
        stem_row(0, 0);
        for (int k = 0; k < OH; k += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (k + u >= OH) break;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int rs = 1 + 2 * u + s2;
                    stem_row(2 * k + rs, rs % 3);
                }
                emit((2 * u) % 3, (2 * u + 1) % 3, (2 * u + 2) % 3, k + u);
            }
        }
        step_barrier();  // the consumers' two trailing steps
        step_barrier();
    } else {
        // ---------------------------------------------------------------- consumer: the residual block on 16 of the 32 channels
        const int cw = wave - strips_x;
        const int sx = cw >> 1, w = cw & 1;
        const int ow = sx * 16 + n;
        const int c0 = 16 * w + 4 * kq;
        const ActBounds dw_bounds = act_bounds(A.dw_act), pw_bounds = act_bounds(A.pw_act);
        const v4f* dw = &dw2_lds[0][c0 >> 2];
        const v4f dwb = *reinterpret_cast<const v4f*>(A.dw_b + c0);
        v4f pa[2];
        {
            const int ch = 16 * w + 4 * (n >> 2) + (n & 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) pa[ks] = reinterpret_cast<const v4f*>(A.pw_w)[(ks * 2 + (ch >> 4)) * 64 + kq * 16 + (ch & 15)];
        }
        const v4f pb = *reinterpret_cast<const v4f*>(A.pw_b + 16 * w + 4 * kq);
        const __amdgpu_buffer_rsrc_t rs_out =
            __builtin_amdgcn_make_buffer_rsrc(A.y + (size_t)chunk * OH * a.OW * 32, 0, OH * a.OW * 32 * 4, 0x00020000);
        const int voff_out = (ow * 32 + 16 * w + 4 * kq) * 4;
        const float* rcol = ring + ow * RP + c0;          // tap j of row r: rcol[((r & 1) * (OW + 2) + j) * RP]
        v4f* xs = xchg + sx * 256;                        // [2][2][64]
        Row4 T[3];
        v4f pacc = {0.0f, 0.0f, 0.0f, 0.0f}, pcen = pacc;
        const int steps = OH + 2;
        for (int t0 = 0; t0 < steps; t0 += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = t0 + u;
                if (t >= steps) break;
                // window slot u <- map row t - 1 (zero rows above and below the map)
                if (t >= 1 && t <= OH) {
                    const float* rp = rcol + (((t - 1) & 1) * (a.OW + 2)) * RP;
#pragma unroll
                    for (int j = 0; j < 3; ++j) T[u].t[j] = *reinterpret_cast<const v4f*>(rp + j * RP);
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) T[u].t[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                }
                const int i0 = (u + 1) % 3, i1 = (u + 2) % 3, i2 = u;   // rows t - 3, t - 2, t - 1
                if (t >= 2) {
                    asm volatile("" ::: "memory");
                    v4f acc = dwb;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        acc = __builtin_elementwise_fma(T[i0].t[j], dw[(0 + j) * 8], acc);
                        acc = __builtin_elementwise_fma(T[i1].t[j], dw[(3 + j) * 8], acc);
                        acc = __builtin_elementwise_fma(T[i2].t[j], dw[(6 + j) * 8], acc);
                    }
                    xs[((t & 1) * 2 + w) * 64 + lane] = act4(acc, dw_bounds);
                }
                if (t >= 3) store16(rs_out, act4(pacc + pcen, pw_bounds), voff_out, (t - 3) * a.OW * 32 * 4);
                step_barrier();
                if (t >= 2) {
                    const v4f f0 = xs[((t & 1) * 2 + 0) * 64 + lane], f1 = xs[((t & 1) * 2 + 1) * 64 + lane];
                    pacc = pb;
#pragma unroll
                    for (int g = 0; g < 4; ++g) mfma_acc(pacc, pa[0][g], f0[g]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) mfma_acc(pacc, pa[1][g], f1[g]);
                    pcen = T[i1].t[1];
                }
            }
        }
        store16(rs_out, act4(pacc + pcen, pw_bounds), voff_out, (OH - 1) * a.OW * 32 * 4);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-alone depthwise 3x3 (the inverted-residual blocks: expand 1x1 -> DEPTHWISE -> squeeze-excite -> project 1x1) as a
// row-streaming kernel.  The one-thread-per-output-quad kernel (bn_f32.hip) loads nine taps per output: 2.4 TB/s.  Here a wave
// owns NCOL = 64 / CQ columns x CQ channel quads (CQ = the largest of 16, 8, 4, 2, 1 dividing C / 4, so that a tap load of the
// wave covers NCOL runs of 16 CQ bytes) and walks down the rows with the 3x3 window of its quad in registers: three 16-byte loads
// per output instead of nine, rows requested two steps ahead, SAME padding = the buffer descriptor's range check.
struct DwStreamArgs {
    const float* x;
    float* y;
    const float* w;     // [3][3][C]
    const float* bias;  // [C]
    int B, H, W, C, OH, OW, TH, pt, pl, act, CQ;
    float* gap_part;    // optional: [B][strips_x * row blocks][C] channel sums of every wave's strip, for the squeeze-excite gate behind the stage
};

// (stride 2 keeps two windows of taps: at the 128 registers of four waves per SIMD it spilled 14 of them — 48 B of scratch per lane —,
// so that instantiation is built for three waves per SIMD)
template <int S>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S == 2 ? 3 : 4))) void f32_dw_stream_kernel(DwStreamArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CQ = a.CQ, NCOL = 64 / CQ;
    const int cq = lane % CQ, n = lane / CQ;
    const int groups = a.C / (4 * CQ);
    const int strips_x = (a.OW + NCOL - 1) / NCOL;
    const int rblocks = (a.OH + a.TH - 1) / a.TH;
    long wid = (long)xcd_tile(blockIdx.x, gridDim.x) * 4 + wave;
    if (wid >= (long)a.B * groups * strips_x * rblocks) return;
    const int g = (int)(wid % groups);
    wid /= groups;
    const int sx = (int)(wid % strips_x);
    wid /= strips_x;
    const int ry = (int)(wid % rblocks);
    const int chunk = (int)(wid / rblocks);
    const int oh0 = ry * a.TH;
    const int nrows = (a.OH - oh0) < a.TH ? (a.OH - oh0) : a.TH;
    const int ow = sx * NCOL + n;
    const bool live = ow < a.OW;
    const int c0 = 4 * (g * CQ + cq);
    const ActBounds bounds = act_bounds(a.act);

    v4f wt[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) wt[i][j] = *reinterpret_cast<const v4f*>(a.w + (i * 3 + j) * a.C + c0);
    const v4f b4 = *reinterpret_cast<const v4f*>(a.bias + c0);

    const int in_chunk_bytes = a.H * a.W * a.C * 4;
    const int row_bytes = a.W * a.C * 4;
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)chunk * a.H * a.W * a.C, 0, in_chunk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * a.C, 0, a.OH * a.OW * a.C * 4, 0x00020000);
    const int iw0 = ow * S - a.pl;
    int voff_in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) voff_in[j] = (live && iw0 + j >= 0 && iw0 + j < a.W) ? ((iw0 + j) * a.C + c0) * 4 : 0x7fff0000;
    const int voff_out = live ? (ow * a.C + c0) * 4 : 0x7fff0000;  // beyond the descriptor's range: the store is dropped
    const int ir0 = oh0 * S - a.pt;
    const int rows_needed = S * (nrows - 1) + 3;

    Row4 raw[2], T[3];
    auto row_ok = [&](int rr) { const int ir = ir0 + rr; return rr < rows_needed && ir >= 0 && ir < a.H; };
    auto issue = [&](int slot, int rr) {
        if (row_ok(rr)) {
            const int soff = (ir0 + rr) * row_bytes;
#pragma unroll
            for (int j = 0; j < 3; ++j) raw[slot].t[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff_in[j], soff, 0));
        }
    };
    auto consume = [&](int slot, int rr, int ti) {
        if (row_ok(rr)) {
            T[ti] = raw[slot];
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) T[ti].t[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    v4f gsum = {0.0f, 0.0f, 0.0f, 0.0f};
    auto emit = [&](int i0, int i1, int i2, int oh) {
        v4f acc = b4;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc = __builtin_elementwise_fma(T[i0].t[j], wt[0][j], acc);
            acc = __builtin_elementwise_fma(T[i1].t[j], wt[1][j], acc);
            acc = __builtin_elementwise_fma(T[i2].t[j], wt[2][j], acc);
        }
        const v4f o = act4(acc, bounds);
        if (live) gsum += o;
        store16(rs_out, o, voff_out, oh * a.OW * a.C * 4);
    };
    constexpr int P = 3 - S;
    issue(0, 0);
    issue(1, 1);
#pragma unroll
    for (int rr = 0; rr < P; ++rr) {
        consume(rr & 1, rr, rr % 3);
        issue(rr & 1, rr + 2);
    }
    constexpr int U = 6 / S;
    for (int k = 0; k < nrows; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u >= nrows) break;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int rs = P + S * u + s;
                consume(rs & 1, S * k + rs, rs % 3);
                issue(rs & 1, S * k + rs + 2);
            }
            emit((S * u) % 3, (S * u + 1) % 3, (S * u + 2) % 3, oh0 + k + u);
        }
    }
    if (a.gap_part) {  // the strip's channel sums: the NCOL columns of a quad are lanes cq, cq + CQ, ...: a butterfly over the lane bits above CQ
        for (int m = CQ; m < 64; m <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gsum[e] += __shfl_xor(gsum[e], m, 64);
        }
        if (n == 0) *reinterpret_cast<v4f*>(a.gap_part + ((size_t)chunk * strips_x * rblocks + (size_t)ry * strips_x + sx) * a.C + c0) = gsum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Inverted-residual blocks (reference models/blocks.py:88-110): expand 1x1 (+bias, activation) -> depthwise 3x3 stride S (+bias,
// activation) in ONE kernel.  The expanded map is the largest tensor of the block (hid = 2 Cin channels at the INPUT resolution:
// 1.5 MB per chunk in stage 1); written by one kernel and read back by the next it is two thirds of the pair's HBM traffic.  Here a
// workgroup owns RB output rows of one chunk and walks down the rows, ten waves in three roles that meet at ONE barrier per hidden row:
//   * loaders (waves 8-9): keep three input rows in flight (768 float4 per row, six per thread) and copy the row that arrived two
//     steps after its request into a three-row staging ring in LDS ([W][Cin + 4]);
//   * producers (waves 0-3): hidden row = W positions x hid channels on the f32 matrix cores.  Every wave owns two position tiles x
//     three channel tiles (W x hid = 6144 in every stage of the alpha = 1.5 net: 128 x 48, 64 x 96, 32 x 192), its A fragments (the
//     packer's fragment-ordered weights) pinned in registers for the whole walk, B fragments read from the staging ring (one
//     16-byte LDS read per lane and 16 channels), the six accumulator chains interleaved; the finished row goes into a four-row ring
//     in LDS, position-major with a pitch of hid + 4 floats and zero border columns (the SAME padding of the hidden map; 16-byte
//     tile writes and tap reads both spread over all banks);
//   * consumers (waves 4-7): a thread owns ONE channel quad (nine taps + bias in registers) and every (240 / quads)-th column, the
//     quad running along the lanes: nine 16-byte LDS reads in flight, the summation order of f32_dw_stream_kernel, one 16-byte store
//     coalesced along the channels.
// At step t the producers write hidden row t while the consumers run the output row that ends at hidden row t - 1.
// Measured (configs[4], 1024 chunks): the six pairs it takes 3.18 -> 1.79 ms (3.0 TB/s on the 2 : 1 write-heavy traffic that is left,
// matrix pipe 35 % busy).  Variants that changed nothing (within 3 %): the producers loading their own B fragments one or two rows
// ahead, the epilogue of row t between the matrix instructions of row t + 1, chains not interleaved.  Slower: two sequential
// workgroups per CU without roles (2.1 ms), taps behind per-lane bounds tests (2.4 ms: nine dependent LDS round trips per output).
struct F32PwDwArgs {
    const float* x;       // [B][H][W][Cin]
    float* y;             // [B][OH][OW][hid]
    const float* pw_w;    // fragment order [Kp/16][hid/16][64][4]
    const float* pw_b;    // [hid]
    const float* dw_w;    // [3][3][hid]
    const float* dw_b;    // [hid]
    int B, H, W, Cin, hid, OH, OW, pt, pl, pw_act, dw_act, RB;
    // optional stem in front (fe != nullptr): x is not read; the loaders compute row hr of CONV 3x3 (1 -> Cin channels, stride ssh x ssw, SAME) from
    // the frontend map fe [B][H0][W0] instead of loading it — the stem map (786 KB per chunk in configs[4]) never exists in HBM either
    const float* fe; const float* stem_w; const float* stem_b;   // [3][3][Cin], [Cin]
    int H0, W0, ssh, ssw, spt, spl, stem_act;
    // optional (gap_part != nullptr): per-workgroup channel sums of the depthwise output, [B][row blocks][hid] — the squeeze-excite gate behind
    // the block pools them (f32_segate_kernel) instead of reading the whole map again
    float* gap_part;
};

// NCW = channel tiles per producer wave: 3 for hid % 48 == 0 (alpha = 1.5: W x hid = 6144), 2 for hid % 32 == 0 (alpha = 1: W x hid = 4096)
template <int NJ, int S, int NCW>
__global__ __launch_bounds__(640) void f32_pwdw_kernel(F32PwDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ring3[];  // [4][W + 2][hid + 4] (column hx at index hx + 1), then [3][W][Cin + 4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int P = a.hid + 4, PI = a.Cin + 4;
    float* stage = ring3 + 4 * (a.W + 2) * P;
    const int rblocks = (a.OH + a.RB - 1) / a.RB;
    const int wid = xcd_tile(blockIdx.x, gridDim.x);
    const int ry = wid % rblocks, chunk = wid / rblocks;
    const int oh0 = ry * a.RB;
    const int nrows = (a.OH - oh0) < a.RB ? (a.OH - oh0) : a.RB;
    const int h_lo = S * oh0 - a.pt;
    const int nhid = S * (nrows - 1) + 3;          // hidden rows h_lo .. h_lo + nhid - 1 (those outside the map are skipped)
    const int nsteps = nhid + 1;
    auto row_ok = [&](int k) { return k >= 0 && k < nhid && h_lo + k >= 0 && h_lo + k < a.H; };

    for (int i = tid; i < 8 * P; i += (int)blockDim.x) {  // border columns of the four ring slots
        const int slot = i / (2 * P), rest = i - slot * 2 * P;
        ring3[(slot * (a.W + 2) + (rest < P ? 0 : a.W + 1)) * P + (rest % P)] = 0.0f;
    }

    if (w < 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int npp = a.W >> 5;
        const int pp = w % npp, tc = w / npp;      // positions 32 pp + 16 u + n, hidden channels 16 NCW tc + 16 c + 4 kq + e
        const int nct = a.hid >> 4;
        const ActBounds pw_bounds = act_bounds(a.pw_act);
        v4f pa[NCW][NJ], pbias[NCW];
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) pa[c][j] = reinterpret_cast<const v4f*>(a.pw_w)[(j * nct + NCW * tc + c) * 64 + lane];
            pbias[c] = *reinterpret_cast<const v4f*>(a.pw_b + 16 * NCW * tc + 16 * c + 4 * kq);
        }
        bool chan_ok[NJ];                           // channels beyond Cin: the zero-padded k-step of Cin = 24
#pragma unroll
        for (int j = 0; j < NJ; ++j) chan_ok[j] = 16 * j + 4 * kq < a.Cin;
        const int src_off = (32 * pp + n) * PI + 4 * kq;
        const int dst_off = (1 + 32 * pp + n) * P + 16 * NCW * tc + 4 * kq;
        // stem mode (a.fe): the wave computes the stem outputs of ITS two position tiles itself, on the matrix cores — the contraction index of
        // MFMA i is the window column (lane group kq < 3), window row i per instruction, so a lane's accumulators end up as the stem channels
        // 16 j + 4 kq .. + 3 of its position: exactly its B fragments of the expand convolution.  Taps of the frontend map (64 KB per chunk,
        // cache resident) are requested a row ahead.  f32_stem_kernel's summation order (bias, taps row by row; a tap outside the map adds 0).
        constexpr bool STEM_OK = NJ <= 2;           // (a stem feeds at most 32 channels; the wide instantiations do not carry this code)
        constexpr int NJS = STEM_OK ? NJ : 1;
        float sa[3][NJS], tcur[2][3], tnext[2][3];
        v4f sbias[NJS];
        const bool stem = STEM_OK && a.fe != nullptr;
        const ActBounds st_bounds = act_bounds(a.stem_act);
        // taps through a range-checked raw buffer over this chunk's map: a tap outside it (SAME padding, rows the strip does not need, lane
        // group 3) gets an offset behind the buffer's end and reads as 0 — no branch around the load, all six in flight (as conditional
        // loads `ok ? fmap[i] : 0` they ran one round trip at a time)
        const __amdgpu_buffer_rsrc_t rs_fe = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(stem ? a.fe + (size_t)chunk * a.H0 * a.W0 : a.pw_b), 0, stem ? a.H0 * a.W0 * 4 : 0, 0x00020000);
        auto taps = [&](float (&v)[2][3], int k) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int ih = (h_lo + k) * a.ssh - a.spt + i, iw = (32 * pp + 16 * u + n) * a.ssw - a.spl + kq;
                    const bool ok = row_ok(k) && kq < 3 && ih >= 0 && ih < a.H0 && iw >= 0 && iw < a.W0;
                    v[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_fe, ok ? (ih * a.W0 + iw) * 4 : 0x7ffffff0, 0, 0));
                }
        };
        if (stem) {
#pragma unroll
            for (int j = 0; j < NJS; ++j) {
                const int ch = 16 * j + n;           // A operand row = stem channel
#pragma unroll
                for (int i = 0; i < 3; ++i) sa[i][j] = (kq < 3 && ch < a.Cin) ? a.stem_w[(i * 3 + kq) * a.Cin + ch] : 0.0f;
                sbias[j] = chan_ok[j] ? *reinterpret_cast<const v4f*>(a.stem_b + 16 * j + 4 * kq) : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            }
            taps(tcur, 0);
        }
        __syncthreads();                            // staging row 0 and the ring's border columns are in place
        for (int t = 0; t < nsteps; ++t) {
            if (stem) taps(tnext, t + 1);
            if (row_ok(t)) {
                const float* src = stage + (t % 3) * a.W * PI + src_off;
                v4f bf[2][NJ];
                if (stem) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < NJS; ++j) {
                            v4f st = sbias[j];
#pragma unroll
                            for (int i = 0; i < 3; ++i) mfma_acc(st, sa[i][j], tcur[u][i]);
                            bf[u][j] = act4(st, st_bounds);
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            bf[u][j] = chan_ok[j] ? *reinterpret_cast<const v4f*>(src + 16 * u * PI + 16 * j) : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                }
                float* dst = ring3 + (((h_lo + t) & 3) * (a.W + 2)) * P + dst_off;
                v4f acc[2][NCW];  // the tiles' chains interleaved: consecutive matrix instructions never depend on each other
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int c = 0; c < NCW; ++c) acc[u][c] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int c = 0; c < NCW; ++c) mfma_acc(acc[u][c], pa[c][j][g], bf[u][j][g]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int c = 0; c < NCW; ++c)  // bias behind the sum, as the stand-alone 1x1 kernels add it: the pair stays bit-identical to them
                        *reinterpret_cast<v4f*>(dst + 16 * u * P + 16 * c) = act4(acc[u][c] + pbias[c], pw_bounds);
            }
            if (stem) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 3; ++i) tcur[u][i] = tnext[u][i];
            }
            __syncthreads();
        }
        __syncthreads();                            // (the consumers' channel sums pass through LDS behind the last row)
    } else if (w < 8) {
        // ------------------------------------------------------------------------------------------------ consumers
        const int dt = tid - 256;
        const ActBounds dw_bounds = act_bounds(a.dw_act);
        const int quads = a.hid >> 2;
        const int dwn = NCW == 3 ? 240 : 256;        // threads of the depthwise stage: a multiple of the quads per position (12 / 24 / 48 or 8 / 16 / 32)
        const int groups = dwn / quads;
        const bool dw_live = dt < dwn;
        const int q = dw_live ? dt % quads : 0, cg = dw_live ? dt / quads : 0;
        v4f wt[3][3], dwb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) wt[dy][dx] = dw_live ? *reinterpret_cast<const v4f*>(a.dw_w + (dy * 3 + dx) * a.hid + 4 * q) : dwb;
        if (dw_live) dwb = *reinterpret_cast<const v4f*>(a.dw_b + 4 * q);
        float* ybase = a.y + ((size_t)chunk * a.OH) * a.OW * a.hid + 4 * q;
        v4f gsum = {0.0f, 0.0f, 0.0f, 0.0f};
        auto depthwise = [&](int oh) {
            if (!dw_live) return;
            const int hr0 = S * oh - a.pt;
            const float* rows[3];
            bool rok[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int hr = hr0 + dy;
                rok[dy] = hr >= 0 && hr < a.H;
                rows[dy] = ring3 + (size_t)((rok[dy] ? hr & 3 : 0) * (a.W + 2) + 1 - a.pl) * P + 4 * q;  // tap dx of output column ox at [(S ox + dx) P]
            }
            const bool interior = rok[0] && rok[1] && rok[2];
            for (int ox = cg; ox < a.OW; ox += groups) {
                const int off = S * ox * P;
                v4f acc = dwb;
                if (interior) {  // nine loads in flight, then the same summation order as below
                    v4f v[3][3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) v[dy][dx] = *reinterpret_cast<const v4f*>(rows[dy] + off + dx * P);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) acc = __builtin_elementwise_fma(v[dy][dx], wt[dy][dx], acc);
                } else {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
                            if (rok[dy]) acc = __builtin_elementwise_fma(*reinterpret_cast<const v4f*>(rows[dy] + off + dx * P), wt[dy][dx], acc);
                }
                const v4f o = act4(acc, dw_bounds);
                gsum += o;
                *reinterpret_cast<v4f*>(ybase + ((size_t)oh * a.OW + ox) * a.hid) = o;
            }
        };
        __syncthreads();
        for (int t = 0; t < nsteps; ++t) {
            if (t >= 3 && (t - 3) % S == 0) depthwise(oh0 + (t - 3) / S);
            __syncthreads();
        }
        // channel sums of this workgroup's rows: the column groups' sums meet in LDS (the ring is free now) and are added in a fixed order
        if (a.gap_part && dw_live) *reinterpret_cast<v4f*>(ring3 + cg * a.hid + 4 * q) = gsum;
        __syncthreads();
        if (a.gap_part && dt < quads) {
            v4f tot = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int g2 = 0; g2 < groups; ++g2) tot += *reinterpret_cast<const v4f*>(ring3 + g2 * a.hid + 4 * dt);
            *reinterpret_cast<v4f*>(a.gap_part + ((size_t)chunk * rblocks + ry) * a.hid + 4 * dt) = tot;
        }
    } else {
        // ------------------------------------------------------------------------------------------------ loaders (two waves)
        // keep THREE input rows in flight (a row is W Cin / 4 = 768 float4: six per thread): row t + 3 is requested at step t and copied
        // into the staging ring at step t + 2, two steps later — the HBM latency (~2 k cycles) is longer than a step.  Row indices are
        // clamped into the map instead of tested: no branch between the loads, so the waits on a register set count exactly the two
        // younger sets; rows outside the map are never read by the producers.
        const int lt = tid - 512;
        const int cq4 = a.Cin >> 2;
        const float* xc = a.x + ((size_t)chunk * a.H) * a.W * a.Cin + 4 * lt;
        constexpr int NLD = NCW == 3 ? 6 : 4;         // float4 per thread and row: W Cin / 4 = 768 (alpha = 1.5) or 512 (alpha = 1)
        int st_off[NLD];
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int idx = lt + 128 * e, pos = idx / cq4;
            st_off[e] = pos * PI + 4 * (idx - pos * cq4);
        }
        auto request = [&](v4f (&r)[NLD], int k) {
            int hr = h_lo + k;
            hr = hr < 0 ? 0 : (hr >= a.H ? a.H - 1 : hr);
            const float* src = xc + (size_t)hr * a.W * a.Cin;
#pragma unroll
            for (int e = 0; e < NLD; ++e) r[e] = *reinterpret_cast<const v4f*>(src + 512 * e);
        };
        auto deposit = [&](const v4f (&r)[NLD], int k) {
            float* dst = stage + (k % 3) * a.W * PI;
#pragma unroll
            for (int e = 0; e < NLD; ++e) *reinterpret_cast<v4f*>(dst + st_off[e]) = r[e];
        };
        v4f r0[NLD], r1[NLD], r2[NLD];
        request(r0, 0);
        request(r1, 1);
        request(r2, 2);
        deposit(r0, 0);
        __syncthreads();
        for (int t = 0; t < nsteps; t += 3) {       // at the top of step t: staging slot t % 3 holds row t, rows t + 1 and t + 2 are in flight
            request(r0, t + 3);
            deposit(r1, t + 1);
            __syncthreads();
            if (t + 1 < nsteps) {
                request(r1, t + 4);
                deposit(r2, t + 2);
                __syncthreads();
            }
            if (t + 2 < nsteps) {
                request(r2, t + 5);
                deposit(r0, t + 3);
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

template <int NW, int COUT, int S, bool RES>
void launch_strip(const DwPwArgs& a, hipStream_t s) {
    const long strips = (long)a.B * (a.OW / 16) * ((a.OH + a.TH - 1) / a.TH);
    hipLaunchKernelGGL((f32_strip_kernel<NW, COUT, S, RES>), dim3((unsigned)strips), dim3(64 * NW), 0, s, a);
}

}  // namespace

bool f32_strip_supported(const DwPwArgs& a) {
    if (!a.has_dw || a.gate || a.OW % 16 || a.sh != a.sw || (a.sh != 1 && a.sh != 2)) return false;
    if (a.res && (a.res != a.x || a.sh != 1 || a.Cin != a.Cout)) return false;
    const bool shape = (a.Cin == 32 && (a.Cout == 32 || a.Cout == 64)) || (a.Cin == 64 && (a.Cout == 64 || a.Cout == 128)) ||
                       (a.Cin == 128 && a.Cout == 128);  // eight waves per strip: ~145 registers, one workgroup per CU, still 0.07 vs 0.09 ms
    return shape && (long)a.H * a.W * a.Cin * 4 < 0x7fff0000L;
}

// rows per strip of the stand-alone depthwise kernel and the strips per chunk that follow from them (the squeeze-excite gate behind the stage
// adds up one partial sum per strip: f32_dw_stream_strips is what its scratch must hold per chunk and channel)
// (with partial sums the strip height must not depend on the batch size: a chunk's scores may not change with the size of the batch it sits in)
static void dw_stream_plan(int B, int C, int OH, int OW, bool batch_independent, int* cq_out, int* th_out) {
    int cq = 16;
    while ((C / 4) % cq) cq >>= 1;
    const int ncol = 64 / cq;
    const long per_row_block = (long)B * (C / (4 * cq)) * ((OW + ncol - 1) / ncol);
    int th = OH;
    while (th > 16) th = (th + 1) / 2;
    while (!batch_independent && th > 4 && per_row_block * ((OH + th - 1) / th) < 8192) th = (th + 1) / 2;
    if (const int v = g_opt.f32_strip_th; v >= 1) th = v < OH ? v : OH;
    *cq_out = cq;
    *th_out = th;
}

int f32_dw_stream_strips(int B, int C, int OH, int OW) {
    int cq, th;
    dw_stream_plan(B, C, OH, OW, true, &cq, &th);
    const int ncol = 64 / cq;
    return ((OW + ncol - 1) / ncol) * ((OH + th - 1) / th);
}

bool launch_f32_dw_stream(const float* x, float* y, int B, int H, int W, int C, int sh, int sw, int act, int OH, int OW, int pt, int pl,
                          const float* w, const float* bias, float* gap_part, hipStream_t s) {
    if (sh != sw || (sh != 1 && sh != 2) || C % 4 || (long)H * W * C * 4 >= 0x7fff0000L || (long)OH * OW * C * 4 >= 0x7fff0000L) return false;
    if (!g_opt.f32_strip) return false;
    int cq, th;
    dw_stream_plan(B, C, OH, OW, gap_part != nullptr, &cq, &th);
    DwStreamArgs a{x, y, w, bias, B, H, W, C, OH, OW, 0, pt, pl, act, cq, gap_part};
    const int ncol = 64 / cq;
    const long per_row_block = (long)B * (C / (4 * cq)) * ((OW + ncol - 1) / ncol);
    a.TH = th;
    const long waves = per_row_block * ((OH + th - 1) / th);
    if (sh == 1)
        hipLaunchKernelGGL(f32_dw_stream_kernel<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(f32_dw_stream_kernel<2>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
    return true;
}

// expand 1x1 + depthwise 3x3 of an inverted-residual block as one kernel (f32_pwdw_kernel): four waves x (2 position tiles x 3 channel tiles)
static int pwdw_ncw(int dW, int dC) {  // channel tiles per producer wave: four waves x (2 position tiles x NCW channel tiles) cover a hidden row
    if (dW % 32 || dW < 32) return 0;
    if (dC % 48 == 0 && (dW / 32) * (dC / 48) == 4) return 3;
    if (dC % 32 == 0 && (dW / 32) * (dC / 32) == 4) return 2;
    return 0;
}

bool f32_pwdw_supported(const DwPwArgs& e, int dH, int dW, int dC, int dsh, int dsw, int dOH, int dOW) {
    if (e.has_dw || e.res || e.gate || e.H != e.OH || e.W != e.OW || e.Cout != dC || e.H != dH || e.W != dW) return false;
    const int ncw = pwdw_ncw(dW, dC);
    if (dsh != dsw || (dsh != 1 && dsh != 2) || !ncw || e.Cin % 4) return false;
    const int nj = (e.Cin + 15) / 16;
    if (ncw == 3 ? (nj != 2 && nj != 3 && nj != 6) || dW * e.Cin != 3072 : (nj != 1 && nj != 2 && nj != 4) || dW * e.Cin != 2048) return false;
    if (dOH != (dH + dsh - 1) / dsh || dOW != (dW + dsw - 1) / dsw) return false;
    const size_t smem = ((size_t)4 * (dW + 2) * (dC + 4) + (size_t)3 * dW * (e.Cin + 4)) * sizeof(float);
    return smem <= 156 * 1024 && (long)e.H * e.W * e.Cin * 4 < 0x7fff0000L;
}

int f32_pwdw_rows(int dOH) {  // output rows per workgroup
    int rb = dOH;
    while (rb > 16) rb = (rb + 1) / 2;
    return rb;
}

bool launch_f32_pwdw(const DwPwArgs& e, const float* dw_w, const float* dw_b, float* y, int dsh, int dOH, int dOW, int dpt, int dpl, int dw_act,
                     const F32StemIn* stem, float* gap_part, hipStream_t s) {
    const size_t smem = ((size_t)4 * (e.W + 2) * (e.Cout + 4) + (size_t)3 * e.W * (e.Cin + 4)) * sizeof(float);
    const int rb = f32_pwdw_rows(dOH);
    F32PwDwArgs a{e.x, y, e.pw_w, e.pw_b, dw_w, dw_b, e.B, e.H, e.W, e.Cin, e.Cout, dOH, dOW, dpt, dpl, e.pw_act, dw_act, rb,
                  nullptr, nullptr, nullptr, 0, 0, 1, 1, 0, 0, 0, gap_part};
    if (stem) {
        a.fe = stem->fe; a.stem_w = stem->w; a.stem_b = stem->b; a.H0 = stem->H0; a.W0 = stem->W0; a.ssh = stem->sh; a.ssw = stem->sw;
        a.spt = stem->pt; a.spl = stem->pl; a.stem_act = stem->act;
    }
    const unsigned blocks = (unsigned)((long)e.B * ((dOH + rb - 1) / rb));
    const int nj = (e.Cin + 15) / 16, ncw = pwdw_ncw(e.W, e.Cout);
#define BN_PWDW(NJV, SV, NCWV)                                                                                                              \
    if (nj == NJV && dsh == SV && ncw == NCWV) {                                                                                            \
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(f32_pwdw_kernel<NJV, SV, NCWV>), smem)) return false;                            \
        hipLaunchKernelGGL((f32_pwdw_kernel<NJV, SV, NCWV>), dim3(blocks), dim3(stem ? 512 : 640), smem, s, a);  /* stem mode: no loader waves */                             \
        return true;                                                                                                                        \
    }
    BN_PWDW(2, 1, 3) BN_PWDW(2, 2, 3) BN_PWDW(3, 1, 3) BN_PWDW(3, 2, 3) BN_PWDW(6, 1, 3) BN_PWDW(6, 2, 3)
    BN_PWDW(1, 1, 2) BN_PWDW(1, 2, 2) BN_PWDW(2, 1, 2) BN_PWDW(2, 2, 2) BN_PWDW(4, 1, 2) BN_PWDW(4, 2, 2)
#undef BN_PWDW
    return false;
}

bool f32_front_strip_supported(int H0, int W0, int C, int N, int OH, int OW) {
    return C == 16 && N == 32 && OW % 16 == 0 && H0 == 2 * OH && W0 == 4 * OW && H0 <= 64 && (long)H0 * W0 * 4 < 0x7fff0000L;
}

void launch_f32_front_strip(F32FrontStripArgs a, hipStream_t s) {
    int th = a.OH;
    while (th > 16) th = (th + 1) / 2;
    while (th > 4 && (long)a.B * (a.OW / 16) * ((a.OH + th - 1) / th) < 4096) th = (th + 1) / 2;
    if (const int v = g_opt.f32_strip_th; v >= 1) th = v < a.OH ? v : a.OH;
    const bool staged = g_opt.f32_front_staged && a.OW / 16 <= 16 && a.W0 % 4 == 0;
    if (staged && g_opt.f32_strip_th < 1) th = a.OH < 8 ? a.OH : 8;  // 19 input rows per block (+19 % halo), 21 KB of LDS
    a.TH = th;
    const long blocks = (long)a.B * ((a.OH + th - 1) / th);
    if (staged) {
        const size_t smem = (size_t)(2 * th + 4) * (a.W0 + 8) * sizeof(float);
        if (smem <= 60000) {
            hipLaunchKernelGGL(f32_front_strip_kernel<true>, dim3((unsigned)blocks), dim3(64 * (a.OW / 16)), smem, s, a);
            return;
        }
    }
    const long waves = blocks * (a.OW / 16);
    hipLaunchKernelGGL(f32_front_strip_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
}

// front block + stage1_ds2 in one kernel: same geometry limits as the strip kernels, the whole chunk's input map in LDS
bool f32_front2_supported(const F32FrontStripArgs& f, const DwPwArgs& d) {
    if (!f32_front_strip_supported(f.H0, f.W0, 16, 32, f.OH, f.OW) || f.W0 % 4 || f.OW / 16 > 4) return false;
    if (!d.has_dw || d.gate || d.res != d.x || d.Cin != 32 || d.Cout != 32 || d.sh != 1 || d.sw != 1 || d.H != f.OH || d.W != f.OW ||
        d.OH != f.OH || d.OW != f.OW || d.pt != 1 || d.pl != 1)
        return false;
    const size_t smem = ((size_t)(f.H0 + 4) * (f.W0 + 8) + 2 * (f.OW + 2) * 36) * 4 + (size_t)(f.OW / 16) * 256 * 16;
    return smem <= 150 * 1024 && (long)f.OH * f.OW * 32 * 4 < 0x7fff0000L;
}

bool launch_f32_front2(const F32FrontStripArgs& f, const DwPwArgs& d, hipStream_t s) {
    const size_t smem = ((size_t)(f.H0 + 4) * (f.W0 + 8) + 2 * (f.OW + 2) * 36) * 4 + (size_t)(f.OW / 16) * 256 * 16;
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(f32_front2_kernel), 150 * 1024)) return false;
    F32Front2Args A{f, d.dw_w, d.dw_b, d.pw_w, d.pw_b, d.y, d.dw_act, d.pw_act};
    hipLaunchKernelGGL(f32_front2_kernel, dim3((unsigned)f.B), dim3(192 * (f.OW / 16)), smem, s, A);
    return true;
}

void launch_f32_strip(DwPwArgs a, hipStream_t s) {
    const int nw = a.Cin / 16;
    // rows per strip (measured at B = 1024): 16 on the 32-row maps (32: fewer, longer strips leave CUs idle at the end; 8: the
    // two-row prologue weighs 25 %), the whole map below that; shorter only while the launch would not fill the chip once
    int th = a.OH;
    while (th > 16) th = (th + 1) / 2;
    while (th > 4 && (long)a.B * (a.OW / 16) * ((a.OH + th - 1) / th) * nw < 4096) th = (th + 1) / 2;
    if (const int v = g_opt.f32_strip_th; v >= 1) th = v < a.OH ? v : a.OH;  // tests: force the rows per strip
    a.TH = th;
    const bool res = a.res != nullptr;
#define BN_FSTRIP(NW, CO, ST, RS) \
    if (nw == NW && a.Cout == CO && a.sh == ST && res == RS) return launch_strip<NW, CO, ST, RS>(a, s);
    BN_FSTRIP(2, 32, 1, true)
    BN_FSTRIP(4, 64, 1, true)
    BN_FSTRIP(2, 32, 1, false)
    BN_FSTRIP(4, 64, 1, false)
    BN_FSTRIP(2, 64, 1, false)
    BN_FSTRIP(2, 32, 2, false)
    BN_FSTRIP(2, 64, 2, false)
    BN_FSTRIP(4, 64, 2, false)
    BN_FSTRIP(4, 128, 1, false)
    BN_FSTRIP(8, 128, 1, true)
    BN_FSTRIP(8, 128, 1, false)
    BN_FSTRIP(4, 128, 2, false)
#undef BN_FSTRIP
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_f32_strip() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&f32_dw_stream_kernel<1>));
}

}  // namespace bn
