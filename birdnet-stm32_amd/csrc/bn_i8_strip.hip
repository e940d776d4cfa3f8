// bn_i8_strip.hip — wave-autonomous INT8 depthwise-separable block for the wide early stages (Cin, Cout in {32, 64}):
//
//   DEPTHWISE_CONV_2D 3x3 (ReLU6 clamp) -> CONV_2D 1x1 on the int8 matrix cores [-> TFLite ADD with the block input]
//
// Same integer semantics as bn_i8_fused.hip / bn_i8.hip (bit-identical results); what changes is who does what.
// The generic fused kernel spends ~950 vector instructions per wave and 64-position tile, most of them on index
// arithmetic, per-tap bounds tests and 64-bit addresses; these layers are bound by the vector ALU, so this kernel is
// organised around issuing as few vector instructions as the arithmetic allows:
//
//   * one WAVE owns a strip of 16 output columns and walks TH output rows downwards; the four waves of a workgroup
//     share nothing but read-only tables, so there is no barrier after the prologue;
//   * lane (n, kq) = (column n of the strip, channel group kq) holds CL = Cin/4 consecutive channels.  Its depthwise
//     outputs ARE its B fragment of v_mfma_i32_16x16x32_i8 (Cin = 32) / 16x16x64 (Cin = 64): no LDS staging;
//   * the pointwise weights are the A operand with their rows permuted by the packer so that the 4 accumulator
//     registers of tile t in lane (n, q) are output channels (Cout/4) q + 4 t + 0..3: a lane ends up with Cout/4
//     CONSECUTIVE output channels of its own column — one 8/16-byte store, no LDS transpose — which for Cin = Cout is
//     exactly the channel group it read, so the residual of the ADD is the centre tap it already holds;
//   * input rows stream through a 3-row register window in byte-transposed form (per channel the three taps of a row in
//     one dword): a new row costs 3 buffer loads (scalar row offset, no address arithmetic) and 6 byte-permutes per
//     channel quad, and serves three output rows; 12 v_dot4_i32_i8 per quad and output row do the 36 MACs;
//   * vertical padding is a wave-uniform branch, horizontal padding two selects per quad and row (lanes 0 / 15 of the
//     outer strips); rows are prefetched two steps ahead;
//   * requantisation constants come prepared by the packer (rounding offset and zero point folded into one addend:
//     ((v + c1 + (v >> 31)) >> e with c1 = 2^(e-1) + (zp << e)), the per-channel ones from LDS as 128-bit reads.
//
// Blocks wider than 64 channels (or with more than 64 outputs) split the CHANNELS over the NW waves of a workgroup: wave w
// runs the depthwise stage for input channels 32 w .. 32 w + 31 of the same strip, the waves swap their 8-byte B fragments
// through LDS (one barrier per output row, double-buffered), and each wave multiplies all NW fragments into ITS Cout/NW
// output channels — per-lane registers stay those of the 32-channel kernel whatever the width.
//
// The packer (models/_lower_i8.py: strip_constants) only emits the constant block when every multiplier is >= 0, every
// shift is a right shift of 1..22 bits (dead channels with larger shifts are canonicalised to multiplier 0 when their
// accumulator bound proves the result is the zero point) — bn_api.hip falls back to the generic kernel otherwise.
#include <stdlib.h>

#include "bn_kernels.h"
#include "bn_requant.h"

namespace bn {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int perm(int s0, int s1, uint32_t sel) { return (int)__builtin_amdgcn_perm((uint32_t)s0, (uint32_t)s1, sel); }
__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
// clamp as ONE instruction; lo <= hi.  (The compiler cannot prove lo <= hi for run-time bounds and emits compare + select + min.)
__device__ __forceinline__ int med3(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}
// first link of a dot4 chain in the three-address form (no move of the bias into the accumulator)
__device__ __forceinline__ int dot4_first(int a, int b, int c) {
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// q = ((srdhm(x, m) + c1 + sign) >> e), c1 = 2^(e-1) + (zp << e): RoundingDivideByPOT(SRDHM(x, m), e) + zp (bn_requant.h)
__device__ __forceinline__ int rq(int x, int m, int c1, int e) {
    const int v = srdhm_pos(x, m);
    return (v + c1 + (v >> 31)) >> e;
}

// Where the clamp's lower bound is at or above the zero point (ReLU / ReLU6 outputs: the packer checks it) a negative v gives a result
// <= zero point with or without the sign term (v + 2^(e-1) < 2^e) and both clamp to the same bound: (v + c1) >> e,
// and with the addend folded into the 64-bit multiply-add: ((x*m + 2^30) >> 31 + c1) >> e == (x*m + 2^30 + c1 * 2^31) >> (31 + e)
// (nested floors) == hi32(x*m + C) >> (e - 1) for e >= 1: v_mad_i64_i32 with a per-channel 64-bit constant C = 2^30 + c1 * 2^31, then ONE
// arithmetic shift of the high dword — 2 instructions + clamp instead of 5 + clamp.  rq64(c1) builds C; kernels keep (C, e - 1) per channel.
__device__ __forceinline__ long rq64(int c1) { return ((long)c1 << 31) + 0x40000000L; }
// The four shifts of a channel quad are packed into the bytes of ONE register (SDWA picks byte `e` as the shift
// count: no unpacking instruction, three registers less per quad)
__device__ __forceinline__ int pack_shifts(v4i sh) { return sh.x | (sh.y << 8) | (sh.z << 16) | (sh.w << 24); }
__device__ __forceinline__ int rq_hi(int x, int m, long c, int e1_packed, int e) {
    const int hi = (int)(((long)x * (long)m + c) >> 32);
    int r;
    switch (e) {  // e is a compile-time constant after unrolling
        case 0: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(e1_packed), "v"(hi)); break;
        case 1: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(e1_packed), "v"(hi)); break;
        case 2: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(e1_packed), "v"(hi)); break;
        default: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(e1_packed), "v"(hi)); break;
    }
    return r;
}

template <int QL> struct RawRow { int t[3][QL]; };   // three taps (columns j = 0..2) of one input row, QL dwords each
template <int QL> struct TRow { int c[QL][4]; };     // per channel: bytes (tap0, tap1, tap2, 0)

template <int QL>
__device__ __forceinline__ RawRow<QL> load_row(__amdgpu_buffer_rsrc_t rsrc, const int voff[3], int soff) {
    RawRow<QL> r;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if constexpr (QL == 2) {
            const v2i v = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[j], soff, 0));
            r.t[j][0] = v.x; r.t[j][1] = v.y;
        } else {
            const v4i v = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j], soff, 0));
            r.t[j][0] = v.x; r.t[j][1] = v.y; r.t[j][2] = v.z; r.t[j][3] = v.w;
        }
    }
    return r;
}

// CW = input channels per wave (32 or 64; 32 when NW > 1), NW = waves sharing a strip (channel split), COUT = all output channels.
// Strips per workgroup: 4 single-wave strips, or one NW-wave strip; with the ADD 8 / 2, so that the 64 KB table is shared by 512
// threads and two workgroups (16 waves) fit a CU.
// W8: maps 8 columns wide (stage 4).  The 16 positions of a wave are 2 row blocks x 8 columns: lanes n < 8 walk the upper half of
// the rows, lanes n >= 8 the lower half, each half streaming its own input rows (row offsets and row padding become per-lane).
template <int CW, int NW, int COUT, int S, bool ADD, bool W8 = false>
// (registers: four waves per SIMD for the shipped shapes; the stride-1 blocks without an ADD that keep 64 output channels per wave and
// the 8-column variants — shapes only other topologies reach — spilled 6-26 registers at that cap and are built for three)
__global__ __launch_bounds__(64 * NW * (ADD ? 8 / NW : (NW == 1 ? 4 : 1)))
__attribute__((amdgpu_waves_per_eu(CW == 32 ? (((S == 1 && !ADD && COUT / NW >= 64) || W8) ? 3 : 4) : 2)))
void i8_strip_kernel(Strip8Args a) {
    constexpr int CIN = CW * NW, CL = CW / 4, QL = CL / 4;
    constexpr int CWO = COUT / NW, NT = CWO / 16, COL = CWO / 4;  // output channels per wave / tiles per wave / per lane
    constexpr int SPB = ADD ? 8 / NW : (NW == 1 ? 4 : 1);
    constexpr int NTHREADS = 64 * NW * SPB;
    static_assert(NW == 1 || CW == 32, "the channel split works on 32-channel slices");
    static_assert(!ADD || (CIN == COUT && S == 1), "the residual is the block input");
    static_assert(NT == 2 || NT == 4, "8 or 16 output channels per lane");
    // constant block (int32 words), see strip_constants() in models/_lower_i8.py; every section starts with the wave index
    constexpr int nDWW = 4 * QL * 12, nDWB = 4 * QL * 4, nDWC = 4 * QL * 12, nPWA = NT * NW * 64 * QL, nPWB = 4 * NT * 4, nPWC = 4 * NT * 12;
    constexpr int kDWW = 0;                   // [w][kq][ql][row 3][e 4]
    constexpr int kDWB = kDWW + NW * nDWW;    // [w][kq][ql][e]
    constexpr int kDWC = kDWB + NW * nDWB;    // [w][kq][ql][kind 3][e]: multiplier, c1, shift
    constexpr int kPWA = kDWC + NW * nDWC;    // [w][t][ks][lane][QL]
    constexpr int kPWB = kPWA + NW * nPWA;    // [w][q][t][reg]
    constexpr int kPWC = kPWB + NW * nPWB;    // [w][q][t][kind 3][reg]
    // LDS (dynamic: the ADD table alone is 64 KB): [ADD table 65536 B] [c_dw] [c_pw] [xchg]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    const unsigned char* add_tab = lds_all;   // [residual byte pattern][own value + 128] -> output byte (packer: add_table)
    v4i* c_dw = reinterpret_cast<v4i*>(lds_all + (ADD ? 65536 : 0));
    v4i* c_pw = c_dw + NW * nDWC / 3;                           // staged as 16 words per quad: multiplier, (C lo, C hi) x 4, shift - 1 (rq_hi)
    v4i* c_dww = c_pw + NW * nPWC / 3;                          // depthwise weights (only when the waves split the channels)
    constexpr bool DWW_LDS = NW > 1;                            // 24 long-lived registers less per lane; LDS reads are free here
    v2i* xchg = reinterpret_cast<v2i*>(c_dww + (DWW_LDS ? NW * nDWW / 4 : 0));  // [buffer][wave of the workgroup][lane]: B fragments
    const int tid = threadIdx.x;
    {
        const v4i* src = reinterpret_cast<const v4i*>(a.cst);
        // requantisation constants: (multiplier, c1, shift) per quad in the blob -> (multiplier, 64-bit addends, shift - 1) for rq_hi;
        // the pointwise stage of a residual block keeps (multiplier, c1, shift): its values feed the ADD table with their sign (rq)
        auto stage = [&](v4i* dst, const v4i* q3, int n_quads, bool fold) {
            for (int i = tid; i < n_quads; i += NTHREADS) {
                const v4i m = q3[3 * i], c1 = q3[3 * i + 1], sh = q3[3 * i + 2];
                dst[4 * i] = m;
                if (fold) {
                    const long c[4] = {rq64(c1.x), rq64(c1.y), rq64(c1.z), rq64(c1.w)};
                    dst[4 * i + 1] = (v4i){(int)c[0], (int)(c[0] >> 32), (int)c[1], (int)(c[1] >> 32)};
                    dst[4 * i + 2] = (v4i){(int)c[2], (int)(c[2] >> 32), (int)c[3], (int)(c[3] >> 32)};
                    dst[4 * i + 3] = (v4i){pack_shifts(sh - 1), 0, 0, 0};
                } else {
                    dst[4 * i + 1] = c1;
                    dst[4 * i + 2] = sh;
                    dst[4 * i + 3] = (v4i){0, 0, 0, 0};
                }
            }
        };
        stage(c_dw, src + kDWC / 4, NW * nDWC / 12, true);
        stage(c_pw, src + kPWC / 4, NW * nPWC / 12, !ADD);
        if constexpr (DWW_LDS)
            for (int i = tid; i < NW * nDWW / 4; i += NTHREADS) c_dww[i] = src[kDWW / 4 + i];
        if constexpr (ADD) {
            const v4i* tsrc = reinterpret_cast<const v4i*>(a.add_tab);
            v4i* tdst = reinterpret_cast<v4i*>(lds_all);
            for (int i = tid; i < 4096; i += NTHREADS) tdst[i] = tsrc[i];
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = NW == 1 ? 0 : wave % NW;  // channel slice of this wave
    const int n = lane & 15, kq = lane >> 4;
    const int strips_x = W8 ? 1 : a.OW >> 4;
    const int rblocks = (a.OH + a.TH - 1) / a.TH;
    int wid, chunk;
    if constexpr (NW == 1) {
        wid = xcd_tile(blockIdx.x, gridDim.x) * SPB + wave;
        if (wid >= a.B * strips_x * rblocks) return;  // no barrier after this point
    } else {
        wid = xcd_tile(blockIdx.x, gridDim.x);         // (strip column, row block, group of SPB chunks): the strips of a workgroup
    }                                                  // share the row block, hence the number of barriers
    const int sx = wid % strips_x;
    wid /= strips_x;
    const int ry = wid % rblocks;
    chunk = wid / rblocks;
    if constexpr (NW > 1 && SPB > 1) {
        chunk = chunk * SPB + wave / NW;
        if (chunk >= a.B) chunk = a.B - 1;             // odd batch: the spare strip repeats the last chunk (same bytes written twice)
    }
    const int oh0 = ry * a.TH;
    const int nrows = (a.OH - oh0) < a.TH ? (a.OH - oh0) : a.TH;
    const int steps = W8 ? nrows >> 1 : nrows;           // W8: two row blocks of nrows / 2 rows side by side (nrows even, see launcher)
    const int rofs = W8 ? (n >> 3) * steps : 0;          // first output row of this lane's block, relative to oh0
    const int ow = W8 ? (n & 7) : sx * 16 + n;

    // per-lane constants in registers
    int dww[QL][3][4], dwb[QL][4];
    {
        const v4i* p = reinterpret_cast<const v4i*>(a.cst + kDWW + w * nDWW) + kq * QL * 3;
        if constexpr (!DWW_LDS) {
#pragma unroll
            for (int ql = 0; ql < QL; ++ql)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const v4i v = p[ql * 3 + i];
                    dww[ql][i][0] = v.x; dww[ql][i][1] = v.y; dww[ql][i][2] = v.z; dww[ql][i][3] = v.w;
                }
        }
        const v4i* pb = reinterpret_cast<const v4i*>(a.cst + kDWB + w * nDWB) + kq * QL;
#pragma unroll
        for (int ql = 0; ql < QL; ++ql) {
            const v4i v = pb[ql];
            dwb[ql][0] = v.x; dwb[ql][1] = v.y; dwb[ql][2] = v.z; dwb[ql][3] = v.w;
        }
    }
    int pwa[NT][NW][QL];
    v4i pwb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ks = 0; ks < NW; ++ks)
#pragma unroll
            for (int ql = 0; ql < QL; ++ql) pwa[t][ks][ql] = a.cst[kPWA + w * nPWA + ((t * NW + ks) * 64 + lane) * QL + ql];
        pwb[t] = reinterpret_cast<const v4i*>(a.cst + kPWB + w * nPWB)[kq * NT + t];
    }
    const v4i* my_dww = c_dww + w * (nDWW / 4) + kq * QL * 3;
    const v4i* my_dw = c_dw + w * (nDWC / 3);
    const v4i* my_pw = c_pw + w * (nPWC / 3);

    const int zp4 = (a.zp_in & 0xff) * 0x01010101;
    const int zprow = (a.zp_in & 0xff) * 0x00010101;
    const int iw0 = ow * S - a.pl;
    const bool left_pad = iw0 < 0, right_pad = iw0 + 2 >= a.W;  // tap j = 1 is always inside (checked by the packer)
    const int in_chunk_bytes = a.H * a.W * CIN;
    const int row_bytes = a.W * CIN;
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x) + (size_t)chunk * in_chunk_bytes, 0, in_chunk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * COUT, 0, a.OH * a.OW * COUT, 0x00020000);
    // padding columns load a valid neighbour instead (their value is replaced by the zero point below)
    const int coff = CW * w + CL * kq;
    const int voff_in[3] = {(left_pad ? 0 : iw0) * CIN + coff, (iw0 + 1) * CIN + coff, (right_pad ? a.W - 1 : iw0 + 2) * CIN + coff};
    const int voff_out = (rofs * a.OW + ow) * COUT + CWO * w + COL * kq;
    const int ir0 = (oh0 + rofs) * S - a.pt;  // first input row of the strip (W8: of this lane's row block)
    const int rows_needed = S * (steps - 1) + 3;

    RawRow<QL> raw[2];
    TRow<QL> T[3];
    int cen[3][QL];

    // row_ok is wave-uniform except in W8 mode, where the two lane halves stream different rows
    auto row_ok = [&](int rr) { const int ir = ir0 + rr; return rr < rows_needed && ir >= 0 && ir < a.H; };
    auto issue = [&](int slot, int rr) {
        if constexpr (W8) {
            if (rr < rows_needed) {
                int ir = ir0 + rr;
                ir = ir < 0 ? 0 : (ir >= a.H ? a.H - 1 : ir);  // padding rows load a valid row, replaced in consume()
                const int vo[3] = {voff_in[0] + ir * row_bytes, voff_in[1] + ir * row_bytes, voff_in[2] + ir * row_bytes};
                raw[slot] = load_row<QL>(rs_in, vo, 0);
            }
        } else {
            if (row_ok(rr)) raw[slot] = load_row<QL>(rs_in, voff_in, (ir0 + rr) * row_bytes);
        }
    };
    auto consume = [&](int slot, int rr, int ti) {
        const bool ok = row_ok(rr);
        if (W8 ? rr < rows_needed : ok) {
#pragma unroll
            for (int ql = 0; ql < QL; ++ql) {
                const int r0 = (left_pad || (W8 && !ok)) ? zp4 : raw[slot].t[0][ql];
                const int r1 = (W8 && !ok) ? zp4 : raw[slot].t[1][ql];
                const int r2 = (right_pad || (W8 && !ok)) ? zp4 : raw[slot].t[2][ql];
                const int lo = perm(r1, r0, 0x05010400u);  // r0.0 r1.0 r0.1 r1.1
                const int hi = perm(r1, r0, 0x07030602u);  // r0.2 r1.2 r0.3 r1.3
                T[ti].c[ql][0] = perm(r2, lo, 0x0c040100u);
                T[ti].c[ql][1] = perm(r2, lo, 0x0c050302u);
                T[ti].c[ql][2] = perm(r2, hi, 0x0c060100u);
                T[ti].c[ql][3] = perm(r2, hi, 0x0c070302u);
                cen[ti][ql] = r1;
            }
        } else {
#pragma unroll
            for (int ql = 0; ql < QL; ++ql) {
#pragma unroll
                for (int e = 0; e < 4; ++e) T[ti].c[ql][e] = zprow;
                cen[ti][ql] = zp4;
            }
        }
    };

    // One output row from window rows (i0, i1, i2) in three parts: depthwise (vector ALU) -> B fragment; matrix cores -> int32
    // accumulators; epilogue (vector ALU: requantisation, ADD table, store).  Single-wave strips run them back to back.  When the
    // waves split the channels (NW > 1) the row loop is software-pipelined: a wave writes its B fragment of row k, then does the
    // EPILOGUE OF ROW k - 1, and only then meets the others at the barrier — the LDS write and the other waves' arrival are
    // hidden behind ~80 vector instructions, and the MFMAs of row k are hidden behind the depthwise stage of row k + 1.
    constexpr bool PIPE = NW > 1 && NT == 2 && !W8;  // (16 accumulator registers of the four-tile variants spill when carried over)
    auto dw_part = [&](int i0, int i1, int i2, int (&bfrag)[QL]) {
        asm volatile("" ::: "memory");  // the per-channel requantisation constants are re-read from LDS every row instead of pinning
                                        // ~48 registers (the kernel is bound by vector-ALU issue, LDS reads are free)
#pragma unroll
        for (int ql = 0; ql < QL; ++ql) {
            const v4i m = my_dw[(kq * QL + ql) * 4 + 0], c01 = my_dw[(kq * QL + ql) * 4 + 1], c23 = my_dw[(kq * QL + ql) * 4 + 2];
            const int e1 = reinterpret_cast<const int*>(my_dw + (kq * QL + ql) * 4 + 3)[0];
            const long cc[4] = {__builtin_bit_cast(long, (v2i){c01.x, c01.y}), __builtin_bit_cast(long, (v2i){c01.z, c01.w}),
                                __builtin_bit_cast(long, (v2i){c23.x, c23.y}), __builtin_bit_cast(long, (v2i){c23.z, c23.w})};
            int qv[4];
            v4i w0, w1, w2;
            if constexpr (DWW_LDS) {
                w0 = my_dww[ql * 3 + 0]; w1 = my_dww[ql * 3 + 1]; w2 = my_dww[ql * 3 + 2];
            } else {
                w0 = (v4i){dww[ql][0][0], dww[ql][0][1], dww[ql][0][2], dww[ql][0][3]};
                w1 = (v4i){dww[ql][1][0], dww[ql][1][1], dww[ql][1][2], dww[ql][1][3]};
                w2 = (v4i){dww[ql][2][0], dww[ql][2][1], dww[ql][2][2], dww[ql][2][3]};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int acc = dot4_first(T[i0].c[ql][e], w0[e], dwb[ql][e]);
                acc = dot4(T[i1].c[ql][e], w1[e], acc);
                acc = dot4(T[i2].c[ql][e], w2[e], acc);
                qv[e] = med3(rq_hi(acc, m[e], cc[e], e1, e), a.dw_lo, a.dw_hi);
            }
            bfrag[ql] = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
        }
    };
    auto mma_part = [&](const int (&bfrag)[QL], const v2i* buf, v4i (&acc)[NT]) {
        long bfs[NW];  // B fragments of every channel slice (QL == 2 whenever NW > 1)
        if constexpr (NW > 1) {
#pragma unroll
            for (int ks = 0; ks < NW; ++ks) bfs[ks] = __builtin_bit_cast(long, buf[ks * 64 + lane]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = pwb[t];
            if constexpr (NW > 1) {
#pragma unroll
                for (int ks = 0; ks < NW; ++ks) {
                    const long af = ((long)(uint32_t)pwa[t][ks][1] << 32) | (uint32_t)pwa[t][ks][0];
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(af, bfs[ks], acc[t], 0, 0, 0);
                }
            } else if constexpr (QL == 2) {
                const long af = ((long)(uint32_t)pwa[t][0][1] << 32) | (uint32_t)pwa[t][0][0];
                const long bf = ((long)(uint32_t)bfrag[1] << 32) | (uint32_t)bfrag[0];
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(af, bf, acc[t], 0, 0, 0);
            } else {
                const v4i af = {pwa[t][0][0], pwa[t][0][1], pwa[t][0][2], pwa[t][0][3]};
                const v4i bf = {bfrag[0], bfrag[1], bfrag[2], bfrag[3]};
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[t], 0, 0, 0);
            }
        }
    };
    auto epi_part = [&](const v4i (&acc)[NT], const int (&cenv)[QL], int oh) {
        int outw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // ADD: (multiplier, c1, shift, -); else (multiplier, C01, C23, packed shifts - 1)
            const v4i m = my_pw[(kq * NT + t) * 4 + 0], c1 = my_pw[(kq * NT + t) * 4 + 1], sh = my_pw[(kq * NT + t) * 4 + 2];
            const int e1 = ADD ? 0 : reinterpret_cast<const int*>(my_pw + (kq * NT + t) * 4 + 3)[0];
            const long cc[4] = {__builtin_bit_cast(long, (v2i){c1.x, c1.y}), __builtin_bit_cast(long, (v2i){c1.z, c1.w}),
                                __builtin_bit_cast(long, (v2i){sh.x, sh.y}), __builtin_bit_cast(long, (v2i){sh.z, sh.w})};
            int qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // ADD: value + 128 (table index, any sign: full rounding), else the int8 value behind a ReLU6 clamp
                int v = med3(ADD ? rq(acc[t][e], m[e], c1[e], sh[e]) : rq_hi(acc[t][e], m[e], cc[e], e1, e), a.pw_lo, a.pw_hi);
                if constexpr (ADD)  // the whole TFLite ADD (two input rescales, sum, output rescale, clamp) is a function of two bytes
                    v = add_tab[(uint32_t)perm(cenv[t % QL], v, 0x0c0c0400u + (e << 8))];
                qv[e] = v;
            }
            outw[t] = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
        }
        const int soff = oh * a.OW * COUT;
        if constexpr (NT == 2) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(int)))) int, (v2i){outw[0], outw[1]}), rs_out, voff_out, soff, 0);
        } else {
            v4i ov = {outw[0], outw[1], outw[2], outw[3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(int)))) int, ov), rs_out, voff_out, soff, 0);
            asm volatile("s_nop 1" : "+v"(ov));  // the data registers of a 16-byte store are not rewritten right behind it (bn_f32_strip.hip: store16)
        }
    };
    v4i pacc[NT];     // pipelined form: accumulators, centre taps and row of the previous step
    int pcen[QL] = {};
    int poh = 0;
    auto emit = [&](int i0, int i1, int i2, int oh, int step) {
        int bfrag[QL];
        dw_part(i0, i1, i2, bfrag);
        if constexpr (NW > 1 && !PIPE) {
            v2i* buf = xchg + ((step & 1) * SPB + wave / NW) * (NW * 64);
            buf[w * 64 + lane] = (v2i){bfrag[0], bfrag[1]};
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            v4i acc[NT];
            mma_part(bfrag, buf, acc);
            epi_part(acc, cen[i1], oh);
        } else if constexpr (PIPE) {
            v2i* buf = xchg + ((step & 1) * SPB + wave / NW) * (NW * 64);
            buf[w * 64 + lane] = (v2i){bfrag[0], bfrag[1]};
            if (step > 0) epi_part(pacc, pcen, poh);
            // LDS only: the prefetched global loads stay in flight across the barrier (a __syncthreads would drain them)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            mma_part(bfrag, buf, pacc);
#pragma unroll
            for (int ql = 0; ql < QL; ++ql) pcen[ql] = cen[i1][ql];
            poh = oh;
        } else {
            v4i acc[NT];
            mma_part(bfrag, nullptr, acc);
            epi_part(acc, cen[i1], oh);
        }
    };

    // rows consumed before the first output row: 3 - S; afterwards S per step.  Relative row rr lives in raw slot rr & 1 and in
    // window slot rr % 3; a slot is reloaded with row rr + 2 as soon as it has been transposed.
    constexpr int P = 3 - S;
    issue(0, 0);
    issue(1, 1);
#pragma unroll
    for (int rr = 0; rr < P; ++rr) {
        consume(rr & 1, rr, rr % 3);
        issue(rr & 1, rr + 2);
    }
    constexpr int U = 6 / S;  // unroll period: window slots (3) x raw slots (2)
    for (int k = 0; k < steps; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u >= steps) break;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int rs = P + S * u + s;  // static part of the relative row index (k is a multiple of 6 rows)
                consume(rs & 1, S * k + rs, rs % 3);
                issue(rs & 1, S * k + rs + 2);
            }
            emit((S * u) % 3, (S * u + 1) % 3, (S * u + 2) % 3, oh0 + k + u, k + u);
        }
    }
    if constexpr (PIPE) {
        if (steps > 0) epi_part(pacc, pcen, poh);  // the last row's epilogue
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// stage1_ds2 of the shipped graph (32 -> 32 channels, stride 1, residual ADD) with the DEPTHWISE 3x3 ON THE MATRIX CORES — the streaming sibling
// of bn_i8_tail2.hip's blocks (round 5).  Same constant block, same strip walk, same pointwise stage / ADD table / store as
// i8_strip_kernel<32, 1, 32, 1, true>; what changes is the depthwise stage:
//   * the B operand of v_mfma_i32_16x16x64_i8 is the NHWC input as it lies in memory: lane (n, g) loads the 16 channels of channel tile ct at
//     column n + g - 1 of the input row (two 16-byte loads per row instead of three 8-byte ones; g = 3 meets zero weights), contraction index
//     16 * (window column) + channel; the A operand is the row's taps as a block-diagonal matrix (lane (m, g): byte m of its 16 = w[row][g][16 ct + m]),
//     built from the packer's tap dwords at kernel start: 2 channel tiles x 3 window rows = 24 registers, what the tap dwords took;
//   * three accumulating matrix instructions per channel tile and output row replace 6 byte permutes + 12 v_dot4 + the transposed window
//     (66 -> 30 vector instructions per row for the depthwise stage; the requantisation stays);
//   * the results — lane (n, g): channels 16 ct + 4 g .. + 3 — are the pointwise product's B fragment with its K order permuted
//     (slot 8 g + j <-> channel 4 g + j, 8 g + 4 + j <-> 16 + 4 g + j); the A fragments are gathered from the packer's in that order;
//   * padding is explicit: rows outside the map are a register of zero points (wave-uniform), the two border lanes of the outer strips select
//     the zero point for their outside column (eight selects per row, in the outer strips only) — no bias variants;
//   * the residual of the ADD is no longer a tap the lane holds: one more 8-byte load per row (the lines are in L1).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void i8_strip_mf_kernel(Strip8Args a) {
    constexpr int CIN = 32, COUT = 32, QL = 2, NT = 2, NW = 1, SPB = 8, NTHREADS = 512;
    constexpr int nDWW = 4 * QL * 12, nDWB = 4 * QL * 4, nDWC = 4 * QL * 12, nPWA = NT * NW * 64 * QL, nPWB = 4 * NT * 4;
    constexpr int kDWW = 0, kDWB = kDWW + nDWW, kDWC = kDWB + nDWB, kPWA = kDWC + nDWC, kPWB = kPWA + nPWA, kPWC = kPWB + nPWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    const unsigned char* add_tab = lds_all;
    v4i* c_dw = reinterpret_cast<v4i*>(lds_all + 65536);   // per channel quad: multiplier, (C lo, C hi) x 4, packed shifts - 1 (rq_hi)
    v4i* c_pw = c_dw + nDWC / 3;                            // per output quad: multiplier, c1, shift (rq: the value + 128 indexes the ADD table)
    const int tid = threadIdx.x;
    {
        const v4i* src = reinterpret_cast<const v4i*>(a.cst);
        for (int i = tid; i < nDWC / 12; i += NTHREADS) {
            const v4i m = src[kDWC / 4 + 3 * i], c1 = src[kDWC / 4 + 3 * i + 1], sh = src[kDWC / 4 + 3 * i + 2];
            const long c[4] = {rq64(c1.x), rq64(c1.y), rq64(c1.z), rq64(c1.w)};
            c_dw[4 * i] = m;
            c_dw[4 * i + 1] = (v4i){(int)c[0], (int)(c[0] >> 32), (int)c[1], (int)(c[1] >> 32)};
            c_dw[4 * i + 2] = (v4i){(int)c[2], (int)(c[2] >> 32), (int)c[3], (int)(c[3] >> 32)};
            c_dw[4 * i + 3] = (v4i){pack_shifts(sh - 1), 0, 0, 0};
        }
        for (int i = tid; i < 4 * NT; i += NTHREADS) {
            c_pw[4 * i] = src[kPWC / 4 + 3 * i];
            c_pw[4 * i + 1] = src[kPWC / 4 + 3 * i + 1];
            c_pw[4 * i + 2] = src[kPWC / 4 + 3 * i + 2];
            c_pw[4 * i + 3] = (v4i){0, 0, 0, 0};
        }
        const v4i* tsrc = reinterpret_cast<const v4i*>(a.add_tab);
        v4i* tdst = reinterpret_cast<v4i*>(lds_all);
        for (int i = tid; i < 4096; i += NTHREADS) tdst[i] = tsrc[i];
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int strips_x = a.OW >> 4;
    const int rblocks = (a.OH + a.TH - 1) / a.TH;
    int wid = xcd_tile(blockIdx.x, gridDim.x) * SPB + wave;
    if (wid >= a.B * strips_x * rblocks) return;  // no barrier after this point
    const int sx = wid % strips_x;
    wid /= strips_x;
    const int ry = wid % rblocks;
    const int chunk = wid / rblocks;
    const int oh0 = ry * a.TH;
    const int steps = (a.OH - oh0) < a.TH ? (a.OH - oh0) : a.TH;
    const int ow = sx * 16 + n;

    // depthwise A operands: lane (m = n, g) of channel tile ct, window row dy: w[dy][g][16 ct + m] at byte m, zeros elsewhere (g = 3: all zero)
    v4i dwa[2][3];
    v4i dwb[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int c = 16 * ct + n;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int taps = a.cst[kDWW + (((c >> 3) * QL + ((c >> 2) & 1)) * 3 + dy) * 4 + (c & 3)];   // bytes (tap0, tap1, tap2, 0) of channel c
            const int b = g < 3 ? ((taps >> (8 * g)) & 0xff) << (8 * (n & 3)) : 0;
            dwa[ct][dy] = (v4i){(n >> 2) == 0 ? b : 0, (n >> 2) == 1 ? b : 0, (n >> 2) == 2 ? b : 0, (n >> 2) == 3 ? b : 0};
        }
        dwb[ct] = reinterpret_cast<const v4i*>(a.cst + kDWB)[4 * ct + g];   // channels 16 ct + 4 g .. + 3
    }
    // pointwise A fragments in the K order of the depthwise results (see above); bias as in i8_strip_kernel
    long pwa[NT];
    v4i pwb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t lo = (uint32_t)a.cst[kPWA + (t * 64 + (n + 16 * (g >> 1))) * QL + (g & 1)];
        const uint32_t hi = (uint32_t)a.cst[kPWA + (t * 64 + (n + 16 * (2 + (g >> 1)))) * QL + (g & 1)];
        pwa[t] = (long)(((unsigned long)hi << 32) | lo);
        pwb[t] = reinterpret_cast<const v4i*>(a.cst + kPWB)[g * NT + t];
    }

    const int zp4 = (a.zp_in & 0xff) * 0x01010101;
    const v4i zrow = {zp4, zp4, zp4, zp4};
    const int icol = ow + g - 1;                                     // the input column this lane reads (g = 3: as g = 2, weights are zero)
    const bool outside = g < 3 && (icol < 0 || icol >= a.W);         // only lanes (0, 0) of the first and (15, 2) of the last strip
    const bool outer = sx == 0 || sx == strips_x - 1;                // (wave-uniform)
    const int in_chunk_bytes = a.H * a.W * CIN, row_bytes = a.W * CIN;
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x) + (size_t)chunk * in_chunk_bytes, 0, in_chunk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * COUT, 0, a.OH * a.OW * COUT, 0x00020000);
    const int ccol = g == 3 ? icol - 1 : icol;
    const int voff_b = (ccol < 0 ? 0 : (ccol >= a.W ? a.W - 1 : ccol)) * CIN;   // (an outside column loads a valid neighbour, replaced below)
    const int voff_r = ow * CIN + 8 * g;                                        // residual: the lane's own 8 output channels of its column
    const int voff_out = ow * COUT + 8 * g;
    const int ir0 = oh0 - a.pt;
    const int rows_needed = steps + 2;

    v4i rawb[2][2];   // rows requested two ahead: [slot][channel tile]
    v2i rawr[2];
    v4i brow[3][2];   // the 3-row window of B operands
    v2i cen[3];
    auto row_ok = [&](int rr) { const int ir = ir0 + rr; return rr < rows_needed && ir >= 0 && ir < a.H; };
    auto issue = [&](int slot, int rr) {
        if (row_ok(rr)) {
            const int so = (ir0 + rr) * row_bytes;
            rawb[slot][0] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff_b, so, 0));
            rawb[slot][1] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff_b + 16, so, 0));
            rawr[slot] = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rs_in, voff_r, so, 0));
        }
    };
    auto consume = [&](int slot, int rr, int ti) {
        if (row_ok(rr)) {
            brow[ti][0] = rawb[slot][0];
            brow[ti][1] = rawb[slot][1];
            if (outer) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) brow[ti][ct][e] = outside ? zp4 : brow[ti][ct][e];
            }
            cen[ti] = rawr[slot];
        } else {
            brow[ti][0] = zrow;
            brow[ti][1] = zrow;
            cen[ti] = (v2i){zp4, zp4};
        }
    };
    auto emit = [&](int i0, int i1, int i2, int oh) {
        asm volatile("" ::: "memory");  // (the per-channel constants are re-read from LDS every row instead of pinning ~48 registers)
        int bfrag[2];
        v4i acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(dwa[ct][0], brow[i0][ct], dwb[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(dwa[ct][1], brow[i1][ct], acc[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(dwa[ct][2], brow[i2][ct], acc[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int q = 4 * ct + g;
            const v4i m = c_dw[q * 4 + 0], c01 = c_dw[q * 4 + 1], c23 = c_dw[q * 4 + 2];
            const int e1 = reinterpret_cast<const int*>(c_dw + q * 4 + 3)[0];
            const long cc[4] = {__builtin_bit_cast(long, (v2i){c01.x, c01.y}), __builtin_bit_cast(long, (v2i){c01.z, c01.w}),
                                __builtin_bit_cast(long, (v2i){c23.x, c23.y}), __builtin_bit_cast(long, (v2i){c23.z, c23.w})};
            int qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[e] = med3(rq_hi(acc[ct][e], m[e], cc[e], e1, e), a.dw_lo, a.dw_hi);
            bfrag[ct] = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
        }
        const long bf = ((long)(uint32_t)bfrag[1] << 32) | (uint32_t)bfrag[0];
        int outw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const v4i pacc = __builtin_amdgcn_mfma_i32_16x16x32_i8(pwa[t], bf, pwb[t], 0, 0, 0);
            const v4i m = c_pw[(g * NT + t) * 4 + 0], c1 = c_pw[(g * NT + t) * 4 + 1], sh = c_pw[(g * NT + t) * 4 + 2];
            int qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = med3(rq(pacc[e], m[e], c1[e], sh[e]), a.pw_lo, a.pw_hi);   // value + 128: the table's column
                qv[e] = add_tab[(uint32_t)perm(cen[i1][t], v, 0x0c0c0400u + (e << 8))];
            }
            outw[t] = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(int)))) int, (v2i){outw[0], outw[1]}), rs_out, voff_out,
                                              oh * a.OW * COUT, 0);
    };

    // relative row rr lives in raw slot rr & 1 and window slot rr % 3; two rows are consumed before the first output row, one per step afterwards
    issue(0, 0);
    issue(1, 1);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        consume(rr & 1, rr, rr % 3);
        issue(rr & 1, rr + 2);
    }
    for (int k = 0; k < steps; k += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (k + u >= steps) break;
            const int rs = 2 + u;
            consume(rs & 1, k + rs, rs % 3);
            issue(rs & 1, k + rs + 2);
            emit(u % 3, (u + 1) % 3, (u + 2) % 3, oh0 + k + u);
        }
    }
}

template <int CW, int NW, int COUT, int S, bool ADD, bool W8 = false>
void launch_strip(const Strip8Args& a, hipStream_t s) {
    constexpr int SPB = ADD ? 8 / NW : (NW == 1 ? 4 : 1);
    constexpr int QL = CW / 16, NT = COUT / NW / 16;
    constexpr size_t smem = (ADD ? 65536 : 0) + (size_t)NW * (4 * QL * 16 + 4 * NT * 16) * 4 + (NW > 1 ? NW * 4 * QL * 12 * 4 + 2 * SPB * NW * 64 * 8 : 0);
    const long per_chunk = (long)(W8 ? 1 : a.OW / 16) * ((a.OH + a.TH - 1) / a.TH);
    const long blocks = NW == 1 ? (a.B * per_chunk + SPB - 1) / SPB : ((a.B + SPB - 1) / SPB) * per_chunk;
    auto kern = i8_strip_kernel<CW, NW, COUT, S, ADD, W8>;
    if (smem > 65536) (void)ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem);  // (a refusal shows as the launch's error)
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW * SPB), smem, s, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Front block in the same style: frontend output [H0][W0] int8 -> CONV_2D 3x3 stem (stride 1x2, 16 channels, ReLU6) ->
// DEPTHWISE_CONV_2D 3x3 stride 2 (ReLU6) -> CONV_2D 1x1 to 32 channels (ReLU6); bit-identical to i8_front_kernel.
//
// A wave owns 16 output columns and walks down the output rows; lane (n, kq) = (output column, channel quad).  The stem
// runs on the matrix cores as well: with the contraction index K = 8 * (window row) + (window column), lane (n, kq) of the
// B operand holds the three bytes fe[sr - 1 + kq][2 sc .. 2 sc + 2] of its OWN input row — every lane streams input row
// (stem row - 1 + kq), so a B fragment is one AND / byte-permute of the two dwords it loaded, and the result lands as the
// four channels 4 q .. 4 q + 3 of stem column sc in lane (n, q): exactly the quad the depthwise stage of that lane needs.
// Three MFMAs per stem row give stem columns 2 ow, 2 ow + 1, 2 ow + 2 (the three taps of the stride-2 depthwise window);
// what remains on the vector ALU is the requantisation of each value and the byte shuffles.
constexpr int kF_STA = 0, kF_STB = 64, kF_STC = 80, kF_DWW = 128, kF_DWB = 176, kF_DWC = 192, kF_PWA = 240, kF_PWB = 368, kF_PWC = 400;

__global__ __launch_bounds__(256) void i8_front_strip_kernel(FrontStrip8Args a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int strips_x = a.OW >> 4;
    const int rblocks = (a.OH + a.TH - 1) / a.TH;
    int wid = xcd_tile(blockIdx.x, gridDim.x) * 4 + wave;
    if (wid >= a.B * strips_x * rblocks) return;
    const int sx = wid % strips_x;
    wid /= strips_x;
    const int ry = wid % rblocks;
    const int chunk = wid / rblocks;
    const int oh0 = ry * a.TH;
    const int nrows = (a.OH - oh0) < a.TH ? (a.OH - oh0) : a.TH;
    const int ow = sx * 16 + n;

    const v4i* c4 = reinterpret_cast<const v4i*>(a.cst);
    const long sta = (long)(uint32_t)a.cst[kF_STA + lane];
    const v4i stb = c4[kF_STB / 4 + kq];
    const v4i stm = c4[kF_STC / 4 + kq * 3 + 0], stc1 = c4[kF_STC / 4 + kq * 3 + 1], ste = c4[kF_STC / 4 + kq * 3 + 2];
    const long stc[4] = {rq64(stc1.x), rq64(stc1.y), rq64(stc1.z), rq64(stc1.w)};
    const int ste1 = pack_shifts(ste - 1);
    int dww[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const v4i v = c4[kF_DWW / 4 + kq * 3 + i];
        dww[i][0] = v.x; dww[i][1] = v.y; dww[i][2] = v.z; dww[i][3] = v.w;
    }
    const v4i dwb = c4[kF_DWB / 4 + kq];
    const v4i dwm = c4[kF_DWC / 4 + kq * 3 + 0], dwc1 = c4[kF_DWC / 4 + kq * 3 + 1], dwe = c4[kF_DWC / 4 + kq * 3 + 2];
    const long dwc[4] = {rq64(dwc1.x), rq64(dwc1.y), rq64(dwc1.z), rq64(dwc1.w)};
    const int dwe1 = pack_shifts(dwe - 1);
    long pwa[2];
    v4i pwb[2], pwm[2], pwc1[2];
    int pwe1[2];
    long pwc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        pwa[t] = (long)(uint32_t)a.cst[kF_PWA + t * 64 + lane];
        pwb[t] = c4[kF_PWB / 4 + kq * 2 + t];
        pwm[t] = c4[kF_PWC / 4 + (kq * 2 + t) * 3 + 0];
        pwc1[t] = c4[kF_PWC / 4 + (kq * 2 + t) * 3 + 1];
        pwe1[t] = pack_shifts(c4[kF_PWC / 4 + (kq * 2 + t) * 3 + 2] - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) pwc[t][e] = rq64(pwc1[t][e]);
    }

    const int zfe4 = (a.zp_fe & 0xff) * 0x01010101;
    const int zst4 = (a.zp_st & 0xff) * 0x01010101;
    const int zstrow = (a.zp_st & 0xff) * 0x00010101;
    const int fe_bytes = a.H0 * a.W0;
    const __amdgpu_buffer_rsrc_t rs_fe =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.fe) + (size_t)chunk * fe_bytes, 0, fe_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * 32, 0, a.OH * a.OW * 32, 0x00020000);
    const bool right_fe = 4 * ow + 4 >= a.W0;       // second dword of the row lies beyond the input: stem padding column
    const bool right_st = 2 * ow + 2 >= a.W0 / 2;   // third stem column lies beyond the stem map: depthwise padding column
    const int voff_out = ow * 32 + 8 * kq;
    const int sr0 = 2 * oh0;                        // first stem row of the strip (depthwise: stride 2, no top padding)
    const int rows_needed = 2 * (nrows - 1) + 3;

    constexpr int FD = 6;  // input rows requested ahead (slot = row % FD, static under the six-row unrolling below)
    v2i raw[FD];
    TRow<1> T[3];

    // stem row sr0 + srel: lane (n, kq) reads input row sr - 1 + kq (the stem's top padding is row -1)
    auto fe_row = [&](int srel) { return sr0 + srel - 1 + kq; };
    auto issue = [&](int slot, int srel) {
        if (srel < rows_needed) {
            int fr = fe_row(srel);
            fr = fr < 0 ? 0 : (fr >= a.H0 ? a.H0 - 1 : fr);  // padding rows load a valid row, replaced below
            raw[slot] = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rs_fe, fr * a.W0 + 4 * ow, 0, 0));
        }
    };
    auto stem_row = [&](int slot, int srel, int ti) {
        if (srel < rows_needed && sr0 + srel < a.H0) {
            const int fr = fe_row(srel);
            const bool ok = fr >= 0 && fr < a.H0;
            const int A = ok ? raw[slot].x : zfe4;
            const int Bd = (ok && !right_fe) ? raw[slot].y : zfe4;
            const int x[3] = {A & 0x00ffffff, perm(Bd, A, 0x0c040302u), Bd & 0x00ffffff};
            int pk[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const v4i acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(sta, (long)(uint32_t)x[j], stb, 0, 0, 0);
                int qv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) qv[e] = med3(rq_hi(acc[e], stm[e], stc[e], ste1, e), a.st_lo, a.st_hi);
                pk[j] = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
            }
            if (right_st) pk[2] = zst4;
            const int lo = perm(pk[1], pk[0], 0x05010400u);
            const int hi = perm(pk[1], pk[0], 0x07030602u);
            T[ti].c[0][0] = perm(pk[2], lo, 0x0c040100u);
            T[ti].c[0][1] = perm(pk[2], lo, 0x0c050302u);
            T[ti].c[0][2] = perm(pk[2], hi, 0x0c060100u);
            T[ti].c[0][3] = perm(pk[2], hi, 0x0c070302u);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) T[ti].c[0][e] = zstrow;
        }
    };
    auto emit = [&](int i0, int i1, int i2, int oh) {
        int qv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int acc = dot4_first(T[i0].c[0][e], dww[0][e], dwb[e]);
            acc = dot4(T[i1].c[0][e], dww[1][e], acc);
            acc = dot4(T[i2].c[0][e], dww[2][e], acc);
            qv[e] = med3(rq_hi(acc, dwm[e], dwc[e], dwe1, e), a.dw_lo, a.dw_hi);
        }
        const long bf = (long)(uint32_t)perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
        int outw[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const v4i acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(pwa[t], bf, pwb[t], 0, 0, 0);
            int ov[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = med3(rq_hi(acc[e], pwm[t][e], pwc[t][e], pwe1[t], e), a.pw_lo, a.pw_hi);
            outw[t] = perm(perm(ov[3], ov[2], 0x0c0c0400u), perm(ov[1], ov[0], 0x0c0c0400u), 0x05040100u);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(int)))) int, (v2i){outw[0], outw[1]}), rs_out, voff_out, oh * a.OW * 32, 0);
    };

#pragma unroll
    for (int rr = 0; rr < FD; ++rr) issue(rr, rr);
    stem_row(0, 0, 0);
    issue(0, FD);
    for (int k = 0; k < nrows; k += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (k + u >= nrows) break;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int rs = 1 + 2 * u + s;  // static part of the relative stem row (k is a multiple of 3 output rows = 6 stem rows)
                stem_row(rs % FD, 2 * k + rs, rs % 3);
                issue(rs % FD, 2 * k + rs + FD);
            }
            emit((2 * u) % 3, (2 * u + 1) % 3, (2 * u + 2) % 3, oh0 + k + u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-alone depthwise 3x3 (inverted-residual blocks of exported graphs: expand 1x1 -> DEPTHWISE -> squeeze-excite -> project 1x1) as
// a row-streaming kernel — the INT8 sibling of f32_dw_stream_kernel.  A wave owns NCOL = 64 / CQ columns x CQ channel quads (CQ = the
// largest of 16, 8, 4, 2, 1 dividing C / 4) and walks down the rows: three dword loads per input row instead of nine per output, the
// rows kept byte-transposed (per channel the three taps of a row in one dword, 6 v_perm per row) so that an output is three
// v_dot4_i32_i8.  The zero point of the input is folded into the bias (padding taps read as the zero point); requantisation is the
// generic exact form (any multiplier / shift).
struct DwStream8Args {
    const int8_t* x; int8_t* y;
    const int8_t* w; const int32_t* bias; const int32_t* mult; const int32_t* shift;   // [3][3][C], [C], [C], [C]
    int B, H, W, C, OH, OW, TH, pt, pl, zp_in, zp_out, amin, amax, CQ, rq_right;
    int32_t* pool;  // [B][C] or null: += sum of the stored bytes per channel (the squeeze-excite MEAN behind the stage; i8_dw_stream_kernel only)
};

// HI (the launcher checked: multipliers >= 0, shifts in [-20, -1], clamp starting at the zero point — ReLU behind the stage): the sign-free
// requantisation clamp(hi32(acc m + C) >> (e - 1)), C = 2^30 + (2^(e-1) + zp 2^e) 2^31 (bn_i8_pw.hip), one multiply-add + shift + clamp per output
struct RqHi {
    long long c[4];
    int sh[4];
};
__device__ __forceinline__ void rq_hi_setup(RqHi& r, const int (&shift)[4], int zp_out) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ex = -shift[e];
        r.c[e] = (1ll << 30) + (((1ll << (ex - 1)) + (long long)zp_out * (1ll << ex)) << 31);
        r.sh[e] = ex - 1;
    }
}

// BN_TAIL_STAMPS (measurement build only, tools/dw_stamps.py): waves from the middle of the grid of the launch with (C, H) = g_dw_sel record when
// they started, had their constants and first rows, finished the row walk and the pooling atomics (s_memrealtime, 10 ns ticks)
#ifdef BN_TAIL_STAMPS
__device__ long long* g_dw_stamps = nullptr;   // [kDwStampWaves][6]
__device__ int g_dw_sel[2] = {0, 0};
constexpr int kDwStampWaves = 4096;
#define BN_DSTAMP(i) do { if (dstamp) dst_[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BN_DSTAMP(i) do {} while (0)
#endif

template <int S, bool HI>
__global__ __launch_bounds__(256) void i8_dw_stream_kernel(DwStream8Args a) {
#ifdef BN_TAIL_STAMPS
    long long dst_[4] = {0, 0, 0, 0};
    const long dslot = ((long)blockIdx.x - (long)gridDim.x / 2) * 4 + (threadIdx.x >> 6);
    const bool dstamp = g_dw_stamps && a.C == g_dw_sel[0] && a.H == g_dw_sel[1] && dslot >= 0 && dslot < kDwStampWaves;
#endif
    BN_DSTAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
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

    // weights as bytes (tap0, tap1, tap2, 0) per window row and channel; bias with the input zero point folded in (filled by load_constants()
    // BEHIND the first row requests: the wave's first rows and its constants travel together — one round trip of prologue instead of two)
    int wr[3][4], bias[4], mult[4], shift[4];
    RqHi rqh;
    auto load_constants = [&]() {
        // nine dword loads (four channels of a tap each) + four 16-byte loads, all in flight together; byte transposes per window row
        int wq[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wq[t] = *reinterpret_cast<const int*>(a.w + t * a.C + c0);
        const v4i b4 = *reinterpret_cast<const v4i*>(a.bias + c0), m4 = *reinterpret_cast<const v4i*>(a.mult + c0), s4 = *reinterpret_cast<const v4i*>(a.shift + c0);
        int sum[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int r0 = wq[3 * i], r1 = wq[3 * i + 1], r2 = wq[3 * i + 2];
            const int lo = perm(r1, r0, 0x05010400u), hi = perm(r1, r0, 0x07030602u);
            wr[i][0] = perm(r2, lo, 0x0c040100u);
            wr[i][1] = perm(r2, lo, 0x0c050302u);
            wr[i][2] = perm(r2, hi, 0x0c060100u);
            wr[i][3] = perm(r2, hi, 0x0c070302u);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] = dot4(wr[i][e], 0x00010101, sum[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bias[e] = b4[e] - a.zp_in * sum[e];
            mult[e] = m4[e];
            shift[e] = s4[e];
        }
        if constexpr (HI) rq_hi_setup(rqh, shift, a.zp_out);
    };
    const int zp4 = (a.zp_in & 0xff) * 0x01010101, zprow = (a.zp_in & 0xff) * 0x00010101;
    const int in_chunk_bytes = a.H * a.W * a.C;
    const int row_bytes = a.W * a.C;
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x) + (size_t)chunk * in_chunk_bytes, 0, in_chunk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * a.C, 0, a.OH * a.OW * a.C, 0x00020000);
    const int iw0 = ow * S - a.pl;
    bool pad[3];
    int voff_in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        pad[j] = !live || iw0 + j < 0 || iw0 + j >= a.W;  // padding columns load a valid byte offset (0) and are replaced by the zero point
        voff_in[j] = pad[j] ? 0 : (iw0 + j) * a.C + c0;
    }
    const int voff_out = live ? ow * a.C + c0 : 0x7fff0000;  // beyond the descriptor's range: the store is dropped
    const int ir0 = oh0 * S - a.pt;
    const int rows_needed = S * (nrows - 1) + 3;

    constexpr int D = 6;  // input rows requested ahead (slot = row % D: static under the six-row unrolling): the walk is bound by load latency, not issue
    int raw[D][3], T[3][4];
    int psum[4] = {0, 0, 0, 0};  // what this lane stored, per channel (pool != null)
    auto row_ok = [&](int rr) { const int ir = ir0 + rr; return rr < rows_needed && ir >= 0 && ir < a.H; };
    auto issue = [&](int slot, int rr) {
        if (row_ok(rr)) {
            const int soff = (ir0 + rr) * row_bytes;
#pragma unroll
            for (int j = 0; j < 3; ++j) raw[slot][j] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff_in[j], soff, 0);
        }
    };
    auto consume = [&](int slot, int rr, int ti) {
        if (row_ok(rr)) {
            const int r0 = pad[0] ? zp4 : raw[slot][0], r1 = pad[1] ? zp4 : raw[slot][1], r2 = pad[2] ? zp4 : raw[slot][2];
            const int lo = perm(r1, r0, 0x05010400u), hi = perm(r1, r0, 0x07030602u);
            T[ti][0] = perm(r2, lo, 0x0c040100u);
            T[ti][1] = perm(r2, lo, 0x0c050302u);
            T[ti][2] = perm(r2, hi, 0x0c060100u);
            T[ti][3] = perm(r2, hi, 0x0c070302u);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) T[ti][e] = zprow;
        }
    };
    auto emit = [&](int i0, int i1, int i2, int oh) {
        int qv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int acc = dot4_first(T[i0][e], wr[0][e], bias[e]);
            acc = dot4(T[i1][e], wr[1][e], acc);
            acc = dot4(T[i2][e], wr[2][e], acc);
            if constexpr (HI) qv[e] = med3((int)(((long long)acc * mult[e] + rqh.c[e]) >> 32) >> rqh.sh[e], a.amin, a.amax);
            else qv[e] = med3(mbqm_u(acc, mult[e], shift[e], (a.rq_right & 1) != 0) + a.zp_out, a.amin, a.amax);
            psum[e] += qv[e];
        }
        const int word = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
        __builtin_amdgcn_raw_buffer_store_b32(word, rs_out, voff_out, oh * a.OW * a.C, 0);
    };
    constexpr int P = 3 - S;
#pragma unroll
    for (int rr = 0; rr < D; ++rr) issue(rr, rr);
    load_constants();
#ifdef BN_TAIL_STAMPS
    asm volatile("" :: "v"(wr[0][0]), "v"(bias[3]) : "memory");  // (the stamp sits behind the constants' arrival)
#endif
    BN_DSTAMP(1);
#pragma unroll
    for (int rr = 0; rr < P; ++rr) {
        consume(rr % D, rr, rr % 3);
        issue(rr % D, rr + D);
    }
    constexpr int U = 6 / S;
    for (int k = 0; k < nrows; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u >= nrows) break;
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
                const int rs = P + S * u + s2;  // (S k is a multiple of six: the slot of row S k + rs is rs % D)
                consume(rs % D, S * k + rs, rs % 3);
                issue(rs % D, S * k + rs + D);
            }
            emit((S * u) % 3, (S * u + 1) % 3, (S * u + 2) % 3, oh0 + k + u);
        }
    }
    BN_DSTAMP(2);
    if (a.pool) {
        // squeeze-excite pooling on the way out: integer sums, so the order (lanes, waves, atomics) does not matter — bit-identical to MEAN over the map
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int v = live ? psum[e] : 0;
            for (int off = CQ; off < 64; off <<= 1) v += __shfl_xor(v, off);
            if (n == 0) atomicAdd(a.pool + (size_t)chunk * a.C + c0 + e, v);
        }
    }
    BN_DSTAMP(3);
#ifdef BN_TAIL_STAMPS
    if (dstamp && lane == 0) {
        long long* o = g_dw_stamps + dslot * 6;
        for (int i = 0; i < 4; ++i) o[i] = dst_[i];
        o[4] = nrows;
        o[5] = S;
    }
#endif
}

// The stem of exported graphs (3x3 convolution of the single-channel map, any stride, C output channels) in the same row-streaming
// form: lane = (column, channel quad), the three window bytes of an input row packed into ONE dword that serves all four channels
// of the lane (three v_dot4_i32_i8 per output instead of nine sign-extend + multiply-add triples), three byte loads per input row.
template <int S, bool HI>
__global__ __launch_bounds__(256) void i8_stem_stream_kernel(DwStream8Args a, int SW) {
    const int tid = threadIdx.x, lane = tid & 63;
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
    int wr[3][4], bias[4], mult[4], shift[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int sum = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k0 = a.w[(i * 3 + 0) * a.C + c0 + e], k1 = a.w[(i * 3 + 1) * a.C + c0 + e], k2 = a.w[(i * 3 + 2) * a.C + c0 + e];
            wr[i][e] = (k0 & 0xff) | ((k1 & 0xff) << 8) | ((k2 & 0xff) << 16);
            sum += k0 + k1 + k2;
        }
        bias[e] = a.bias[c0 + e] - a.zp_in * sum;
        mult[e] = a.mult[c0 + e];
        shift[e] = a.shift[c0 + e];
    }
    RqHi rqh;
    if constexpr (HI) rq_hi_setup(rqh, shift, a.zp_out);
    const int zp = a.zp_in & 0xff, zprow = zp * 0x00010101;
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x) + (size_t)chunk * a.H * a.W, 0, a.H * a.W, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)chunk * a.OH * a.OW * a.C, 0, a.OH * a.OW * a.C, 0x00020000);
    const int iw0 = ow * SW - a.pl;
    bool pad[3];
    int voff_in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        pad[j] = !live || iw0 + j < 0 || iw0 + j >= a.W;
        voff_in[j] = pad[j] ? 0 : iw0 + j;
    }
    const int voff_out = live ? ow * a.C + c0 : 0x7fff0000;
    const int ir0 = oh0 * S - a.pt;
    const int rows_needed = S * (nrows - 1) + 3;
    constexpr int D = 6;
    int raw[D][3], T[3];
    auto row_ok = [&](int rr) { const int ir = ir0 + rr; return rr < rows_needed && ir >= 0 && ir < a.H; };
    auto issue = [&](int slot, int rr) {
        if (row_ok(rr)) {
            const int soff = (ir0 + rr) * a.W;
#pragma unroll
            for (int j = 0; j < 3; ++j) raw[slot][j] = __builtin_amdgcn_raw_buffer_load_b8(rs_in, voff_in[j], soff, 0);
        }
    };
    auto consume = [&](int slot, int rr, int ti) {
        if (row_ok(rr)) {
            const int b0 = pad[0] ? zp : (raw[slot][0] & 0xff), b1 = pad[1] ? zp : (raw[slot][1] & 0xff), b2 = pad[2] ? zp : (raw[slot][2] & 0xff);
            T[ti] = b0 | (b1 << 8) | (b2 << 16);
        } else {
            T[ti] = zprow;
        }
    };
    auto emit = [&](int i0, int i1, int i2, int oh) {
        int qv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int acc = dot4_first(T[i0], wr[0][e], bias[e]);
            acc = dot4(T[i1], wr[1][e], acc);
            acc = dot4(T[i2], wr[2][e], acc);
            if constexpr (HI) qv[e] = med3((int)(((long long)acc * mult[e] + rqh.c[e]) >> 32) >> rqh.sh[e], a.amin, a.amax);
            else qv[e] = med3(mbqm_u(acc, mult[e], shift[e], (a.rq_right & 1) != 0) + a.zp_out, a.amin, a.amax);
        }
        const int word = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
        __builtin_amdgcn_raw_buffer_store_b32(word, rs_out, voff_out, oh * a.OW * a.C, 0);
    };
    constexpr int P = 3 - S;
#pragma unroll
    for (int rr = 0; rr < D; ++rr) issue(rr, rr);
#pragma unroll
    for (int rr = 0; rr < P; ++rr) {
        consume(rr % D, rr, rr % 3);
        issue(rr % D, rr + D);
    }
    constexpr int U = 6 / S;
    for (int k = 0; k < nrows; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u >= nrows) break;
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
                const int rs = P + S * u + s2;  // (S k is a multiple of six: the slot of row S k + rs is rs % D)
                consume(rs % D, S * k + rs, rs % 3);
                issue(rs % D, S * k + rs + D);
            }
            emit((S * u) % 3, (S * u + 1) % 3, (S * u + 2) % 3, oh0 + k + u);
        }
    }
}

}  // namespace

// channel split of a block: waves per strip (1 = a wave holds all input channels); 0 = no strip kernel for this shape
int i8_strip_waves(int Cin, int Cout, int stride, int OW, bool add) {
    if (stride != 1 && stride != 2) return 0;
    if (OW == 8) {  // stage 4: two row blocks x 8 columns per wave
        if (add) return (stride == 1 && Cin == 256 && Cout == 256) ? 8 : 0;
        return (stride == 2 && Cin == 128 && Cout == 256) ? 4 : 0;
    }
    if (OW % 16) return 0;
    if (add && (stride != 1 || Cin != Cout)) return 0;
    if (Cin == 64 && Cout == 64) return 2;  // measured: two waves x 32 channels (136 VGPRs) beat one wave x 64 (230) by 5 %
    if ((Cin == 32 || Cin == 64) && (Cout == 32 || Cout == 64)) return 1;
    if (add) return Cin == 128 ? 4 : 0;
    if (Cin == 64 && Cout == 128) return 2;
    if (Cin == 128 && Cout == 128) return 4;
    return 0;
}

bool i8_strip_supported(int Cin, int Cout, int stride, int OW, bool add) { return i8_strip_waves(Cin, Cout, stride, OW, add) != 0; }
// (8-wide maps additionally need an even height: checked by the packer, which only then emits the constant block)

void launch_i8_strip(Strip8Args a, int Cin, int Cout, int stride, hipStream_t s) {
    const bool add = a.add.enabled != 0;
    const int nw = i8_strip_waves(Cin, Cout, stride, a.OW, add);
    // rows per wave: as tall as possible while the launch still fills the chip a few times over
    int th = a.OH;
    if (a.OW != 8) {
        // two rounds of resident waves (4 per SIMD x 1024 SIMDs) are enough to fill the chip; halving the rows beyond that only adds prologues
        // (stage2_ds1 at 4096 chunks: 8 rows per wave 2.079 ms per step, 16 rows 2.058 — tools/ab_option.py i8_strip_th)
        while (th > 4 && (long)a.B * (a.OW / 16) * ((a.OH + th - 1) / th) * nw < 8192) th = (th + 1) / 2;
        if (const int v = g_opt.i8_strip_th; v >= 1) th = v < a.OH ? v : a.OH;  // tests: force the rows per wave
    }  // 8-wide maps: the whole (even) height, half per lane group
    a.TH = th;
#define BN_STRIP(CW, NW, CO, ST, AD) \
    if (Cin == CW * NW && nw == NW && Cout == CO && stride == ST && add == AD) return launch_strip<CW, NW, CO, ST, AD>(a, s);
    if (Cin == 32 && nw == 1 && Cout == 32 && stride == 1 && add && g_opt.i8_strip_mfdw && a.pt == 1 && a.pl == 1 && a.OW % 16 == 0 && a.W == a.OW && a.H == a.OH) {
        // stage1_ds2 of the shipped graph: the depthwise stage on the matrix cores (i8_strip_mf_kernel)
        const long per_chunk = (long)(a.OW / 16) * ((a.OH + a.TH - 1) / a.TH);
        const long blocks = (a.B * per_chunk + 7) / 8;
        const size_t smem = 65536 + (size_t)(4 * 2 * 16 + 4 * 2 * 16) * 4;
        (void)ensure_dynamic_lds(reinterpret_cast<const void*>(i8_strip_mf_kernel), smem);
        hipLaunchKernelGGL(i8_strip_mf_kernel, dim3((unsigned)blocks), dim3(512), smem, s, a);
        return;
    }
    BN_STRIP(32, 1, 32, 1, true)
    BN_STRIP(32, 4, 128, 1, true)
    BN_STRIP(32, 2, 64, 1, true)
    BN_STRIP(32, 2, 64, 1, false)
    BN_STRIP(32, 2, 64, 2, false)
    BN_STRIP(32, 1, 32, 1, false)
    BN_STRIP(32, 1, 64, 1, false)
    BN_STRIP(64, 1, 32, 1, false)
    BN_STRIP(32, 4, 128, 1, false)
    BN_STRIP(32, 2, 128, 1, false)
    BN_STRIP(32, 1, 32, 2, false)
    BN_STRIP(32, 1, 64, 2, false)
    BN_STRIP(64, 1, 32, 2, false)
    BN_STRIP(32, 4, 128, 2, false)
    BN_STRIP(32, 2, 128, 2, false)
#undef BN_STRIP
    if (a.OW == 8 && (a.OH & 1) == 0) {
        if (Cin == 128 && Cout == 256 && stride == 2 && !add) return launch_strip<32, 4, 256, 2, false, true>(a, s);
        if (Cin == 256 && Cout == 256 && stride == 1 && add) return launch_strip<32, 8, 256, 1, true, true>(a, s);
    }
}

bool i8_front_strip_supported(int H0, int W0, int C, int N, int OH, int OW) {
    return C == 16 && N == 32 && OW % 16 == 0 && H0 == 2 * OH && W0 == 4 * OW && W0 % 4 == 0;
}

void launch_i8_front_strip(FrontStrip8Args a, hipStream_t s) {
    int th = a.OH;
    while (th > 4 && (long)a.B * (a.OW / 16) * ((a.OH + th - 1) / th) < 8192) th = (th + 1) / 2;
    if (const int v = g_opt.i8_strip_th; v >= 1) th = v < a.OH ? v : a.OH;
    a.TH = th;
    const long waves = (long)a.B * (a.OW / 16) * ((a.OH + th - 1) / th);
    hipLaunchKernelGGL(i8_front_strip_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Inverted-residual blocks of exported graphs: expand CONV_2D 1x1 (fused activation) -> DEPTHWISE_CONV_2D 3x3 stride S in ONE kernel, the
// INT8 sibling of f32_pwdw_kernel (bn_f32_strip.hip).  The expanded map (hid = 2 Cin channels at the input resolution) is the largest
// tensor of the block; here it lives two rows at a time in LDS.  A workgroup of twelve waves owns RB output rows of one chunk; every
// thread works in both stages, ONE barrier per hidden row:
//   * expand: the hidden row's (W / 16) x (hid / 16) tiles on the int8 matrix cores, two per wave (A = the packer's weight fragments,
//     pinned in registers; B = 16 bytes per lane straight from the input row, the next row requested before this one is multiplied;
//     bytes beyond Cin meet zero weight columns), exact requantisation, the four channels of a lane as one dword into the ring
//     [2][W + 2][hid] (border columns = the zero point: the SAME padding of the hidden map);
//   * depthwise: i8_dw_stream_kernel's walk with the ring in place of memory — a thread owns (column, channel quad) items, the 3 x 3
//     window byte-transposed in registers (a hidden row is read from LDS exactly once, right behind the barrier of its own step, so two
//     ring rows are enough), three v_dot4_i32_i8 per output, exact requantisation, dword stores coalesced along the channels.
// Bit-identical to i8_pw_wave_kernel / i8_dwpw_kernel followed by i8_dw_stream_kernel (same integer arithmetic).
// MEASURED SLOWER than those two (configs[4] in INT8, 1024 chunks: the six pairs 1.18 -> 1.83 ms; 0.566 vs 0.315 ms on the widest): the int8
// maps are a quarter of the float32 ones, so the two kernels were never far from their vector-ALU time, and here that work (exact
// requantisation of 1.5 outputs per input byte) runs at ~36 % ALU utilisation between one barrier per row.  Kept behind option i8_pwdw
// (default 0) with its tests; the float32 sibling is where the fusion pays (bn_f32_strip.hip).
struct PwDw8Args {
    const int8_t* x; int8_t* y;
    const int8_t* pw_w; const int32_t* pw_b; const int32_t* pw_mult; const int32_t* pw_shift;  // fragments [KS][hid/16][64][16]; [hid] x 3
    const int8_t* dw_w; const int32_t* dw_b; const int32_t* dw_mult; const int32_t* dw_shift;  // [3][3][hid]; [hid] x 3
    int B, H, W, Cin, hid, OH, OW, pt, pl, RB;
    int h_zp, h_amin, h_amax;      // the hidden map's quantisation: output of the expand convolution = input of the depthwise one
    int o_zp, o_amin, o_amax;
};

template <int S, int KS>
__global__ __launch_bounds__(768) void i8_pwdw_kernel(PwDw8Args a) {
    extern __shared__ __attribute__((aligned(16))) int ring8[];  // [2][W + 2][hid / 4] dwords, then the constant tables
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int hq = a.hid >> 2;                      // channel quads (dwords) per position
    const int rowd = (a.W + 2) * hq;                // dwords per ring row
    const int rblocks = (a.OH + a.RB - 1) / a.RB;
    const int wid = xcd_tile(blockIdx.x, gridDim.x);
    const int ry = wid % rblocks, chunk = wid / rblocks;
    const int oh0 = ry * a.RB;
    const int nrows = (a.OH - oh0) < a.RB ? (a.OH - oh0) : a.RB;
    const int h_lo = S * oh0 - a.pt;
    const int nhid = S * (nrows - 1) + 3;
    auto row_ok = [&](int k) { return k >= 0 && k < nhid && h_lo + k >= 0 && h_lo + k < a.H; };
    const int zp4 = (a.h_zp & 0xff) * 0x01010101, zprow = (a.h_zp & 0xff) * 0x00010101;

    // per-quad constants in LDS (in registers they spilled: 72 dwords per thread): depthwise [hq][6 x v4i] = the three weight rows as
    // (tap0, tap1, tap2, 0) bytes per channel, bias with the hidden zero point folded in, multipliers, shifts; expand [hq][3 x v4i]
    v4i* dwt = reinterpret_cast<v4i*>(ring8 + 2 * rowd);
    v4i* pwt = dwt + 6 * hq;
    for (int quad = tid; quad < hq; quad += 768) {
        const int c0 = 4 * quad;
        v4i wrow[3], bb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int sum = 0;
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) {
                const int k0 = a.dw_w[(ii * 3 + 0) * a.hid + c0 + e], k1 = a.dw_w[(ii * 3 + 1) * a.hid + c0 + e], k2 = a.dw_w[(ii * 3 + 2) * a.hid + c0 + e];
                wrow[ii][e] = (k0 & 0xff) | ((k1 & 0xff) << 8) | ((k2 & 0xff) << 16);
                sum += k0 + k1 + k2;
            }
            bb[e] = a.dw_b[c0 + e] - a.h_zp * sum;
        }
        dwt[6 * quad + 0] = wrow[0]; dwt[6 * quad + 1] = wrow[1]; dwt[6 * quad + 2] = wrow[2]; dwt[6 * quad + 3] = bb;
        dwt[6 * quad + 4] = *reinterpret_cast<const v4i*>(a.dw_mult + c0);
        dwt[6 * quad + 5] = *reinterpret_cast<const v4i*>(a.dw_shift + c0);
        pwt[3 * quad + 0] = *reinterpret_cast<const v4i*>(a.pw_b + c0);
        pwt[3 * quad + 1] = *reinterpret_cast<const v4i*>(a.pw_mult + c0);
        pwt[3 * quad + 2] = *reinterpret_cast<const v4i*>(a.pw_shift + c0);
    }
    for (int i = tid; i < 4 * hq; i += 768) {       // border columns of both ring rows
        const int slot = i / (2 * hq), rest = i - slot * 2 * hq;
        ring8[slot * rowd + (rest < hq ? 0 : (a.W + 1) * hq) + (rest % hq)] = zp4;
    }

    // ---- expand stage: tiles t = 2 wave + u -> position tile t % npt, channel tile t / npt
    const int npt = a.W >> 4, nct = a.hid >> 4, ntiles = npt * nct;
    v4i af[2][KS];
    int pos_t[2], dst_t[2], cq_t[2];
    bool live_t[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = 2 * wave + u;
        live_t[u] = t < ntiles;
        const int ptile = live_t[u] ? t % npt : 0, ct = live_t[u] ? t / npt : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[u][ks] = reinterpret_cast<const v4i*>(a.pw_w)[((size_t)ks * nct + ct) * 64 + lane];
        cq_t[u] = 4 * ct + q;                       // the lane's channel quad of this tile
        pos_t[u] = 16 * ptile + r;
        dst_t[u] = (pos_t[u] + 1) * hq + 4 * ct + q;
    }
    const int row_bytes = a.W * a.Cin;
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x) + (size_t)chunk * a.H * row_bytes, 0, a.H * row_bytes, 0x00020000);
    auto request = [&](v4i (&bf)[2][KS], int k) {
        if (!row_ok(k)) return;
        const int soff = (h_lo + k) * row_bytes;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int koff = 64 * ks + 16 * q;
                bf[u][ks] = (v4i){0, 0, 0, 0};
                if (live_t[u] && koff < a.Cin) bf[u][ks] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_in, pos_t[u] * a.Cin + koff, soff, 0));
            }
    };
    auto expand = [&](const v4i (&bf)[2][KS], int k) {
        if (!row_ok(k)) return;
        int* dst = ring8 + (k & 1) * rowd;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live_t[u]) continue;
            v4i acc = pwt[3 * cq_t[u] + 0];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u][ks], bf[u][ks], acc, 0, 0, 0);
            const v4i cm = pwt[3 * cq_t[u] + 1], cs = pwt[3 * cq_t[u] + 2];
            int packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) packed |= (med3(mbqm(acc[e], cm[e], cs[e]) + a.h_zp, a.h_amin, a.h_amax) & 0xff) << (8 * e);
            dst[dst_t[u]] = packed;
        }
    };

    // ---- depthwise stage: items i = tid + 768 u -> (column i / hq, channel quad i % hq)
    const int items = a.OW * hq;
    int T[2][3][4], src_d[2], out_off[2], quad_i[2];
    bool live_i[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 768 * u;
        live_i[u] = i < items;
        const int ow = live_i[u] ? i / hq : 0, quad = live_i[u] ? i - ow * hq : 0;
        quad_i[u] = quad;
        src_d[u] = (S * ow - a.pl + 1) * hq + quad;           // tap j of a ring row at [src_d + j hq]
        out_off[u] = ow * a.hid + 4 * quad;
    }
    int8_t* ybase = a.y + (size_t)chunk * a.OH * a.OW * a.hid;
    auto consume = [&](int k, int ti) {                       // hidden row k -> window slot ti of every item
        const int* src = ring8 + (k & 1) * rowd;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live_i[u]) continue;
            if (row_ok(k)) {
                const int r0 = src[src_d[u]], r1 = src[src_d[u] + hq], r2 = src[src_d[u] + 2 * hq];
                const int lo = perm(r1, r0, 0x05010400u), hi = perm(r1, r0, 0x07030602u);
                T[u][ti][0] = perm(r2, lo, 0x0c040100u);
                T[u][ti][1] = perm(r2, lo, 0x0c050302u);
                T[u][ti][2] = perm(r2, hi, 0x0c060100u);
                T[u][ti][3] = perm(r2, hi, 0x0c070302u);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) T[u][ti][e] = zprow;
            }
        }
    };
    auto emit = [&](int i0, int i1, int i2, int oh) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live_i[u]) continue;
            const v4i* c = dwt + 6 * quad_i[u];
            const v4i w0 = c[0], w1 = c[1], w2 = c[2], bb = c[3], mm = c[4], ss = c[5];
            int qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int acc = dot4_first(T[u][i0][e], w0[e], bb[e]);
                acc = dot4(T[u][i1][e], w1[e], acc);
                acc = dot4(T[u][i2][e], w2[e], acc);
                qv[e] = med3(mbqm(acc, mm[e], ss[e]) + a.o_zp, a.o_amin, a.o_amax);
            }
            const int word = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
            *reinterpret_cast<int*>(ybase + (size_t)oh * a.OW * a.hid + out_off[u]) = word;
        }
    };

    // ---- the walk: step k = expand row k (its B fragments were requested a step ago), barrier, rows into the windows, output row if due
    v4i bfa[2][KS], bfb[2][KS];
    request(bfa, 0);
    __syncthreads();                                          // border columns
    auto step = [&](const v4i (&cur)[2][KS], v4i (&nxt)[2][KS], int k, int ti, int i0, int i1) {
        request(nxt, k + 1);
        expand(cur, k);
        __syncthreads();
        consume(k, ti);
        if (k >= 2 && (k - 2) % S == 0 && (k - 2) / S < nrows) emit(i0, i1, ti, oh0 + (k - 2) / S);
    };
    for (int k = 0; k < nhid; k += 6) {                       // (window slots and the two B buffers rotate with periods 3 and 2)
        step(bfa, bfb, k + 0, 0, 1, 2);
        if (k + 1 < nhid) step(bfb, bfa, k + 1, 1, 2, 0);
        if (k + 2 < nhid) step(bfa, bfb, k + 2, 2, 0, 1);
        if (k + 3 < nhid) step(bfb, bfa, k + 3, 0, 1, 2);
        if (k + 4 < nhid) step(bfa, bfb, k + 4, 1, 2, 0);
        if (k + 5 < nhid) step(bfb, bfa, k + 5, 2, 0, 1);
    }
}

bool i8_pwdw_supported(const DwPw8Args& e, const I8ConvGeom& d) {
    if (!g_opt.i8_strip || e.has_dw || e.transposed || e.add.enabled || e.lut || e.qx || e.gate || e.H != e.OH || e.W != e.OW) return false;
    if (d.H != e.H || d.W != e.W || d.C != e.Cout || d.sh != d.sw || (d.sh != 1 && d.sh != 2) || d.zp_in != e.pw_zp_out) return false;
    if (e.W % 16 || e.Cout % 16 || e.Cin % 4 || e.Cin > 128 || (e.W / 16) * (e.Cout / 16) > 24) return false;
    if ((long)d.OW * (e.Cout / 4) > 2 * 768 || (size_t)2 * (e.W + 2) * e.Cout > 60000) return false;
    return (long)e.H * e.W * e.Cin < 0x7fff0000L && (long)d.OH * d.OW * d.C < 0x7fff0000L;
}

bool launch_i8_pwdw(const DwPw8Args& e, const I8ConvGeom& d, const int8_t* dw_w, const int32_t* dw_b, const int32_t* dw_mult, const int32_t* dw_shift,
                    int8_t* y, hipStream_t s) {
    int rb = d.OH;
    while (rb > 16) rb = (rb + 1) / 2;
    PwDw8Args a{e.x, y, e.pw_w, e.pw_b, e.pw_mult, e.pw_shift, dw_w, dw_b, dw_mult, dw_shift, e.B, e.H, e.W, e.Cin, e.Cout, d.OH, d.OW, d.pt, d.pl, rb,
                e.pw_zp_out, e.pw_amin, e.pw_amax, d.zp_out, d.amin, d.amax};
    const unsigned blocks = (unsigned)((long)e.B * ((d.OH + rb - 1) / rb));
    const size_t smem = (size_t)2 * (e.W + 2) * e.Cout + (size_t)(e.Cout / 4) * 9 * 16;  // ring + per-quad constants
    const int ks = (e.Cin + 63) / 64;
#define BN_PWDW8(SV, KSV) \
    if (d.sh == SV && ks == KSV) { hipLaunchKernelGGL((i8_pwdw_kernel<SV, KSV>), dim3(blocks), dim3(768), smem, s, a); return true; }
    BN_PWDW8(1, 1) BN_PWDW8(2, 1) BN_PWDW8(1, 2) BN_PWDW8(2, 2)
#undef BN_PWDW8
    return false;
}

bool launch_i8_dw_stream(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias, const int32_t* mult,
                         const int32_t* shift, hipStream_t s, int32_t* pool) {
    if (!g_opt.i8_strip || g.sh != g.sw || (g.sh != 1 && g.sh != 2) || g.C % 4 || (long)g.H * g.W * g.C >= 0x7fff0000L ||
        (long)g.OH * g.OW * g.C >= 0x7fff0000L)
        return false;
    int cq = 16;
    while ((g.C / 4) % cq) cq >>= 1;
    DwStream8Args a{x, y, w, bias, mult, shift, B, g.H, g.W, g.C, g.OH, g.OW, 0, g.pt, g.pl, g.zp_in, g.zp_out, g.amin, g.amax, cq, g.rq_right, pool};
    const int ncol = 64 / cq;
    const long per_row_block = (long)B * (g.C / (4 * cq)) * ((g.OW + ncol - 1) / ncol);
    int th = g.OH;
    while (th > 64) th = (th + 1) / 2;  // (a wave's prologue — weights, constants, the first rows — is paid once per th rows)
    while (th > 4 && per_row_block * ((g.OH + th - 1) / th) < 8192) th = (th + 1) / 2;
    if (const int v = g_opt.i8_strip_th; v >= 1) th = v < g.OH ? v : g.OH;
    a.TH = th;
    const long waves = per_row_block * ((g.OH + th - 1) / th);
    const bool hi = g_opt.i8_pw_forms && (g.rq_right & 3) == 3 && g.amin >= g.zp_out;  // (bit 1: every shift in [-20, -1])
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (g.sh == 1) {
        if (hi) hipLaunchKernelGGL((i8_dw_stream_kernel<1, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((i8_dw_stream_kernel<1, false>), grid, dim3(256), 0, s, a);
    } else {
        if (hi) hipLaunchKernelGGL((i8_dw_stream_kernel<2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((i8_dw_stream_kernel<2, false>), grid, dim3(256), 0, s, a);
    }
    return true;
}

bool launch_i8_stem_stream(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias, const int32_t* mult,
                           const int32_t* shift, hipStream_t s) {
    if (!g_opt.i8_strip || (g.sh != 1 && g.sh != 2) || g.sw < 1 || g.sw > 2 || g.C % 4 || (long)g.OH * g.OW * g.C >= 0x7fff0000L) return false;
    int cq = 16;
    while ((g.C / 4) % cq) cq >>= 1;
    DwStream8Args a{x, y, w, bias, mult, shift, B, g.H, g.W, g.C, g.OH, g.OW, 0, g.pt, g.pl, g.zp_in, g.zp_out, g.amin, g.amax, cq, g.rq_right, nullptr};
    const int ncol = 64 / cq;
    const long per_row_block = (long)B * (g.C / (4 * cq)) * ((g.OW + ncol - 1) / ncol);
    int th = g.OH;
    while (th > 64) th = (th + 1) / 2;  // (as in launch_i8_dw_stream: the prologue is paid once per th rows; 64 rows measured 0.118 -> 0.093 ms on the 64 x 128 stem)
    while (th > 4 && per_row_block * ((g.OH + th - 1) / th) < 8192) th = (th + 1) / 2;
    if (const int v = g_opt.i8_strip_th; v >= 1) th = v < g.OH ? v : g.OH;
    a.TH = th;
    const long waves = per_row_block * ((g.OH + th - 1) / th);
    const bool hi = g_opt.i8_pw_forms && (g.rq_right & 3) == 3 && g.amin >= g.zp_out;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (g.sh == 1) {
        if (hi) hipLaunchKernelGGL((i8_stem_stream_kernel<1, true>), grid, dim3(256), 0, s, a, g.sw);
        else hipLaunchKernelGGL((i8_stem_stream_kernel<1, false>), grid, dim3(256), 0, s, a, g.sw);
    } else {
        if (hi) hipLaunchKernelGGL((i8_stem_stream_kernel<2, true>), grid, dim3(256), 0, s, a, g.sw);
        else hipLaunchKernelGGL((i8_stem_stream_kernel<2, false>), grid, dim3(256), 0, s, a, g.sw);
    }
    return true;
}

// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_i8_strip() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&i8_front_strip_kernel));
}

}  // namespace bn

#ifdef BN_TAIL_STAMPS
// debug export of the stamps build only: stamps of the row-streaming depthwise launch with (C, H) ([4096][6] int64, zeroed by the caller)
extern "C" __attribute__((visibility("default"))) int bn_debug_dw_stamps(long long* d_buf, int C, int H) {
    const int sel[2] = {C, H};
    if (hipMemcpyToSymbol(HIP_SYMBOL(bn::g_dw_sel), sel, sizeof sel) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(bn::g_dw_stamps), &d_buf, sizeof d_buf) == hipSuccess ? 0 : -1;

}
#endif
