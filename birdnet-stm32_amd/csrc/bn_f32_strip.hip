// bn_f32_strip.hip — float32 depthwise-separable block of the wide early stages (Cin, Cout in {32, 64}) as row-streaming
// strips, the float32 sibling of bn_i8_strip.hip:
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

struct Row4 { v4f t[3]; };  // the three taps (columns j = 0..2) of one input row, one channel quad

template <int NW, int COUT, int S, bool RES>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu((NW == 2 && COUT == 32) ? 4 : 3))) void f32_strip_kernel(DwPwArgs a) {
    constexpr int CIN = 16 * NW, CWO = COUT / NW, NT = CWO / 16;
    static_assert(NT == 1 || NT == 2, "16 or 32 output channels per wave");
    static_assert(NW == 2 || NW == 4, "two or four waves per strip");
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
        v4f f[NW];
#pragma unroll
        for (int ks = 0; ks < NW; ++ks) f[ks] = buf[ks][lane];
        v4f o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            o[t] = pb[t];
#pragma unroll
            for (int ks = 0; ks < NW; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[t][ks][g], f[ks][g], o[t], 0, 0, 0);
            // One chain after the other: the scheduler otherwise interleaves the two tiles' chains (t0, t1, t0, t1, ...), and with ONE
            // independent MFMA between dependent ones an accumulator register of the first chain came out short of a term in output
            // columns 12-15 (1 launch in ~50 on MI355X; the single-chain variants never failed in hundreds of launches).
            __builtin_amdgcn_sched_barrier(0);
        }
        // The B operands must stay untouched until the MFMAs have READ them, and on MI355X v_mfma_f32_16x16x4_f32 (8 passes) reads B
        // pass by pass: output columns 12-15 use the values B holds ~30 cycles after issue.  The compiler assumes operands are
        // consumed at issue — it re-used a dying B register as the destination of a chain's last MFMA (`v_mfma_f32_16x16x4_f32
        // v[46:49], v33, v49, v[58:61]`: columns 12-15 wrong in every run) and scheduled `v_mov_b32 v6, s33` right behind an MFMA
        // reading v6 (wrong in 1 run of 100).  Here every B register is an in/out operand of a 32-cycle wait placed after the last
        // MFMA, so nothing can write one before that; tools/mfma_overlap_check.py scans the assembly for the first pattern.
        if constexpr (NW == 2)
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(f[0]), "+v"(f[1]));
        else
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            v4f r = o[t];
            if constexpr (RES) r += T[i1].t[1];  // centre tap = the block input at this position, channels 16 w + 4 q + 0..3
            r = act4(r, pw_bounds);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, r), rs_out,
                                                   voff_out + 16 * t, oh * a.OW * COUT * 4, 0);
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

template <int NW, int COUT, int S, bool RES>
void launch_strip(const DwPwArgs& a, hipStream_t s) {
    const long strips = (long)a.B * (a.OW / 16) * ((a.OH + a.TH - 1) / a.TH);
    hipLaunchKernelGGL((f32_strip_kernel<NW, COUT, S, RES>), dim3((unsigned)strips), dim3(64 * NW), 0, s, a);
}

}  // namespace

bool f32_strip_supported(const DwPwArgs& a) {
    if (!a.has_dw || a.gate || a.OW % 16 || a.sh != a.sw || (a.sh != 1 && a.sh != 2)) return false;
    if (a.res && (a.res != a.x || a.sh != 1 || a.Cin != a.Cout)) return false;
    // 128 -> 128 (eight waves per strip) needs ~145 registers per lane: one workgroup per CU, no faster than the tile kernel
    const bool shape = (a.Cin == 32 && (a.Cout == 32 || a.Cout == 64)) || (a.Cin == 64 && (a.Cout == 64 || a.Cout == 128));
    return shape && (long)a.H * a.W * a.Cin * 4 < 0x7fff0000L;
}

void launch_f32_strip(DwPwArgs a, hipStream_t s) {
    const int nw = a.Cin / 16;
    // rows per strip (measured at B = 1024): 16 on the 32-row maps (32: fewer, longer strips leave CUs idle at the end; 8: the
    // two-row prologue weighs 25 %), the whole map below that; shorter only while the launch would not fill the chip once
    int th = a.OH;
    while (th > 16) th = (th + 1) / 2;
    while (th > 4 && (long)a.B * (a.OW / 16) * ((a.OH + th - 1) / th) * nw < 4096) th = (th + 1) / 2;
    if (const char* e = getenv("BN_F32_STRIP_TH")) {  // tests: force the rows per strip
        const int v = atoi(e);
        if (v >= 1) th = v < a.OH ? v : a.OH;
    }
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
    BN_FSTRIP(4, 128, 2, false)
#undef BN_FSTRIP
}

}  // namespace bn
