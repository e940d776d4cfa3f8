// bn_i8_tail.hip — the back half of the INT8 graph as ONE kernel: every block behind stage 2
//
//   DEPTHWISE_CONV_2D 3x3 (ReLU6) -> CONV_2D 1x1 on the int8 matrix cores [-> TFLite ADD with the block input]        (x n_layers)
//   -> MEAN -> FULLY_CONNECTED -> LOGISTIC / DEQUANTIZE
//
// with the activation maps of kTailG chunks resident in LDS from the first block to the scores (reference operators: SURVEY.md
// Appendix B ops #36-#55; models/dscnn.py:209-261 of the reference).  Same integer semantics as the per-block kernels
// (bn_i8_strip.hip / bn_i8_fused.hip), bit-identical results; what changes is where the maps live and how the waves synchronise.
//
// The per-block kernels of these small maps (8 x 16 and 4 x 8 positions) spend 46-70 % of their wave cycles parked at the per-row
// barrier that lets the waves of a strip exchange their depthwise outputs, and each of them reloads its constants per two chunks
// (profiles/r02_i8_b4096_digest.md).  Here a workgroup of 16 waves owns kTailG = 4 chunks at a time:
//
//   * a WAVE owns a tile of 16 positions (one row of a 16-wide map, two rows of an 8-wide one) from the depthwise stage to the
//     stored output: lane (n, kq) = (position n of the tile, lane group kq) reads the nine taps of ITS channels straight from the
//     map in LDS (one ds_read_b32 per tap and channel quad, immediate offsets), and its depthwise outputs ARE its B fragments of
//     v_mfma_i32_16x16x64_i8 for all Cin channels — no exchange between waves, so the only workgroup barrier is the one between
//     blocks;
//   * the pointwise weights of the block sit in LDS in fragment order (one ds_read_b128 per A operand, 1 KB per wave, conflict-free);
//     the K order is permuted by the packer so that the quads of lane groups kq and kq ^ 1 lie 16 quads apart: with a map pitch of
//     Cin + 4 bytes the 32 lanes of an LDS access cycle then hit 32 different banks;
//   * output channels land four consecutive per lane (natural row order of the A operand): requantise, [ADD], pack, one
//     ds_write_b32 into the next map.  The ADD cannot use the per-block kernels' 64 KB table here (the maps need the LDS): its two
//     single-byte input rescales are 256-entry tables, the sum is requantised in registers;
//   * SAME padding: a tap outside the map reads a row of zero points kept in LDS (an address select per tap and tile, no
//     per-element test);
//   * the first block streams its taps from global memory (its input map, 32 KB per chunk, never enters LDS);
//   * after the last block: MEAN per (chunk, channel) thread, FULLY_CONNECTED per (chunk, class) thread, table + dequantise.
//
// LDS plan per block (computed on the host, bn::tail_plan): input map, output map, pointwise weights, depthwise / pointwise
// constants, ADD tables, zero-point row — first fit into 160 KB; blocks whose maps do not fit make the plan fall back to the
// per-block operators (bn_api.hip).
#include "bn_tail_common.h"

namespace bn {
namespace {

// BN_TAIL_STAMPS (tools/tail_stamps.py builds lib/libbirdnet_hip_stamps.so with it; never defined in the production library): every wave
// of the first workgroups records, per block, when it entered, finished staging, left the staging barrier, how long its depthwise and
// pointwise phases took, and when it left the end barrier (s_memrealtime: 100 MHz) — the attribution of the kernel's parked cycles.
#ifdef BN_TAIL_STAMPS
__device__ long long* g_tail_stamps = nullptr;   // [workgroup < 8][group < 4][block < 8][wave 16][6]
constexpr int kStampWg = 8, kStampGrp = 4;
__device__ __forceinline__ long long now_ticks() { return (long long)__builtin_amdgcn_s_memrealtime(); }
#define BN_STAMP(var) const long long var = now_ticks()
#else
#define BN_STAMP(var) do {} while (0)
#endif

// first channel quad of lane group kq (mirrors _tail_quad_base in models/_lower_i8.py)
template <int CIN>
__device__ __forceinline__ int pq_base(int kq) {
    if constexpr (CIN == 128) return ((kq & 1) << 4) | ((kq >> 1) << 3);
    else if constexpr (CIN == 256) return kq << 4;
    else return kq << 2;
}

// One block for the kTailG chunks of the workgroup.  `lds` = the workgroup's LDS; maps are [chunk][position][C + 4 bytes].
// The geometry is a template parameter (H x W input map, stride S): positions turn into shifts and the padding tests into
// comparisons with constants; tail_plan() only accepts the four shapes instantiated below.
template <int CIN, int COUT, int S, int H, int W, bool ADD, bool SRCG>
__device__ __forceinline__ void tail_block(const Tail8Layer& L, const Tail8Args& a, unsigned char* lds, int chunk0, int stamp_slot = -1) {
    BN_STAMP(st0);
    constexpr int KS = CIN / 64;        // k-steps of v_mfma_i32_16x16x64_i8
    constexpr int NTILES = COUT / 16;
    constexpr int PIN = CIN + 4, POUT = COUT + 4;
    constexpr int OH = H / S, OW = W / S, PER_CHUNK = OH * OW;
    constexpr int TILES = kTailG * PER_CHUNK / 16;
    constexpr int NGRP = TILES >= kTailWaves ? 1 : kTailWaves / TILES;  // few tiles: split them over groups of output channels
    constexpr int NT_PER = NTILES / NGRP;
    constexpr int kSkewBit = CIN == 64 ? 2 : 4;  // the bit of a quad's index that tells lane group kq from kq ^ 1 (pq_base)
    constexpr int PT = S == 1 ? 1 : 0, PL = PT;  // TF SAME padding of a 3x3 window on even maps: 1 / 1 at stride 1, 0 / 1 at stride 2
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;

    // ---- stage the block's constants: pointwise weights, depthwise / pointwise constants, ADD tables, zero-point row ----------
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // the staging copies' 64-bit pointers (constant block + 16 tid, + 112 tid) are recomputed per block: hoisted to
                                       // the top of the kernel they stayed live across all blocks and two registers spilled at the 128-register cap
        const v4i* g = reinterpret_cast<const v4i*>(a.cst);
        v4i* dst = reinterpret_cast<v4i*>(lds + L.w_off);
#pragma unroll
        for (int i = 0; i < (CIN * COUT / 16 + kTailThreads - 1) / kTailThreads; ++i)
            if (i * kTailThreads + tid < CIN * COUT / 16) dst[i * kTailThreads + tid] = g[L.g_w / 4 + i * kTailThreads + tid];
        if (tid < CIN / 4) {   // per channel quad: three weight rows, bias, then the requantisation constants in rq_hi form (8 x 16 bytes)
            const v4i* q = g + L.g_dwc / 4 + tid * 7;
            // (one more 16-byte slot of skew per lane group's run of quads: the lanes of kq and kq ^ 1 share a ds_read_b128 group and their
            // quads lie a multiple of 256 bytes apart — without it every read of these constants is a 2-way bank conflict)
            v4i* d = reinterpret_cast<v4i*>(lds + L.dwc_off) + tid * 8 + (tid >> kSkewBit);
            d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
            stage_rq(d + 4, q[4], q[5], q[6]);
        }
        if (tid < COUT / 4) {  // per (tile, lane group): bias + constants (5 x 16 bytes); a residual block keeps (multiplier, c1, shift) for rq
            const v4i* q = g + L.g_pwc / 4 + tid * 4;
            v4i* d = reinterpret_cast<v4i*>(lds + L.pwc_off) + tid * 5;
            d[0] = q[0];
            if constexpr (ADD) { d[1] = q[1]; d[2] = q[2]; d[3] = q[3]; d[4] = (v4i){0, 0, 0, 0}; }
            else stage_rq(d + 1, q[1], q[2], q[3]);
        }
        if (ADD && tid < 128) reinterpret_cast<v4i*>(lds + L.lut_off)[tid] = g[L.g_lut / 4 + tid];
        if (!SRCG && tid < PIN / 4) reinterpret_cast<int*>(lds + L.zp_off)[tid] = (L.zp_in & 0xff) * 0x01010101;
    }
    BN_STAMP(st1);
    __syncthreads();
    BN_STAMP(st2);
#ifdef BN_TAIL_STAMPS
    long long t_dw = 0, t_pw = 0;
#endif

    const int zp4 = (L.zp_in & 0xff) * 0x01010101;
    const int qb = pq_base<CIN>(kq);
    const v4i* dwc = reinterpret_cast<const v4i*>(lds + L.dwc_off) + qb * 8 + (qb >> kSkewBit);
    const v4i* pwc = reinterpret_cast<const v4i*>(lds + L.pwc_off) + kq * 5;
    const int* lut = reinterpret_cast<const int*>(lds + L.lut_off);
    const v4i* wl = reinterpret_cast<const v4i*>(lds + L.w_off) + lane;
    const int dw_lo = L.dw_lo, dw_hi = L.dw_hi, pw_lo = L.pw_lo, pw_hi = L.pw_hi;
    const int add_m = L.add_m, add_e1 = L.add_e - 1;
    const long add_c = rq64(L.add_c1);

    // A wave takes UPW tiles at a time (two when every wave has two): the depthwise / pointwise constants and the A operands are
    // read from LDS once for both (the LDS pipe is the second-busiest unit of this kernel after the vector ALU).
    constexpr int UPW = (TILES * NGRP) % (2 * kTailWaves) == 0 && !SRCG ? 2 : 1;
    for (int u0 = wave * UPW; u0 < TILES * NGRP; u0 += kTailWaves * UPW) {
        int p[UPW], grp[UPW];
        int taddr[UPW][9];
        unsigned okmask = 0;
        int chunk = 0;
#pragma unroll
        for (int t = 0; t < UPW; ++t) {
            const int u = u0 + t;
            const int tile = u / NGRP;
            grp[t] = u % NGRP;
            p[t] = tile * 16 + n;                    // position over the kTailG chunks
            const int g = p[t] / PER_CHUNK, pc = p[t] % PER_CHUNK;
            const int oy = pc / OW, ox = pc % OW;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int iy = oy * S - PT + dy, ix = ox * S - PL + dx;
                    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    if constexpr (SRCG) {
                        taddr[t][dy * 3 + dx] = ok ? (iy * W + ix) * CIN + 4 * qb : 0;
                        okmask |= ok ? 1u << (dy * 3 + dx) : 0u;
                    } else if constexpr (!(UPW == 2 && S == 1 && OW == 16)) {  // (the shared-row form below has its own twelve addresses)
                        taddr[t][dy * 3 + dx] = (ok ? L.x_off + ((g * H + iy) * W + ix) * PIN : L.zp_off) + 4 * qb;
                    }
                }
            chunk = chunk0 + g;
            if (chunk >= a.B) chunk = a.B - 1;       // ragged last group: the spare slots repeat the last chunk
        }
        // (only the first block reads global memory; the other instantiations never use the descriptor)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int8_t*>(a.x) + (SRCG ? (size_t)chunk * H * W * CIN : 0), 0, SRCG ? H * W * CIN : 0, 0x00020000);

        // ---- depthwise 3x3 for all CIN channels of this lane's position(s) -> B fragments -------------------------------------
        BN_STAMP(sa);
        v4i bf[UPW][KS];
        // Two tiles of a wave at stride 1 on a 16-wide map are two vertically adjacent output rows of one chunk (u0 is even, a chunk has
        // an even number of rows): they share two of their three input rows.  Four rows of taps are read and byte-transposed once for
        // both (12 LDS reads and 24 permutes per channel quad instead of 18 and 36), and 12 tap addresses stay live instead of 18 —
        // that form ran into the 128-register cap of the 16-wave workgroup (16 spilled registers, 68 B of scratch per lane in round 2).
        constexpr bool ROWS4 = UPW == 2 && S == 1 && OW == 16 && !SRCG;
        int raddr[ROWS4 ? 4 : 1][3];
        if constexpr (ROWS4) {
            const int g = p[0] / PER_CHUNK, oy = (p[0] % PER_CHUNK) / OW, ox = n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int iy = oy - 1 + r, ix = ox - 1 + dx;
                    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    raddr[r][dx] = (ok ? L.x_off + ((g * H + iy) * W + ix) * PIN : L.zp_off) + 4 * qb;
                }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int frag[UPW][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qi = 4 * ks + j;   // quad qb + qi: channels 4 (qb + qi) ..
                const v4i bias = dwc[qi * 8 + 3];
                int acc[UPW][4];
#pragma unroll
                for (int t = 0; t < UPW; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] = bias[e];
                if constexpr (ROWS4) {
                    v4i wprev = (v4i){0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int rr[3];
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) rr[dx] = *reinterpret_cast<const int*>(lds + raddr[r][dx] + 4 * qi);
                        const int lo = perm(rr[1], rr[0], 0x05010400u), hi = perm(rr[1], rr[0], 0x07030602u);  // bytes (tap0, tap1, tap2, 0) per channel
                        const int tr[4] = {perm(rr[2], lo, 0x0c040100u), perm(rr[2], lo, 0x0c050302u), perm(rr[2], hi, 0x0c060100u), perm(rr[2], hi, 0x0c070302u)};
                        v4i w = wprev;
                        if (r < 3) {  // input row r is window row r of the upper output row ...
                            w = dwc[qi * 8 + r];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[0][e] = dot4(tr[e], w[e], acc[0][e]);
                        }
                        if (r > 0) {  // ... and window row r - 1 of the lower one
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[1][e] = dot4(tr[e], wprev[e], acc[1][e]);
                        }
                        wprev = w;
                    }
                } else {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const v4i w = dwc[qi * 8 + dy];
#pragma unroll
                    for (int t = 0; t < UPW; ++t) {
                        int r[3];
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            if constexpr (SRCG) {
                                const int v = __builtin_amdgcn_raw_buffer_load_b32(rs, taddr[t][dy * 3 + dx] + 4 * qi, 0, 0);
                                r[dx] = (okmask >> (dy * 3 + dx)) & 1 ? v : zp4;
                            } else {
                                r[dx] = *reinterpret_cast<const int*>(lds + taddr[t][dy * 3 + dx] + 4 * qi);
                            }
                        }
                        const int lo = perm(r[1], r[0], 0x05010400u), hi = perm(r[1], r[0], 0x07030602u);  // bytes (tap0, tap1, tap2, 0) per channel
                        acc[t][0] = dot4(perm(r[2], lo, 0x0c040100u), w[0], acc[t][0]);
                        acc[t][1] = dot4(perm(r[2], lo, 0x0c050302u), w[1], acc[t][1]);
                        acc[t][2] = dot4(perm(r[2], hi, 0x0c060100u), w[2], acc[t][2]);
                        acc[t][3] = dot4(perm(r[2], hi, 0x0c070302u), w[3], acc[t][3]);
                    }
                }
                }
                const v4i m = dwc[qi * 8 + 4], c01 = dwc[qi * 8 + 5], c23 = dwc[qi * 8 + 6];
                const int e1 = reinterpret_cast<const int*>(dwc + qi * 8 + 7)[0];
                const long cc[4] = {pair(c01.x, c01.y), pair(c01.z, c01.w), pair(c23.x, c23.y), pair(c23.z, c23.w)};
#pragma unroll
                for (int t = 0; t < UPW; ++t) {
                    int qv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) qv[e] = med3(rq_hi(acc[t][e], m[e], cc[e], e1, e), dw_lo, dw_hi);
                    frag[t][j] = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
                }
            }
#pragma unroll
            for (int t = 0; t < UPW; ++t) bf[t][ks] = (v4i){frag[t][0], frag[t][1], frag[t][2], frag[t][3]};
        }

        // ---- pointwise 1x1 on the matrix cores, requantise, [ADD], store into the next map ------------------------------------
        // accumulator rows 4 kq .. 4 kq + 3 of tile nt = output channels 16 nt + 4 kq ..; the tiles of a wave share the channel group
#ifdef BN_TAIL_STAMPS
        asm volatile("" :: "v"(bf[0][0]) : "memory");  // (the stamp sits behind the depthwise results, not in front of them)
#endif
        BN_STAMP(sb);
        for (int tt = 0; tt < NT_PER; ++tt) {
            const int nt = grp[0] * NT_PER + tt;
            const v4i* pc4 = pwc + nt * 20;
            v4i acc[UPW];
#pragma unroll
            for (int t = 0; t < UPW; ++t) acc[t] = pc4[0];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i af = wl[(nt * KS + ks) * 64];
#pragma unroll
                for (int t = 0; t < UPW; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf[t][ks], acc[t], 0, 0, 0);
            }
            const v4i m = pc4[1], c1 = pc4[2], sh = pc4[3];  // ADD: (multiplier, c1, shift); else (multiplier, C01, C23) + packed shifts
            const int e1 = ADD ? 0 : reinterpret_cast<const int*>(pc4 + 4)[0];
            const long cc[4] = {pair(c1.x, c1.y), pair(c1.z, c1.w), pair(sh.x, sh.y), pair(sh.z, sh.w)};
#pragma unroll
            for (int t = 0; t < UPW; ++t) {
                const int yrow = L.y_off + p[t] * POUT + 4 * kq;
                int res = 0;
                if constexpr (ADD) res = *reinterpret_cast<const int*>(lds + L.x_off + p[t] * PIN + 4 * kq + 16 * nt);  // residual: same position
                int qv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // with the ADD: value + 128 = index of the second table (any sign: full rounding)
                    int v = med3(ADD ? rq(acc[t][e], m[e], c1[e], sh[e]) : rq_hi(acc[t][e], m[e], cc[e], e1, e), pw_lo, pw_hi);
                    if constexpr (ADD) {
                        // the ADD's own rescale: its clamp starts at the zero point (fused ReLU6; the packer checks it), so the sign term of the
                        // rounding shift is not needed and the result is the high dword of one 64-bit multiply-add, shifted (rq_hi's form with
                        // wave-uniform constants): 4 instructions instead of 7
                        const int sa = lut[(res >> (8 * e)) & 0xff], sb = lut[256 + v];
                        v = med3((int)(((long)(sa + sb) * (long)add_m + add_c) >> 32) >> add_e1, L.add_lo, L.add_hi);
                    }
                    qv[e] = v;
                }
                *reinterpret_cast<int*>(lds + yrow + 16 * nt) = perm(perm(qv[3], qv[2], 0x0c0c0400u), perm(qv[1], qv[0], 0x0c0c0400u), 0x05040100u);
            }
        }
#ifdef BN_TAIL_STAMPS
        asm volatile("" ::: "memory");
        const long long sc = now_ticks();
        t_dw += sb - sa;
        t_pw += sc - sb;
#endif
    }
    BN_STAMP(st4);
    __syncthreads();
#ifdef BN_TAIL_STAMPS
    if (stamp_slot >= 0 && g_tail_stamps && (threadIdx.x & 63) == 0) {
        long long* o = g_tail_stamps + ((size_t)stamp_slot * kTailWaves + (threadIdx.x >> 6)) * 6;
        const long long st5 = now_ticks();
        o[0] = st1 - st0; o[1] = st2 - st1; o[2] = t_dw; o[3] = t_pw; o[4] = st5 - st4; o[5] = st5 - st0;
    }
#endif
}

__global__ __launch_bounds__(kTailThreads) void i8_tail_kernel(Tail8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int ngroups = (a.B + kTailG - 1) / kTailG;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int chunk0 = grp * kTailG;
        for (int li = 0; li < a.n_layers; ++li) {
            const Tail8Layer& L = a.L[li];
            int slot = -1;
#ifdef BN_TAIL_STAMPS
            const int gi = (grp - (int)blockIdx.x) / (int)gridDim.x;
            if ((int)blockIdx.x < kStampWg && gi < kStampGrp && li < 8) slot = ((int)blockIdx.x * kStampGrp + gi) * 8 + li;
#endif
            if (L.Cin == 64) tail_block<64, 128, 2, 16, 32, false, true>(L, a, lds, chunk0, slot);
            else if (L.Cin == 128 && L.Cout == 128) tail_block<128, 128, 1, 8, 16, true, false>(L, a, lds, chunk0, slot);
            else if (L.Cin == 128) tail_block<128, 256, 2, 8, 16, false, false>(L, a, lds, chunk0, slot);
            else tail_block<256, 256, 1, 4, 8, true, false>(L, a, lds, chunk0, slot);
        }
        tail_head<kTailThreads, 4>(a, lds, chunk0);
        __syncthreads();  // the next group overwrites the maps
    }
}

// first-fit allocator over the workgroup's LDS: `used` holds [begin, end) intervals that must stay intact
struct Span {
    int b, e;
};
int first_fit(std::vector<Span>& used, int bytes, int cap) {
    bytes = (bytes + 15) & ~15;
    int pos = 0;
    for (;;) {
        bool moved = false;
        for (const Span& s : used)
            if (pos < s.e && pos + bytes > s.b) {
                pos = (s.e + 15) & ~15;
                moved = true;
            }
        if (!moved) break;
    }
    if (pos + bytes > cap) return -1;
    used.push_back({pos, pos + bytes});
    return pos;
}

}  // namespace

// Build the kernel arguments from the descriptor table the packer wrote (24 words per block + 16 head words) and plan the LDS of
// every block.  Returns false when the topology is not one the kernel takes or a block's maps do not fit 160 KB.
bool tail_plan(const int32_t* desc, int n_words, int n_layers, Tail8Args& a) {
    constexpr int LW = 24, HW = 16, CAP = 160 * 1024;
    if (n_layers < 1 || n_layers > 8 || n_words != LW * n_layers + HW) return false;
    a.n_layers = n_layers;
    int cur_off = -1;   // where the current input map lives (-1: global memory)
    int lds_need = 0;
    for (int i = 0; i < n_layers; ++i) {
        const int32_t* d = desc + LW * i;
        Tail8Layer& L = a.L[i];
        L.H = d[0]; L.W = d[1]; L.Cin = d[2]; L.Cout = d[3]; L.S = d[4]; L.OH = d[5]; L.OW = d[6]; L.pt = d[7]; L.pl = d[8]; L.has_add = d[9];
        L.zp_in = d[10]; L.dw_lo = d[11]; L.dw_hi = d[12]; L.pw_lo = d[13]; L.pw_hi = d[14];
        L.add_m = d[15]; L.add_c1 = d[16]; L.add_e = d[17]; L.add_lo = d[18]; L.add_hi = d[19];
        L.g_w = d[20]; L.g_dwc = d[21]; L.g_pwc = d[22]; L.g_lut = d[23];
        const bool first = i == 0;
        auto is = [&](int cin, int cout, int st, int hh, int ww, int add) {
            return L.Cin == cin && L.Cout == cout && L.S == st && L.H == hh && L.W == ww && L.has_add == add && L.pt == (st == 1) && L.pl == (st == 1);
        };
        // the four instantiations of tail_block (bn_i8_tail.hip: i8_tail_kernel)
        const bool shape_ok = (first && is(64, 128, 2, 16, 32, 0)) || (!first && (is(128, 128, 1, 8, 16, 1) || is(128, 256, 2, 8, 16, 0) || is(256, 256, 1, 4, 8, 1)));
        if (!shape_ok || (L.S != 1 && L.S != 2) || L.OH != (L.H + L.S - 1) / L.S || L.OW != (L.W + L.S - 1) / L.S || (L.OH * L.OW) % 16 ||
            L.pt < 0 || L.pt > 1 || L.pl < 0 || L.pl > 1 || L.H < 1 || L.W < 1 || L.H * L.W > 4096)
            return false;
        if (L.has_add && (L.Cin != L.Cout || L.S != 1 || L.g_lut < 0 || L.add_e < 1 || L.add_e > 22 || L.add_m < 0)) return false;
        if (i > 0 && (L.H != a.L[i - 1].OH || L.W != a.L[i - 1].OW || L.Cin != a.L[i - 1].Cout)) return false;
        const int tiles = kTailG * L.OH * L.OW / 16;
        if (tiles < kTailWaves && (kTailWaves % tiles || (L.Cout / 16) % (kTailWaves / tiles))) return false;
        if ((L.g_w | L.g_dwc | L.g_pwc) & 3 || (L.has_add && (L.g_lut & 3))) return false;
        // offsets into the constant block start at 0 (tail_const_words bounds them from above: a stale or hostile blob must not make the
        // kernel read in front of the block), and every clamp whose result indexes a table or is stored as a byte is an int8 range
        // (+ 128 where the pointwise value of a residual block is kept as the index of the second ADD table)
        if (L.g_w < 0 || L.g_dwc < 0 || L.g_pwc < 0 || (L.has_add && L.g_lut < 0)) return false;
        const int off = L.has_add ? 128 : 0;
        if (L.dw_lo < -128 || L.dw_hi > 127 || L.dw_lo > L.dw_hi || L.pw_lo < -128 + off || L.pw_hi > 127 + off || L.pw_lo > L.pw_hi) return false;
        if (L.has_add && (L.add_lo < -128 || L.add_hi > 127 || L.add_lo > L.add_hi)) return false;
        std::vector<Span> used;
        L.x_off = cur_off;
        if (cur_off >= 0) used.push_back({cur_off, cur_off + kTailG * L.H * L.W * (L.Cin + 4)});
        L.y_off = first_fit(used, kTailG * L.OH * L.OW * (L.Cout + 4), CAP);
        L.w_off = first_fit(used, L.Cin * L.Cout, CAP);
        L.dwc_off = first_fit(used, L.Cin * 32 + 64, CAP);   // staged in rq_hi form: 8 x 16 bytes per channel quad (+ up to four slots of skew),
        L.pwc_off = first_fit(used, L.Cout * 20, CAP);  // 5 x 16 bytes per four output channels
        L.lut_off = L.has_add ? first_fit(used, 2048, CAP) : 0;
        L.zp_off = first_fit(used, L.Cin + 16, CAP);
        if (L.y_off < 0 || L.w_off < 0 || L.dwc_off < 0 || L.pwc_off < 0 || L.lut_off < 0 || L.zp_off < 0) return false;
        if (i == n_layers - 1) {
            a.mean_off = first_fit(used, kTailG * L.Cout, CAP);
            if (a.mean_off < 0) return false;
        }
        for (const Span& s : used) lds_need = s.e > lds_need ? s.e : lds_need;
        cur_off = L.y_off;
    }
    const int32_t* h = desc + LW * n_layers;
    a.mean_zp_in = h[0]; a.mean_mult = h[1]; a.mean_shift = h[2]; a.mean_zp_out = h[3];
    a.fc_zp_out = h[4]; a.fc_lo = h[5]; a.fc_hi = h[6]; a.g_fcw = h[7]; a.g_fcb = h[8]; a.g_fcm = h[9]; a.g_fcs = h[10]; a.g_hlut = h[11];
    a.head_zp_fc = h[12]; a.head_zp_out = h[13]; a.P = h[14]; a.C = h[15];
    if (a.g_fcw < 0 || a.g_fcb < 0 || a.g_fcm < 0 || a.g_fcs < 0 || a.g_hlut < -1 || (a.g_fcw & 3)) return false;
    if (a.fc_lo < -128 || a.fc_hi > 127 || a.fc_lo > a.fc_hi) return false;  // (the classifier byte + 128 indexes the head's table)
    const Tail8Layer& last = a.L[n_layers - 1];
    if (a.P != last.OH * last.OW || a.C != last.Cout || a.C % 4 || a.NC < 1 || kTailG * a.NC > kTailThreads * 4) return false;
    // the head's LDS copy of the classifier matrix overlays the last block's pointwise weights (4 x 1024 threads x 16 bytes are staged at most)
    const int fc_bytes = a.NC * (a.C / 4 + 1) * 4;
    a.fcw_off = (a.C % 16 == 0 && fc_bytes <= last.Cin * last.Cout && a.NC * (a.C / 16) <= 4 * kTailThreads) ? last.w_off : -1;
    a.lds_bytes = lds_need;
    return true;
}

// words of the constant block the kernel reads (for the load-time check of the tensor's size)
long tail_const_words(const Tail8Args& a) {
    long need = 0;
    auto upto = [&](long off, long words) { need = off + words > need ? off + words : need; };
    for (int i = 0; i < a.n_layers; ++i) {
        const Tail8Layer& L = a.L[i];
        upto(L.g_w, (long)L.Cin * L.Cout / 4);
        upto(L.g_dwc, (long)L.Cin * 7);
        upto(L.g_pwc, (long)L.Cout * 4);
        if (L.has_add) upto(L.g_lut, 512);
    }
    upto(a.g_fcw, (long)a.NC * a.C / 4);
    upto(a.g_fcb, a.NC);
    upto(a.g_fcm, a.NC);
    upto(a.g_fcs, a.NC);
    if (a.g_hlut >= 0) upto(a.g_hlut, 64);
    return need;
}

#ifdef BN_TAIL_STAMPS
// debug export of the stamps build only: where the kernel writes its stamps ([8][4][8][16][6] int64, zeroed by the caller)
extern "C" __attribute__((visibility("default"))) int bn_debug_tail_stamps(long long* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamps), &d_buf, sizeof d_buf) == hipSuccess ? 0 : -1;
}
#endif

bool launch_i8_tail(Tail8Args a, hipStream_t s) {
    if (!g_opt.i8_tail_fclds) a.fcw_off = -1;
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(i8_tail_kernel), 160 * 1024)) return false;
    const int ngroups = (a.B + kTailG - 1) / kTailG;
    int cus = 256;
    const int grid = ngroups < cus ? ngroups : cus;  // one workgroup per CU (its LDS), each walks over its share of the chunk groups
    hipLaunchKernelGGL(i8_tail_kernel, dim3(grid), dim3(kTailThreads), (size_t)a.lds_bytes, s, a);
    return true;
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_i8_tail() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&i8_tail_kernel));
}

}  // namespace bn
