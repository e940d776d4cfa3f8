// bn_i8_tail2.hip — the back half of the INT8 graph as ONE kernel, second form: the DEPTHWISE stage runs on the matrix cores too.
//
//   DEPTHWISE_CONV_2D 3x3 (ReLU6) -> CONV_2D 1x1 [-> TFLite ADD with the block input]        (x n_layers)
//   -> MEAN -> FULLY_CONNECTED -> LOGISTIC / DEQUANTIZE
//
// Same operators, same integer results as bn_i8_tail.hip (reference: SURVEY.md Appendix B ops #36-#55; models/dscnn.py:28-84,
// 209-261 of the reference).  What changes is which pipe does the depthwise multiply-accumulates.
//
// Why.  profiles/r04_i8_b4096_digest.md: i8_tail_kernel issues 33.6 k vector instructions per wave against 640 matrix instructions;
// its vector ALU is busy 59 % of the launch, its matrix pipe 4.7 %.  Per depthwise output the vector form needs a tap read per
// channel quad and window position, 1.5 byte permutes, 0.75 dot products, then the requantisation: ~12 instructions.  Here a
// 3 x 3 window row over a tile of 16 channels x 16 positions is ONE v_mfma_i32_16x16x64_i8:
//
//   A (16 x 64, weights)      row m = channel m of the tile, contraction index k = 16 g + c': window column g (g = 3: zero), channel c';
//                             A[m][16 g + c'] = w[dy][g][channel m] where c' = m, else 0  — block-diagonal, one byte in sixteen is used
//   B (64 x 16, activations)  column n = position n of the tile; lane (n, g) holds the 16 channels of the tile at input position
//                             (row + dy, column n + g - 1): ONE ds_read_b128 from the NHWC map in LDS, no transposition
//   D (16 x 16)               lane (n, g) holds channels 4 g .. 4 g + 3 at position n: four requantisations, one dword of NHWC bytes
//
// and a window is three of them (dy = 0, 1, 2) into one accumulator whose start value is the folded bias.  The matrix pipe does 16 x
// the useful work and does not care (it was idle); the vector ALU is left with the requantisation alone: 3.75 instructions per
// depthwise output.  A wave's depthwise results for the channel tiles 4 ks .. 4 ks + 3 ARE its B fragment of the pointwise
// convolution's k-step ks (the packer orders the contraction index of the pointwise weights accordingly), as in the first form.
//
// Geometry: a workgroup of 8 waves (up to 256 registers per lane) owns kTailG = 4 chunks; a wave owns UPW consecutive tiles of 16
// positions (four rows of a 16-wide map of one chunk; one tile = two rows of an 8-wide map) so that the A operands and the
// per-channel constants it reads from LDS serve four tiles and vertically adjacent tiles share their input rows (six B reads per
// channel tile for four tiles).  Maps live in LDS with a pitch of C + 16 bytes (16-byte reads 144 / 272 bytes apart are
// conflict-free) and are updated IN PLACE: all depthwise results of a block sit in registers before a barrier, then the pointwise
// results overwrite the input map — which frees the LDS the expanded depthwise weights need (3 KB per channel tile).
//
// The residual ADD is arithmetic here (no table reads: the LDS array is the busiest shared unit once the depthwise taps come as
// 16-byte reads): the block's own value enters with multiplier 2^30 / shift 0 (it has the larger scale: (v - z) << 19 exactly), the
// residual byte's rescale is one unsigned 64-bit multiply-add and a shift (tail2_constants in models/_lower_i8.py proves both forms on
// all 256 bytes), the sum's rescale is the sign-free form the other stages use.
//
// Constants never make a wave wait for memory: a block's depthwise part (expanded weights + constants) is requested while the PREVIOUS
// block's pointwise phase runs and written to LDS behind it, its pointwise part (weights + constants) is requested before the block's
// own depthwise phase and written behind that — two barriers per block, none of them behind a load (tail2_plan places the parts so that
// nothing live is overwritten).
//
// The first block streams its taps from global memory (range-checked 16-byte buffer loads: a tap outside the map reads 0; what the
// folded bias assumed for it — the zero point — is taken back through per-border variants of the bias).
#include "bn_tail_common.h"

namespace bn {
namespace {

// BN_TAIL_STAMPS (the measurement build `make stamps`, never the production library): per block and wave, when it entered, had issued its
// staging copies, left the staging barrier, finished the depthwise phase, left the middle barrier, finished the pointwise phase, left the end
// barrier (s_memrealtime: 100 MHz) — tools/tail2_stamps.py.
#ifdef BN_TAIL_STAMPS
__device__ long long* g_tail2_stamps = nullptr;   // [workgroup < 8][group < 4][block < 8][wave 8][8]
// (only the stamps right behind a barrier are taken — 0, 4, 6: a stamp in the middle of a phase needs a scheduling fence and changes what it measures;
// the stamped launch is within 2 % of the production one)
#define BN_T2STAMP(i) do { if ((i) == 0 || (i) == 4 || (i) == 6) st[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BN_T2STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ long upair(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | lo); }

// Constant parts travel global -> LDS without passing through registers (global_load_lds_dwordx4: every lane's 16 bytes land at the wave's
// LDS base + 16 lane): requested before a compute phase, complete at the barrier behind it (__syncthreads waits for the wave's outstanding
// vector-memory operations, which is what orders the LDS-DMA for the readers).  n16 = sixteen-byte pieces; MAXP rounds of kTail2Threads pieces.
template <int MAXP>
__device__ __forceinline__ void part_request(const int32_t* src, unsigned char* dst, int n16, int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int idx = i * kTail2Threads + tid;
        if (idx < n16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (size_t)idx),
                                             (__attribute__((address_space(3))) void*)(dst + (i * kTail2Threads + wave * 64) * 16), 16, 0, 0);
    }
}
constexpr int kTail2MaxDw16 = 7 * kTail2Threads;   // sixteen-byte pieces of the largest depthwise part a block may prefetch (256 channels: 3392)

__device__ __forceinline__ int tail2_dw_bytes(const Tail2Layer& L, bool first) { return (L.Cin / 16) * (3 * 1024 + (first ? 8 : 5) * 64); }

// MultiplyByQuantizedMultiplier for either sign in four instructions (the block's own term of a residual ADD: no activation, no clamp):
//   hi = (x M' + C) >> 32 with M' = the dword 2 m read as signed (= 2 m - 2^32) and C = (2^(e-1) << 32) + 2^31
//      = SRDHM(x, m) + 2^(e-1) - x          (x M' = 2 x m - x 2^32; SRDHM(x, m) = (2 x m + 2^31) >> 32)
//   v  = (hi + x + (x >> 31)) >> e          (the sign of SRDHM(x, m) is the sign of x for 2^30 < m < 2^31: tail2_constants checks)
// The four shifts of a channel quad sit in the bytes of one register.
__device__ __forceinline__ int rq_signed(int x, int m2, long c, int e_packed, int e) {
    const int hi = (int)(((long)x * (long)m2 + c) >> 32);
    const int s = hi + x + (x >> 31);
    int r;
    switch (e) {
        case 0: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(e_packed), "v"(s)); break;
        case 1: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(e_packed), "v"(s)); break;
        case 2: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(e_packed), "v"(s)); break;
        default: asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(e_packed), "v"(s)); break;
    }
    return r;
}

// MEAN -> FULLY_CONNECTED -> LOGISTIC / DEQUANTIZE for the workgroup's chunks (reference operators #52-#55).
//   MEAN: thread (chunk slot, channel quad) walks the positions with one dword read each (waves 0-3);
//   FULLY_CONNECTED on the matrix cores: wave w owns the class tile 16 w .., its A fragments come straight from memory (requested before the
//   MEAN, 1 KB per k-step, contiguous), B = the pooled vectors from LDS (column n = chunk slot n & 3: columns 4..15 repeat, lanes n < 4 store).
__device__ __forceinline__ void tail2_head(const Tail2Args& a_, unsigned char* lds, int chunk0, int stamp_slot = -1) {
#ifdef BN_TAIL_STAMPS
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st[0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // (the head's dozen scalar arguments are read from the kernel-argument segment HERE, through a pointer the optimiser cannot see through:
    // hoisted to the top of the kernel they stayed live across every block and the scalar registers spilled)
    const Tail2Args* ap = (const Tail2Args*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const Tail2Args& a = *ap;
    (void)a_;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const Tail2Layer& L = a.L[a.n_layers - 1];
    const int C = a.C, KS = C / 64, nct = (a.NC + 15) / 16, pitch = C + 16;
    const bool fc_wave = wave < nct;
    // (waves without a class tile repeat the last tile's requests: uniform code, nothing stored)
    const v4i* gw = reinterpret_cast<const v4i*>(a.cst + a.g_fcw) + (fc_wave ? wave : nct - 1) * KS * 64 + lane;
    const v4i af0 = gw[0], af1 = KS > 1 ? gw[64] : af0, af2 = KS > 2 ? gw[128] : af0, af3 = KS > 3 ? gw[192] : af0;
    const v4i* fc = reinterpret_cast<const v4i*>(a.cst + a.g_fcb) + (fc_wave ? wave : nct - 1) * 12 + g;   // [kind][g]
    const v4i fc_b = fc[0], fc_mu = fc[4], fc_sh = fc[8];
    for (int i = tid; i < kTailG * (C / 4); i += kTail2Threads) {
        const int gq = i / (C / 4), cq = i - gq * (C / 4);
        const int* src = reinterpret_cast<const int*>(lds + L.y_off + gq * a.P * pitch + 4 * cq);
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = 0;
        for (; k + 8 <= a.P; k += 8) {   // eight independent reads in flight
            int x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = src[(k + u) * (pitch / 4)];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += (int)(int8_t)x[u]; s1 += (int)(int8_t)(x[u] >> 8); s2 += (int)(int8_t)(x[u] >> 16); s3 += x[u] >> 24; }
        }
        for (; k < a.P; ++k) {
            const int x = src[k * (pitch / 4)];
            s0 += (int)(int8_t)x; s1 += (int)(int8_t)(x >> 8); s2 += (int)(int8_t)(x >> 16); s3 += x >> 24;
        }
        const int q[4] = {mean_q(s0, a.P, a.mean_zp_in, a.mean_mult, a.mean_shift, a.mean_zp_out), mean_q(s1, a.P, a.mean_zp_in, a.mean_mult, a.mean_shift, a.mean_zp_out),
                          mean_q(s2, a.P, a.mean_zp_in, a.mean_mult, a.mean_shift, a.mean_zp_out), mean_q(s3, a.P, a.mean_zp_in, a.mean_mult, a.mean_shift, a.mean_zp_out)};
        reinterpret_cast<int*>(lds + a.mean_off)[i] = pack4(q);
    }
    __syncthreads();
#ifdef BN_TAIL_STAMPS
    st[4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if (fc_wave) {
        v4i acc = fc_b;
        const v4i mu = fc_mu, sh = fc_sh;
        const unsigned char* mv = lds + a.mean_off + (n & (kTailG - 1)) * C + 16 * g;
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af0, *reinterpret_cast<const v4i*>(mv), acc, 0, 0, 0);
        if (KS > 1) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1, *reinterpret_cast<const v4i*>(mv + 64), acc, 0, 0, 0);
        if (KS > 2) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af2, *reinterpret_cast<const v4i*>(mv + 128), acc, 0, 0, 0);
        if (KS > 3) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af3, *reinterpret_cast<const v4i*>(mv + 192), acc, 0, 0, 0);
        const int chunk = chunk0 + n;
        if (n < kTailG && chunk < a.B) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * wave + 4 * g + r;
                if (j < a.NC) {
                    const int qv = clampi(mbqm(acc[r], mu[r], sh[r]) + a.fc_zp_out, a.fc_lo, a.fc_hi);
                    const size_t o = (size_t)chunk * a.NC + j;
                    if (a.logits) a.logits[o] = (float)(qv - a.head_zp_fc) * a.s_fc;
                    if (a.g_hlut >= 0) {
                        const int ov = reinterpret_cast<const int8_t*>(a.cst + a.g_hlut)[qv + 128];
                        a.scores[o] = (float)(ov - a.head_zp_out) * a.s_head;
                    } else {
                        a.scores[o] = (float)(qv - a.head_zp_fc) * a.s_fc;
                    }
                }
            }
        }
    }
#ifdef BN_TAIL_STAMPS
    if (stamp_slot >= 0 && g_tail2_stamps && (threadIdx.x & 63) == 0) {
        long long* o = g_tail2_stamps + ((size_t)stamp_slot * kTail2Waves + (threadIdx.x >> 6)) * 8;
        o[0] = st[0]; o[4] = st[4];
    }
#endif
}

// One block for the G chunks of the workgroup; maps are [chunk][position][C + 16 bytes], input and output at the same place.
// A wave owns TPW tiles of 16 positions stacked vertically in ONE column strip of ONE chunk (a 32-wide map has two strips; a tile of an
// 8-wide map is two rows) and walks them in passes of at most four.
template <int G, int CIN, int COUT, int S, int H, int W, bool ADD, bool SRCG>
__device__ __forceinline__ void tail2_block(const Tail2Layer& L, const Tail2Args& a, unsigned char* lds, int chunk0, int nxi, int stamp_slot = -1) {
#ifdef BN_TAIL_STAMPS
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    BN_T2STAMP(0);
    constexpr int KS = (CIN + 63) / 64, NCT = CIN / 16, NT = COUT / 16;   // (32 input channels: the k-step is half empty, zero weights)
    constexpr int PIN = CIN + 16, POUT = COUT + 16;
    constexpr int OH = H / S, OW = W / S, PER_CHUNK = OH * OW;
    constexpr int TILES = G * PER_CHUNK / 16;
    static_assert(TILES % kTail2Waves == 0, "tiles per wave");
    constexpr int TPW = TILES / kTail2Waves;       // tiles of a wave
    constexpr int UPW = TPW < 4 ? TPW : 4;         // ... of which it walks UPW at a time
    constexpr int PASSES = TPW / UPW;
    static_assert(TPW % UPW == 0, "passes");
    constexpr int TR = OW == 8 ? 2 : 1;            // output rows of a tile
    constexpr int STRIPS = OW == 8 ? 1 : OW / 16;  // column strips of the map
    constexpr int WPC = kTail2Waves / G;           // waves per chunk
    static_assert(kTail2Waves % G == 0 && WPC % STRIPS == 0 && (WPC / STRIPS) * TPW * TR == OH, "a wave's tiles: one strip of one chunk");
    constexpr int PSTEP = TR * OW;                 // positions from a tile to the one below it
    constexpr int NR = (UPW - 1) * TR * S + 3;     // input rows a wave reads per channel tile and pass
    constexpr int PT = S == 1 ? 1 : 0, PL = PT;    // TF SAME padding of a 3x3 window on even maps: 1 / 1 at stride 1, 0 / 1 at stride 2
    constexpr int DWK = SRCG ? 8 : 5;              // v4i per (channel tile, lane group) of depthwise constants
    constexpr int W_BYTES = KS * 64 * COUT, DWA_BYTES = NCT * 3 * 1024, DWC_BYTES = NCT * DWK * 64, PWC_BYTES = NT * 5 * 64;
    constexpr int PW16 = (W_BYTES + PWC_BYTES) / 16, PWP = (PW16 + kTail2Threads - 1) / kTail2Threads;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;

    // ---- request the block's pointwise part (its depthwise part and zero-point row are in LDS already) ------------------------------
    part_request<PWP>(a.cst + L.g_cst + (DWA_BYTES + DWC_BYTES) / 4, lds + L.pw_off, PW16, tid);
    BN_T2STAMP(1);
    BN_T2STAMP(2);

    const v4i* wl = reinterpret_cast<const v4i*>(lds + L.pw_off) + lane;
    const v4i* dwa = reinterpret_cast<const v4i*>(lds + L.dw_off) + lane;
    const v4i* dwc = reinterpret_cast<const v4i*>(lds + L.dw_off + DWA_BYTES) + g;
    const v4i* pwc = reinterpret_cast<const v4i*>(lds + L.pw_off + W_BYTES) + g;
    const int dw_lo = L.dw_lo, dw_hi = L.dw_hi, pw_lo = L.pw_lo, pw_hi = L.pw_hi;

    // ---- where this wave's tiles are --------------------------------------------------------------------------------------------
    const int gch = wave / WPC, wi = wave % WPC;               // chunk slot; which of the chunk's vertical runs
    const int cx0 = (wi % STRIPS) * 16;                        // first column of the strip
    const int oy_base = (wi / STRIPS) * TPW * TR;              // first output row of the run
    const int lr = OW == 8 ? n >> 3 : 0, ox = OW == 8 ? n & 7 : cx0 + n;
    int chunk = chunk0 + gch;
    if (chunk >= a.B) chunk = a.B - 1;                         // ragged last group: the spare slots repeat the last chunk
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t*>(a.x) + (SRCG ? (size_t)chunk * H * W * CIN : 0), 0, SRCG ? H * W * CIN : 0, 0x00020000);
    v4i bf[TPW][KS];
    if constexpr (CIN % 64 != 0) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) bf[t][KS - 1] = (v4i){0, 0, 0, 0};
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
    const int oy0 = oy_base + pass * UPW * TR;                 // first output row of the pass
    int raddr[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int iy = (oy0 + lr) * S - PT + r, ix = ox * S - PL + g;
        const bool ok = g < 3 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        if constexpr (SRCG) raddr[r] = ok ? (iy * W + ix) * CIN : 0x40000000;   // (past the buffer: the load returns 0)
        else raddr[r] = ok ? L.x_off + ((gch * H + iy) * W + ix) * PIN : L.zp_off;
    }
    int bkind[UPW];   // first block: which bias a tile's lanes start from (0 inside, 5 right border, 6 bottom, 7 both)
#pragma unroll
    for (int t = 0; t < UPW; ++t) {
        const bool right = ox == OW - 1, bottom = oy0 + t * TR + lr == OH - 1;
        bkind[t] = SRCG ? (right ? (bottom ? 7 : 5) : (bottom ? 6 : 0)) * 4 : 0;
    }

    // ---- depthwise 3x3 on the matrix cores: all CIN channels of this wave's positions -> B fragments of the pointwise stage -------
    // (taps from memory: the rows of up to four channel tiles — all of a 64-channel block — are requested before the first is used: one round trip)
    constexpr int AHEAD = NCT < 4 ? NCT : 4;
    v4i grow[SRCG ? NCT : 1][SRCG ? NR : 1];
    if constexpr (SRCG) {
#pragma unroll
        for (int ct = 0; ct < AHEAD && ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < NR; ++r) grow[ct][r] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, raddr[r] + 16 * ct, 0, 0));
    }
    // Software pipeline over the channel tiles, written out (left to itself the scheduler merges all iterations of this loop and runs out of
    // registers): iteration ct requests the operands of tile ct + 1, issues the matrix instructions of tile ct and requantises tile ct - 1.
    constexpr int NB0 = SRCG ? UPW : 1;   // start values (the folded bias): one per tile where the border decides which, else one for all
    v4i af[2][3], brow[2][NR], acc[2][UPW], b0[2][NB0], rqm[2], rq01[2], rq23[2];
    int rqe[2];
    auto request = [&](int ct) {   // operands of channel tile ct: A fragments, input rows, start values
        const int b = ct & 1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) af[b][dy] = dwa[(ct * 3 + dy) * 64];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if constexpr (SRCG) {
                brow[b][r] = grow[ct][r];
                if (ct + AHEAD < NCT) grow[ct + AHEAD][r] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, raddr[r] + 16 * (ct + AHEAD), 0, 0));
            } else {
                brow[b][r] = *reinterpret_cast<const v4i*>(lds + raddr[r] + 16 * ct);
            }
        }
        const v4i* dc = dwc + ct * DWK * 4;
#pragma unroll
        for (int t = 0; t < NB0; ++t) b0[b][t] = SRCG ? dc[bkind[t]] : dc[0];
    };
    auto request_rq = [&](int ct) {   // its requantisation constants (needed one iteration later than the rest)
        const int b = ct & 1;
        const v4i* dc = dwc + ct * DWK * 4;
        rqm[b] = dc[4]; rq01[b] = dc[8]; rq23[b] = dc[12];
        rqe[b] = reinterpret_cast<const int*>(dc + 16)[0];
    };
    request(0);
#pragma unroll
    for (int ct = 0; ct <= NCT; ++ct) {
        if (ct + 1 < NCT) request(ct + 1);
        if (ct < NCT) {
            const int b = ct & 1;
            request_rq(ct);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int t = 0; t < UPW; ++t)
                    acc[b][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[b][dy], brow[b][t * TR * S + dy], dy ? acc[b][t] : b0[b][SRCG ? t : 0], 0, 0, 0);
        }
        if (ct > 0) {
            const int pt = ct - 1, b = pt & 1, c = b;
            const long cc[4] = {pair(rq01[c].x, rq01[c].y), pair(rq01[c].z, rq01[c].w), pair(rq23[c].x, rq23[c].y), pair(rq23[c].z, rq23[c].w)};
#pragma unroll
            for (int t = 0; t < UPW; ++t) {
                int qv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) qv[e] = med3(rq_hi(acc[b][t][e], rqm[c][e], cc[e], rqe[c], e), dw_lo, dw_hi);
                bf[pass * UPW + t][pt >> 2][pt & 3] = pack4(qv);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }  // pass
    BN_T2STAMP(3);
    __syncthreads();   // the pointwise part has landed, and every wave has read its taps: the map may be overwritten
    BN_T2STAMP(4);
    // the successor's depthwise part (block 0 of the next group behind the last block): requested now, complete at the end barrier
    const Tail2Layer& N = a.L[nxi < 0 ? 0 : nxi];
    const int nx_n16 = nxi < 0 ? 0 : tail2_dw_bytes(N, nxi == 0) / 16;
    part_request<kTail2MaxDw16 / kTail2Threads>(a.cst + N.g_cst, lds + N.dw_off, nx_n16, tid);

    // ---- pointwise 1x1 on the matrix cores, requantise, [ADD], store into the map -------------------------------------------------
    const int add_m = L.add_m, add_e1 = L.add_e - 1;
    const long add_c = rq64(L.add_c1);
    const uint32_t res_m = (uint32_t)L.res_m;
    const long res_c = upair((uint32_t)L.res_c_lo, (uint32_t)L.res_c_hi);
    const int res_k = L.res_k;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
    const int pbase = (gch * OH + oy_base + pass * UPW * TR) * OW + (OW == 8 ? 0 : cx0) + n;   // this lane's position in the first tile of the pass
    // Software pipeline over the tiles of 16 output channels, written out like the depthwise loop's: iteration nt requests the A fragments and
    // start values of tile nt + 2 and the epilogue's operands (constants, residual bytes) of tile nt + 1, issues the matrix instructions of tile
    // nt + 1 and runs the epilogue of tile nt — no LDS round trip and no matrix-pipe latency in front of the vector work.
    v4i paf[2][KS], pb0[2], pacc[2][UPW], pm[2], pc01[2], pc23[2];
    int pe1[2], pres[2][ADD ? UPW : 1];
    auto request_a = [&](int nt) {
        const int b = nt & 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) paf[b][ks] = wl[(nt * KS + ks) * 64];
        pb0[b] = pwc[nt * 20];
    };
    auto request_e = [&](int nt) {
        const int b = nt & 1;
        const v4i* pc = pwc + nt * 20;
        pm[b] = pc[4]; pc01[b] = pc[8]; pc23[b] = pc[12];   // (multiplier, C01, C23) + packed shifts; with the ADD: the signed form's operands (rq_signed)
        pe1[b] = reinterpret_cast<const int*>(pc + 16)[0];
        if constexpr (ADD) {
#pragma unroll
            for (int t = 0; t < UPW; ++t) pres[b][t] = *reinterpret_cast<const int*>(lds + L.x_off + (pbase + PSTEP * t) * PIN + 4 * g + 16 * nt);
        }
    };
    auto matrix = [&](int nt) {
        const int b = nt & 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < UPW; ++t) pacc[b][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(paf[b][ks], bf[pass * UPW + t][ks], ks ? pacc[b][t] : pb0[b], 0, 0, 0);
    };
    request_a(0);
    request_e(0);
    if (NT > 1) request_a(1);
    matrix(0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int b = nt & 1;
        if (nt + 1 < NT) {
            request_e(nt + 1);
            matrix(nt + 1);
        }
        if (nt + 2 < NT) request_a(nt + 2);   // (into the fragments' slot of tile nt: its matrix instructions were issued an iteration ago)
        const long cc[4] = {pair(pc01[b].x, pc01[b].y), pair(pc01[b].z, pc01[b].w), pair(pc23[b].x, pc23[b].y), pair(pc23[b].z, pc23[b].w)};
#pragma unroll
        for (int t = 0; t < UPW; ++t) {
            const int p = pbase + PSTEP * t;
            const int xr = ADD ? pres[b][t] ^ (int)0x80808080u : 0;  // residual bytes + 128
            int qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (ADD) {
                    const int v = rq_signed(pacc[b][t][e], pm[b][e], cc[e], pe1[b], e);                       // the block's own value minus its zero point
                    const uint32_t xt = (uint32_t)perm(0, xr, ((uint32_t)e << 24) | 0x000c0c0cu);            // (b + 128) << 24
                    const int f = (int)((uint32_t)(((unsigned long)xt * (unsigned long)res_m + (unsigned long)res_c) >> 32) >> res_k);
                    const int total = (v << 19) + f;
                    qv[e] = med3((int)(((long)total * (long)add_m + add_c) >> 32) >> add_e1, L.add_lo, L.add_hi);
                } else {
                    qv[e] = med3(rq_hi(pacc[b][t][e], pm[b][e], cc[e], pe1[b], e), pw_lo, pw_hi);
                }
            }
            *reinterpret_cast<int*>(lds + L.y_off + p * POUT + 4 * g + 16 * nt) = pack4(qv);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }  // pass
    BN_T2STAMP(5);
    if (nxi > 0 && tid < (N.Cin + 16) / 4) reinterpret_cast<int*>(lds + N.zp_off)[tid] = (N.zp_in & 0xff) * 0x01010101;   // (block 0 has no zero-point row)
    __syncthreads();
    BN_T2STAMP(6);
#ifdef BN_TAIL_STAMPS
    if (stamp_slot >= 0 && g_tail2_stamps && (threadIdx.x & 63) == 0) {
        long long* o = g_tail2_stamps + ((size_t)stamp_slot * kTail2Waves + (threadIdx.x >> 6)) * 8;
        for (int i = 0; i < 7; ++i) o[i] = st[i];
    }
#endif
}

__global__ __launch_bounds__(kTail2Threads) void i8_tail2_kernel(Tail2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int ngroups = (a.B + kTailG - 1) / kTailG;
    {   // the first block's depthwise part, once; afterwards every block finds its own staged by its predecessor
        part_request<kTail2MaxDw16 / kTail2Threads>(a.cst + a.L[0].g_cst, lds + a.L[0].dw_off, tail2_dw_bytes(a.L[0], true) / 16, (int)threadIdx.x);
        __syncthreads();
    }
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int chunk0 = grp * kTailG;
        const bool more = grp + (int)gridDim.x < ngroups;
        for (int li = 0; li < a.n_layers; ++li) {
            const Tail2Layer& L = a.L[li];
            const bool last = li == a.n_layers - 1;
            const Tail2Layer& N = a.L[last ? 0 : li + 1];
            const int nx = last ? (more ? 0 : -1) : li + 1;   // whose depthwise part this block stages (-1: nothing follows)
            (void)N;
            int slot = -1;
#ifdef BN_TAIL_STAMPS
            const int gi = (grp - (int)blockIdx.x) / (int)gridDim.x;
            if ((int)blockIdx.x < 8 && gi < 4 && li < 8 && g_tail2_stamps) slot = ((int)blockIdx.x * 4 + gi) * 8 + li;
#endif
            if (L.Cin == 64) tail2_block<kTailG, 64, 128, 2, 16, 32, false, true>(L, a, lds, chunk0, nx, slot);
            else if (L.Cin == 128 && L.Cout == 128) tail2_block<kTailG, 128, 128, 1, 8, 16, true, false>(L, a, lds, chunk0, nx, slot);
            else if (L.Cin == 128) tail2_block<kTailG, 128, 256, 2, 8, 16, false, false>(L, a, lds, chunk0, nx, slot);
            else tail2_block<kTailG, 256, 256, 1, 4, 8, true, false>(L, a, lds, chunk0, nx, slot);
        }
        int hslot = -1;
#ifdef BN_TAIL_STAMPS
        const int gi = (grp - (int)blockIdx.x) / (int)gridDim.x;
        if ((int)blockIdx.x < 8 && gi < 4 && g_tail2_stamps) hslot = ((int)blockIdx.x * 4 + gi) * 8 + 6;
#endif
        tail2_head(a, lds, chunk0, hslot);
        __syncthreads();  // the next group overwrites the maps
#ifdef BN_TAIL_STAMPS
        if (hslot >= 0 && (threadIdx.x & 63) == 0) g_tail2_stamps[((size_t)hslot * kTail2Waves + (threadIdx.x >> 6)) * 8 + 6] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    }
}

// Stage 2 of the shipped graph with the same blocks: kMidG = 2 chunks per workgroup (a 16 x 32 map of 64 channels is 40 KB in LDS), the first
// block's taps from memory, the last map written back for the tail kernel (coalesced 16-byte pieces).
__global__ __launch_bounds__(kTail2Threads) void i8_mid2_kernel(Tail2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int ngroups = (a.B + kMidG - 1) / kMidG;
    part_request<kTail2MaxDw16 / kTail2Threads>(a.cst + a.L[0].g_cst, lds + a.L[0].dw_off, tail2_dw_bytes(a.L[0], true) / 16, (int)threadIdx.x);
    __syncthreads();
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int chunk0 = grp * kMidG;
        const bool more = grp + (int)gridDim.x < ngroups;
        for (int li = 0; li < a.n_layers; ++li) {
            const Tail2Layer& L = a.L[li];
            const bool last = li == a.n_layers - 1;
            const int nx = last ? (more ? 0 : -1) : li + 1;
            if (li == 0) tail2_block<kMidG, 32, 64, 2, 32, 64, false, true>(L, a, lds, chunk0, nx);
            else tail2_block<kMidG, 64, 64, 1, 16, 32, true, false>(L, a, lds, chunk0, nx);
        }
        // the last map: [chunk slot][position][C + 16] in LDS -> [chunk][position][C] in memory
        const Tail2Layer& L = a.L[a.n_layers - 1];
        const int c16 = L.Cout / 16, per = a.P * c16;
        for (int i = threadIdx.x; i < kMidG * per; i += kTail2Threads) {
            const int gq = i / per, r = i - gq * per, pos = r / c16, c = r - pos * c16;
            if (chunk0 + gq < a.B)
                reinterpret_cast<v4i*>(a.y + ((size_t)(chunk0 + gq) * a.P + pos) * L.Cout)[c] =
                    *reinterpret_cast<const v4i*>(lds + L.y_off + (gq * a.P + pos) * (L.Cout + 16) + 16 * c);
        }
        __syncthreads();  // the next group overwrites the map
    }
}

int tail2_dw_part(const Tail2Layer& L, bool first) { return (L.Cin / 16) * (3 * 1024 + (first ? 8 : 5) * 64); }
int tail2_pw_part(const Tail2Layer& L) { return (L.Cin + 63) / 64 * 64 * L.Cout + (L.Cout / 16) * 5 * 64; }

struct Span2 {
    int b, e;
};
int first_fit2(const std::vector<Span2>& used, int bytes, int cap) {
    bytes = (bytes + 15) & ~15;
    int pos = 0;
    for (;;) {
        bool moved = false;
        for (const Span2& s : used)
            if (pos < s.e && pos + bytes > s.b) {
                pos = (s.e + 15) & ~15;
                moved = true;
            }
        if (!moved) break;
    }
    return pos + bytes > cap ? -1 : pos;
}
bool overlap2(int b0, int e0, int b1, int e1) { return b0 < e1 && b1 < e0; }

}  // namespace

// Kernel arguments and LDS plan from the packer's descriptor table (32 words per block + 16 head words).  A block's map sits at LDS
// offset 0 (input and output in place).  Its depthwise part and zero-point row are written while the PREVIOUS block's pointwise phase
// runs: they avoid that block's output map (= this block's input) and pointwise part; its pointwise part is written while its own
// depthwise phase runs: it avoids the map, the zero-point row and the depthwise part.  The first block's depthwise part is written behind
// the last block of the previous group and stays through the head: it avoids the last map, the last pointwise part, the pooled vector and
// the head's copy of the classifier matrix.  false = not a topology / size the kernel takes (the plan keeps i8_tail_kernel).
bool tail2_plan(const int32_t* desc, int n_words, int n_layers, Tail2Args& a, bool mid) {
    constexpr int LW = kTail2LayerWords, CAP = 160 * 1024;
    const int HW = mid ? 0 : 16, G = mid ? kMidG : kTailG;
    if (n_layers < 1 || n_layers > 8 || n_words != LW * n_layers + HW) return false;
    a.n_layers = n_layers;
    int lds_need = 0;
    Span2 prev_pw{0, 0};
    auto grow = [&](int end) { lds_need = end > lds_need ? end : lds_need; };
    // The first block's depthwise part is the one piece whose neighbours in time are at BOTH ends of the chain (it is written behind the last
    // block's pointwise phase): placed by first fit like the others it can collide with the last block's parts, so a second attempt puts it at
    // the top of the LDS.
    for (int attempt = 0; attempt < 2; ++attempt) {
    lds_need = 0;
    prev_pw = {0, 0};
    for (int i = 0; i < n_layers; ++i) {
        const int32_t* d = desc + LW * i;
        Tail2Layer& L = a.L[i];
        L.H = d[0]; L.W = d[1]; L.Cin = d[2]; L.Cout = d[3]; L.S = d[4]; L.OH = d[5]; L.OW = d[6]; L.pt = d[7]; L.pl = d[8]; L.has_add = d[9];
        L.zp_in = d[10]; L.dw_lo = d[11]; L.dw_hi = d[12]; L.pw_lo = d[13]; L.pw_hi = d[14];
        L.add_m = d[15]; L.add_c1 = d[16]; L.add_e = d[17]; L.add_lo = d[18]; L.add_hi = d[19];
        L.res_m = d[20]; L.res_c_lo = d[21]; L.res_c_hi = d[22]; L.res_k = d[23]; L.g_cst = d[24];
        const bool first = i == 0;
        auto is = [&](int cin, int cout, int st, int hh, int ww, int add) {
            return L.Cin == cin && L.Cout == cout && L.S == st && L.H == hh && L.W == ww && L.has_add == add && L.pt == (st == 1) && L.pl == (st == 1) &&
                   L.OH == hh / st && L.OW == ww / st;
        };
        // the instantiations of tail2_block: four in i8_tail2_kernel, two in i8_mid2_kernel
        if (!mid && !((first && is(64, 128, 2, 16, 32, 0)) || (!first && (is(128, 128, 1, 8, 16, 1) || is(128, 256, 2, 8, 16, 0) || is(256, 256, 1, 4, 8, 1))))) return false;
        if (mid && !((first && is(32, 64, 2, 32, 64, 0)) || (!first && is(64, 64, 1, 16, 32, 1)))) return false;
        if (i > 0 && (L.H != a.L[i - 1].OH || L.W != a.L[i - 1].OW || L.Cin != a.L[i - 1].Cout)) return false;
        if (L.g_cst < 0 || (L.g_cst & 3)) return false;
        if (first && L.zp_in != -128) return false;   // (zero-filled taps + border biases assume it; the packer checks the same)
        // every clamp whose result is stored as a byte is an int8 range (with the ADD the block's own term is not clamped: pw_lo / pw_hi unused)
        if (L.dw_lo < -128 || L.dw_hi > 127 || L.dw_lo > L.dw_hi || L.pw_lo > L.pw_hi) return false;
        if (!L.has_add && (L.pw_lo < -128 || L.pw_hi > 127)) return false;
        if (L.has_add && (L.add_lo < -128 || L.add_hi > 127 || L.add_lo > L.add_hi || L.add_e < 1 || L.add_e > 22 || L.add_m < 0 ||
                          L.res_m < 0 || L.res_k < 3 || L.res_k > 19))
            return false;
        if (tail2_dw_part(L, first) / 16 > kTail2MaxDw16) return false;
        const int in_bytes = first ? 0 : G * L.H * L.W * (L.Cin + 16), out_bytes = G * L.OH * L.OW * (L.Cout + 16);
        const Span2 map{0, in_bytes > out_bytes ? in_bytes : out_bytes};
        L.x_off = first ? -1 : 0;
        L.y_off = 0;
        std::vector<Span2> used{map};
        if (prev_pw.e > prev_pw.b) used.push_back(prev_pw);
        L.zp_off = 0;
        if (!first) {
            L.zp_off = first_fit2(used, L.Cin + 16, CAP);
            if (L.zp_off < 0) return false;
            used.push_back({L.zp_off, L.zp_off + L.Cin + 16});
        }
        L.dw_off = first && attempt ? (CAP - tail2_dw_part(L, first)) & ~15 : first_fit2(used, tail2_dw_part(L, first), CAP);
        if (L.dw_off < 0) return false;
        const Span2 dw{L.dw_off, L.dw_off + tail2_dw_part(L, first)};
        used = {map, dw};
        if (!first) used.push_back({L.zp_off, L.zp_off + L.Cin + 16});
        L.pw_off = first_fit2(used, tail2_pw_part(L), CAP);
        if (L.pw_off < 0) return false;
        prev_pw = {L.pw_off, L.pw_off + tail2_pw_part(L)};
        grow(map.e); grow(dw.e); grow(prev_pw.e);
    }
    {
        const Tail2Layer& lastL = a.L[n_layers - 1];
        const Span2 dw0{a.L[0].dw_off, a.L[0].dw_off + tail2_dw_part(a.L[0], true)};
        const Span2 map_lastL{0, G * lastL.OH * lastL.OW * (lastL.Cout + 16)};
        if (!overlap2(dw0.b, dw0.e, map_lastL.b, map_lastL.e) && !overlap2(dw0.b, dw0.e, prev_pw.b, prev_pw.e)) break;
        if (attempt) return false;
    }
    }  // attempt
    const Tail2Layer& last = a.L[n_layers - 1];
    // the first block's depthwise part of the NEXT group is written behind the last block's pointwise phase and lies there through the head
    const Span2 dw0{a.L[0].dw_off, a.L[0].dw_off + tail2_dw_part(a.L[0], true)};
    const Span2 map_last{0, G * last.OH * last.OW * (last.Cout + 16)};
    if (overlap2(dw0.b, dw0.e, map_last.b, map_last.e) || overlap2(dw0.b, dw0.e, prev_pw.b, prev_pw.e)) return false;
    if (mid) {
        a.P = last.OH * last.OW;
        a.C = last.Cout;
        a.fcw_off = -1;
        a.lds_bytes = lds_need;
        return a.C % 16 == 0;
    }
    const int32_t* h = desc + LW * n_layers;
    a.mean_zp_in = h[0]; a.mean_mult = h[1]; a.mean_shift = h[2]; a.mean_zp_out = h[3];
    a.fc_zp_out = h[4]; a.fc_lo = h[5]; a.fc_hi = h[6]; a.g_fcw = h[7]; a.g_fcb = h[8]; a.g_fcm = h[9]; a.g_fcs = h[10]; a.g_hlut = h[11];
    a.head_zp_fc = h[12]; a.head_zp_out = h[13]; a.P = h[14]; a.C = h[15];
    if (a.g_fcw < 0 || a.g_fcb < 0 || a.g_hlut < -1 || (a.g_fcw & 3) || (a.g_fcb & 3)) return false;
    if (a.fc_lo < -128 || a.fc_hi > 127 || a.fc_lo > a.fc_hi) return false;
    if (a.P != last.OH * last.OW || a.C != last.Cout || a.C % 64 || a.C > 256 || a.NC < 1 || (a.NC + 15) / 16 > kTail2Waves) return false;
    std::vector<Span2> used{map_last, dw0};
    a.mean_off = first_fit2(used, kTailG * last.Cout, CAP);
    if (a.mean_off < 0) return false;
    used.push_back({a.mean_off, a.mean_off + kTailG * last.Cout});
    grow(a.mean_off + kTailG * last.Cout);
    a.fcw_off = -1;   // (the classifier's fragments come straight from memory)
    a.lds_bytes = lds_need;
    return true;
}

namespace {
int tail2_cst_bytes(const Tail2Layer& L, bool first) { return tail2_dw_part(L, first) + tail2_pw_part(L); }
}  // namespace

#ifdef BN_TAIL_STAMPS
extern "C" __attribute__((visibility("default"))) int bn_debug_tail2_stamps(long long* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tail2_stamps), &d_buf, sizeof d_buf) == hipSuccess ? 0 : -1;
}
#endif

long tail2_const_words(const Tail2Args& a, bool mid) {
    long need = 0;
    auto upto = [&](long off, long words) { need = off + words > need ? off + words : need; };
    for (int i = 0; i < a.n_layers; ++i) upto(a.L[i].g_cst, tail2_cst_bytes(a.L[i], i == 0) / 4);
    if (mid) return need;
    const long nct = (a.NC + 15) / 16;
    upto(a.g_fcw, nct * (a.C / 64) * 256);
    upto(a.g_fcb, nct * 48);
    if (a.g_hlut >= 0) upto(a.g_hlut, 64);
    return need;
}

bool launch_i8_tail2(Tail2Args a, hipStream_t s) {
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(i8_tail2_kernel), 160 * 1024)) return false;
    const int ngroups = (a.B + kTailG - 1) / kTailG;
    const int grid = ngroups < 256 ? ngroups : 256;  // one workgroup per CU (its LDS), each walks over its share of the chunk groups
    hipLaunchKernelGGL(i8_tail2_kernel, dim3(grid), dim3(kTail2Threads), (size_t)a.lds_bytes, s, a);
    return true;
}

bool launch_i8_mid2(Tail2Args a, hipStream_t s) {
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(i8_mid2_kernel), 160 * 1024)) return false;
    const int ngroups = (a.B + kMidG - 1) / kMidG;
    const int grid = ngroups < 256 ? ngroups : 256;
    hipLaunchKernelGGL(i8_mid2_kernel, dim3(grid), dim3(kTail2Threads), (size_t)a.lds_bytes, s, a);
    return true;
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_i8_tail2() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&i8_tail2_kernel));
}

}  // namespace bn
