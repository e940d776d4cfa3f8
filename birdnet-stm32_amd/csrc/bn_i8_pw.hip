// bn_i8_pw.hip — dense INT8 1x1 convolutions (Cin = 192, 384, 512 or 768: the expand / project / embedding convolutions of the late
// stages of exported inverted-residual graphs, reference birdnet_stm32/models/blocks.py:49-133 behind the converter) on the int8 matrix
// cores, with the squeeze-excite MUL (blocks.py:27-46) applied while the projection loads its input.
//
// These are real GEMMs (configs[4], 1024 chunks: M = 131 072 positions, K = 384, N = 192) that ran through the generic 64-position tile
// kernel at 4 % of the matrix peak and 10 % of the HBM peak, plus a separate i8_scale launch for every gate.  By arithmetic they are
// HBM-bound (19 GOP against 100 MB per launch), so the form is the wave-level one of i8_pw_wave_kernel — no activation tile in LDS, no
// barrier in the data path — with what the wide layers need on top:
//
//   * the activations of a wave's 16 positions ARE the B operand of v_mfma_i32_16x16x64_i8 as they lie in memory: lane (position r,
//     k quarter q) holds the 16 channel bytes 64 s + 16 q .. of its position per k-step (one range-checked 16-byte buffer load each,
//     all k-steps of the NEXT group in flight while the current one is multiplied);
//   * the weights of the workgroup's slice of output channels live in LDS for the whole launch, in CONSUMPTION order (the staging
//     loop gathers the packer's fragments once, so the A operand of (k-step, tile) is one linear ds_read_b128 per lane, conflict-free).
//     Slices are at most 72 KB (two workgroups of eight waves per CU); the workgroups of all slices of one position range share an
//     XCD, so the activations a second slice reads come from that XCD's L2;
//   * rows of a tile are permuted (like i8_pw_wave_kernel) so that lane (r, q) ends with the slice's channels ns/4 * q .. + ns/4 of its
//     position: 16-byte stores, 16-byte residual loads;
//   * the squeeze-excite gate: with zero points -128 on both sides (a ReLU6 map times a LOGISTIC output: every exported graph) both
//     factors are non-negative, the MUL's requantisation needs no sign term and is the high dword of ONE 64-bit multiply-add with the
//     output zero point folded into the addend:  clamp(hi32((x+128)(g+128) M + C) >> (e-1)),  C = 2^30 + (2^(e-1) + zo 2^e) 2^31
//     (nested floors; derivation in bn_i8_strip.hip) — about 5 vector instructions per input byte instead of 12.  Other zero points
//     take the literal MultiplyByQuantizedMultiplier form.  Bit-identical to MUL -> CONV_2D as separate operators either way.
//
// Integer arithmetic is exact and order-free, so results are bit-identical to the tile kernel (tests/test_conversion.py per tensor
// against the INT8 oracle; tests/test_gpu_sweeps.py against the tile kernel through option i8_pw_lds = 0).
#include "bn_kernels.h"
#include "bn_requant.h"

namespace bn {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kPwLdsThreads = 1024;  // one workgroup of sixteen waves per CU: the slice's weights are staged once per CU, not twice
constexpr int kPwLdsWaves = kPwLdsThreads / 64;
constexpr size_t kPwLdsBudget = 154 * 1024;  // weights + constants of a slice (+ 2 KB of ADD tables) inside a CU's 160 KB

struct PwLdsGeom {
    long n_pos;     // B * OH * OW
    int ns;         // output channels per slice (multiple of 16)
    int n_slices;
    int walkers;    // position walkers = gridDim.x / n_slices
    int tab;        // the ADD as one lookup in the 64 KB table (DwPw8Args::add_tab), staged behind the constants
};

template <bool ADD, bool GATE, int KS, int NCT, bool HI>  // K = 64 KS input channels, slices of 16 NCT output channels; HI: sign-free requantisation
__global__ __launch_bounds__(kPwLdsThreads) void i8_pw_lds_kernel(DwPw8Args a, PwLdsGeom g) {
    extern __shared__ __attribute__((aligned(16))) int lds_raw[];
    __shared__ int add_lut[2][256];
    constexpr int K = 64 * KS, ns = 16 * NCT, cpl = 4 * NCT;
    const int tid = threadIdx.x;
    const int N = a.Cout;
    // workgroup -> (XCD, slice, walker): consecutive workgroup ids go round the eight XCDs, so the slices of one walker share an XCD
    const int wg = blockIdx.x;
    const int xcd = wg & 7, j = wg >> 3;
    const int slice = j % g.n_slices;
    const int walker = (j / g.n_slices) * 8 + xcd;
    const int n0 = slice * ns;
    v4i* wl = reinterpret_cast<v4i*>(lds_raw);       // [KS][NCT][64 lanes]
    v4i* cst = wl + (size_t)KS * NCT * 64;           // [4 lane quarters][NCT][bias, multiplier, shift | addend low, addend high]
    const unsigned char* tabl = reinterpret_cast<const unsigned char*>(lds_raw) + (size_t)ns * (K + 20);  // (g.tab) the ADD table behind them
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.x), 0, (int)(g.n_pos * K), 0x00020000);
    const long n_groups = g.n_pos / 16, stride = (long)g.walkers * kPwLdsWaves;
    const int P = a.OH * a.OW;
    auto fetch = [&](long grp, v4i (&dst)[KS]) {  // (a group past the end reads behind the buffer: zeros, never used)
        const long base = grp < n_groups ? (grp * 16 + r) * (long)K + 16 * q : g.n_pos * (long)K;
#pragma unroll
        for (int s = 0; s < KS; ++s) dst[s] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)base + 64 * s, 0, 0));
    };
    long grp = (long)walker * kPwLdsWaves + wave;
    v4i bfr[KS];
    fetch(grp, bfr);  // the first group's bytes travel while the weights are staged
    // sign-free requantisation (ReLU6 outputs: the clamp starts at the zero point, a negative value ends at the lower bound either way):
    // clamp(hi32(acc m + C) >> (e - 1)),  C = 2^30 + (2^(e-1) + zp 2^e) 2^31 — three instructions per output instead of eight (HI: the launcher
    // checked that every shift lies in [-20, -1] and that the clamp starts at the zero point)
    {
        const v4i* wp = reinterpret_cast<const v4i*>(a.pw_w);  // packer's fragment order [K/64][N/16][64 lanes]
        const int n_ct_all = N >> 4;
        constexpr int TOT = KS * NCT * 64, NLD = (TOT + kPwLdsThreads - 1) / kPwLdsThreads;
        v4i tmp[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {  // every gather in flight before the first LDS write (one L2 round trip, not NLD)
            const int i = tid + k * kPwLdsThreads;
            const int ln = i & 63, ct = (i >> 6) % NCT, s = (i >> 6) / NCT;
            const int rr = ln & 15, qq = ln >> 4;
            const int ch = n0 + cpl * (rr >> 2) + (rr & 3) + 4 * ct;  // the channel lane rr's A row stands for in tile ct
            tmp[k] = i < TOT ? wp[((size_t)s * n_ct_all + (ch >> 4)) * 64 + qq * 16 + (ch & 15)] : (v4i){0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * kPwLdsThreads;
            if (i < TOT) wl[i] = tmp[k];
        }
        for (int i = tid; i < 4 * NCT; i += kPwLdsThreads) {
            const int qq = i / NCT, ct = i % NCT;
            const int ch = n0 + cpl * qq + 4 * ct;
            const v4i bb = *reinterpret_cast<const v4i*>(a.pw_b + ch), mm = *reinterpret_cast<const v4i*>(a.pw_mult + ch);
            v4i ss = *reinterpret_cast<const v4i*>(a.pw_shift + ch);
            v4i clo = (v4i){0, 0, 0, 0}, chi = (v4i){0, 0, 0, 0};
            if constexpr (HI) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ex = -ss[e];  // 1 .. 20 (checked at load)
                    const long long C = (1ll << 30) + (((1ll << (ex - 1)) + (long long)a.pw_zp_out * (1ll << ex)) << 31);
                    clo[e] = (int)(unsigned)(C & 0xffffffffll);
                    chi[e] = (int)(C >> 32);
                    ss[e] = ex - 1;
                }
            }
            cst[5 * i + 0] = bb;
            cst[5 * i + 1] = mm;
            cst[5 * i + 2] = ss;
            cst[5 * i + 3] = clo;
            cst[5 * i + 4] = chi;
        }
        if (ADD && g.tab) {
            const v4i* tsrc = reinterpret_cast<const v4i*>(a.add_tab);
            v4i* tdst = reinterpret_cast<v4i*>(lds_raw) + (size_t)ns * (K + 20) / 16;
            v4i tv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) tv[k] = tsrc[tid + kPwLdsThreads * k];
#pragma unroll
            for (int k = 0; k < 4; ++k) tdst[tid + kPwLdsThreads * k] = tv[k];
        }
        if (ADD && !g.tab && tid < 256) {
            const int v = (int)(int8_t)tid;
            add_lut[0][tid] = mbqm((v - a.add.z1) * (1 << 20), a.add.m1, a.add.s1);
            add_lut[1][tid] = mbqm((v - a.pw_zp_out) * (1 << 20), a.add.m2, a.add.s2);
        }
    }
    __syncthreads();
    // the gate's constants: C = 2^30 + (2^(e-1) + zo 2^e) 2^31, shift e - 1 (uniform: the MUL is quantised per tensor)
    const int ge = GATE ? -a.g_shift : 1, gsh = ge - 1;
    const long long gC = (1ll << 30) + (((1ll << (ge - 1)) + (long long)a.g_zo * (1ll << ge)) << 31);
    // The B operands of a group are requested one group ahead IN PLACE: k-step s of the next group is loaded into bfr[s] as soon as the
    // matrix instructions of k-step s of this group are issued (no second buffer: 24 / 48 registers instead of 48 / 96).
    // U matrix instructions per step, their A operands requested a step ahead; the A operands walk linearly through the LDS copy
    // (the gated twelve-tile slices have no registers left for four: two)
    constexpr int U = (GATE && NCT == 12) ? 2 : 4;
    constexpr int STEPS = KS * NCT / U;
    for (; grp < n_groups; grp += stride) {
        const long pos = grp * 16 + r;
        const long nxt = grp + stride;
        const int nbase = nxt < n_groups ? (int)((nxt * 16 + r) * (long)K + 16 * q) : (int)(g.n_pos * (long)K);
        int8_t* yrow = a.y + pos * N + n0 + cpl * q;
        const int8_t* rrow = ADD ? a.res + pos * N + n0 + cpl * q : nullptr;
        const int8_t* gb = GATE ? a.gate + (grp * 16 / P) * K + 16 * q : nullptr;  // (a group of 16 positions lies inside one chunk: P % 16 == 0)
        v4i acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = cst[5 * (q * NCT + ct) + 0];
        v4i af[U], an[U];
#pragma unroll
        for (int u = 0; u < U; ++u) af[u] = wl[u * 64 + lane];
        v4i b = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int s = (U * i) / NCT, ct0 = (U * i) % NCT;
            if (ct0 == 0) {
                b = bfr[s];
                if constexpr (GATE) {
                    // squeeze-excite MUL on the way in: (x + 128)(g + 128) >= 0, so the requantisation is one multiply-add's high dword
                    const v4i gv = *reinterpret_cast<const v4i*>(gb + 64 * s);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned xu = (unsigned)b[d] ^ 0x80808080u, gu = (unsigned)gv[d] ^ 0x80808080u;  // byte + 128 = byte - zero point
                        int gq[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int p = (int)((xu >> (8 * e)) & 0xff) * (int)((gu >> (8 * e)) & 0xff);
                            gq[e] = med3i((int)(((long long)p * a.g_mult + gC) >> 32) >> gsh, a.g_amin, a.g_amax);
                        }
                        b[d] = pack4(gq);
                        __builtin_amdgcn_sched_barrier(0);  // (one dword at a time: the sixteen requantisations in parallel cost 40 more registers)
                    }
                }
            }
            if (i + 1 < STEPS) {
#pragma unroll
                for (int u = 0; u < U; ++u) an[u] = wl[(U * (i + 1) + u) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc[ct0 + u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u], b, acc[ct0 + u], 0, 0, 0);
            if (ct0 + U == NCT)  // k-step s is through: its registers take the next group's bytes
                bfr[s] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_x, nbase + 64 * s, 0, 0));
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist every LDS read of the tile to the top: 288 registers)
#pragma unroll
            for (int u = 0; u < U; ++u) af[u] = an[u];
        }
#pragma unroll
        for (int ct0 = 0; ct0 < NCT; ct0 += 4) {
            v4i rv4 = (v4i){0, 0, 0, 0};
            if (ADD) rv4 = *reinterpret_cast<const v4i*>(rrow + 4 * ct0);
            v4i outw;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cidx = q * NCT + ct0 + u;
                const v4i m = cst[5 * cidx + 1], sh = cst[5 * cidx + 2];
                v4i clo = (v4i){0, 0, 0, 0}, chi = (v4i){0, 0, 0, 0};
                if constexpr (HI) {
                    clo = cst[5 * cidx + 3];
                    chi = cst[5 * cidx + 4];
                }
                int qv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (HI) {
                        const long long C = (long long)(((unsigned long long)(unsigned)chi[e] << 32) | (unsigned)clo[e]);
                        qv[e] = med3i((int)(((long long)acc[ct0 + u][e] * m[e] + C) >> 32) >> sh[e], a.pw_amin, a.pw_amax);
                    } else if (ADD && g.tab) {  // (uniform) own value + 128 = the table's column; the ADD is one byte read
                        const int own = med3i(mbqm_right(acc[ct0 + u][e], m[e], sh[e]) + (a.pw_zp_out + 128), a.pw_amin + 128, a.pw_amax + 128);
                        qv[e] = tabl[__builtin_amdgcn_perm((unsigned)rv4[u], (unsigned)own, 0x0c0c0400u + (e << 8))];
                    } else {  // every multiplier >= 0 and every shift < 0 (checked at load): the branch-free signed form
                        qv[e] = med3i(mbqm_right(acc[ct0 + u][e], m[e], sh[e]) + a.pw_zp_out, a.pw_amin, a.pw_amax);
                    }
                    if (ADD && !g.tab) {
                        const int sa = add_lut[0][(rv4[u] >> (8 * e)) & 0xff];
                        const int sb = add_lut[1][qv[e] & 0xff];
                        qv[e] = med3i(mbqm(sa + sb, a.add.mo, a.add.so) + a.add.zo, a.add.amin, a.add.amax);  // (uniform parameters: the form is chosen once)
                    }
                }
                outw[u] = pack4(qv);
            }
            *reinterpret_cast<v4i*>(yrow + 4 * ct0) = outw;  // (cpl = 4 NCT is a multiple of 16: NCT % 4 == 0)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// slice width: 192 output channels where Cout and the register budget allow, else 128, else 64
int pick_slice(int K, int N) {
    const int widest = K <= 384 ? 192 : K <= 512 ? 128 : 64;  // registers: 4 K / 64 for the activations + ns / 4 accumulators per lane, 128 in all
    for (int ns : {192, 128, 64})
        if (ns <= widest && N % ns == 0 && (size_t)ns * (K + 20) <= kPwLdsBudget) return ns;
    return 0;
}

}  // namespace

bool i8_pw_lds_supported(const DwPw8Args& a) {
    if (!g_opt.i8_pw_lds || !g_opt.i8_strip || !(a.rq_right & 4)) return false;  // (pointwise multipliers >= 0, right shifts only: the kernel's two requantisation forms)
    const long n_pos = (long)a.B * a.OH * a.OW;
    if (a.has_dw || a.transposed || a.lut || a.qx || a.sh != 1 || a.sw != 1 || a.H != a.OH || a.W != a.OW) return false;
    if (a.Cin != 192 && a.Cin != 384 && a.Cin != 512 && a.Cin != 768) return false;
    if (n_pos % 16 || n_pos * a.Cin >= 0x7fff0000L || n_pos * a.Cout >= 0x7fff0000L) return false;
    if (a.gate) {
        if (((long)a.OH * a.OW) % 16) return false;  // a group of 16 positions must lie inside one chunk
        // the one-multiply-add form of the MUL: both zero points -128 (factors >= 0), right shift 1..20 (the addend stays inside 64 bits)
        if (a.g_zx != -128 || a.g_zg != -128 || a.g_mult < 0 || a.g_shift > -1 || a.g_shift < -20) return false;
        const int ns = pick_slice(a.Cin, a.Cout);
        if (!ns || a.Cout / ns > 2) return false;  // every slice repeats the MUL of all input bytes: beyond two slices a separate i8_scale pass is cheaper
    }
    return pick_slice(a.Cin, a.Cout) > 0;
}

void launch_i8_pw_lds(const DwPw8Args& a, hipStream_t s) {
    PwLdsGeom g;
    g.n_pos = (long)a.B * a.OH * a.OW;
    g.ns = pick_slice(a.Cin, a.Cout);
    g.n_slices = a.Cout / g.ns;
    // persistent: one workgroup per CU; walkers in multiples of the eight XCDs, no more than there are groups of 16 positions per wave
    const long groups = g.n_pos / 16;
    long walkers = 256 / g.n_slices / 8 * 8;
    if (walkers < 8) walkers = 8;
    const long need = ((groups + kPwLdsWaves - 1) / kPwLdsWaves + 7) / 8 * 8;
    if (walkers > need) walkers = need;
    g.walkers = (int)walkers;
    const unsigned blocks = (unsigned)(walkers * g.n_slices);
    g.tab = a.add.enabled && a.add_tab && g_opt.i8_add_tab && (a.rq_right & 4) && (size_t)g.ns * (a.Cin + 20) + 65536 <= kPwLdsBudget ? 1 : 0;
    size_t smem = (size_t)g.ns * (a.Cin + 20) + (g.tab ? 65536 : 0);
    // the sign-free form is for operators behind a ReLU only: with a residual ADD the block's own value is linear (any sign) and the kernel's
    // epilogue applies the ADD in its other branch — ADD + HI would drop the residual; option i8_pw_forms = 0 forces the general form
    const bool hi = g_opt.i8_pw_forms && (a.rq_right & 2) && a.pw_amin >= a.pw_zp_out && !a.add.enabled;
    // a runtime that refuses the LDS limit with the ADD table gets the two-table form (64 KB less); a second refusal surfaces as the launch error
#define BN_PWL1(ADDV, GATEV, KSV, NCTV, HIV)                                                                                    \
    do {                                                                                                                        \
        const void* fn = (const void*)i8_pw_lds_kernel<ADDV, GATEV, KSV, NCTV, HIV>;                                            \
        if (smem > 64 * 1024 && !ensure_dynamic_lds(fn, smem) && g.tab) {                                                       \
            g.tab = 0;                                                                                                          \
            smem -= 65536;                                                                                                      \
            if (smem > 64 * 1024) (void)ensure_dynamic_lds(fn, smem);                                                           \
        }                                                                                                                       \
        hipLaunchKernelGGL((i8_pw_lds_kernel<ADDV, GATEV, KSV, NCTV, HIV>), dim3(blocks), dim3(kPwLdsThreads), smem, s, a, g); \
    } while (0)
#define BN_PWL(ADDV, GATEV, KSV, NCTV)                   \
    do {                                                 \
        if (hi) BN_PWL1(ADDV, GATEV, KSV, NCTV, true);   \
        else BN_PWL1(ADDV, GATEV, KSV, NCTV, false);     \
    } while (0)
#define BN_PWL_AG(KSV, NCTV)                                \
    do {                                                    \
        if (a.add.enabled) {                                \
            if (a.gate) BN_PWL(true, true, KSV, NCTV);      \
            else BN_PWL(true, false, KSV, NCTV);            \
        } else {                                            \
            if (a.gate) BN_PWL(false, true, KSV, NCTV);     \
            else BN_PWL(false, false, KSV, NCTV);           \
        }                                                   \
    } while (0)
#define BN_PWL_N(KSV)                                       \
    do {                                                    \
        if (g.ns == 192) BN_PWL_AG(KSV, 12);                \
        else if (g.ns == 128) BN_PWL_AG(KSV, 8);            \
        else BN_PWL_AG(KSV, 4);                             \
    } while (0)
    if (a.Cin == 192) BN_PWL_N(3);
    else if (a.Cin == 384) BN_PWL_N(6);
    else if (a.Cin == 512) {
        if (g.ns == 128) BN_PWL_AG(8, 8);
        else BN_PWL_AG(8, 4);
    }
    else BN_PWL_AG(12, 4);
#undef BN_PWL_N
#undef BN_PWL_AG
#undef BN_PWL
#undef BN_PWL1
}

// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_i8_pw() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&i8_pw_lds_kernel<false, false, 3, 4, false>));
}

}  // namespace bn
