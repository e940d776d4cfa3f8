// bn_f32_pw.hip — plain float32 1x1 convolutions of the wide layers (Cin > 128) as a persistent kernel in three roles
//
//   y[p][n] = act( sum_k (x[p][k] * gate[chunk(p)][k]) * W[k][n] + b[n] [+ res[p][n]] )
//
// (reference: the expand / project convolutions of the inverted-residual blocks and the embedding convolution, models/dscnn.py:99-140,
// 230-258; squeeze-excite multiply of models/blocks.py:75-100 applied on load.)
//
// The tile kernel these layers ran through (f32_dwpw_kernel<RG, CT, false>, bn_f32_fused.hip) does load -> multiply -> store per workgroup and
// relies on the other workgroups of the CU to fill the gaps.  Measured on configs[4]'s 384 -> 192 projection (tools/config5_bench.py, ablations
// of one launch): matrix phase alone 0.145 ms, memory phases alone 0.085 ms, skeleton 0.042 ms — and the full kernel 0.26 ms, their SUM: three
// resident workgroups per CU do not overlap their phases (staggered starts and raised priority for the memory phases changed nothing).
// Here the overlap is by construction.  One 512-thread workgroup per CU walks over 64-position tiles:
//
//   * waves 0-3 (MATRIX waves, one per SIMD) do nothing but matrix instructions: 64 rows x CT column tiles each (or 2 x 2 waves of 32 rows x
//     48 columns for 96 columns), A fragments from LDS one k-step ahead, B fragments (weights in fragment order, L2) FOUR k-steps ahead in a
//     rotating register set that runs on across slices and tiles (the weight stream is cyclic), accumulators dumped into a separate LDS
//     tile behind a tile's last slice;
//   * waves 4-5 (LOADERS) keep the next TWO 128-channel slices of the activations in flight while the current one is multiplied: range-checked
//     buffer loads in straight-line steps (what does not apply to a step is requested behind the buffer's end), the gate requested at the
//     start of the step and applied on the way into the other half of a double-buffered LDS tile;
//   * waves 6-7 (EPILOGUE) request a tile's residual at its first step and, behind its last barrier, add bias and residual, clamp and store
//     whole rows while the matrix waves already multiply the next tile;
//   * one LDS-only barrier per slice is the only synchronisation; waves 4-7 run at raised priority (beside a wave that streams matrix
//     instructions the other wave of the SIMD otherwise loses every issue arbitration).
//
// Why three roles, which barrier, why no branch in a loader step: DESIGN.md §4 (kernel table) lists what each cost when it was missing.
// Same arithmetic in the same order as the tile kernel (v_mfma_f32_16x16x4_f32 chains over k in ascending order, bias, residual,
// activation): results are bit-identical to it (tests/test_gpu_sweeps.py: test_f32_pw_ws_kernel_matches_the_tile_kernel).
#include <type_traits>

#include "bn_kernels.h"

namespace bn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The workgroup's barrier orders LDS traffic only (tiles, accumulators): global loads and stores stay in flight across it.  __syncthreads()
// is a workgroup-scope fence as well: for the waves that store it put `s_waitcnt vmcnt(0)` in front of every barrier — the epilogue waves
// arrived late by a store round trip at the first barrier behind each tile and the matrix waves waited for them (5.5 us per tile).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kKC = 128;            // contraction channels per slice
constexpr int kS4 = kKC / 4 + 1;    // row stride of an activation buffer in float4 (one of padding: conflict-free fragment reads)
constexpr int kABuf = 64 * kS4;     // one activation buffer, in float4

struct PwArgs {
    const float* x;      // [P][K]
    const float* w;      // fragment order [K/16][N/16][64] float4 (pack_f32_fragments)
    const float* b;      // [N]
    const float* gate;   // [chunks][K] or null
    const float* res;    // [P][N] or null
    float* y;            // [P][N]
    int P, K, N, hw_shift, act, n_tiles, n_chunks;
};

typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t rs, int voff, int soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, soff, 0);
}

// NS = columns per workgroup (column slice), CT = 16-column tiles per matrix wave: the four matrix waves are WN = NS / (16 CT) across the columns
// times WM = 4 / WN down the rows, RG = 4 / WM row groups of 16 positions each (NS = 192, 128: 4 x 1 waves, all 64 rows; NS = 96: 2 x 2).
template <int NS, int CT, bool GATE, bool RES>
__global__ __launch_bounds__(512) void f32_pw_ws_kernel(PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    constexpr int SO = NS + 4, Q4 = NS / 4, WN = NS / (16 * CT), WM = 4 / WN, RG = 4 / WM;
    static_assert(WN * CT * 16 == NS && WM * WN == 4 && RG * WM == 4, "four matrix waves cover 64 rows x NS columns");
    f32x4* abuf = reinterpret_cast<f32x4*>(lds_raw);  // [2][64][kS4]
    float* otile = lds_raw + 2 * kABuf * 4;            // [64][SO]: the accumulators of the tile that has just been multiplied
    float* bias_l = otile + 64 * SO;                   // [NS]: this column slice's bias
    const int K = a.K, N = a.N, P = a.P;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = (K + kKC - 1) / kKC;  // slices per tile (K % 64 == 0, K > 128: the launcher checks)
    const int n_base = blockIdx.y * NS;
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_tiles * S;  // steps of this workgroup: step g = slice g % S of its tile g / S
    if (G <= 0) return;
    for (int i = tid; i < NS; i += 512) bias_l[i] = a.b[n_base + i];  // (visible behind the first barrier)

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, P * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? a.gate : a.x), 0, GATE ? a.n_chunks * K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? a.res : a.x), 0, RES ? P * N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, P * N * 4, 0x00020000);
    // Byte offsets go into the VECTOR offset (the range check of a raw buffer access covers the vector offset, not the scalar one): rows
    // behind the last position read zeros / are not written; the columns a 64-channel last slice does not have land in tile columns nobody
    // reads.  kOff marks an access that must not happen in this step: behind every buffer (sizes < 2^31: the launcher checks).
    constexpr int kOff = (int)0x80000000;
    auto tile_p0 = [&](int ti) __attribute__((always_inline)) { return ((int)blockIdx.x + ti * (int)gridDim.x) * 64; };
    const int n_bar = G + (G & 1);  // barriers behind the first one: the loaders' steps come in pairs

    // The matrix waves need one issue slot per 32 cycles; the loader / epilogue wave that shares their SIMD is the younger one and loses every
    // arbitration at equal priority (measured: ~25 cycles per instruction in the epilogue).  Static priority for waves 4-7.
    if (wave >= 4) __builtin_amdgcn_s_setprio(2);
    if (wave >= 6) {
        // ------------------------------------------------------------------ waves 6-7: epilogue ------------------------------------
        // Per tile: request the residual at the tile's first step, sit out the tile's S barriers, then bias + residual + activation and
        // whole-row stores while the consumers multiply the next tile.  (Loads and stores of their own waves: the loaders' wait counters
        // never see a store.  In one role with the loads every step queued behind 12 stores per lane and the counter's 63 slots filled up:
        // issuing a step's 16 loads took 4 us.)
        const int et = tid - 384;
        constexpr int EP2 = 64 * Q4 / 128;
        f32x4 rp[RES ? EP2 : 1];
        const bool clamp = a.act != 0;                                           // act 1: max(v, 0); act 2: min(max(v, 0), 6)
        const float hi = a.act == 2 ? 6.0f : __builtin_inff();
        lds_barrier();
        for (int ti = 0; ti < my_tiles; ++ti) {
            const int base = (tile_p0(ti) * N + n_base) * 4;
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < EP2; ++i) {
                    const int item = et + 128 * i, p = item / Q4, c4 = item - p * Q4;
                    rp[i] = buf_load4(rs_r, base + (p * N + 4 * c4) * 4, 0);
                }
            }
            for (int s = 0; s < S; ++s) lds_barrier();
            // four rows at a time: their eight LDS reads in flight together; the activation's bounds are picked once (a per-element
            // `switch` on the run-time activation cost 260 scalar branches per tile)
            constexpr int GRP = 4;
            static_assert(EP2 % GRP == 0, "items per epilogue thread");
#pragma unroll
            for (int i0 = 0; i0 < EP2; i0 += GRP) {
                f32x4 t[GRP], b[GRP];
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int item = et + 128 * (i0 + i), p = item / Q4, c4 = item - p * Q4;
                    t[i] = *reinterpret_cast<const f32x4*>(otile + p * SO + 4 * c4);
                    b[i] = *reinterpret_cast<const f32x4*>(bias_l + 4 * c4);
                }
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int item = et + 128 * (i0 + i), p = item / Q4, c4 = item - p * Q4;
                    f32x4 o = t[i] + b[i];
                    if constexpr (RES) o += rp[i0 + i];
                    if (clamp) {  // (wave-uniform)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(o[e], 0.0f), hi);
                    }
                    buf_store4(rs_y, base + (p * N + 4 * c4) * 4, 0, o);
                }
            }
        }
        if (G & 1) lds_barrier();
        return;
    }

    if (wave >= 4) {
        // ------------------------------------------------------------------ waves 4-5: loaders -------------------------------------
        // Straight-line steps, the same instruction sequence every time (what does not apply — no slice g + 2 — is requested at kOff): the
        // compiler's wait counters stay exact.  With per-item or per-step branches it fell back to `s_waitcnt vmcnt(0)` at the top of
        // every step.
        const int lt = tid - 256;
        // item u of this thread: position p = lt / 32 + 4 u of the tile, channel quad cq = lt % 32 of the slice
        const int p0 = lt >> 5, cq = lt & 31;
        const int xoff = (p0 * K + 4 * cq) * 4;  // + u * 4 K * 4
        f32x4 v[2][16], gt[GATE ? 16 : 1];  // two slices in flight: slice g + 1 (requested a step ago) and g + 2; the gate of the slice committed at the end of the step
        auto fill_issue = [&](auto SET, bool on, int ti, int s) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            const int base = on ? (tile_p0(ti) * K + s * kKC) * 4 : kOff;
#pragma unroll
            for (int u = 0; u < 16; ++u) v[set][u] = buf_load4(rs_x, base + xoff + u * 4 * K * 4, 0);
        };
        auto gate_issue = [&](bool on, int ti, int s) __attribute__((always_inline)) {
            if constexpr (GATE) {
                const int P0 = tile_p0(ti), c0 = P0 >> a.hw_shift;
                const int base = on ? (c0 * K + s * kKC) * 4 : kOff;
#pragma unroll
                for (int u = 0; u < 16; ++u) gt[u] = buf_load4(rs_g, base + ((((P0 + p0 + 4 * u) >> a.hw_shift) - c0) * K + 4 * cq) * 4, 0);
            }
        };
        auto fill_commit = [&](auto SET, int g) __attribute__((always_inline)) {  // (behind the last step: zeros into a buffer nobody reads any more)
            constexpr int set = decltype(SET)::value;
            f32x4* dst = abuf + (g & 1) * kABuf + p0 * kS4 + cq;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if constexpr (GATE) dst[u * 4 * kS4] = v[set][u] * gt[u];
                else dst[u * 4 * kS4] = v[set][u];
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // (tile, slice) of step g + d, d = 1, 2 (steps run over the slices of the workgroup's tiles in order)
        auto step_of = [&](int ti, int s, int d, int& to, int& so) __attribute__((always_inline)) {
            so = s + d;
            to = ti;
            while (so >= S) {
                so -= S;
                ++to;
            }
        };
        fill_issue(S0{}, true, 0, 0);
        gate_issue(true, 0, 0);
        fill_commit(S0{}, 0);
        {
            int t1, s1;
            step_of(0, 0, 1, t1, s1);
            fill_issue(S1{}, 1 < G, t1, s1);
        }
        lds_barrier();
        int ti = 0, s = 0;
        auto step = [&](auto CUR, auto NXT, int g) __attribute__((always_inline)) {  // set CUR held slice g (free now), NXT holds slice g + 1
            int t1, s1, t2, s2;
            step_of(ti, s, 1, t1, s1);
            step_of(ti, s, 2, t2, s2);
            gate_issue(g + 1 < G, t1, s1);  // (first: the commit below then waits for these loads only, not for the slice behind them)
            fill_issue(CUR, g + 2 < G, t2, s2);
            fill_commit(NXT, g + 1);
            lds_barrier();
            ti = t1;
            s = s1;
        };
        for (int g = 0; g < n_bar; g += 2) {  // (an odd G: one idle step at the end; consumers and epilogue waves add the matching barrier)
            step(S0{}, S1{}, g);
            step(S1{}, S0{}, g + 1);
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers ---------------------------------------------
    const int lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wn = wave % WN, wm = wave / WN, row0 = wm * RG * 16;
    const int n_ct = N >> 4, ct0 = blockIdx.y * (NS / 16) + wn * CT, KS = K >> 4;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.w);
    f32x4 bq[4][CT];
    int jn = 0;  // next k-step of the (cyclic) weight stream to request
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int c = 0; c < CT; ++c) bq[u][c] = wp[((size_t)jn * n_ct + ct0 + c) * 64 + lane];
        jn = jn + 1 == KS ? 0 : jn + 1;
    }
    f32x4 acc[RG][CT];
    lds_barrier();
    int s = 0;
    for (int g = 0; g < G; ++g) {
        if (s == 0) {
#pragma unroll
            for (int gg = 0; gg < RG; ++gg)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[gg][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int k0 = s * kKC, ksteps = ((K - k0) < kKC ? (K - k0) : kKC) >> 4;  // 8, or 4 in a last slice of 64 channels
        const f32x4* tile = abuf + (g & 1) * kABuf;
        f32x4 af[RG], afn[RG];
#pragma unroll
        for (int gg = 0; gg < RG; ++gg) af[gg] = tile[(row0 + 16 * gg + r) * kS4 + q];
        for (int j = 0; j < ksteps; j += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int gg = 0; gg < RG; ++gg) afn[gg] = tile[(row0 + 16 * gg + r) * kS4 + ((4 * (j + u + 1) + q) & 31)  /* behind the last k-step: a harmless wrap, the value is dropped */];
                __builtin_amdgcn_sched_barrier(0);  // keep the four LDS reads AHEAD of the 48 matrix instructions (the scheduler sinks them to the block's end,
                                                    // where every k-step then waits out the LDS latency: 0.219 -> ? ms on configs[4]'s 384 -> 192 projection)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int gg = 0; gg < RG; ++gg)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            acc[gg][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[gg][e], bq[u][c][e], acc[gg][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CT; ++c) bq[u][c] = wp[((size_t)jn * n_ct + ct0 + c) * 64 + lane];
                jn = jn + 1 == KS ? 0 : jn + 1;
#pragma unroll
                for (int gg = 0; gg < RG; ++gg) af[gg] = afn[gg];
            }
        }
        if (s == S - 1) {
#pragma unroll
            for (int gg = 0; gg < RG; ++gg)
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) otile[(row0 + 16 * gg + 4 * q + reg) * SO + (wn * CT + c) * 16 + r] = acc[gg][c][reg];
        }
        lds_barrier();
        s = s + 1 == S ? 0 : s + 1;
    }
    if (G & 1) lds_barrier();  // (the producers' steps come in pairs)
}

template <int NS, int CT, bool GATE, bool RES>
bool launch_one(const PwArgs& a, int slices, hipStream_t s) {
    const size_t smem = (size_t)2 * kABuf * 16 + (size_t)64 * (NS + 4) * sizeof(float) + NS * sizeof(float);
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&f32_pw_ws_kernel<NS, CT, GATE, RES>), smem)) return false;
    int per_slice = 256 / slices;  // one persistent workgroup per CU
    if (per_slice < 1) per_slice = 1;
    if (per_slice > a.n_tiles) per_slice = a.n_tiles;
    hipLaunchKernelGGL((f32_pw_ws_kernel<NS, CT, GATE, RES>), dim3(per_slice, slices), dim3(512), smem, s, a);
    return true;
}
template <int NS, int CT>
bool launch_ct(const PwArgs& a, int slices, hipStream_t s) {
    if (a.gate) return a.res ? launch_one<NS, CT, true, true>(a, slices, s) : launch_one<NS, CT, true, false>(a, slices, s);
    return a.res ? launch_one<NS, CT, false, true>(a, slices, s) : launch_one<NS, CT, false, false>(a, slices, s);
}

}  // namespace

// Plain 1x1 convolution (no depthwise stage, stride 1) over P = B * H * W positions; false: not this kernel's shape, the caller uses the tile kernel.
bool launch_f32_pw_ws(const DwPwArgs& d, hipStream_t s) {
    if (d.has_dw || d.sh != 1 || d.sw != 1 || d.H != d.OH || d.W != d.OW) return false;
    const int K = d.Cin, N = d.Cout, HW = d.H * d.W;
    if (K % 64 || K <= kKC || (HW & (HW - 1))) return false;
    const long P = (long)d.B * HW;
    if (P <= 0 || P * (long)(K > N ? K : N) * 4 > 0x7fffffffL) return false;  // (buffer descriptors and 32-bit byte offsets)
    int hw_shift = 0;
    while ((1 << hw_shift) < HW) ++hw_shift;
    PwArgs a{d.x, d.pw_w, d.pw_b, d.gate, d.res, d.y, (int)P, K, N, hw_shift, d.pw_act, (int)((P + 63) / 64), d.B};
    if (N % 192 == 0) return launch_ct<192, 3>(a, N / 192, s);
    if (N % 128 == 0) return launch_ct<128, 2>(a, N / 128, s);  // (four column tiles per consumer wave do not fit 256 registers)
    if (N % 96 == 0) return launch_ct<96, 3>(a, N / 96, s);     // (2 x 2 matrix waves of 32 rows x 48 columns)
    return false;
}

// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_f32_pw() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&f32_pw_ws_kernel<192, 3, false, false>));
}

}  // namespace bn
