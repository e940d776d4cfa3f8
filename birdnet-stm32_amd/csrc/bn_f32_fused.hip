// bn_f32_fused.hip — float32 depthwise-separable block as ONE kernel on gfx950:
//
//     [depthwise 3x3 (+folded BN bias, ReLU6)]  ->  LDS tile  ->  pointwise 1x1 on the matrix cores
//                                                    (+folded BN bias, +residual, ReLU6)
//
// Reference semantics: ds_conv_block of birdnet_stm32/models/dscnn.py:28-84 (DW -> BN -> ReLU6 -> PW -> BN
// [-> Add] -> ReLU6), and any plain 1x1 convolution (inverted-residual expand/project, embedding conv:
// birdnet_stm32/models/blocks.py:88-131, dscnn.py:248-253) with the depthwise stage switched off.
//
// One 256-thread workgroup owns 64 output positions (a TH x TW spatial tile of NB chunks, TH*TW*NB = 64)
// and all Cout channels:
//   phase 1 (vector ALU): the depthwise outputs of the tile, [64][Cin] float32, are computed straight
//           from the NHWC input (float4 over channels) into LDS — they never touch HBM;
//   phase 2 (matrix cores): v_mfma_f32_16x16x4_f32 (exact f32 FMA chain).  A fragments come from the LDS tile
//           as one ds_read_b128 per 16 contraction channels; B fragments come from the weight matrix that
//           the packer stored in fragment order, so each wave-instruction reads 1 KiB contiguous from L2.
// The contraction index inside each group of 16 channels is permuted identically for A and B
// (k = 16 j + 4 (lane >> 4) + e for the e-th MFMA), which only re-orders the float32 summation.
#include "bn_kernels.h"

namespace bn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}

constexpr int kKC = 128;  // contraction channels staged in LDS at a time (256: one barrier pair less per 256 channels, but 64 KB tiles — measured slower:
                          // configs[4] 7.04 -> 6.86 ms, shipped float32 net 0.945 -> 0.923 ms per 1024 chunks; the epilogue in two column passes
                          // for still more workgroups per CU changed nothing)

// Per-position bookkeeping, computed once per workgroup (integer divisions are costly on the vector ALU).
struct PosInfo {
    int in_base;   // element offset of tap (0,0) of this position in x (may point outside the image: see mask)
    int out_base;  // element offset of channel 0 of this position in y / res, or -1 if the chunk is out of range
    int mask;      // bit t set = tap t (t = 3 i + j) lies inside the image
    int pad;
};

template <int RG, int CT, bool HAS_DW>
__global__ __launch_bounds__(256) void f32_dwpw_kernel(DwPwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    __shared__ PosInfo pos[64];
    f32x4* lds4 = reinterpret_cast<f32x4*>(lds_raw);  // activation tile [64][kc/4 + 1] float4, later the output tile
    const int K = a.Cin, N = a.Cout;
    const int tid = threadIdx.x;

    // ---- which 64 positions -------------------------------------------------------------------------
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
        PosInfo pi;
        pi.pad = 0;
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
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    constexpr int WM = 4 / RG;  // waves along the 64 rows (RG row groups of 16 per wave), RG waves along the columns
    const int wm = wave % WM, wn = wave / WM;
    const int row0 = wm * RG * 16;
    const int ct0 = blockIdx.y * (RG * CT) + wn * CT;  // first 16-column tile of this wave (blockIdx.y = column slice)
    const int n_ct = N >> 4;                           // column tiles in the packed weights

    f32x4 acc[RG][CT];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.pw_w);  // [K/16][N/16][64 lanes] float4

    // The residual rows this thread will add in the epilogue are requested now, so that their HBM/L2 latency hides behind
    // the depthwise and matrix phases (narrow slices only: the prefetch costs NS/16 float4 registers).
    constexpr int NS = RG * CT * 16;  // columns of this workgroup's slice
    constexpr int Q4 = NS / 4;        // float4 per output row
    constexpr int EP = (64 * Q4) / 256;  // epilogue items per thread
    constexpr bool PREFETCH_RES = NS <= 32;  // at 64 columns the extra registers cost more occupancy than the latency they hide (measured)
    float4 res_pf[PREFETCH_RES ? EP : 1];
    if (PREFETCH_RES && a.res) {
#pragma unroll
        for (int i = 0; i < EP; ++i) {
            const int item = tid + 256 * i;
            const int p = item / Q4, c4 = item - p * Q4;
            const int ob = pos[p].out_base;
            res_pf[i] = ob >= 0 ? *reinterpret_cast<const float4*>(a.res + (long)ob + blockIdx.y * NS + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    for (int k0 = 0; k0 < K; k0 += kKC) {
        const int kc = (K - k0) < kKC ? (K - k0) : kKC;
        const int kq = kc >> 2;   // float4 groups per position
        const int kcp = (kc + 15) & ~15;  // whole matrix-core k-steps; columns kc..kcp-1 are zero and meet zero weight rows (1x1 convs with Cin % 16 != 0)
        const int S4 = (kcp >> 2) + 1;    // row stride of the tile in float4 units (one float4 of padding)
        if (k0) __syncthreads();  // the previous slice's fragments have been read

        // ---- phase 1: fill the LDS tile [64][kc] ---------------------------------------------------------
        // 256 % kq == 0 (kc in {16,32,64,128,256}): a thread keeps one channel quad and walks positions, so the
        // nine depthwise weights stay in registers; otherwise items are dealt round-robin.
        const bool fixed_cq = (256 % kq) == 0;
        const int items = 64 * kq;
        const int cq_fixed = tid % kq;
        float4 wgt[9];
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (HAS_DW && fixed_cq) {
            bias = *reinterpret_cast<const float4*>(a.dw_b + k0 + 4 * cq_fixed);
#pragma unroll
            for (int t = 0; t < 9; ++t) wgt[t] = *reinterpret_cast<const float4*>(a.dw_w + t * K + k0 + 4 * cq_fixed);
        }
        if (!HAS_DW) {
            // plain 1x1 convolution: the tile is a copy (times the squeeze-excite gate).  All of a thread's loads of the slice are requested
            // before the first is used — item by item every load paid the full memory latency (eight round trips per 128-channel slice)
            constexpr int NPF = (RG * CT >= 8) ? 4 : 1;  // (narrow slices are HBM-bound: the extra registers cost them occupancy — 48 columns 0.43 -> 0.47 ms)
            for (int item0 = tid; item0 < items; item0 += 256 * NPF) {
                float4 v[NPF];
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const int item = item0 + 256 * u;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (item < items) {
                        const int p = item / kq, cq = fixed_cq ? cq_fixed : item - p * kq;
                        if (pos[p].mask) v[u] = *reinterpret_cast<const float4*>(a.x + (long)pos[p].in_base + k0 + 4 * cq);
                    }
                }
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const int item = item0 + 256 * u;
                    if (item >= items) break;
                    const int p = item / kq, cq = fixed_cq ? cq_fixed : item - p * kq;
                    float4 accv = v[u];
                    if (a.gate && pos[p].mask) {
                        const int chunk = pos[p].in_base / (a.H * a.W * K);
                        const float4 g = *reinterpret_cast<const float4*>(a.gate + (size_t)chunk * K + k0 + 4 * cq);
                        accv.x *= g.x;
                        accv.y *= g.y;
                        accv.z *= g.z;
                        accv.w *= g.w;
                    }
                    lds4[p * S4 + cq] = (f32x4){accv.x, accv.y, accv.z, accv.w};
                }
            }
        }
        if (HAS_DW && fixed_cq) {
            // depthwise stage, a thread keeps its channel quad: the taps of TWO positions are requested before the first is multiplied (one
            // position at a time every output paid a full memory round trip: eight per 128-channel slice)
            auto taps = [&](const PosInfo& pi, float4 (&v9)[9]) {
                const float* xin = a.x + (long)pi.in_base + k0 + 4 * cq_fixed;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int t = i * 3 + j;
                        v9[t] = (pi.mask >> t) & 1 ? *reinterpret_cast<const float4*>(xin + (i * a.W + j) * K) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
            };
            auto finish = [&](const PosInfo& pi, const float4 (&v9)[9], int p) {
                float4 accv = bias;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    accv.x = fmaf(v9[t].x, wgt[t].x, accv.x);
                    accv.y = fmaf(v9[t].y, wgt[t].y, accv.y);
                    accv.z = fmaf(v9[t].z, wgt[t].z, accv.z);
                    accv.w = fmaf(v9[t].w, wgt[t].w, accv.w);
                }
                accv.x = act_f(accv.x, a.dw_act);
                accv.y = act_f(accv.y, a.dw_act);
                accv.z = act_f(accv.z, a.dw_act);
                accv.w = act_f(accv.w, a.dw_act);
                if (pi.mask == 0) accv = make_float4(0.f, 0.f, 0.f, 0.f);
                lds4[p * S4 + cq_fixed] = (f32x4){accv.x, accv.y, accv.z, accv.w};
            };
            for (int item0 = tid; item0 < items; item0 += 512) {
                const int p0 = item0 / kq, p1 = (item0 + 256) / kq;
                const bool two = item0 + 256 < items;
                const PosInfo pi0 = pos[p0], pi1 = pos[two ? p1 : p0];
                float4 va[9], vb[9];
                taps(pi0, va);
                if (two) taps(pi1, vb);
                finish(pi0, va, p0);
                if (two) finish(pi1, vb, p1);
            }
        }
        for (int item = tid; HAS_DW && !fixed_cq && item < items; item += 256) {
            const int p = item / kq;  // power-of-two kq in the fixed case: a shift
            const int cq = fixed_cq ? cq_fixed : item - p * kq;
            const PosInfo pi = pos[p];
            float4 accv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (HAS_DW) {
                if (!fixed_cq) {
                    bias = *reinterpret_cast<const float4*>(a.dw_b + k0 + 4 * cq);
#pragma unroll
                    for (int t = 0; t < 9; ++t) wgt[t] = *reinterpret_cast<const float4*>(a.dw_w + t * K + k0 + 4 * cq);
                }
                const float* xin = a.x + (long)pi.in_base + k0 + 4 * cq;
                float4 v9[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int t = i * 3 + j;
                        v9[t] = (pi.mask >> t) & 1 ? *reinterpret_cast<const float4*>(xin + (i * a.W + j) * K)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                accv = bias;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    accv.x = fmaf(v9[t].x, wgt[t].x, accv.x);
                    accv.y = fmaf(v9[t].y, wgt[t].y, accv.y);
                    accv.z = fmaf(v9[t].z, wgt[t].z, accv.z);
                    accv.w = fmaf(v9[t].w, wgt[t].w, accv.w);
                }
                accv.x = act_f(accv.x, a.dw_act);
                accv.y = act_f(accv.y, a.dw_act);
                accv.z = act_f(accv.z, a.dw_act);
                accv.w = act_f(accv.w, a.dw_act);
                if (pi.mask == 0) accv = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (pi.mask) {
                accv = *reinterpret_cast<const float4*>(a.x + (long)pi.in_base + k0 + 4 * cq);
                if (a.gate) {
                    const int chunk = pi.in_base / (a.H * a.W * K);
                    const float4 g = *reinterpret_cast<const float4*>(a.gate + (size_t)chunk * K + k0 + 4 * cq);
                    accv.x *= g.x;
                    accv.y *= g.y;
                    accv.z *= g.z;
                    accv.w *= g.w;
                }
            }
            lds4[p * S4 + cq] = (f32x4){accv.x, accv.y, accv.z, accv.w};
        }
        if (kcp != kc) {
            const int padq = (kcp - kc) >> 2;
            for (int item = tid; item < 64 * padq; item += 256) lds4[(item / padq) * S4 + kq + item % padq] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();

        // ---- phase 2: [64 x kc] x [kc x N] on the matrix cores, B fragments one k-step ahead -------------
        const int ksteps = kcp >> 4, j0 = k0 >> 4;
        f32x4 bf[CT], bnext[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) bf[c] = wp[((size_t)j0 * n_ct + ct0 + c) * 64 + lane];
        // plain 1x1 convolutions: the A fragments one k-step ahead as well (the LDS read of step j + 1 is issued before the matrix instructions of
        // step j); the depthwise variants have no registers to spare for it (264 > 256: one workgroup per CU) and keep the read inside the step
        if constexpr (!HAS_DW) {
            f32x4 af[RG], afn[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) af[g] = lds4[(row0 + 16 * g + r) * S4 + q];
            for (int j = 0; j < ksteps; ++j) {
                if (j + 1 < ksteps) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) bnext[c] = wp[((size_t)(j0 + j + 1) * n_ct + ct0 + c) * 64 + lane];
#pragma unroll
                    for (int g = 0; g < RG; ++g) afn[g] = lds4[(row0 + 16 * g + r) * S4 + 4 * (j + 1) + q];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < RG; ++g)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][e], bf[c][e], acc[g][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CT; ++c) bf[c] = bnext[c];
#pragma unroll
                for (int g = 0; g < RG; ++g) af[g] = afn[g];
            }
        } else {
            for (int j = 0; j < ksteps; ++j) {
                if (j + 1 < ksteps) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) bnext[c] = wp[((size_t)(j0 + j + 1) * n_ct + ct0 + c) * 64 + lane];
                }
                f32x4 af[RG];
#pragma unroll
                for (int g = 0; g < RG; ++g) af[g] = lds4[(row0 + 16 * g + r) * S4 + 4 * j + q];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < RG; ++g)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][e], bf[c][e], acc[g][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CT; ++c) bf[c] = bnext[c];
            }
        }
    }

    // ---- epilogue: accumulators -> LDS [64][NS + 4] -> bias, residual, activation -> whole-row stores --------
    constexpr int SO = NS + 4;
    __syncthreads();                  // everyone is done reading the activation tile
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) lds_raw[(row0 + 16 * g + 4 * q + reg) * SO + (wn * CT + c) * 16 + r] = acc[g][c][reg];
    __syncthreads();
    const int n_base = blockIdx.y * NS;
#pragma unroll
    for (int i = 0; i < EP; ++i) {
        const int item = tid + 256 * i;
        const int p = item / Q4, c4 = item - p * Q4;
        const int ob = pos[p].out_base;
        if (ob < 0) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds_raw + p * SO + 4 * c4);
        const float4 b = *reinterpret_cast<const float4*>(a.pw_b + n_base + 4 * c4);
        float4 o = make_float4(v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w);
        const long off = (long)ob + n_base + 4 * c4;
        if (a.res) {
            const float4 rv = PREFETCH_RES ? res_pf[PREFETCH_RES ? i : 0] : *reinterpret_cast<const float4*>(a.res + off);
            o.x += rv.x;
            o.y += rv.y;
            o.z += rv.z;
            o.w += rv.w;
        }
        o.x = act_f(o.x, a.pw_act);
        o.y = act_f(o.y, a.pw_act);
        o.z = act_f(o.z, a.pw_act);
        o.w = act_f(o.w, a.pw_act);
        *reinterpret_cast<float4*>(a.y + off) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Wave-autonomous variant for the narrow early layers (Cin, Cout <= 64: stage 1 and 2 of the shipped net).
// Each of the four waves owns 16 of the workgroup's 64 positions from the depthwise inputs to the stores: its depthwise
// outputs, its 16-row A fragments, all Cout columns, its epilogue rows.  Nothing is exchanged between waves, so there is
// no workgroup barrier: the waves drift apart and one wave's HBM round trip overlaps the others' arithmetic (with
// barriers every wave waited for the slowest four times per tile).  Each wave reads the whole (small) weight matrix.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTA>
__global__ __launch_bounds__(256) void f32_dwpw_wave_kernel(DwPwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    constexpr int N = 16 * CTA;
    constexpr int Q4 = N / 4;             // float4 per output row
    constexpr int EP = (16 * Q4) / 64;    // epilogue items per lane
    const int K = a.Cin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int SA = (K > N ? K : N) + 4;   // row stride of the wave's tile in floats
    const int S4 = SA >> 2;
    float* at = lds_raw + wave * 16 * SA;
    f32x4* at4 = reinterpret_cast<f32x4*>(at);
    __shared__ PosInfo pos_all[64];
    PosInfo* pos = pos_all + 16 * wave;

    // ---- this wave's 16 positions: rows of the tile --------------------------------------------------------------
    if (lane < 16) {
        const int tiles_x = a.OW / a.TW, tiles_y = a.OH / a.TH;
        int bid = xcd_tile(blockIdx.x, gridDim.x);
        const int tx0 = (bid % tiles_x) * a.TW;
        bid /= tiles_x;
        const int ty0 = (bid % tiles_y) * a.TH;
        const int chunk = bid / tiles_y;  // NB == 1
        const int rr = 16 * wave + lane;
        const int oh = ty0 + rr / a.TW, ow = tx0 + rr % a.TW;
        PosInfo pi;
        pi.pad = 0;
        pi.out_base = ((chunk * a.OH + oh) * a.OW + ow) * N;
        const int ih0 = oh * a.sh - a.pt, iw0 = ow * a.sw - a.pl;
        pi.in_base = ((chunk * a.H + ih0) * a.W + iw0) * K;
        int mask = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (ih0 + i >= 0 && ih0 + i < a.H && iw0 + j >= 0 && iw0 + j < a.W) mask |= 1 << (i * 3 + j);
        pi.mask = mask;
        pos[lane] = pi;
    }
    wave_sync();

    // residual rows and bias of the epilogue: requested now, used at the very end
    float4 res_pf[EP];
    float4 pwb[EP];
#pragma unroll
    for (int i = 0; i < EP; ++i) {
        const int item = lane + 64 * i;
        const int p = item / Q4, c4 = item - p * Q4;
        pwb[i] = *reinterpret_cast<const float4*>(a.pw_b + 4 * c4);
        res_pf[i] = a.res ? *reinterpret_cast<const float4*>(a.res + (long)pos[p].out_base + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---- depthwise 3x3 of the 16 positions -> the wave's tile [16][K] -------------------------------------------------
    const int kq = K >> 2;             // 64 % kq == 0 (launcher): a lane keeps one channel quad
    const int cq = lane % kq;
    const int ppp = 64 / kq;           // positions per pass
    const int p_first = lane / kq;
    float4 wgt[9];
    const float4 bias = *reinterpret_cast<const float4*>(a.dw_b + 4 * cq);
#pragma unroll
    for (int t = 0; t < 9; ++t) wgt[t] = *reinterpret_cast<const float4*>(a.dw_w + t * K + 4 * cq);
    for (int p = p_first; p < 16; p += ppp) {
        const PosInfo pi = pos[p];
        const float* xin = a.x + (long)pi.in_base + 4 * cq;
        float4 v9[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = i * 3 + j;
                v9[t] = (pi.mask >> t) & 1 ? *reinterpret_cast<const float4*>(xin + (i * a.W + j) * K) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float4 accv = bias;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            accv.x = fmaf(v9[t].x, wgt[t].x, accv.x);
            accv.y = fmaf(v9[t].y, wgt[t].y, accv.y);
            accv.z = fmaf(v9[t].z, wgt[t].z, accv.z);
            accv.w = fmaf(v9[t].w, wgt[t].w, accv.w);
        }
        at4[p * S4 + cq] = (f32x4){act_f(accv.x, a.dw_act), act_f(accv.y, a.dw_act), act_f(accv.z, a.dw_act), act_f(accv.w, a.dw_act)};
    }
    wave_sync();

    // ---- [16 x K] x [K x N] on the matrix cores ---------------------------------------------------------------------------
    f32x4 acc[CTA];
#pragma unroll
    for (int c = 0; c < CTA; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.pw_w);  // [K/16][N/16][64 lanes] float4
    const int ksteps = K >> 4;
    for (int j = 0; j < ksteps; ++j) {
        const f32x4 af = at4[r * S4 + 4 * j + q];
        f32x4 bf[CTA];
#pragma unroll
        for (int c = 0; c < CTA; ++c) bf[c] = wp[((size_t)j * CTA + c) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CTA; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[c][e], acc[c], 0, 0, 0);
    }
    wave_sync();  // every lane has read its A fragments: the tile can take the outputs
#pragma unroll
    for (int c = 0; c < CTA; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) at[(4 * q + reg) * SA + 16 * c + r] = acc[c][reg];
    wave_sync();

    // ---- bias, residual, activation, whole-row stores -----------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < EP; ++i) {
        const int item = lane + 64 * i;
        const int p = item / Q4, c4 = item - p * Q4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(at + p * SA + 4 * c4);
        float4 o = make_float4(v[0] + pwb[i].x, v[1] + pwb[i].y, v[2] + pwb[i].z, v[3] + pwb[i].w);
        if (a.res) {
            o.x += res_pf[i].x;
            o.y += res_pf[i].y;
            o.z += res_pf[i].z;
            o.w += res_pf[i].w;
        }
        o.x = act_f(o.x, a.pw_act);
        o.y = act_f(o.y, a.pw_act);
        o.z = act_f(o.z, a.pw_act);
        o.w = act_f(o.w, a.pw_act);
        *reinterpret_cast<float4*>(a.y + (long)pos[p].out_base + 4 * c4) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Front block: frontend output [64][256] (one channel) -> stem 3x3 (stride 1x2, +BN, ReLU6) -> depthwise 3x3
// stride 2 (+BN, ReLU6) -> pointwise 1x1 (+BN, ReLU6), in ONE kernel.  The stem activation (512 KB per chunk in
// float32, the largest tensor of the network) lives only in LDS: a workgroup computes the 17x17x16 stem patch its
// 8x8 output tile needs from a 19x35 patch of the frontend output.  Same matrix-core tail as f32_dwpw_kernel.
// Reference: stem_conv/stem_bn/stem_relu + stage1_ds1 of birdnet_stm32/models/dscnn.py:198-202,28-84.
struct FrontArgs {
    const float* fe;      // [B][H0][W0]
    float* y;             // [B][OH][OW][N]
    const float* stem_w;  // [3][3][C]
    const float* stem_b;  // [C]
    const float* dw_w;    // [3][3][C]
    const float* dw_b;    // [C]
    const float* pw_w;    // fragment order [C/16][N/16][64][4]
    const float* pw_b;    // [N]
    int B, H0, W0, SH, SW, C, N, OH, OW;  // stem map SH x SW, block output OH x OW
    int stem_act, dw_act, pw_act;
    // audio path: `fe` holds UN-normalised mel energies; the patch load applies relu((x - mn * wsum[row]) / rng) and the
    // magnitude scaling (the frontend without per-sample max normalisation), saving a separate pass over the tensor
    const float* minmax;  // [B][2] or null
    const float* wsum;    // [H0]
    const float* magp;    // [NP][H0]
    int mag;
    int tpw;              // horizontally adjacent tiles one workgroup walks (divides OW / 8)
};

template <int RG, int CT>
__global__ __launch_bounds__(256, 4) void f32_front_kernel(FrontArgs a) {  // 4 waves per SIMD: without the bound the allocator takes 121 registers (3 waves)
    constexpr int TS = 17;            // stem patch edge for an 8x8 tile of a stride-2 depthwise
    constexpr int FH = 19, FW = 35;   // frontend patch: stem stride (1, 2), 3x3
    constexpr int C = 16;             // stem channels = contraction width of the pointwise
    constexpr int NS = RG * CT * 16;
    constexpr int PF = (FH * FW + 255) / 256;  // patch elements per thread
    __shared__ float fe_t[FH][FW + 1];
    __shared__ __attribute__((aligned(16))) float stem_t[TS * TS][C];
    __shared__ __attribute__((aligned(16))) float tile[64 * (C + 4)];  // activation tile [64][C + 4]
    float* otile = &stem_t[0][0];  // output tile [64][NS + 4]: the stem patch is dead once the depthwise stage has run (28.5 KB -> 5 workgroups per CU)
    static_assert(64 * (NS + 4) <= TS * TS * C, "output tile must fit the stem patch");
    const int tid = threadIdx.x;
    // A workgroup walks `tpw` horizontally adjacent 8x8 tiles of one chunk.  The next tile's frontend patch (three floats per
    // thread) is fetched into registers while the current tile is computed, so only the first patch's HBM latency is exposed.
    const int tiles_x = a.OW / 8, tiles_y = a.OH / 8;
    int bid = xcd_tile(blockIdx.x, gridDim.x) * a.tpw;
    const int tx_first = (bid % tiles_x) * 8;
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * 8;
    const int chunk = bid / tiles_y;

    const int r_base = 2 * ty0 - 1;  // stem pad_top 1
    const float* fe = a.fe + (size_t)chunk * a.H0 * a.W0;
    float mn = 0.0f, inv_rng = 1.0f;
    if (a.minmax) {
        mn = a.minmax[2 * chunk];
        inv_rng = 1.0f / (float)((double)(a.minmax[2 * chunk + 1] - mn) + 1e-10);  // one division per workgroup, not per element
    }
    // per-row constants of the finalisation (mel bin = patch row): wsum and the magnitude-scaling rows, staged in LDS once —
    // fetching them per element through the vector L1 would cost ten loads for every patch element
    __shared__ float rowc[FH][12];
    if (a.minmax) {
        for (int i = tid; i < FH * 12; i += 256) {
            const int rr = i / 12, c = i - rr * 12;
            const int gr = r_base + rr;
            float v = 0.0f;
            if (gr >= 0 && gr < a.H0) v = c == 0 ? a.wsum[gr] : (c <= 10 ? a.magp[(c - 1) * a.H0 + gr] : 0.0f);
            rowc[rr][c] = v;
        }
    }
    // Stem and depthwise weights + biases go to LDS once per workgroup; pointwise fragment and bias stay in registers.  Fetching
    // them from global inside every phase of every tile exposed an L2 round trip per phase (waves were stalled ~85 % of their life).
    __shared__ __attribute__((aligned(16))) float wts[2][10][C];  // [stem | depthwise][9 taps + bias][C]
    if (tid < 2 * 10 * (C / 4)) {
        const int which = tid / (10 * (C / 4)), rem = tid - which * 10 * (C / 4);
        const int k = rem / (C / 4), c4 = rem - k * (C / 4);
        const float* src = which ? (k < 9 ? a.dw_w + k * C : a.dw_b) : (k < 9 ? a.stem_w + k * C : a.stem_b);
        *reinterpret_cast<float4*>(&wts[which][k][4 * c4]) = *reinterpret_cast<const float4*>(src + 4 * c4);
    }
    // rows 2*ty0-1 .., cols 4*tx0 .. of the frontend map (zero outside = the stem's SAME padding, pad_left 0)
    float pf[PF];
    auto fetch_patch = [&](int tx0) {
        const int c_base = 4 * tx0;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = tid + 256 * u;
            const int rr = i / FW, cc = i - rr * FW;
            const int gr = r_base + rr, gc = c_base + cc;
            pf[u] = (i < FH * FW && gr >= 0 && gr < a.H0 && gc >= 0 && gc < a.W0) ? fe[gr * a.W0 + gc] : __uint_as_float(0x7fc00000u);
        }
    };
    fetch_patch(tx_first);

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    constexpr int WM = 4 / RG;
    const int wm = wave % WM, wn = wave / WM;
    const int row0 = wm * RG * 16;
    const int ct0 = blockIdx.y * (RG * CT) + wn * CT;
    const int n_ct = a.N >> 4;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.pw_w);
    const int n_base = blockIdx.y * NS;
    f32x4 bfrag[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bfrag[c] = wp[((size_t)0 * n_ct + ct0 + c) * 64 + lane];
    constexpr int Q4 = NS / 4;
    static_assert(256 % Q4 == 0, "each thread keeps one bias quad");
    const float4 pwb = *reinterpret_cast<const float4*>(a.pw_b + n_base + 4 * (tid % Q4));

    for (int t = 0; t < a.tpw; ++t) {
        const int tx0 = tx_first + 8 * t;
        __syncthreads();  // rowc ready (first pass) / previous tile's epilogue done with `tile`, stem phase done with fe_t
        // ---- frontend patch registers -> LDS, finalised on the way (NaN marks "outside the map": stays an exact zero) ------
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = tid + 256 * u;
            if (i >= FH * FW) continue;
            const int rr = i / FW, cc = i - rr * FW;
            float v = pf[u];
            if (v != v) {
                v = 0.0f;
            } else if (a.minmax) {
                const float* rc = rowc[rr];
                const float y = fmaxf((v - mn * rc[0]) * inv_rng, 0.0f);
                if (a.mag == 1) {  // pwl: rows k0, k1..3, w1..3, b1..3
                    v = y * rc[1];
#pragma unroll
                    for (int qq = 0; qq < 3; ++qq) v += rc[2 + qq] * fmaxf(rc[5 + qq] * y + rc[8 + qq], 0.0f);
                } else if (a.mag == 2) {  // pcen-like: rows agc, k1, sw, sb, k2
                    const float y0 = fmaxf(y - rc[1] * y, 0.0f);
                    v = fmaxf(rc[2] * y0 + rc[5] * fmaxf(rc[3] * y0 + rc[4], 0.0f), 0.0f);
                } else if (a.mag == 3) {
                    v = 10.0f * logf(fmaxf(y, 1e-6f)) / logf(10.0f);
                } else {
                    v = y;
                }
            }
            fe_t[rr][cc] = v;
        }
        __syncthreads();
        if (t + 1 < a.tpw) fetch_patch(tx0 + 8);  // in flight during the stem / depthwise / pointwise phases below

        // ---- stem patch: stem rows 2*ty0 .. +16, cols 2*tx0 .. +16; zero where the stem map ends (depthwise SAME padding) ---
        {
            const int cq = tid & 3;
            float4 w9[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w9[k] = *reinterpret_cast<const float4*>(&wts[0][k][4 * cq]);
            const float4 b4 = *reinterpret_cast<const float4*>(&wts[0][9][4 * cq]);
            for (int sp = tid >> 2; sp < TS * TS; sp += 64) {
                const int sr = sp / TS, sc = sp - sr * TS;
                float4 acc = b4;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float v = fe_t[sr + i][2 * sc + j];
                        const float4 w = w9[i * 3 + j];
                        acc.x = fmaf(v, w.x, acc.x);
                        acc.y = fmaf(v, w.y, acc.y);
                        acc.z = fmaf(v, w.z, acc.z);
                        acc.w = fmaf(v, w.w, acc.w);
                    }
                const bool inside = (2 * ty0 + sr) < a.SH && (2 * tx0 + sc) < a.SW;
                acc.x = inside ? act_f(acc.x, a.stem_act) : 0.0f;
                acc.y = inside ? act_f(acc.y, a.stem_act) : 0.0f;
                acc.z = inside ? act_f(acc.z, a.stem_act) : 0.0f;
                acc.w = inside ? act_f(acc.w, a.stem_act) : 0.0f;
                *reinterpret_cast<float4*>(&stem_t[sp][4 * cq]) = acc;
            }
        }
        __syncthreads();

        // ---- depthwise 3x3 stride 2 (pad 0 / 1) from the stem patch -> activation tile [64][C] ----------------------------
        constexpr int S4 = C / 4 + 1;
        f32x4* lds4 = reinterpret_cast<f32x4*>(tile);
        {
            const int cq = tid & 3, p = tid >> 2;  // 64 positions x 4 channel quads = 256 items
            const int py = p >> 3, px = p & 7;
            float4 acc = *reinterpret_cast<const float4*>(&wts[1][9][4 * cq]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float4 v = *reinterpret_cast<const float4*>(&stem_t[(2 * py + i) * TS + 2 * px + j][4 * cq]);
                    const float4 w = *reinterpret_cast<const float4*>(&wts[1][i * 3 + j][4 * cq]);
                    acc.x = fmaf(v.x, w.x, acc.x);
                    acc.y = fmaf(v.y, w.y, acc.y);
                    acc.z = fmaf(v.z, w.z, acc.z);
                    acc.w = fmaf(v.w, w.w, acc.w);
                }
            lds4[p * S4 + cq] = (f32x4){act_f(acc.x, a.dw_act), act_f(acc.y, a.dw_act), act_f(acc.z, a.dw_act), act_f(acc.w, a.dw_act)};
        }
        __syncthreads();

        // ---- pointwise on the matrix cores (one k-step of 16) --------------------------------------------------------------
        f32x4 acc[RG][CT];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                acc[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 af = lds4[(row0 + 16 * g + r) * S4 + q];
                const f32x4 bf = bfrag[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], acc[g][c], 0, 0, 0);
            }
        constexpr int SO = NS + 4;
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) otile[(row0 + 16 * g + 4 * q + reg) * SO + (wn * CT + c) * 16 + r] = acc[g][c][reg];
        __syncthreads();
        for (int item = tid; item < 64 * Q4; item += 256) {
            const int p = item / Q4, c4 = item - p * Q4;
            const int oh = ty0 + (p >> 3), ow = tx0 + (p & 7);
            const f32x4 v = *reinterpret_cast<const f32x4*>(otile + p * SO + 4 * c4);
            const float4 b = pwb;
            float4 o = make_float4(act_f(v[0] + b.x, a.pw_act), act_f(v[1] + b.y, a.pw_act), act_f(v[2] + b.z, a.pw_act),
                                   act_f(v[3] + b.w, a.pw_act));
            *reinterpret_cast<float4*>(a.y + (((size_t)chunk * a.OH + oh) * a.OW + ow) * a.N + n_base + 4 * c4) = o;
        }
    }
}

template <int RG, int CT>
void launch_cfg(const DwPwArgs& a, hipStream_t s) {
    const int tiles = (a.OH / a.TH) * (a.OW / a.TW) * ((a.B + a.NB - 1) / a.NB);
    const int slices = (a.Cout / 16) / (RG * CT);
    const int kc = ((a.Cin < kKC ? a.Cin : kKC) + 15) & ~15, ns = RG * CT * 16;
    const size_t smem = (size_t)64 * ((kc > ns ? kc : ns) + 4) * sizeof(float);
    if (a.has_dw)
        hipLaunchKernelGGL((f32_dwpw_kernel<RG, CT, true>), dim3(tiles, slices), dim3(256), smem, s, a);
    else
        hipLaunchKernelGGL((f32_dwpw_kernel<RG, CT, false>), dim3(tiles, slices), dim3(256), smem, s, a);
}

}  // namespace

bool f32_front_supported(int H0, int W0, int C, int N, int OH, int OW) {
    // stem stride (1,2) pad (1,0); depthwise stride 2 pad (0,0): the shapes of the reference's stem + stage1_ds1
    return C == 16 && N == 32 && OH % 8 == 0 && OW % 8 == 0 && H0 == 2 * OH && W0 == 4 * OW;
}

void launch_f32_front(const float* fe, float* y, int B, int H0, int W0, int C, int N, int OH, int OW, int stem_act,
                      int dw_act, int pw_act, const float* stem_w, const float* stem_b, const float* dw_w, const float* dw_b,
                      const float* pw_w, const float* pw_b, const float* minmax, const float* wsum, const float* magp, int mag,
                      hipStream_t s) {
    if (g_opt.f32_strip && f32_front_strip_supported(H0, W0, C, N, OH, OW)) {
        launch_f32_front_strip(F32FrontStripArgs{fe, y, stem_w, stem_b, dw_w, dw_b, pw_w, pw_b, minmax, wsum, magp, B, H0, W0, OH, OW, 0,
                                                 stem_act, dw_act, pw_act, mag}, s);
        return;
    }
    FrontArgs a{fe, y, stem_w, stem_b, dw_w, dw_b, pw_w, pw_b, B, H0, W0, H0, W0 / 2, C, N, OH, OW, stem_act, dw_act, pw_act,
                minmax, wsum, magp, mag, 1};
    const int forced = g_opt.front_tpw;
    const int tiles_x = OW / 8;
    int tpw = forced > 0 ? forced : 8;
    while (tiles_x % tpw) --tpw;
    a.tpw = tpw;
    const int tiles = (OH / 8) * tiles_x * B;
    hipLaunchKernelGGL((f32_front_kernel<2, 1>), dim3(tiles / tpw, 1), dim3(256), 0, s, a);
}

bool f32_dwpw_supported(int Cin, int Cout) { return Cin % 4 == 0 && Cout % 16 == 0 && Cin >= 4 && Cin <= 2048; }  // Cin % 16 != 0 only without the depthwise stage

// Each workgroup covers RG*CT column tiles of 16 (4 waves = (4/RG) along the 64 rows x RG along the columns);
// wider layers are cut into column slices (grid.y), each recomputing the cheap depthwise stage.
template <int CTA>
static void launch_wave(const DwPwArgs& a, hipStream_t s) {
    const int tiles = (a.OH / a.TH) * (a.OW / a.TW) * a.B;
    const int sa = (a.Cin > a.Cout ? a.Cin : a.Cout) + 4;
    hipLaunchKernelGGL((f32_dwpw_wave_kernel<CTA>), dim3(tiles), dim3(256), (size_t)64 * sa * sizeof(float), s, a);
}

void launch_f32_dwpw(const DwPwArgs& a, hipStream_t s) {
    const int ct_total = a.Cout / 16;
    const int wave_variant = g_opt.wave_dwpw;
    const int strip_variant = g_opt.f32_strip;  // (tests switch it inside one process through bn_set_option)
    if (strip_variant && f32_strip_supported(a)) return launch_f32_strip(a, s);
    if (g_opt.f32_pw_ws && !a.has_dw && launch_f32_pw_ws(a, s)) return;
    if (wave_variant && a.has_dw && a.NB == 1 && a.Cin <= 64 && a.Cout <= 64 && 64 % (a.Cin / 4) == 0 && !a.gate) {
        switch (ct_total) {
            case 1: launch_wave<1>(a, s); return;
            case 2: launch_wave<2>(a, s); return;
            case 3: launch_wave<3>(a, s); return;
            case 4: launch_wave<4>(a, s); return;
        }
    }
    static const int kSlices[] = {24, 16, 12, 8, 6, 4, 3, 2, 1};
    int slice = 1;
    // 24 tiles (384 output channels) in one slice need 99 KB of LDS for the epilogue: one workgroup of four waves per CU, nothing overlaps
    // its load / multiply / store phases.  Two slices of 12 read the input tile twice but run two workgroups per CU: 384-wide layers of the
    // alpha = 1.5 net 0.289 -> 0.234 ms (measured; three slices of 8 and four of 6 are slower again)
    const int cap = g_opt.f32_tile_slice > 0 ? g_opt.f32_tile_slice : 16;
    for (int v : kSlices)
        if (ct_total % v == 0 && v <= cap) {
            slice = v;
            break;
        }
    switch (slice) {
        case 24: launch_cfg<4, 6>(a, s); break;
        case 16: launch_cfg<4, 4>(a, s); break;
        case 12: launch_cfg<4, 3>(a, s); break;
        case 8: launch_cfg<4, 2>(a, s); break;
        case 6: launch_cfg<2, 3>(a, s); break;
        case 4: launch_cfg<4, 1>(a, s); break;
        case 3: launch_cfg<1, 3>(a, s); break;
        case 2: launch_cfg<2, 1>(a, s); break;
        default: launch_cfg<1, 1>(a, s); break;
    }
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_f32_fused() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&f32_front_kernel<2, 1>));
}

}  // namespace bn
