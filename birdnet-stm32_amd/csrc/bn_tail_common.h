// bn_tail_common.h — helpers shared by the two fused-tail kernels (bn_i8_tail.hip, bn_i8_tail2.hip): byte permutes, the requantisation
// forms, and the MEAN -> FULLY_CONNECTED -> head part behind the last block.
#pragma once
#include "bn_kernels.h"
#include "bn_requant.h"

namespace bn {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int perm(int s0, int s1, uint32_t sel) { return (int)__builtin_amdgcn_perm((uint32_t)s0, (uint32_t)s1, sel); }
__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
__device__ __forceinline__ int med3(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}
// RoundingDivideByPOT(SRDHM(x, m), e) + zp with the rounding offset and the zero point in one addend (bn_i8_strip.hip: rq)
__device__ __forceinline__ int rq(int x, int m, int c1, int e) {
    const int v = srdhm_pos(x, m);
    return (v + c1 + (v >> 31)) >> e;
}

// where the clamp's lower bound is at or above the zero point the sign term is not needed, the addend folds into the 64-bit
// multiply-add and the result is the HIGH dword shifted by e - 1 (bn_i8_strip.hip: rq_hi; the packer checks the clamp and e >= 1).
// The four shifts of a channel quad sit in the bytes of one register, SDWA picks byte `e`.
__device__ __forceinline__ long rq64(int c1) { return ((long)c1 << 31) + 0x40000000L; }
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
// (multiplier, c1, shift) of the constant block -> (multiplier, C01, C23, packed shifts - 1) in LDS
__device__ __forceinline__ void stage_rq(v4i* dst, v4i m, v4i c1, v4i sh) {
    const long c[4] = {rq64(c1.x), rq64(c1.y), rq64(c1.z), rq64(c1.w)};
    dst[0] = m;
    dst[1] = (v4i){(int)c[0], (int)(c[0] >> 32), (int)c[1], (int)(c[1] >> 32)};
    dst[2] = (v4i){(int)c[2], (int)(c[2] >> 32), (int)c[3], (int)(c[3] >> 32)};
    dst[3] = (v4i){pack_shifts(sh - 1), 0, 0, 0};
}
__device__ __forceinline__ long pair(int lo, int hi) { return __builtin_bit_cast(long, (v2i){lo, hi}); }

// MEAN + FULLY_CONNECTED + head for the workgroup's chunks.  (As a real function call it costs a stack copy of the arguments: 1168 B of
// scratch per lane — it stays inlined.)
template <int NTHREADS, int PAD, class Args>
__device__ __forceinline__ void tail_head(const Args& a, unsigned char* lds, int chunk0) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // keeps this part's index arithmetic from being hoisted to the top of the kernel and held in registers across
                                   // the blocks, which run at the 128-register cap of a 16-wave workgroup (it was what spilled)
    // The classifier matrix (NC x C bytes: 25.6 KB) comes into LDS for the four chunks of the group: as 64 dependent dword loads per
    // (chunk, class) thread, 256 bytes apart between neighbouring threads, the head took 15 % of a group's time with 60 % of the threads idle.
    // Requested here (coalesced 16-byte loads, one round trip for the whole workgroup), written behind the MEAN; it overlays the last
    // block's pointwise weights, which nobody reads any more (the block's end barrier is behind us).  Rows of C / 4 + 1 dwords: thread j
    // walks row j, so neighbouring threads sit on neighbouring banks.
    const auto& L = a.L[a.n_layers - 1];
    const int c16 = a.C / 16, n_w16 = a.NC * c16, pitchw = a.C / 4 + 1;
    constexpr int NW = 4096 / NTHREADS;   // staging rounds: up to 4096 sixteen-byte pieces of the classifier matrix
    v4i wreg[NW];
    if (a.fcw_off >= 0) {
        const v4i* gw = reinterpret_cast<const v4i*>(a.cst + a.g_fcw);
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * NTHREADS;
            wreg[u] = i < n_w16 ? gw[i] : (v4i){0, 0, 0, 0};
        }
    }
    // ---- MEAN over the positions of the last map: one thread per (chunk slot, channel) ---------------------------------------
    const int pitch = a.C + PAD;
    for (int i = tid; i < kTailG * a.C; i += NTHREADS) {
        const int g = i / a.C, c = i - g * a.C;
        const int8_t* src = reinterpret_cast<const int8_t*>(lds + L.y_off + g * a.P * pitch + c);
        int s = 0;
        for (int k = 0; k < a.P; ++k) s += src[k * pitch];
        reinterpret_cast<int8_t*>(lds + a.mean_off)[i] = (int8_t)mean_q(s, a.P, a.mean_zp_in, a.mean_mult, a.mean_shift, a.mean_zp_out);
    }
    if (a.fcw_off >= 0) {
        int* wl = reinterpret_cast<int*>(lds + a.fcw_off);
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + u * NTHREADS;
            if (i < n_w16) {
                int* d = wl + (i / c16) * pitchw + 4 * (i % c16);
                d[0] = wreg[u].x; d[1] = wreg[u].y; d[2] = wreg[u].z; d[3] = wreg[u].w;
            }
        }
    }
    __syncthreads();
    // ---- FULLY_CONNECTED + head: one thread per (chunk slot, class) ----------------------------------------------------------
    for (int i = tid; i < kTailG * a.NC; i += NTHREADS) {
        const int g = i / a.NC, j = i - g * a.NC;
        const int chunk = chunk0 + g;
        if (chunk >= a.B) continue;
        const int* xr = reinterpret_cast<const int*>(lds + a.mean_off + g * a.C);
        const int* wr = a.fcw_off >= 0 ? reinterpret_cast<const int*>(lds + a.fcw_off) + j * pitchw : a.cst + a.g_fcw + j * (a.C / 4);
        int acc = a.cst[a.g_fcb + j];
        for (int k = 0; k < a.C / 4; ++k) acc = dot4(xr[k], wr[k], acc);
        const int qv = clampi(mbqm(acc, a.cst[a.g_fcm + j], a.cst[a.g_fcs + j]) + a.fc_zp_out, a.fc_lo, a.fc_hi);
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


}  // namespace
}  // namespace bn
