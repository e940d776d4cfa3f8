// bn_exact_dft.h — one STFT element the way the reference evaluates it (float64 window product and DFT, complex64, numpy's |.|), by a
// 16-lane DPP row.  Shared by the exactness pass's own kernels (bn_stft_exact.hip) and by the mel mixer, which re-evaluates the
// elements it finds in doubt itself (bn_i8_fused.hip).  Every function sets `fp contract(off)` for its own body: the reference rounds
// the window product before it enters the sum, a fused multiply-add would not.
#pragma once
#include <hip/hip_runtime.h>

#include "bn_kernels.h"
#include "bn_quant_in.h"

namespace bn {

// float64 sums over the 16 lanes of a DPP row (quad swaps, rotations by 4 and 8): no LDS traffic
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long bits = __builtin_bit_cast(long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row16_sum_d(double v) {
#pragma clang fp contract(off)
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x124>(v);
    v += dpp_d<0x128>(v);
    return v;
}

// LDS copy of the twiddles as (cos, sin) pairs: ONE 16-byte gather per term.  Entry e sits at slot e + (e >> 4): a lane group walks the
// table with stride k, and without the skew every k that is a multiple of 16 would put its 16 lanes on one bank.
struct ExactTabs {
    double2 cs[512 + 32];
};
struct ExactTabsW : ExactTabs {  // with the window as well (callers short of registers: LaneWindow costs 32)
    double hann[256];            // hann[n] = hann[512 - n]: n = 0..255 is all a lane's pairs need
};
__device__ __forceinline__ int cs_slot(int e) { return e + (e >> 4); }
__device__ __forceinline__ void stage_tabs(ExactTabs& tl, const StftTables& tb) {
    for (int i = threadIdx.x; i < 512; i += blockDim.x) tl.cs[cs_slot(i)] = make_double2(tb.cs64[i], tb.cs64[(i + 384) & 511]);  // sin(a) = cos(a - pi/2); sin(0) = 0 exactly
}
__device__ __forceinline__ void stage_tabs(ExactTabsW& tl, const StftTables& tb) {
    stage_tabs(static_cast<ExactTabs&>(tl), tb);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tl.hann[i] = tb.hann64[i];
}
// the window values of a lane's 16 sample pairs n = (lane & 15) + 16 i (hann[512 - n] = hann[n]): fetched once, kept in registers
struct LaneWindow {
    double w[16];
    __device__ __forceinline__ void load(const StftTables& tb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = tb.hann64[(threadIdx.x & 15) + 16 * i];
    }
    __device__ __forceinline__ double at(int i) const { return w[i]; }
};
struct LdsWindow {  // the same values read from ExactTabsW::hann at each use
    const double* h;
    __device__ __forceinline__ double at(int i) const { return h[(threadIdx.x & 15) + 16 * i]; }
};

// |X_k| of frame t the way the reference evaluates it: float64 window product, float64 DFT, complex64, numpy's |.|.
// A GROUP of 16 lanes (one DPP row) evaluates one element; the four groups of a wave work on four elements at once.  Samples
// n and 512 - n share their cosine and have opposite sines (and the same window value), so a lane takes 16 such pairs:
//   re = sum_n (xw[n] + xw[512 - n]) cos(2 pi k n / 512),   im = -sum_n (xw[n] - xw[512 - n]) sin(2 pi k n / 512),   n = 1..255,
// with xw[0] + (-1)^k xw[256] riding on n = 0 (cos = 1, sin = 0).  Tree sum inside the row; every lane of the group returns the value.
// twiddles straight from the (cos, sin) table in global memory (8 KB, L1 / L2 resident) and the window derived from them
// (0.5 - 0.5 cos has one rounding, like the host's table): for callers that re-evaluate a dozen elements and would spend more on
// staging 12 KB into LDS behind a barrier than on the gathers
struct GlobalTabs {
    const double2* cs2;
};
__device__ __forceinline__ double2 twiddle(const ExactTabs& tl, int e) { return tl.cs[cs_slot(e)]; }
__device__ __forceinline__ double2 twiddle(const GlobalTabs& tl, int e) { return tl.cs2[e]; }
struct GlobalWindow {
    const double2* cs2;
    __device__ __forceinline__ double at(int i) const { return fma(-0.5, cs2[(threadIdx.x & 15) + 16 * i].x, 0.5); }
};

// The frame's samples a lane needs: n = gl + 16 i and their mirror partners 512 - n (n = 0 pairs with 256).
struct RowSamples {
    float xa[16], xb[16];
};
__device__ __forceinline__ void exact_row_load(RowSamples& r, const float* __restrict__ x, int T, int hop, int t) {
    const int gl = threadIdx.x & 15;
    // range-checked raw buffer loads over exactly this chunk: samples before / behind it read as 0 (librosa's centre padding) with no
    // branch around the load, so all 32 loads of a lane are in flight together (as conditional loads they ran one round trip at a time:
    // 20 us per element)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, T * 4, 0x00020000);
    const int base = (t * hop - 256) * 4;  // byte offset of the frame's first sample (negative = out of range as unsigned)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = gl + 16 * i;
        r.xa[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 4 * n, 0, 0));
        r.xb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 4 * (n == 0 ? 256 : 512 - n), 0, 0));
    }
}
// SPLIT: the sixteen terms in two groups of eight as far as the scheduler is concerned (half the twiddles and window values in flight at a time:
// for a caller that keeps a second sample set in registers); same operations in the same order.
template <bool SPLIT = false, class Tabs, class Window>
__device__ __forceinline__ float exact_row_value(const Tabs& tl, const Window& lw, const RowSamples& r, int k) {
#pragma clang fp contract(off)
    const int gl = threadIdx.x & 15;
    double re = 0.0, im = 0.0;
    int idx = k * gl;  // k n mod 512, n = gl + 16 i
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double w = lw.at(i);
        const double va = (double)r.xa[i] * w;
        double vb = (double)r.xb[i] * ((i == 0 && gl == 0) ? 1.0 : w);  // n = 0 pairs with n = 256: hann[256] = 1
        if (i == 0 && gl == 0 && (k & 1)) vb = -vb;
        const double2 c = twiddle(tl, idx & 511);
        re = fma(va + vb, c.x, re);
        im = fma(va - vb, c.y, im);
        idx += 16 * k;
        if (SPLIT && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    re = row16_sum_d(re);
    im = row16_sum_d(im);
    return numpy_cabsf((float)re, (float)im);
}
// (load + value in one: the callers that evaluate a handful of elements; stft_minmax_exact_kernel, which may walk hundreds, requests the next
// element's samples before it evaluates the current one)
template <class Tabs, class Window>
__device__ __forceinline__ float exact_mag_row(const Tabs& tl, const Window& lw, const float* __restrict__ x, int T, int hop, int t, int k) {
    RowSamples r;
    exact_row_load(r, x, T, hop, t);
    return exact_row_value(tl, lw, r, k);
}

}  // namespace bn
