// bn_stft_exact.hip — the float64 side of the STFT: what makes the INT8 path from audio BIT-EXACT with the reference's arithmetic.
//
// The reference computes a chunk's spectrogram as np.abs(librosa.stft(y, 512, hop)) — window product and real FFT in float64,
// the result stored as complex64, |.| by numpy's float32 formula (birdnet_stm32/audio/spectrogram.py:106-115) — normalises it with
// its float32 min / max (:12-21) and the INT8 graph's QUANTIZE (op #0) rounds it to bytes.  The production STFT
// (stft512_mag_kernel, bn_stft.hip) is a float32 FFT: its magnitudes S' differ from the reference's S by ~1e-6 of the peak, which
// flipped 3e-6 of the quantised bytes in round 2.  Integer work has to be identical, so bn_infer_audio now runs
//
//   K1  stft512_mag_kernel<false, GUARD>   S' + a per-frame bound eps_t >= |S' - S| (bn_quant_in.h) + per-tile candidates for the
//                                          chunk's extrema (elements within the bound of the tile's largest / smallest value)
//   K2  stft_minmax_exact_kernel           the candidates that can still be the chunk's max / min re-evaluated in float64
//                                          (512-term DFT per element, a wave per element) -> the EXACT float32 min / max
//   K3  i8_mel_mfma_kernel<QIN, flagging>  quantises S' with the exact min / max, finds every element whose byte could differ for
//                                          some S within [S' - eps, S' + eps] (the quantiser is monotone: test the distance of its
//                                          argument to the next rounding boundary) and re-evaluates those elements in float64 ITSELF
//                                          (512-term DFT by a 16-lane row, bn_exact_dft.h) before the tile is multiplied
//   K4  stft512_f64_list_kernel + K5  i8_mel_mfma_kernel<QIN, worklist>   only for chunks in which a workgroup of K3 found more elements
//                                          in doubt than it keeps (none for sane audio; K3 lists them): whole chunk in float64, its blocks
//                                          through the mixer once more.
//
// Elements that are not listed have the same byte for every S the bound allows, listed ones are exact: the bytes equal the
// oracle's (oracle/stft.py + oracle/int8_graph.py) as long as the float64 DFT here and numpy's float64 FFT round to the same
// complex64 value (they differ by ~1e-16 relative; an element for which that matters is a ~1e-9 event).
//
// stft512_f64_kernel computes a WHOLE spectrogram that way (every bin a float64 DFT): the public bn_stft_mag_exact, and
// bn_infer_audio's INT8 route whenever the guarded fast path does not apply (debug plans, layout / kernel A-B options).
#include "bn_kernels.h"
#include "bn_quant_in.h"
#include "bn_exact_dft.h"

#pragma clang fp contract(off)

namespace bn {

namespace {

constexpr int kFT = 16;  // frames per tile (= stft512_mag_kernel's workgroup)


__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void wave_lds_sync() { wave_sync(); }  // a wave's LDS writes visible to its own lanes: wave-level, no workgroup barrier

__device__ __forceinline__ size_t spec_offset(int W, bool tile_major, int k, int t) {
    return tile_major ? (size_t)(t / kFT) * 257 * kFT + (size_t)k * kFT + (t % kFT) : (size_t)k * W + t;
}

// ------------------------------------------------------------------------------------------------ whole spectrogram in float64
// A 512-point real transform per frame as ONE 256-point complex float64 FFT (z[n] = xw[2n] + i xw[2n+1], four radix-4 Stockham passes
// through LDS, one butterfly per lane and pass) plus the split pass X[k] = E[k] + W^k O[k].  A WAVE owns a frame — its ping / pong
// buffers are its own, so the passes meet at wave-level barriers only — and the four waves of a workgroup share the 16 frames of a
// tile.  Round 3 evaluated every bin as a 255-term float64 DFT (67 MFLOP per chunk, LDS-bound at 7.7 us per chunk); the FFT needs 2.6
// MFLOP.  Both agree with numpy's float64 FFT to ~1e-16 of the frame's norm, i.e. they round to the same complex64 except for elements
// that sit that close to a rounding boundary (the argument of bn_stft_exact.hip's header applies unchanged); the window product is
// rounded to float64 before it enters the transform, as in the reference (fp contraction is off for this file).
struct F64Lds {
    double cs[512];          // cos(2 pi e / 512); sin(2 pi e / 512) = cs[(e + 384) & 511]
    double2 buf[4][2][320];  // per wave: ping / pong (256 elements, one gap per four: fslot)
    double2 win[256];        // window values of sample pairs (2 n, 2 n + 1): read per frame (in registers they cost the third workgroup per CU)
    float red_min[4], red_max[4];
};

__device__ __forceinline__ double2 cmul_tw(const double2 v, const double c, const double sn) {  // v * (c - i sn)
    return make_double2(v.x * c + v.y * sn, v.y * c - v.x * sn);
}

// element i of a wave's 256-point buffer sits at slot i + (i >> 2) (one 16-byte gap per four): the radix-4 passes write with strides of 4 and 16
// elements, which without the skew put the eight lanes of a ds_write_b128 group on a quarter of the banks (4-way conflicts in passes 0 and 1)
__device__ __forceinline__ int fslot(int i) { return i + (i >> 2); }

// The 16 frames of tile (b, tile); 256 threads.  cs must be staged (the caller's loop; the barrier at the top orders it).
// everything that depends on the lane only, loaded once per workgroup and kept in registers across tiles and frames: the window values of the
// lane's 8 samples, the twiddles of its butterfly in passes 1-3 and of its 4 (+1) bins in the split pass
struct F64Regs {
    double2 tw[3][3], tws[4];
    __device__ __forceinline__ void load(const F64Lds& L, const StftTables& tb) {  // after cs is staged and a barrier
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = lane + 64 * i;
            tws[i] = make_double2(L.cs[n], L.cs[(n + 384) & 511]);
        }
#pragma unroll
        for (int pass = 1; pass < 4; ++pass) {
            const int e1 = (lane & ((1 << (2 * pass)) - 1)) * (128 >> (2 * pass));  // 512 (j mod Ns) / (4 Ns)
#pragma unroll
            for (int r = 1; r < 4; ++r) tw[pass - 1][r - 1] = make_double2(L.cs[r * e1], L.cs[(r * e1 + 384) & 511]);
        }
    }
};

__device__ __forceinline__ void f64_tile(F64Lds& L, const F64Regs& R, const float* __restrict__ audio, int T, int hop, int W,
                                         float* __restrict__ spec, float* minmax, bool tile_major, int b, int tile) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t0 = tile * kFT;
    const float* x = audio + (size_t)b * T;
    float* out = spec + (size_t)b * 257 * W;
    float lmin = __uint_as_float(0x7f800000u), lmax = 0.0f;
    __syncthreads();  // cs staged; the previous tile's reduction scratch read
    double2* A = L.buf[wave][0];
    double2* Bf = L.buf[wave][1];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, T * 4, 0x00020000);
    const double2 (&tw)[3][3] = R.tw;
    const double2 (&tws)[4] = R.tws;
    // A wave takes FOUR CONSECUTIVE frames of the tile (4 wave .. 4 wave + 3): in the tile-major layout their magnitudes of one bin are 16 contiguous
    // bytes, written as one dwordx4 per bin when the four are done (a scalar store per frame and bin was 64 four-byte pieces per instruction).
    // The samples of the next frame are requested before the current one is transformed (the walk was a load round trip + a transform per frame).
    auto request = [&](int t, float (&xs)[4][2]) {
        // range-checked raw buffer loads over exactly this chunk: samples before / behind it read as 0 (librosa's centre padding), no selects,
        // all eight loads of the lane in flight together; a frame past the spectrogram's end requests nothing real (offsets behind the buffer)
        const long base = (long)t * hop - 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = t < W ? (int)((base + 2 * (lane + 64 * i)) * 4) : 0x7ffffff0;
            xs[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
            xs[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off < 0x7ffffff0 ? off + 4 : off, 0, 0));
        }
    };
    float xs[4][2], xn[4][2];
    float mag[5][4];
    const int f_lo = t0 + 4 * wave;
    request(f_lo, xs);
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
        const int t = f_lo + fi;
        if (fi < 3) request(t + 1, xn);
        if (t < W) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double2 wv = L.win[lane + 64 * i];
            A[fslot(lane + 64 * i)] = make_double2((double)xs[i][0] * wv.x, (double)xs[i][1] * wv.y);
        }
        wave_lds_sync();
        double2* src = A;
        double2* dst = Bf;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int Ns = 1 << (2 * pass);
            const int jm = lane & (Ns - 1);
            double2 v0 = src[fslot(lane)], v1 = src[fslot(lane + 64)], v2 = src[fslot(lane + 128)], v3 = src[fslot(lane + 192)];
            if (pass) {
                v1 = cmul_tw(v1, tw[pass - 1][0].x, tw[pass - 1][0].y);
                v2 = cmul_tw(v2, tw[pass - 1][1].x, tw[pass - 1][1].y);
                v3 = cmul_tw(v3, tw[pass - 1][2].x, tw[pass - 1][2].y);
            }
            const double2 a0 = make_double2(v0.x + v2.x, v0.y + v2.y), a1 = make_double2(v0.x - v2.x, v0.y - v2.y);
            const double2 a2 = make_double2(v1.x + v3.x, v1.y + v3.y), d13 = make_double2(v1.x - v3.x, v1.y - v3.y);
            const double2 a3 = make_double2(d13.y, -d13.x);  // -i (v1 - v3)
            const int j0 = ((lane - jm) << 2) + jm;
            dst[fslot(j0)] = make_double2(a0.x + a2.x, a0.y + a2.y);
            dst[fslot(j0 + Ns)] = make_double2(a1.x + a3.x, a1.y + a3.y);
            dst[fslot(j0 + 2 * Ns)] = make_double2(a0.x - a2.x, a0.y - a2.y);
            dst[fslot(j0 + 3 * Ns)] = make_double2(a1.x - a3.x, a1.y - a3.y);
            wave_lds_sync();
            double2* tmp = src;
            src = dst;
            dst = tmp;
        }
        // four passes: the transform is back in A (= src).  Split pass: E = (Z_k + conj Z_-k) / 2, O = (Z_k - conj Z_-k) / (2 i), X_k = E + W^k O
        // (the halves are taken once, on the sum: scaling by 1/2 is exact and commutes with every rounding, so 2 E + W (2 O) halved is bit for bit
        // E + W O.  The Nyquist bin X[256] = Re Z[0] - Im Z[0] comes out of lane 0's k = 0 step: as a fifth step it cost every lane a masked pass.)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            const double2 zk = src[fslot(k)], zm = src[fslot((256 - k) & 255)];
            const double2 S2 = make_double2(zk.x + zm.x, zk.y - zm.y);   // 2 E
            const double2 D2 = make_double2(zk.x - zm.x, zk.y + zm.y);
            const double2 O2 = make_double2(D2.y, -D2.x);                // 2 O
            const double2 WO = cmul_tw(O2, tws[i].x, tws[i].y);
            const float m = numpy_cabsf((float)(0.5 * (S2.x + WO.x)), (float)(0.5 * (S2.y + WO.y)));
            mag[i][fi] = m;
            lmin = fminf(lmin, m);
            lmax = fmaxf(lmax, m);
            if (i == 0) {  // (only lane 0's value is stored and enters the reduction)
                const float ny = fabsf((float)(zk.x - zk.y));
                mag[4][fi] = ny;
                if (lane == 0) {
                    lmin = fminf(lmin, ny);
                    lmax = fmaxf(lmax, ny);
                }
            }
        }
        wave_lds_sync();  // the next frame overwrites A
        }
        if (fi < 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xs[i][0] = xn[i][0];
                xs[i][1] = xn[i][1];
            }
        }
    }
    if (f_lo < W) {
        const bool four = tile_major && f_lo + 3 < W;  // (W is a multiple of 4 in every shipped geometry; the scalar path covers the rest)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            if (i == 4 && lane != 0) break;
            if (four) {
                *reinterpret_cast<float4*>(out + spec_offset(W, true, k, f_lo)) = make_float4(mag[i][0], mag[i][1], mag[i][2], mag[i][3]);
            } else {
#pragma unroll
                for (int fi = 0; fi < 4; ++fi)
                    if (f_lo + fi < W) out[spec_offset(W, tile_major, k, f_lo + fi)] = mag[i][fi];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lmin = fminf(lmin, __shfl_xor(lmin, off));
        lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    }
    if (lane == 0) {
        L.red_min[wave] = lmin;
        L.red_max[wave] = lmax;
    }
    __syncthreads();
    if (tid == 0) {
        const float mn = fminf(fminf(L.red_min[0], L.red_min[1]), fminf(L.red_min[2], L.red_min[3]));
        const float mx = fmaxf(fmaxf(L.red_max[0], L.red_max[1]), fmaxf(L.red_max[2], L.red_max[3]));
        atomicMin(reinterpret_cast<unsigned int*>(minmax + 2 * b), __float_as_uint(mn));  // magnitudes are >= 0
        atomicMax(reinterpret_cast<unsigned int*>(minmax + 2 * b + 1), __float_as_uint(mx));
    }
}

__global__ __launch_bounds__(256, 3) void stft512_f64_kernel(StftTables tb, const float* __restrict__ audio, int T, int hop, int W,
                                                          float* __restrict__ spec, float* minmax, int tile_major) {
    __shared__ F64Lds L;
    for (int i = threadIdx.x; i < 512; i += 256) L.cs[i] = tb.cs64[i];
    L.win[threadIdx.x] = make_double2(tb.hann64[2 * threadIdx.x], tb.hann64[2 * threadIdx.x + 1]);
    __syncthreads();
    F64Regs R;
    R.load(L, tb);
    f64_tile(L, R, audio, T, hop, W, spec, minmax, tile_major != 0, blockIdx.y, blockIdx.x);
}

// The chunks of a list (those the guarded pass gives up on: flat spectra, pure tones, signals far below the error bound — more
// elements in doubt than is worth recomputing one by one) as whole float64 spectrograms; their frames' bounds become 0 = exact.
__global__ __launch_bounds__(256, 3) void stft512_f64_list_kernel(StftTables tb, const float* __restrict__ audio, int T, int hop, int W,
                                                               float* __restrict__ spec, float* minmax, int tile_major, const int* __restrict__ list,
                                                               const int* __restrict__ n_list, float* __restrict__ eps) {
    __shared__ F64Lds L;
    const int n_tiles = (W + kFT - 1) / kFT, n = *n_list * n_tiles;
    if ((int)blockIdx.x >= n) return;
    for (int i = threadIdx.x; i < 512; i += 256) L.cs[i] = tb.cs64[i];
    L.win[threadIdx.x] = make_double2(tb.hann64[2 * threadIdx.x], tb.hann64[2 * threadIdx.x + 1]);
    __syncthreads();
    F64Regs R;
    R.load(L, tb);
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int b = list[w / n_tiles], tile = w % n_tiles;
        f64_tile(L, R, audio, T, hop, W, spec, minmax, tile_major != 0, b, tile);
        if (eps && threadIdx.x < kFT && tile * kFT + threadIdx.x < W) eps[(size_t)b * W + tile * kFT + threadIdx.x] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------ K2: exact per-chunk min / max
// One WAVE per chunk, four chunks per workgroup (they share the float64 tables in LDS and nothing else: wave-level syncs only).
// L = max over tiles of (largest S' - eps) is a lower bound of the true maximum, so only elements with S' + eps >= L can be it;
// U = min (smallest S' + eps) likewise for the minimum.  A frame of zeros (eps = 0) has S' = S = 0 exactly and is never listed: it
// shows as U = 0 (then the minimum is 0) and cannot hold the maximum unless every frame is zero.  The tiles' records name THREADS
// of stft512_mag_kernel (frame f = id >> 4, bins j + 16 k2 with j = id & 15, thread j = 0 also bin 256) whose largest / smallest value
// came within the bound of the tile's extremum; the lanes test those threads' bins against L / U in parallel and the few that pass
// are re-evaluated in float64 four at a time.  A chunk with more than kGuardBudget of them (or an overflowing record: flat spectra,
// pure stationary tones, signals far below the bound) goes to stft512_f64_list_kernel as a whole.
constexpr int kK2Thr = 192;      // candidate threads of the minimum a wave keeps
constexpr int kK2MaxThr = 1024;  // ... of the maximum (64 tiles x kGuardCand at most)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void stft_minmax_exact_kernel(StftTables tb, const float* __restrict__ audio, int T, int hop, int W,
                                                                float* __restrict__ spec, int tile_major, StftGuard g, int n_tiles,
                                                                float* __restrict__ minmax, int B) {
    __shared__ ExactTabsW tl;          // (the window from LDS at each use: in registers it costs the 32 that the second sample set below needs)
    const LdsWindow lw{tl.hann};
    __shared__ int cand_s[4][kGuardMaxBudget + 64 + 64];  // passing candidates: is_max << 31 | frame << 16 | bin (maximum from the front, minimum from the back)
    __shared__ unsigned short thr_s[4][kK2MaxThr + kK2Thr + 64];   // candidate threads: tile << 8 | thread id; those of the maximum first (n_thr_max of them).
                                                                   // 16 bits each: with 32 the arrays of this kernel left room for three workgroups per CU —
                                                                   // 768 of a 4096-chunk launch's 1024, i.e. a second scheduling round (22 -> 27 us)
    stage_tabs(tl, tb);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    int* cand = cand_s[wave];
    unsigned short* thr = thr_s[wave];
    const float* x = audio + (size_t)b * T;
    float* S = spec + (size_t)b * 257 * W;
    const float* eps = g.eps + (size_t)b * W;
    const int* rec = g.rec + (size_t)b * n_tiles * kGuardRec;
    if (lane == 0) {  // (the first operator's flagging starts from clean counters)
        g.count[b] = 0;
        g.dirty[b] = 0;
        if (b == 0) {
            *g.n_work = 0;
            g.n_hard[1] = 0;
        }
    }
    float L = 0.0f, U = __uint_as_float(0x7f800000u), Lm = __uint_as_float(0x7f800000u);
    int n_max = 0, n_min = 0;  // lane = tile
    if (lane < n_tiles) {
        const int4 h = *reinterpret_cast<const int4*>(rec + lane * kGuardRec);
        L = fmaxf(__int_as_float(h.x), 0.0f);
        U = __int_as_float(h.y);
        n_max = h.z;
        n_min = h.w;
        Lm = __int_as_float(rec[lane * kGuardRec + kRecExtra]);  // no element of the tile lies below this
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        L = fmaxf(L, __shfl_xor(L, off));
        U = fminf(U, __shfl_xor(U, off));
        Lm = fminf(Lm, __shfl_xor(Lm, off));
    }
    // (wave-uniform after the reductions: kept in scalar registers — the evaluation loop below runs at the register cap)
    L = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(L)));
    U = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(U)));
    Lm = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(Lm)));
    const bool min_is_zero = U == 0.0f;
    if (min_is_zero) n_min = 0;
    bool hard = __ballot(n_max > kGuardCand) != 0;
    // The MINIMUM of noise-free or flat spectra has hundreds of candidates (every near-zero bin lies within the bound of every other): settling
    // them one by one is the whole chunk in float64 (rounds 3-4 did that).  With option stft_minint the minimum is ENCLOSED instead:
    //     Lm <= min <= Um,   Lm = the smallest lower end guard_lo(S') of any element (K1's records),
    //                        Um = min(U, the exact values of the few candidates evaluated here)   (the minimum is <= every exact value).
    // minmax[2 b] = Um, g.mn_lo[b] = Lm; the mel mixer widens its band of doubt by what Um - Lm can move a quantiser argument and decides the
    // elements it re-evaluates for BOTH ends (i8_mel_mfma_kernel; docs/exactness.md "The minimum as an interval").
    const bool may_wide = g.min_interval != 0 && !min_is_zero;
    bool wide = may_wide && __ballot(n_min > kGuardCand) != 0;
    if (!may_wide && __ballot(n_min > kGuardCand) != 0) hard = true;
    if (n_min > kGuardCand) n_min = kGuardCand;   // (an overflowing record still holds its first kGuardCand threads: any of them gives an upper end)
    constexpr int kWideEval = 64;                 // candidates of the minimum evaluated in interval mode: every lane's best — the smallest exact value among
                                                  // them is the interval's upper end, and the narrower the interval the fewer elements the mixer cannot decide
    // 1. collect the recorded threads: lane = tile, slot s of its record per step.  Threads of the maximum first (their overflow gives the chunk up),
    //    then those of the minimum (their overflow only makes the minimum an interval).
    int n_thr = 0;
    for (int s_ = 0; s_ < kGuardCand && !hard; ++s_) {
        // (the record holds, per thread, the upper / lower end of what its extreme element can be: most tiles' entries stop here)
        if (!__ballot(s_ < n_max)) break;
        const bool p_max = s_ < n_max && __int_as_float(rec[lane * kGuardRec + kRecVals + s_]) >= L;
        const unsigned long long m_max = __ballot(p_max);
        const unsigned long long below = (1ull << lane) - 1;
        if (n_thr + __popcll(m_max) > kK2MaxThr) {
            hard = true;
            break;
        }
        if (p_max) thr[n_thr + __popcll(m_max & below)] = (unsigned short)((lane << 8) | rec[lane * kGuardRec + kRecIds + s_]);
        n_thr += __popcll(m_max);
    }
    const int n_thr_max = n_thr;
    if (wide && !hard) {   // the threads that gave their tiles' upper ends first (lane = tile): the likeliest holders of the minimum
        const int arg = lane < n_tiles ? rec[lane * kGuardRec + kRecExtra + 1] : -1;
        const bool p = arg >= 0 && __int_as_float(rec[lane * kGuardRec + 1]) <= U + 4.0f * (U - fmaxf(Lm, 0.0f));   // (tiles whose upper end is near the chunk's)
        const unsigned long long m = __ballot(p);
        if (p) thr[n_thr + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)((lane << 8) | arg);
        n_thr += __popcll(m);
    }
    for (int s_ = 0; s_ < kGuardCand && !hard; ++s_) {
        if (!__ballot(s_ < n_min)) break;
        const bool p_min = s_ < n_min && __int_as_float(rec[lane * kGuardRec + kRecVals + kGuardCand + s_]) <= U;
        const unsigned long long m_min = __ballot(p_min);
        const unsigned long long below = (1ull << lane) - 1;
        if (n_thr - n_thr_max + __popcll(m_min) > kK2Thr) {
            if (!may_wide) hard = true;
            wide = may_wide;   // (the threads collected so far are candidates enough)
            break;
        }
        if (p_min) thr[n_thr + __popcll(m_min & below)] = (unsigned short)((lane << 8) | rec[lane * kGuardRec + kRecIds + kGuardCand + s_]);
        n_thr += __popcll(m_min);
    }
    wave_sync();
    // 2. their bins against L / U: candidates of the maximum from the front of cand[], those of the minimum from the back
    int n_cmax = 0, n_cmin = 0;
    constexpr int kCandCap = kGuardMaxBudget + 64 + 64;
    float best_s = __uint_as_float(0x7f800000u);   // interval mode: this lane's candidate of the minimum with the smallest S'
    int best_code = -1;
    for (int e0 = 0; e0 < n_thr * 17 && !hard; e0 += 64) {
        const int e = e0 + lane;
        bool pred = false;
        int code = 0;
        if (e < n_thr * 17) {
            const int c = thr[e / 17], k2 = e % 17;
            const bool is_max = e / 17 < n_thr_max;
            const int id = c & 0xff, tile = c >> 8, j = id & 15, t = tile * kFT + (id >> 4);
            const int k = j + 16 * k2;
            if (k2 < 16 || j == 0) {
                const float s1 = S[spec_offset(W, tile_major != 0, k, t)], ee = eps[t];
                pred = is_max ? (guard_hi(s1, ee) >= L) : (guard_lo(s1, ee) <= U);
                code = (is_max ? (int)0x80000000 : 0) | (t << 16) | k;
                if (pred && !is_max && s1 < best_s) {
                    best_s = s1;
                    best_code = code;
                }
            }
        }
        const unsigned long long m_max = __ballot(pred && code < 0), m_min = __ballot(pred && code >= 0);
        const unsigned long long below = (1ull << lane) - 1;
        if (n_cmax + __popcll(m_max) > kGuardMaxBudget) hard = true;
        else if (pred && code < 0) cand[n_cmax + __popcll(m_max & below)] = code;
        n_cmax += __popcll(m_max);
        if (!hard && n_cmin < kGuardBudget / 2) {   // (beyond the budget nothing more is kept: the minimum becomes an interval below)
            const int pos = n_cmin + __popcll(m_min & below);
            if (pred && code >= 0 && pos < kGuardBudget / 2) cand[kCandCap - 1 - pos] = code;
        }
        n_cmin += __popcll(m_min);
    }
    if (!hard && n_cmin > kGuardBudget / 2) {
        if (may_wide) wide = true; else hard = true;
    }
    wave_sync();
    if (wide && !hard) {
        // interval mode: every lane's best candidate instead of the first ones met — the smaller the exact values evaluated below, the narrower the
        // interval, the fewer elements the mel mixer cannot decide (tone + 0.01 % noise: 16 % of the chunks handed over with the first 8, 2 % so)
        const unsigned long long m = __ballot(best_code >= 0);
        if (best_code >= 0) cand[kCandCap - 1 - __popcll(m & ((1ull << lane) - 1))] = best_code;
        n_cmin = __popcll(m);
        wave_sync();
    }
    float mx = 0.0f, mn = min_is_zero ? 0.0f : __uint_as_float(0x7f800000u);
    if (!hard) {
        const int grp = lane >> 4, gl = lane & 15;
        const int keep_min = wide ? (n_cmin < kWideEval ? n_cmin : kWideEval) : n_cmin;   // (interval mode: a few candidates are enough for an upper end)
        const int n_cand = n_cmax + keep_min;
        // Four candidates per round (a 16-lane row each).  A stationary tone has hundreds of candidates of the maximum and a round is one dependent
        // round trip to memory: the samples of round r + 1 are requested before round r is evaluated (two sample sets, alternating; the values and
        // their summation order are those of exact_mag_row).
        auto code_of = [&](int ci) { return ci < n_cand ? (ci < n_cmax ? cand[ci] : cand[kCandCap - 1 - (ci - n_cmax)]) : 0; };
        RowSamples rs[2];
        exact_row_load(rs[0], x, T, hop, (code_of(grp) >> 16) & 0x7fff);
        for (int c0 = 0; c0 < n_cand; c0 += 8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ci = c0 + 4 * h + grp;
                if (c0 + 4 * h >= n_cand) break;   // (wave-uniform)
                const bool act = ci < n_cand;
                const int code = code_of(ci);
                const int t = (code >> 16) & 0x7fff, k = code & 0xffff;
                if (c0 + 4 * h + 4 < n_cand) exact_row_load(rs[h ^ 1], x, T, hop, (code_of(ci + 4) >> 16) & 0x7fff);
                __builtin_amdgcn_sched_barrier(0);   // (requests first, then this round's arithmetic: merged, the two rounds need 167 registers)
                const float ex = exact_row_value<true>(tl, lw, rs[h], k);
                __builtin_amdgcn_sched_barrier(0);
                if (act) {
                    if (gl == 0) S[spec_offset(W, tile_major != 0, k, t)] = ex;
                    if (code < 0) mx = fmaxf(mx, ex); else mn = fminf(mn, ex);
                }
            }
        }
        // (the permute addresses of this reduction are rebuilt from an opaque copy of the lane id: shared with the reduction at the top of the
        // kernel they stay live across the loop above and are the three registers that spill at the cap)
        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_ds_bpermute((lane_b ^ off) << 2, __float_as_int(mx))));
            mn = fminf(mn, __int_as_float(__builtin_amdgcn_ds_bpermute((lane_b ^ off) << 2, __float_as_int(mn))));
        }
        if (wide) mn = fminf(mn, U);   // U bounds the minimum from above as well
    }
    if (lane == 0) {
        if (hard) {  // the whole chunk in float64 (stft512_f64_list_kernel reduces into minmax)
            mn = __uint_as_float(0x7f800000u);
            mx = 0.0f;
            g.hard[atomicAdd(g.n_hard, 1)] = b;
        }
        minmax[2 * b] = mn;
        minmax[2 * b + 1] = mx;
        // (an interval of width 0 is an exact minimum)
        g.mn_lo[b] = (!hard && wide && fmaxf(Lm, 0.0f) < mn) ? fmaxf(Lm, 0.0f) : -1.0f;
    }
}

// test hook: the bytes QUANTIZE makes of the spectrogram as it lies in the workspace, [B][257][W] frequency-major
__global__ void spec_bytes_kernel(const float* __restrict__ spec, const float* __restrict__ minmax, int W, int tile_major, float qscale, int qzp,
                                  int8_t* __restrict__ out) {
    const int b = blockIdx.y;
    QuantIn qi;
    qi.set(minmax + 2 * b, qscale, qzp);
    const float* S = spec + (size_t)b * 257 * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 257 * W; i += gridDim.x * blockDim.x) {
        const int k = i / W, t = i - k * W;
        out[(size_t)b * 257 * W + i] = (int8_t)qi.q(S[spec_offset(W, tile_major != 0, k, t)]);
    }
}

}  // namespace

// workgroups of the list kernel: exactly what the device keeps resident (registers and LDS allow two or three per CU) — the kernel strides over
// its work list by gridDim.x, so a grid that needs a second scheduling round would leave the first round's workgroups idle for half the time
// (cached per DEVICE under a mutex, like ensure_dynamic_lds: a process may drive several devices, from several threads)
static int list_grid() {
    static std::mutex mu;
    static std::map<int, int> grids;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    std::lock_guard<std::mutex> lock(mu);
    int& grid = grids[dev];
    if (!grid) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stft512_f64_list_kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (dev < 0 || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        grid = per_cu * prop.multiProcessorCount;
    }
    return grid;
}

void launch_stft512_f64(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, float* minmax, hipStream_t s,
                        bool tile_major) {
    const int n_tiles = (W + kFT - 1) / kFT;
    hipLaunchKernelGGL(stft512_f64_kernel, dim3(n_tiles, B), dim3(256), 0, s, tb, audio, T, hop, W, spec, minmax,
                       (tile_major && W % kFT == 0) ? 1 : 0);
}

void launch_spec_bytes(const float* spec, const float* minmax, int B, int W, bool tile_major, float qscale, int qzp, int8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(spec_bytes_kernel, dim3(32, B), dim3(256), 0, s, spec, minmax, W, tile_major ? 1 : 0, qscale, qzp, out);
}

void launch_stft_minmax_exact(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, bool tile_major,
                              const StftGuard& g, float* minmax, hipStream_t s) {
    hipLaunchKernelGGL(stft_minmax_exact_kernel, dim3((B + 3) / 4), dim3(256), 0, s, tb, audio, T, hop, W, spec, tile_major ? 1 : 0, g,
                       (W + kFT - 1) / kFT, minmax, B);  // (the caller keeps W <= 1024: one lane per tile record)
    // chunks the wave gave up on (none for ordinary audio: the 256 workgroups read the count and leave)
    hipLaunchKernelGGL(stft512_f64_list_kernel, dim3(list_grid()), dim3(256), 0, s, tb, audio, T, hop, W, spec, minmax, tile_major ? 1 : 0, g.hard, g.n_hard,
                       g.eps);
}

void launch_stft_fix(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, bool tile_major, const StftGuard& g,
                     const float* minmax, float qscale, int qzp, hipStream_t s) {
    (void)qscale;
    (void)qzp;
    (void)B;
    // chunks a workgroup of the mel mixer gave up on (it listed them itself): whole float64 spectrograms; the caller then runs the mixer's
    // work-list form over their blocks
    // (minmax is exact already: the atomics of this pass find the same values)
    hipLaunchKernelGGL(stft512_f64_list_kernel, dim3(list_grid()), dim3(256), 0, s, tb, audio, T, hop, W, spec, const_cast<float*>(minmax), tile_major ? 1 : 0,
                       g.hard + g.hard_cap, g.n_hard + 1, (float*)nullptr);
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_stft_exact() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&stft_minmax_exact_kernel));
}

}  // namespace bn
