// bn_stft.hip — batched 512-point STFT magnitude for gfx950.
//
// Replaces the reference's per-chunk numpy/librosa call
//   np.abs(librosa.stft(y, n_fft=512, hop_length=len(y)//W, win_length=512, window='hann'))[:, :W]
// (reference: birdnet_stm32/audio/spectrogram.py:61,106-115,133) and its min-max
// normalisation (:12-21,149).  Framing: centre zero padding of 256 samples, periodic Hann,
// frame t = samples [t*hop-256, t*hop+256).
//
// One 512-point real FFT = one 256-point complex FFT on z[m] = x[2m] + i x[2m+1] followed by
// the split post-pass.  The 256-point FFT is two radix-16 passes: 16 lanes own one frame, each
// lane runs a 16-point FFT entirely in registers, the 16x16 transpose between the passes and the
// k <-> 256-k pairing of the post-pass go through LDS.  A 256-thread workgroup therefore
// transforms 16 consecutive frames of one chunk; magnitudes are staged in LDS and written as
// frequency-major rows so that the global stores are 64-byte runs.
#include <cstdlib>

#include "bn_kernels.h"
#include "bn_quant_in.h"

namespace bn {

namespace {

constexpr int kFT = 16;    // frames per workgroup
constexpr int kFS = 274;   // floats reserved per frame in the exchange buffer (one component at a time, see the kernel)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// Complex values live in 64-bit register pairs end to end so that the compiler emits packed-f32 instructions
// (v_pk_add/mul/fma_f32) with the swizzles folded into op_sel: a 16-point FFT is ~100 vector instructions.
#define BN_SWAP(a) __builtin_shufflevector(a, a, 1, 0)
#define BN_XX(a) __builtin_shufflevector(a, a, 0, 0)
#define BN_YY(a) __builtin_shufflevector(a, a, 1, 1)

// a * b for complex b given as (b, b_rot) with b_rot = (-b.y, b.x)
__device__ __forceinline__ v2f cmulr(v2f a, v2f b, v2f brot) { return __builtin_elementwise_fma(BN_YY(a), brot, BN_XX(a) * b); }
__device__ __forceinline__ v2f mul_neg_i(v2f a) {  // a * (-i) = (a.y, -a.x)
    v2f t = BN_SWAP(a);
    t.y = -t.y;
    return t;
}

// a + (-i) b = (a.x + b.y, a.y - b.x) and a - (-i) b = (a.x - b.y, a.y + b.x) as ONE packed add each: the swap of b's halves goes into op_sel, the sign into
// neg_hi / neg_lo (the compiler built (b.y, -b.x) with a v_xor and moves first: ~65 + 100 of the STFT kernel's 794 vector instructions).  Same roundings:
// x - y and x + (-y) are the same IEEE operation.
__device__ __forceinline__ v2f add_neg_i(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_neg_i(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// complex a * b and a * b + c with the rotation (-b.y, b.x) of the second product folded into op_sel / neg_lo — the same two roundings per component as
// cmulr (t = a.x b, then fma(a.y, rot b, t)), without building rot b
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ v2f cmul_add(v2f a, v2f b, v2f c) {  // fma(a.y, rot b, fma(a.x, b, c))
    v2f t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(b), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a + conj b and a - conj b
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// forward 4-point DFT, natural order in and out
__device__ __forceinline__ void fft4(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {
    const v2f s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = x1 - x3;
    x0 = s02 + s13;
    x2 = s02 - s13;
    x1 = add_neg_i(d02, d13);
    x3 = sub_neg_i(d02, d13);
}

#define BN_TW(c, s) (v2f){c, s}, (v2f){-(s), c}
// forward 16-point DFT in registers: n = 4p+q, k = r+4s; the result X[r+4s] is left in a[4r+s]
__device__ __forceinline__ void fft16(v2f (&a)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
    for (int q = 0; q < 4; ++q) fft4(a[q], a[4 + q], a[8 + q], a[12 + q]);
    // a[4r+q] *= W16^(q r)
    a[5] = cmulr(a[5], BN_TW(c1, -s1));
    a[6] = cmulr(a[6], BN_TW(h, -h));
    a[7] = cmulr(a[7], BN_TW(s1, -c1));
    a[9] = cmulr(a[9], BN_TW(h, -h));
    a[10] = mul_neg_i(a[10]);
    a[11] = cmulr(a[11], BN_TW(-h, -h));
    a[13] = cmulr(a[13], BN_TW(s1, -c1));
    a[14] = cmulr(a[14], BN_TW(-h, -h));
    a[15] = cmulr(a[15], BN_TW(-c1, s1));
#pragma unroll
    for (int r = 0; r < 4; ++r) fft4(a[4 * r], a[4 * r + 1], a[4 * r + 2], a[4 * r + 3]);
}
// cos / sin of n pi/8 (window angle step between a lane's consecutive samples) and of n pi/16 (split-pass twiddle step)
__device__ constexpr float kCos8[16] = {1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508984f, 6.123233995736766e-17f, -0.38268343236508973f, -0.70710678118654746f, -0.92387953251128674f, -1.0f, -0.92387953251128685f, -0.70710678118654768f, -0.38268343236509034f, -1.8369701987210297e-16f, 0.38268343236509f, 0.70710678118654735f, 0.92387953251128652f};
__device__ constexpr float kSin8[16] = {0.0f, 0.38268343236508978f, 0.70710678118654746f, 0.92387953251128674f, 1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508989f, 1.2246467991473532e-16f, -0.38268343236508967f, -0.70710678118654746f, -0.92387953251128652f, -1.0f, -0.92387953251128663f, -0.70710678118654768f, -0.38268343236509039f};
__device__ constexpr float kCos16[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f, 6.123233995736766e-17f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f, -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f};
__device__ constexpr float kSin16[16] = {0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254546f, 0.70710678118654757f, 0.55557023301960218f, 0.38268343236508989f, 0.19509032201612861f};

// index of X[k] in the array fft16 leaves behind
__device__ __forceinline__ constexpr int fidx(int k) { return 4 * (k & 3) + (k >> 2); }

__global__ void minmax_init_kernel(float* minmax, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        minmax[2 * i] = __uint_as_float(0x7f800000u);  // +inf
        minmax[2 * i + 1] = 0.0f;
    }
}

// With MEL_OUT the band-sparse mel mixer is applied while the magnitudes of the tile are still in LDS and the kernel
// writes [B][M][W] un-normalised mel energies (64 KB/chunk) instead of the [B][257][W] spectrogram (263 KB/chunk).
// min/max of the magnitudes still go to `minmax`: the reference's min-max normalisation commutes with the linear mixer,
//   mel((S - mn)/rng)[m] = (mel(S)[m] - mn * sum_f w[f][m]) / rng,   and is applied by the consumer.
struct MelOut {
    const float* wvals;  // band-sparse mixer values
    const int* bands;    // [3][M] start, len, offset
    float* out;          // [B][M][W]
    int M;
    int square;          // mix |S|^2 instead of |S| (power mel spectrogram of the MFCC mode)
};

// The 16 lanes that own a frame sit in ONE wave, and every LDS exchange of the FFT stays inside that frame's slice of
// `xch`, so no workgroup barrier is needed between the passes: LDS instructions of a wave execute in issue order, the
// compiler only has to keep that order.  Waves of a workgroup therefore drift apart and overlap loads with arithmetic.
__device__ __forceinline__ void frame_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// reductions over the 16 lanes of a frame (one DPP row): quad swaps, then rotations by 4 and 8 — vector-ALU only, no LDS traffic
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    const int i = __builtin_bit_cast(int, v);  // (every source lane of these controls exists: with bound_ctrl the compiler may fold the move into the ALU op)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, true));
}
#define BN_ROW16(OP, v)                \
    v = OP(v, dpp_f<0xB1>(v));  /* quad_perm [1,0,3,2] */ \
    v = OP(v, dpp_f<0x4E>(v));  /* quad_perm [2,3,0,1] */ \
    v = OP(v, dpp_f<0x124>(v)); /* row_ror:4 */           \
    v = OP(v, dpp_f<0x128>(v)); /* row_ror:8 */
__device__ __forceinline__ float addf(float a, float b) { return a + b; }
__device__ __forceinline__ float row16_sum(float v) { BN_ROW16(addf, v) return v; }
__device__ __forceinline__ float row16_max(float v) { BN_ROW16(fmaxf, v) return v; }
__device__ __forceinline__ float row16_min(float v) { BN_ROW16(fminf, v) return v; }

// GUARD (INT8 audio path, bn_stft_exact.hip): besides the magnitudes S' the kernel writes, per frame, the bound eps_t of
// bn_quant_in.h on |S' - S| against the reference's float64 evaluation and, per tile, a record of the THREADS (frame, 16 bins)
// that can hold the chunk's largest / smallest value: with Lt = max (S' - eps) and Ut = min (S' + eps) over the tile, those whose
// largest element reaches Lt within the bound, resp. whose smallest comes down to Ut.  The chunk's min / max themselves are left
// to stft_minmax_exact_kernel (no atomics here).
template <bool MEL_OUT, bool GUARD = false>
__global__ __launch_bounds__(256) void stft512_mag_kernel(StftTables tb, const float* __restrict__ audio, int T, int hop,
                                                          int W, float* __restrict__ spec, float* minmax, MelOut mel,
                                                          int tiles_per_wg, StftGuard guard) {
    // The magnitude tile [257][kFT + 1] (17.5 KB) re-uses the exchange buffer (34.8 KB): every lane takes its sixteen conjugate
    // partners into registers, one workgroup barrier later the buffer is free.  35 KB instead of 52 KB of LDS = four
    // workgroups per CU instead of three.
    // The exchange buffer holds ONE float per complex element: real and imaginary parts take turns (write re, read re, write im,
    // read im — the syncs are wave-level).  17.5 KB instead of 35 KB of LDS lets five workgroups instead of four share a CU (the
    // kernel is latency-bound: 87 registers, ~3.5 k cycles of arithmetic in a 20 k-cycle workgroup lifetime).
    __shared__ float xch[kFT][kFS];
    float (*mag)[kFT + 1] = reinterpret_cast<float (*)[kFT + 1]>(&xch[0][0]);
    static_assert(sizeof(float) * 257 * (kFT + 1) <= sizeof(float) * kFT * kFS, "magnitude tile must fit the exchange buffer");
    __shared__ float red_min[4], red_max[4];
    __shared__ float g_red[3][kFT], g_eps[kFT];  // GUARD: per-frame (lower end of the largest, upper end of the smallest, lower end of the smallest element), eps (-1: frame beyond W)
    __shared__ int g_cnt[2], g_rec[2 * kGuardCand], g_arg;
    __shared__ float g_recv[2 * kGuardCand], g_LU[3];
    // mel mixer tables (MEL_OUT): up to 1024 band-sparse values and 3 x 128 band entries (the launcher refuses more mel bins)
    constexpr int kMelW = MEL_OUT ? 1024 : 1, kMelT = MEL_OUT ? 384 : 1;
    __shared__ float mel_w[kMelW];
    __shared__ int mel_tab[kMelT];
    bool mel_in_lds = false;
    if constexpr (MEL_OUT) {
        const int n_w = mel.bands[3 * mel.M - 1] + mel.bands[2 * mel.M - 1];  // offset + length of the last band
        mel_in_lds = n_w <= kMelW;
        for (int i = threadIdx.x; i < 3 * mel.M; i += 256) mel_tab[i] = mel.bands[i];
        if (mel_in_lds)
            for (int i = threadIdx.x; i < n_w; i += 256) mel_w[i] = mel.wvals[i];
    }  // visible after the workgroup barriers of the FFT below

    const int b = blockIdx.y;
    const int f = threadIdx.x >> 4;
    const int j = threadIdx.x & 15;
    const float* x = audio + (size_t)b * T;

    // Window and split-pass twiddles are rebuilt from ONE per-lane base angle each by the angle-addition formulas with
    // compile-time constants: table gathers through the vector L1 (40 KB per wave and tile) were the bottleneck of this
    // kernel, the extra ~100 packed multiply-adds are free next to it.
    //   w[32 n1 + 2 j + e] = 0.25 - 0.25 cos(theta_je + n1 pi/8)        (0.5 * periodic Hann)
    //   t[j + 16 k2]       = -i exp(-i (alpha_j + k2 pi/16))
    const v4f wb = reinterpret_cast<const v4f*>(tb.window)[j];  // (-0.25 cos th_j0, -0.25 cos th_j1, 0.25 sin th_j0, 0.25 sin th_j1)
    const v2f wa = {wb.x, wb.y}, wsn = {wb.z, wb.w};
    const v4f tbase = reinterpret_cast<const v4f*>(tb.tw512)[j];  // (-sin a_j, -cos a_j, -cos a_j, sin a_j)
    const v2f tp = {tbase.x, tbase.y}, tq = {tbase.z, tbase.w};

    // Raw buffer loads through a descriptor that covers exactly this chunk: samples before the chunk (negative offset = huge
    // unsigned) or after it fail the hardware range check and read as 0, which is librosa's centre padding — no address
    // clamping, no selects, all 32 loads of a tile in flight at once.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, T * 4, 0x00020000);
    (void)tiles_per_wg;  // one tile per workgroup (see stft_tiles_per_wg)
    auto fetch = [&](int tile, v2f (&dst)[16]) {
        const int off0 = (int)(((long)(tile * kFT + f) * hop - 256 + 2 * j) * 4);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            dst[n1].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off0 + 128 * n1, 0, 0));
            dst[n1].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off0 + 128 * n1 + 4, 0, 0));
        }
    };
    float lmin = __uint_as_float(0x7f800000u), lmax = 0.0f;

    {
        const int tile = blockIdx.x;
        const int t0 = tile * kFT;
        const int t = t0 + f;
        // pass 1: lane j owns z[16 n1 + j], n1 = 0..15
        v2f a[16];
        fetch(tile, a);
        if (GUARD && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) guard.n_hard[0] = 0;  // (stft_minmax_exact_kernel's list of given-up chunks)
        float frame_ss = 0.0f;  // GUARD: sum of the frame's squared samples
        if constexpr (GUARD) {
            v2f ss = {0.0f, 0.0f};
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) ss = __builtin_elementwise_fma(a[n1], a[n1], ss);
            frame_ss = row16_sum(ss.x + ss.y);
        }
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const v2f wn = __builtin_elementwise_fma(wsn, (v2f){kSin8[n1], kSin8[n1]},
                                                     __builtin_elementwise_fma(wa, (v2f){kCos8[n1], kCos8[n1]}, (v2f){0.25f, 0.25f}));
            a[n1] = a[n1] * wn;
        }
        fft16(a);
        {
            // inter-pass twiddles W256^(j k1), k1 = 0..15: powers of w = W256^j built from one table entry per lane by
            // square-and-multiply (at most four roundings deep) instead of sixteen strided gathers
            const v4f wl = reinterpret_cast<const v4f*>(tb.tw256)[j];
            v2f p[16];
            p[1] = (v2f){wl.x, wl.y};
#define BN_CM(u, v) cmul(u, v)
            p[2] = BN_CM(p[1], p[1]);
            p[3] = BN_CM(p[2], p[1]);
            p[4] = BN_CM(p[2], p[2]);
            p[5] = BN_CM(p[4], p[1]);
            p[6] = BN_CM(p[4], p[2]);
            p[7] = BN_CM(p[4], p[3]);
            p[8] = BN_CM(p[4], p[4]);
#pragma unroll
            for (int k1 = 9; k1 < 16; ++k1) p[k1] = BN_CM(p[8], p[k1 - 8]);
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) a[fidx(k1)] = BN_CM(a[fidx(k1)], p[k1]);
#undef BN_CM
        }
        // 16 x 16 transpose through LDS, real parts first: a[n].x may be overwritten as soon as every x of the frame has been written
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) xch[f][k1 * 17 + j] = a[fidx(k1)].x;
        frame_sync();
        float re_in[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) re_in[n2] = xch[f][j * 17 + n2];
        frame_sync();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) xch[f][k1 * 17 + j] = a[fidx(k1)].y;
        frame_sync();

        // pass 2: lane j is now k1; gathers over n2
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) a[n2] = (v2f){re_in[n2], xch[f][j * 17 + n2]};
        fft16(a);  // a[fidx(k2)] = Z[j + 16 k2] (scaled by 0.5 through the window)
        frame_sync();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) xch[f][j + 16 * k2] = a[fidx(k2)].x;
        frame_sync();

        // split post-pass: X[k] = E - i W512^k O with E = Z[k] + conj Z[256-k], O = Z[k] - conj Z[256-k] (the 1/2 is in Z)
        const bool live = t < W;
        float tmin = __uint_as_float(0x7f800000u), tmax = 0.0f;
        v2f partner[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) partner[k2].x = xch[f][(256 - (j + 16 * k2)) & 255];
        frame_sync();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) xch[f][j + 16 * k2] = a[fidx(k2)].y;
        frame_sync();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) partner[k2].y = xch[f][(256 - (j + 16 * k2)) & 255];
        __syncthreads();  // all frames have their partners: the buffer becomes the magnitude tile
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = j + 16 * k2;
            const v2f z = a[fidx(k2)];
            const v2f pc = partner[k2];
            const v2f e = add_conj(z, pc), o = sub_conj(z, pc);
            const v2f tk = __builtin_elementwise_fma(tq, (v2f){kSin16[k2], kSin16[k2]}, tp * (v2f){kCos16[k2], kCos16[k2]});  // -i W512^k
            const v2f xr = cmul_add(o, tk, e);  // o.y (-tk.y, tk.x) + (o.x tk + e)
            const v2f sq = xr * xr;
            const float m = __builtin_amdgcn_sqrtf(sq.x + sq.y);
            mag[k][f] = m;
            tmin = fminf(tmin, m);
            tmax = fmaxf(tmax, m);
            if (k == 0) {  // lane 0, k2 = 0: the Nyquist bin is Re Z[0] - Im Z[0]
                const float ny = fabsf(e.x - o.y);
                mag[256][f] = ny;
                tmin = fminf(tmin, ny);
                tmax = fmaxf(tmax, ny);
            }
        }
        if (live) {
            lmin = fminf(lmin, tmin);
            lmax = fmaxf(lmax, tmax);
        }
        if constexpr (GUARD) {
            // the frame's bound on |S' - S| (bn_quant_in.h): eps(S') = eps_f + kGuardRel S' with eps_f = kGuardL2 ||x||_2 + kGuardPeak max_k S'
            const float peak = row16_max(tmax), low = row16_min(tmin);
            const float eps_f = __builtin_amdgcn_sqrtf(frame_ss) * guard.k_l2 + peak * guard.k_peak;
            if (j == 0) {
                g_red[0][f] = live ? guard_lo(peak, eps_f) : 0.0f;
                g_red[1][f] = live ? guard_hi(low, eps_f) : __uint_as_float(0x7f800000u);
                g_red[2][f] = live ? guard_lo(low, eps_f) : __uint_as_float(0x7f800000u);   // (guard_lo is monotone: the lower end of the frame's smallest element)
                g_eps[f] = live ? eps_f : -1.0f;
                if (live) guard.eps[(size_t)b * W + t] = eps_f;
            }
            if (threadIdx.x < 2) g_cnt[threadIdx.x] = 0;
            if (threadIdx.x == 2) g_arg = -1;
        }
        __syncthreads();
        if constexpr (GUARD) {
            const float Lt = fmaxf(row16_max(g_red[0][threadIdx.x & 15]), 0.0f), Ut = row16_min(g_red[1][threadIdx.x & 15]);
            g_LU[0] = Lt;  // (every thread writes the same values: read back for the record below)
            g_LU[1] = Ut;
            g_LU[2] = fmaxf(row16_min(g_red[2][threadIdx.x & 15]), 0.0f);  // no element of the tile lies below this (stft_minmax_exact_kernel: interval minimum)
            // candidates for the chunk's extrema: a THREAD whose largest (smallest) of its 16 bins j + 16 k2 of frame f comes within the
            // bound of the tile's extremum is recorded as (j, f) with the upper (lower) end of what that element can be;
            // stft_minmax_exact_kernel looks at its bins (re-reading the tile here, even only in the waves with a hit, cost 30 us of
            // the kernel's 450)
            const float e_f = g_eps[f];
            if (e_f > 0.0f) {  // (frames of zeros are exact, frames beyond W do not count)
                const float top = guard_hi(tmax, e_f), bot = guard_lo(tmin, e_f);
                if (top >= Lt) {
                    const int sl = atomicAdd(&g_cnt[0], 1);
                    if (sl < kGuardCand) {
                        g_rec[sl] = threadIdx.x;
                        g_recv[sl] = top;
                    }
                }
                // the thread whose smallest element gives the tile's upper end Ut is recorded on its own as well: an overflowing record keeps the first
                // kGuardCand candidates it meets, and the interval form of the minimum wants THE candidate most likely to be it (any writer will do)
                if (guard_hi(tmin, e_f) <= Ut) g_arg = threadIdx.x;
                if (bot <= Ut) {
                    const int sl = atomicAdd(&g_cnt[1], 1);
                    if (sl < kGuardCand) {
                        g_rec[kGuardCand + sl] = threadIdx.x;
                        g_recv[kGuardCand + sl] = bot;
                    }
                }
            }
        }

        if (!MEL_OUT) {
            // frequency-major rows, 16 consecutive frames each
            float* out = spec + (size_t)b * 257 * W;
            if (tiles_per_wg < 0) {
                // tile-major layout [W/16][257][16] (private to bn_infer_audio, read by i8_mel_mfma_kernel<QIN>): the workgroup's
                // tile is ONE contiguous 16 KB block — 256-byte store instructions instead of four 64-byte pieces 1 KB apart
                // (0.548 -> 0.466 ms per 4096 chunks, 4.84 TB/s)
                float* ob = out + (size_t)(t0 / kFT) * 257 * kFT;
                for (int idx = threadIdx.x; idx < 257 * kFT; idx += 256) ob[idx] = mag[idx / kFT][idx % kFT];
            } else
            for (int idx = threadIdx.x; idx < 257 * kFT; idx += 256) {
                const int k = idx / kFT, ff = idx % kFT;
                if (t0 + ff < W) out[(size_t)k * W + t0 + ff] = mag[k][ff];
            }
            if constexpr (GUARD) {
                __syncthreads();
                static_assert(kGuardRec <= 256, "one record entry per thread");
                if (threadIdx.x < kGuardRec) {
                    const int i = threadIdx.x;
                    const int v = i < 2 ? __float_as_int(g_LU[i]) : i < kRecIds ? g_cnt[i - 2] : i < kRecExtra ? g_rec[i - kRecIds]
                                : i == kRecExtra ? __float_as_int(g_LU[2]) : i == kRecExtra + 1 ? g_arg : i < kRecVals ? 0 : __float_as_int(g_recv[i - kRecVals]);
                    guard.rec[((size_t)b * gridDim.x + blockIdx.x) * kGuardRec + i] = v;
                }
                return;  // min / max of the chunk: stft_minmax_exact_kernel
            }
        } else {
            // thread = (frame ff, mel bins j, 31 - j, 32 + j, ...): the serpentine order gives every thread about the same number of
            // band bins (low bands are 2-3 bins wide, the top ones 30).  Mixer values and band tables come from LDS (staged at
            // kernel start): fetched per multiply-add through the vector L1 the loop cost 0.04 of the kernel's 0.19 ms.
            float* out = mel.out + (size_t)b * mel.M * W;
            const int ff = threadIdx.x & 15, jj = threadIdx.x >> 4;
            for (int mb = 0; mb < mel.M; mb += 16) {
                const int m = mb + ((mb & 16) ? 15 - jj : jj);
                if (m >= mel.M) continue;
                const int s0 = mel_tab[m], len = mel_tab[mel.M + m], off = mel_tab[2 * mel.M + m];
                float acc = 0.0f;
                if (mel_in_lds) {
                    const float* wv = mel_w + off;
                    if (mel.square) {
                        for (int i = 0; i < len; ++i) {
                            const float v = mag[s0 + i][ff];
                            acc = fmaf(v * v, wv[i], acc);
                        }
                    } else {
                        for (int i = 0; i < len; ++i) acc = fmaf(mag[s0 + i][ff], wv[i], acc);
                    }
                } else {  // more mixer values than the LDS table holds: through the vector L1
                    const float* wv = mel.wvals + off;
                    for (int i = 0; i < len; ++i) {
                        const float v = mag[s0 + i][ff];
                        acc = fmaf(mel.square ? v * v : v, wv[i], acc);
                    }
                }
                if (t0 + ff < W) out[(size_t)m * W + t0 + ff] = acc;
            }
        }
    }

    // per-chunk min / max (magnitudes are >= 0, so the unsigned bit patterns order like the floats)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lmin = fminf(lmin, __shfl_xor(lmin, off));
        lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red_min[wave] = lmin;
        red_max[wave] = lmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mn = fminf(fminf(red_min[0], red_min[1]), fminf(red_min[2], red_min[3]));
        const float mx = fmaxf(fmaxf(red_max[0], red_max[1]), fmaxf(red_max[2], red_max[3]));
        atomicMin(reinterpret_cast<unsigned int*>(minmax + 2 * b), __float_as_uint(mn));
        atomicMax(reinterpret_cast<unsigned int*>(minmax + 2 * b + 1), __float_as_uint(mx));
    }
}

// (S - min) / (max - min + 1e-10): float32 subtraction and division, the 1e-10 added in double
// and rounded back, which is what numpy does with a float32 array and a Python float.
__global__ void spec_normalize_kernel(float* spec, const float* __restrict__ minmax, int per_chunk) {
    const int b = blockIdx.y;
    const float mn = minmax[2 * b], mx = minmax[2 * b + 1];
    const float rng = (float)((double)(mx - mn) + 1e-10);
    float* p = spec + (size_t)b * per_chunk;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_chunk; i += gridDim.x * blockDim.x)
        p[i] = (p[i] - mn) / rng;
}

}  // namespace

void launch_minmax_init(float* minmax, int B, hipStream_t s) {
    hipLaunchKernelGGL(minmax_init_kernel, dim3((B + 255) / 256), dim3(256), 0, s, minmax, B);
}

// One tile per workgroup: a loop that prefetches the next tile's samples needs > 170 VGPRs and drops the kernel to 2 waves per SIMD
// (measured 0.27 ms against 0.20 ms per 1024 chunks); the kernel's last argument only carries the tile-major flag (-1) now.
static int stft_tiles_per_wg(int, int) { return 1; }

void launch_stft512(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, float* minmax,
                    hipStream_t s, bool tile_major, const StftGuard* guard) {
    const int n_tiles = (W + kFT - 1) / kFT;
    const int tpw = stft_tiles_per_wg(B, n_tiles);
    if (guard)
        hipLaunchKernelGGL((stft512_mag_kernel<false, true>), dim3(n_tiles, B), dim3(256), 0, s, tb, audio, T, hop, W, spec, minmax, MelOut{},
                           (tile_major && W % kFT == 0) ? -1 : tpw, *guard);
    else
        hipLaunchKernelGGL((stft512_mag_kernel<false>), dim3((n_tiles + tpw - 1) / tpw, B), dim3(256), 0, s, tb, audio, T, hop, W, spec,
                           minmax, MelOut{}, (tile_major && W % kFT == 0) ? -1 : tpw, StftGuard{});
}

bool launch_stft512_mel(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* mel_out, int M,
                        const float* wvals, const int* bands, float* minmax, hipStream_t s, int square) {
    if (M > 128) return false;  // band tables are staged in LDS
    const int n_tiles = (W + kFT - 1) / kFT;
    const int tpw = stft_tiles_per_wg(B, n_tiles);
    hipLaunchKernelGGL((stft512_mag_kernel<true>), dim3((n_tiles + tpw - 1) / tpw, B), dim3(256), 0, s, tb, audio, T, hop, W, nullptr,
                       minmax, MelOut{wvals, bands, mel_out, M, square}, tpw, StftGuard{});
    return true;
}

void launch_spec_normalize(float* spec, const float* minmax, int B, int per_chunk, hipStream_t s) {
    hipLaunchKernelGGL(spec_normalize_kernel, dim3(32, B), dim3(256), 0, s, spec, minmax, per_chunk);
}


// bn_preload_kernels (bn_api.hip): asking for one kernel's attributes makes the runtime load this file's device code object now instead of at the
// first launch of one of its kernels.
void preload_stft() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&minmax_init_kernel));
}

}  // namespace bn
