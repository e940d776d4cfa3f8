// bn_kernels.h — host-callable launchers of the gfx950 kernels (internal, not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "bn_blob.h"

namespace bn {

// Run-time switches of the launchers (A/B runs, tests).  The process default is set through the C ABI (bn_set_option; the BN_* environment
// variables of the same names seed it ONCE when the library is loaded — nothing on a launch path reads the environment); a context may
// override single switches for itself (bn_ctx_set_option), so two models in one process can run under different options.  What the launchers
// read is `g_opt`: a per-thread copy that every API entry point refreshes from the default and its context's overrides (bn_api.hip: check_device).
struct Options {
    int f32_strip = 1;         // float32 row-streaming strip kernels (0: tile kernels everywhere)
    int f32_strip_th = 0;      // force the rows per strip (0: the launcher's choice)
    int f32_front_staged = 1;  // LDS-staged form of the float32 front strip kernel
    int i8_pwdw = 0;           // 1: expand 1x1 + depthwise 3x3 of exported inverted-residual graphs as one kernel (i8_pwdw_kernel; bit-identical, but
                               // measured 1.5-1.8 x SLOWER than the two kernels: off by default)
    int f32_pw_ws = 1;         // plain 1x1 convolutions with Cin > 128 through the persistent producer / consumer kernel (bn_f32_pw.hip)
    int f32_tile_slice = 0;    // > 0: cap on the 16-column tiles per workgroup slice of the f32 tile kernel (0: the default cap of 16)
    int f32_pwdw = 2;          // expand 1x1 + depthwise 3x3 of inverted-residual blocks as one kernel (the expanded map stays in LDS); 2: it also
                               // hands the squeeze-excite gate behind it per-row-block channel sums (1: the gate pools the map itself, bit-identical to 0)
    int f32_front2 = 1;        // front block + first residual block as one kernel (the map between them stays in LDS)
    int front_tpw = 0;         // tiles per workgroup of the float32 front tile kernel (0: auto)
    int wave_dwpw = 1;         // wave-autonomous variant of the small float32 fused block
    int i8_strip = 1;          // INT8 strip kernels (0: generic fused block everywhere)
    int i8_strip_th = 0;       // force the rows per wave of the INT8 strip kernels (0: auto)
    int i8_dw_pool = 1;        // the row-streaming depthwise kernel of exported graphs adds up what it stores for the squeeze-excite MEAN behind it
                               // (integer sums, order-free: bit-identical); 0: i8_segate_kernel reads the whole map again
    int i8_pw_lds = 1;         // dense 1x1 convolutions (Cin 192 / 384 / 768) of exported INT8 graphs through i8_pw_lds_kernel (bn_i8_pw.hip): weights of a
                               // slice of output channels resident in LDS, squeeze-excite MUL applied on load (0: tile kernel + i8_scale)
    int i8_pw_forms = 1;       // i8_pw_wave_kernel / i8_dw_stream_kernel pick their requantisation form at compile time where the operator's constants allow
                               // (sign-free one-multiply-add form behind ReLU, branch-free right-shift form elsewhere); 0: the runtime-uniform general code
    int i8_add_tab = 1;        // projections with a residual ADD (i8_pw_wave / i8_pw_lds kernels): the ADD as one lookup in a 64 KB table in LDS
                               // (1024-thread workgroups, one per CU); 0: two 256-entry rescale tables + the output requantisation on the vector ALU
    int i8_tail_fclds = 1;     // the fused tail's head reads the classifier matrix from an LDS copy (0: from memory, 64 dependent loads per thread)
    int i8_tail = 1;           // stage 3-4 + MEAN + FC + head of the INT8 graph as one kernel (0: one launch per block)
    int i8_mid = 1;            // stage 2 of the shipped INT8 graph (a stride-2 block + two residual blocks on a 16 x 32 map) as one kernel with the maps of two
                               // chunks in LDS (i8_mid2_kernel: depthwise stage on the matrix cores); 0: one strip kernel per block
    int i8_tail_mfdw = 1;      // ... with the depthwise stage on the matrix cores (i8_tail2_kernel) where the plan carries its constants; 0: i8_tail_kernel
    int i8_mel_generic = 0;    // run the mel mixer through the generic fused block
    int stft_rowmajor = 0;     // keep the reference spectrogram layout inside bn_infer_audio (default: tile-major)
    int i8_strip_mfdw = 1;     // stage1_ds2-shaped blocks (32 -> 32 channels, stride 1, residual ADD): depthwise 3x3 on the matrix cores (i8_strip_mf_kernel); 0: i8_strip_kernel
    int stft_exact = 2;        // INT8 plans from audio: 2 = float32 STFT + float64 pass over the doubtful elements (bit-exact input bytes,
                               // bn_stft_exact.hip; plans / options the guarded kernels do not cover take 1), 1 = every bin as a float64
                               // DFT (same bytes, ~10 x slower), 0 = plain float32 STFT (round 2: ~3e-6 of the input bytes off by one)
    int stft_guard = 0;        // the bound |S' - S| <= eps the exactness pass works with (bn_quant_in.h): 0 = empirical (4 x the largest error seen over 2.2e11
                               // elements), 1 = proven (the worst case over all rounding patterns, derived in docs/exactness.md: ~25 x wider, most chunks
                               // end on the float64 routes), 2 = TEST ONLY: the empirical constants / 1024 and no quantiser slack — too small on purpose, so that the audit has something to find
    int stft_audit = 0;        // 1: the guarded mel mixer also re-evaluates in float64 the elements it did NOT flag but that lie within 4 bounds of a rounding
                               // boundary (the near misses) and counts those whose kept byte is wrong — bn_debug_guard_stats out[5] audited, out[6] violations
    int stft_minint = 1;       // 1: a chunk whose MINIMUM has too many candidates to settle one by one (noise-free tones, chirps, flat spectra: every near-zero bin
                               // is one) keeps the fast path with the minimum ENCLOSED in an interval that widens the band of doubt (docs/exactness.md);
                               // 0: such a chunk is recomputed as a whole float64 spectrogram (rounds 3-4)
    int stft_flagcap = 1022;   // elements in doubt a workgroup of the INT8 mel mixer re-evaluates itself before it hands the chunk over (tests lower it)
    int ingest_blk = 0;        // outputs per workgroup of the resampler (0: auto)
    int ingest_generic = 0;    // generic polyphase kernel instead of the phase-per-thread form
};
extern thread_local Options g_opt;

// A kernel that needs more than the default 64 KB of dynamic LDS has its limit raised with hipFuncSetAttribute — per DEVICE (the attribute
// belongs to the current device's copy of the function; bn_ctx_create takes a device index) and only when the request grows.  false: the
// runtime refused (the caller falls back or reports the launch error).
inline bool ensure_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> allowed;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = allowed[{kernel, dev}];
    if (bytes <= have) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    have = bytes;
    return true;
}

// Load-time validation of a packed plan (bn_plan_check.hip): every operator's geometry against the slot and tensor sizes.
bool check_plan(const BlobHeader& h, const std::vector<SlotRec>& slots, const std::vector<TensorRec>& tensors,
                const std::vector<OpRec>& ops, std::string& err);

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Tiles that share input rows
// (3x3 halos, the residual of the same positions) should meet in ONE L2, so the linear tile index is permuted: XCD k walks
// the k-th eighth of the tile range.  A bijection on [0, nb); the tail nb % 8 maps to itself.
__device__ __forceinline__ int xcd_tile(int b, int nb) {
    const int nb8 = nb & ~7;
    return b < nb8 ? (b & 7) * (nb8 >> 3) + (b >> 3) : b;
}

// Read-only tables every STFT launch needs (built once per context, in double, stored f32).
struct StftTables {
    const float* window;    // [16][4] per-lane window base: (-0.25 cos th_j0, -0.25 cos th_j1, 0.25 sin th_j0, 0.25 sin th_j1),
                            //         th_je = 2 pi (2 j + e) / 512; the 1/2 of the real-FFT split is folded into the window
    const float4* tw256;    // [256] (w, w_rot): w = exp(-2 pi i p / 256), w_rot = (-w.y, w.x)
    const float4* tw512;    // [16] per-lane split-pass base: (-sin a_j, -cos a_j, -cos a_j, sin a_j), a_j = 2 pi j / 512
    const double* hann64;   // [512] 0.5 - 0.5 cos(2 pi n / 512): the reference's float64 window (bn_stft_exact.hip)
    const double* cs64;     // [512] cos(2 pi j / 512) with exact symmetries (sin by index shift)
    const double2* cs2;     // [512] (cos, sin)(2 pi j / 512): the same values as pairs
};

// Exactness pass of the INT8 audio path (bn_stft_exact.hip explains the five kernels).
// Record of one (chunk, 16-frame tile) of stft512_mag_kernel<., GUARD> for stft_minmax_exact_kernel, in ints:
//   [0] L bits  [1] U bits  [2] n_max  [3] n_min   (lower end of the tile's largest element, upper end of its smallest, candidate threads counted)
//   [kRecIds ..)   kGuardCand thread ids of the maximum, then kGuardCand of the minimum
//   [kRecExtra ..) lower end of the tile's smallest element (bits), the thread that gave U, 2 unused
//   [kRecVals ..)  per recorded thread the upper (maximum) / lower (minimum) end of what its extreme element can be
// kGuardCand = 62 (round 5; 30 before): a STATIONARY tone has every frame's peak within the bound of every other's — 16 frames x 2-4 threads of a
// tile are candidates of the maximum, and an overflowing record gave such chunks to the float64 STFT as a whole.
constexpr int kGuardCand = 62;
constexpr int kRecIds = 4, kRecExtra = kRecIds + 2 * kGuardCand, kRecVals = kRecExtra + 4;
constexpr int kGuardRec = kRecVals + 2 * kGuardCand;   // 256
constexpr int kGuardBudget = 48;    // candidates of the MINIMUM stft_minmax_exact_kernel settles one by one (half of it) before it encloses the minimum in an interval
constexpr int kGuardMaxBudget = 1024;   // float64 re-evaluations it spends on a chunk's MAXIMUM before it gives the chunk up (a wave per chunk: ~1.3 us per four)
struct StftGuard {
    float* eps;    // [B][W] per-frame bound on |S' - S|
    int* rec;      // [B][ceil(W / 16)][kGuardRec]
    float* mn_lo;  // [B] >= 0: stft_minmax_exact_kernel could only ENCLOSE the chunk's minimum (flat / noise-free spectra: too many candidates) —
                   // minmax[2 b] is an upper end, this the lower end; -1: minmax[2 b] is the exact minimum
    int* count;    // [B] elements found in doubt; > cap: a workgroup of the mixer gave up on the chunk
    int cap;
    int* dirty;    // [B] bit per 64-frame block whose quantised bytes changed
    int* work;     // [B * ceil(W / 64)] dirty (chunk, block) pairs
    int* n_work;   // [1]
    int* hard;     // [2][hard_cap] chunks recomputed as whole float64 spectrograms: behind the min / max pass, behind the fix pass
    int* n_hard;   // [2]
    int hard_cap;
    const float* audio;  // [B][T] the chunks' samples, their geometry and the float64 tables: the mel mixer re-evaluates the elements it finds
    int T, hop;          // in doubt itself (set per call by bn_infer_audio)
    StftTables tabs;
    float k_l2, k_peak;  // the frame part of the bound: eps_f = k_l2 ||x_t||_2 + k_peak max_k S'_tk (option stft_guard picks the constants)
    int* audit;          // [2] elements audited, violations (option stft_audit; null: no audit)
    float audit_scale;   // the audit's band = kAuditBands x (audit_scale eps + slack): 1, or what the test mode divided the constants by
    float slack_scale;   // 1; 0 in the test mode (stft_guard = 2), which drops the quantiser's own slack as well so that real errors escape the band
    int min_interval;    // option stft_minint: enclose the minimum instead of giving the chunk up (0: round-4 behaviour)
    int flag_cap;        // elements in doubt a workgroup of the mel mixer keeps (<= its LDS list; option stft_flagcap: tests lower it to reach the give-up path)
};

// ---- STFT ------------------------------------------------------------------------------
void launch_minmax_init(float* minmax, int B, hipStream_t s);
// tile_major: spectrogram written as [W/16][257][16] instead of [257][W] (needs W % 16 == 0; private layout of bn_infer_audio)
void launch_stft512(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec,
                    float* minmax, hipStream_t s, bool tile_major = false, const StftGuard* guard = nullptr);
// every bin as a float64 DFT, |.| by numpy's formula: the reference's values (bn_stft_exact.hip); minmax as launch_stft512
void launch_stft512_f64(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, float* minmax, hipStream_t s,
                        bool tile_major = false);
void launch_spec_bytes(const float* spec, const float* minmax, int B, int W, bool tile_major, float qscale, int qzp, int8_t* out, hipStream_t s);
void launch_stft_minmax_exact(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, bool tile_major,
                              const StftGuard& g, float* minmax, hipStream_t s);
void launch_stft_fix(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* spec, bool tile_major, const StftGuard& g,
                     const float* minmax, float qscale, int qzp, hipStream_t s);
bool launch_stft512_mel(const StftTables& tb, const float* audio, int B, int T, int hop, int W, float* mel_out, int M,
                        const float* wvals, const int* bands, float* minmax, hipStream_t s, int square = 0);
// per-chunk finishing pass of the precomputed-frontend spectrogram modes (bn_melspec.hip)
size_t melspec_finish_lds_bytes(int M, int W, int Wout, int mode, int mag, int n_mfcc);
bool launch_melspec_finish(const float* mel, float* out, const float* dct, int B, int M, int W, int Wout, int mode, int mag,
                           int n_mfcc, double pcen_b, hipStream_t s);
void launch_spec_normalize(float* spec, const float* minmax, int B, int per_chunk, hipStream_t s);

// ---- ingest and pooling (bn_ingest.hip) ---------------------------------------------------
size_t ingest_resample_lds_bytes(int up, int down, int hpp, int blk);
int ingest_resample_block(int up, int down, int hpp);
size_t ingest_partial_elems(int n_files, long max_out, int up, int down, int hpp);
void launch_ingest_resample(const void* pcm, int fmt, int ch, const long* in_off, const long* out_off, int n_files,
                            long max_out, const float* taps, int up, int down, int hpp, int n_pre_remove, float* mono,
                            float* partial, float* peak, hipStream_t s);
void launch_ingest_chunks(const float* mono, const float* peak, const long* src, const int* valid, const int* file,
                          int n_chunks, int T, float* out, hipStream_t s);
void launch_chunk_peaknorm(const float* x, float* y, int B, int T, float eps, hipStream_t s);
void launch_pool_scores(const float* scores, const long* seg, int F, int C, int method, float beta, float* out,
                        hipStream_t s);

// ---- float32 plan ------------------------------------------------------------------------
void launch_f32_mel(const float* spec, const float* minmax, float* out, float* smax, int B, int F, int W, int M,
                    const float* wvals, const int* bands, const float* magp, int mag, int norm, hipStream_t s);
void launch_f32_melfin(const float* melraw, const float* minmax, float* out, int B, int M, int W, const float* wsum,
                       const float* magp, int mag, int norm, hipStream_t s);
void launch_f32_rawfe(const float* x, float* out, int B, int T, int W, int M, int stride, int pad_left, const float* fb,
                      const float* bias, const float* magp, int mag, hipStream_t s);
void launch_f32_mag(float* x, const float* smax, int B, int M, int W, const float* magp, int mag, hipStream_t s);
void launch_u32_fill(uint32_t* p, uint32_t v, int n, hipStream_t s);
void launch_f32_stem(const float* x, float* y, int B, int H, int W, int Cout, int sh, int sw, int act, int OH, int OW,
                     int pt, int pl, const float* w, const float* bias, hipStream_t s);
// returns true when the row-streaming kernel ran (only it writes gap_part, the per-strip channel sums for a squeeze-excite gate behind the stage)
bool launch_f32_dw(const float* x, float* y, int B, int H, int W, int C, int sh, int sw, int act, int OH, int OW,
                   int pt, int pl, const float* w, const float* bias, float* gap_part, hipStream_t s);
void launch_f32_pw(const float* x, const float* res, const float* gate, float* y, int B, int P, int Cin, int Cout,
                   int act, const float* w, const float* bias, hipStream_t s);
void launch_f32_segate(const float* x, float* gate, int B, int P, int C, int Cr, const float* w1, const float* w2,
                       const float* part /* [B][R][C] partial channel sums instead of x, or null */, int R, hipStream_t s);
void launch_f32_scale(const float* x, const float* gate, float* y, int B, int P, int C, hipStream_t s);
void launch_f32_gap(const float* x, float* y, int B, int P, int C, hipStream_t s);
void launch_f32_dense(const float* x, float* scores, float* logits, int B, int Cin, int Cout, int act, const float* w,
                      const float* bias, hipStream_t s);
void launch_f32_attnpool(const float* x, float* y, int B, int P, int C, const float* score, hipStream_t s);

// Fused depthwise 3x3 -> pointwise 1x1 block (bn_f32_fused.hip); has_dw = 0 gives a plain 1x1 convolution.
struct DwPwArgs {
    const float* x;      // [B][H][W][Cin]
    const float* res;    // [B][OH][OW][Cout] or null
    const float* gate;   // [B][Cin] or null (squeeze-excite gate; only without the depthwise stage)
    float* y;            // [B][OH][OW][Cout]
    const float* dw_w;   // [3][3][Cin]
    const float* dw_b;   // [Cin]
    const float* pw_w;   // fragment order [Cin/16][Cout/16][64][4]
    const float* pw_b;   // [Cout]
    int B, H, W, Cin, Cout, sh, sw, pt, pl, OH, OW, TH, TW, NB, dw_act, pw_act, has_dw;
};
bool f32_dwpw_supported(int Cin, int Cout);
// stem 3x3 (1 -> C channels, stride 1x2) + depthwise 3x3 stride 2 + pointwise C -> N in one kernel (bn_f32_fused.hip)
bool f32_front_supported(int H0, int W0, int C, int N, int OH, int OW);
void launch_f32_front(const float* fe, float* y, int B, int H0, int W0, int C, int N, int OH, int OW, int stem_act,
                      int dw_act, int pw_act, const float* stem_w, const float* stem_b, const float* dw_w, const float* dw_b,
                      const float* pw_w, const float* pw_b, const float* minmax, const float* wsum, const float* magp, int mag,
                      hipStream_t s);
void launch_f32_gap_dense(const float* x, float* scores, float* logits, int B, int P, int Cin, int Cout, int act, const float* w,
                          const float* bias, hipStream_t s);
void launch_f32_dwpw(const DwPwArgs& a, hipStream_t s);
// persistent two-role kernel for plain 1x1 convolutions with Cin > 128 (bn_f32_pw.hip); false: not its shape
bool launch_f32_pw_ws(const DwPwArgs& a, hipStream_t s);
// row-streaming strip kernel for the wide early blocks (bn_f32_strip.hip); launch_f32_dwpw picks it when supported
struct F32FrontStripArgs {
    const float* fe;      // [B][H0][W0]: frontend map, or raw mel energies when minmax != null
    float* y;             // [B][OH][OW][32]
    const float* stem_w; const float* stem_b;  // [3][3][16], [16]
    const float* dw_w; const float* dw_b;      // [3][3][16], [16]
    const float* pw_w; const float* pw_b;      // fragment order [1][2][64][4], [32]
    const float* minmax; const float* wsum; const float* magp;  // finalising mode (see f32_front_kernel)
    int B, H0, W0, OH, OW, TH, stem_act, dw_act, pw_act, mag;
};
bool f32_front_strip_supported(int H0, int W0, int C, int N, int OH, int OW);
void launch_f32_front_strip(F32FrontStripArgs a, hipStream_t s);
// stand-alone depthwise 3x3 as a row-streaming kernel (bn_f32_strip.hip); false = shape not taken, use launch_f32_dw's own kernel
int f32_dw_stream_strips(int B, int C, int OH, int OW);  // partial sums per chunk and channel that launch_f32_dw_stream writes into gap_part
bool launch_f32_dw_stream(const float* x, float* y, int B, int H, int W, int C, int sh, int sw, int act, int OH, int OW, int pt, int pl,
                          const float* w, const float* bias, float* gap_part /* [B][f32_dw_stream_strips][C] or null */, hipStream_t s);
bool f32_strip_supported(const DwPwArgs& a);
void launch_f32_strip(DwPwArgs a, hipStream_t s);
// front block + the residual block behind it (32 -> 32, stride 1) in one kernel, the map between them in LDS (bn_f32_strip.hip)
bool f32_pwdw_supported(const DwPwArgs& expand, int dH, int dW, int dC, int dsh, int dsw, int dOH, int dOW);
struct F32StemIn {  // a stem convolution (3x3, one input channel) computed inside f32_pwdw_kernel instead of being read from memory
    const float* fe;  // [B][H0][W0]
    const float* w;   // [3][3][C]
    const float* b;   // [C]
    int H0, W0, sh, sw, pt, pl, act;
};
bool launch_f32_pwdw(const DwPwArgs& expand, const float* dw_w, const float* dw_b, float* y, int dsh, int dOH, int dOW, int dpt, int dpl, int dw_act,
                     const F32StemIn* stem, float* gap_part /* [B][ceil(dOH / f32_pwdw_rows(dOH))][hid] or null */, hipStream_t s);
int f32_pwdw_rows(int dOH);
bool f32_front2_supported(const F32FrontStripArgs& f, const DwPwArgs& d);
bool launch_f32_front2(const F32FrontStripArgs& f, const DwPwArgs& d, hipStream_t s);

// ---- INT8 plan -----------------------------------------------------------------------------
void launch_i8_quant(const float* spec, const float* minmax, int8_t* out, int B, int F, int W, int Kp, int zp,
                     int fill, float scale, hipStream_t s);
void launch_i8_mel(const int8_t* x, int8_t* y, int B, int W, int Kp, int M, int zp_out, int amin, int amax,
                   const int8_t* w, const int32_t* bias, const int32_t* mult, const int32_t* shift, const int8_t* lut,
                   hipStream_t s);
struct I8ConvGeom {
    int H, W, C, sh, sw, OH, OW, pt, pl, zp_in, zp_out, amin, amax;
    int rq_right = 0;  // every multiplier >= 0 and every shift < 0 (set at load): the row-streaming kernels take the branch-free requantisation
};
void launch_i8_stem(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias,
                    const int32_t* mult, const int32_t* shift, hipStream_t s);
void launch_i8_dw(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias,
                  const int32_t* mult, const int32_t* shift, hipStream_t s);
// the same as a row-streaming kernel (bn_i8_strip.hip); false = shape not taken, use launch_i8_dw
bool launch_i8_dw_stream(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias, const int32_t* mult,
                         const int32_t* shift, hipStream_t s, int32_t* pool = nullptr);  // pool: [B][C] int32, += the sum of every output byte (zeroed by the caller)
bool launch_i8_stem_stream(const int8_t* x, int8_t* y, int B, const I8ConvGeom& g, const int8_t* w, const int32_t* bias, const int32_t* mult,
                           const int32_t* shift, hipStream_t s);  // the single-channel 3x3 stem in the same form
struct I8AddParams {
    int enabled, z1, m1, s1, m2, s2, mo, so, zo, amin, amax;
};
void launch_i8_pw(const int8_t* x, const int8_t* res, int8_t* y, int B, int P, int Cin, int Cout, int zp_out, int amin,
                  int amax, const I8AddParams& add, const int8_t* w, const int32_t* bias, const int32_t* mult,
                  const int32_t* shift, hipStream_t s);
void launch_i8_mean(const int8_t* x, int8_t* y, int B, int P, int C, int zp_in, int mult, int shift, int zp_out,
                    hipStream_t s);
// w: [Cout][Cin rounded up to a multiple of 4], zero padded; lut: optional int8 table applied to the result (LOGISTIC behind the layer)
void launch_i8_fc(const int8_t* x, int8_t* y, int B, int Cin, int Cout, int zp_out, int amin, int amax, const int8_t* w,
                  const int32_t* bias, const int32_t* mult, const int32_t* shift, const int8_t* lut, hipStream_t s);
// MEAN -> FULLY_CONNECTED -> FULLY_CONNECTED (optional tables behind each) of a squeeze-excite gate as one kernel per chunk (C % 4 == 0)
void launch_i8_segate(const int8_t* x, int8_t* y, int B, int P, int C, int zp_in, int mean_mult, int mean_shift, int mean_zp, int R, int zo1, int amin1,
                      int amax1, const int8_t* w1, const int32_t* b1, const int32_t* m1, const int32_t* s1, const int8_t* lut1, int zo2, int amin2,
                      int amax2, const int8_t* w2, const int32_t* b2, const int32_t* m2, const int32_t* s2, const int8_t* lut2, hipStream_t s, int32_t* sums = nullptr);  // sums: [B][C] channel sums of x already taken (read, then zeroed for the next user); null: the kernel pools x itself
void launch_i8_scale(const int8_t* x, const int8_t* gate, int8_t* y, int B, int P, int C, int zx, int zg, int mult, int shift, int zo,
                     int amin, int amax, hipStream_t s);
// per-chunk max -> denominator byte -> DIV table row [-> per-channel table]: [C][W] int8 -> [C][W] int8 (W a multiple of 4)
void launch_i8_rawfe(const float* x, int8_t* y, int B, int T, int W, int M, int stride, int pad_left, float q_scale, int q_zp, int zp_out, int amin,
                     int amax, const int8_t* w, const int32_t* bias, const int32_t* mult, const int32_t* shift, const int8_t* lut, hipStream_t s);
void launch_i8_maxnorm(const int8_t* x, int8_t* y, int B, int C, int W, const int8_t* den_tab, const int8_t* div_tab, const int8_t* lut, hipStream_t s);
// attention pooling of an exported graph: p = the operator's parameters (P C fc_bias fc_mult fc_shift fc_zo form zx za mul_mult mul_shift mul_zo mul_lo
// mul_hi sum_mult sum_shift sum_zo), w the score vector, table the softmax tables (models/_quant.py: softmax_tables); false = geometry not taken
bool launch_i8_attnpool(const int8_t* x, int8_t* y, int B, const int* p, const int8_t* w, const int32_t* table, hipStream_t s);
void launch_i8_head_softmax(const int8_t* x, float* scores, float* logits, int B, int C, int zp_fc, float s_fc, float beta, hipStream_t s);
void launch_i8_head(const int8_t* x, float* scores, float* logits, int B, int C, int zp_fc, int zp_out, float s_fc,
                    float s_out, const int8_t* lut, hipStream_t s);

void launch_debug_requant(const int32_t* x, const int32_t* mult, const int32_t* shift, int n, int mode, int zp, int32_t* out, hipStream_t s);

// Fused INT8 depthwise 3x3 -> pointwise 1x1 block on the int8 matrix cores (bn_i8_fused.hip).
// has_dw = 0: plain 1x1 convolution; transposed = 1: output [chunk][n][position] with an optional per-channel table
// (the frontend's mel mixer + PWL).
struct DwPw8Args {
    const int8_t* x;
    const int8_t* res;
    int8_t* y;
    const int8_t* dw_w;       // [3][3][Cin]
    const int32_t* dw_b;      // [Cin], holds bias - zp_in * sum of the nine weights
    const int32_t* dw_mult;
    const int32_t* dw_shift;
    const int8_t* pw_w;       // fragment order [Kp/64][Cout/16][64 lanes][16]
    const int32_t* pw_b;      // [Cout], zero point folded
    const int32_t* pw_mult;
    const int32_t* pw_shift;
    const int8_t* lut;        // [Cout][256] or null
    int B, H, W, Cin, Cout, sh, sw, pt, pl, OH, OW, TH, TW, NB, has_dw, transposed;
    int dw_zp_in, dw_zp_out, dw_amin, dw_amax, pw_zp_out, pw_amin, pw_amax;
    I8AddParams add;
    int rq_right;  // every multiplier >= 0 and every shift < 0 in this operator (set at load): branch-free requantisation
    // mel mixer only: QUANTIZE fused into the load — x is unused, the input is the float32 spectrogram [B][qF][W]
    const float* qx;       // null = int8 input in x
    const float* qminmax;  // [B][2] per-chunk min / max for the (S - min) / (max - min + 1e-10) normalisation, or null
    float qscale;
    int qzp, qfill, qF;
    int qtiled;            // the spectrogram is tile-major [W/16][qF][16] (written by launch_stft512(..., tile_major))
    // exactness pass of the audio path (bn_stft_exact.hip): qmode 1 lists the elements whose byte may differ from the reference's,
    // qmode 2 runs only the (chunk, 64-frame block) pairs of qguard.work
    int qmode;
    StftGuard qguard;
    // plain 1x1 convolution behind a squeeze-excite MUL (i8_pw_wave_kernel only): x is the UNSCALED map, the gate is applied on load
    const int8_t* gate;    // [B][Cin] or null
    int g_zx, g_zg, g_mult, g_shift, g_zo, g_amin, g_amax;
    // plain 1x1 convolution + ADD: the whole ADD as a table [256 residual byte patterns][own value + 128] (packer: add_table) or null
    const int8_t* add_tab;
};
bool i8_dwpw_supported(int Cin, int Cout);
bool i8_pwdw_supported(const DwPw8Args& expand, const I8ConvGeom& dw);
bool launch_i8_pwdw(const DwPw8Args& expand, const I8ConvGeom& dw, const int8_t* dw_w, const int32_t* dw_b, const int32_t* dw_mult, const int32_t* dw_shift,
                    int8_t* y, hipStream_t s);
bool i8_pw_lds_supported(const DwPw8Args& a);   // bn_i8_pw.hip
void launch_i8_pw_lds(const DwPw8Args& a, hipStream_t s);
bool i8_pw_wave_takes(const DwPw8Args& a);  // the wave-level 1x1 convolution kernel (the only one that applies DwPw8Args::gate) runs this operator
bool i8_mel_mfma_supported(const DwPw8Args& a);
// Wave-autonomous strip kernel for the same block at Cin, Cout in {32, 64} (bn_i8_strip.hip); `cst` is the constant block
// the packer prepares (models/_lower_i8.py: strip_constants).  With the ADD the residual must be the block input x.
struct Strip8Args {
    const int8_t* x;
    int8_t* y;
    const int32_t* cst;
    int B, H, W, OH, OW, TH, pt, pl;
    int zp_in, dw_lo, dw_hi;
    int pw_lo, pw_hi, pw_zp_out;  // with the ADD: clamp bounds + 128 (the block's own value is kept as a table index)
    I8AddParams add;
    const int8_t* add_tab;  // [256][256]: (residual byte pattern, own value + 128) -> output of the TFLite ADD (packer: add_table)
};
struct FrontStrip8Args {
    const int8_t* fe;   // [B][H0][W0]
    int8_t* y;          // [B][OH][OW][32]
    const int32_t* cst; // models/_lower_i8.py: front_strip_constants
    int B, H0, W0, OH, OW, TH;
    int zp_fe, st_lo, st_hi, zp_st, dw_lo, dw_hi, pw_lo, pw_hi;
};
bool i8_front_strip_supported(int H0, int W0, int C, int N, int OH, int OW);
void launch_i8_front_strip(FrontStrip8Args a, hipStream_t s);
bool i8_strip_supported(int Cin, int Cout, int stride, int OW, bool add);
void launch_i8_strip(Strip8Args a, int Cin, int Cout, int stride, hipStream_t s);
// The back half of the INT8 graph as one kernel with the maps in LDS (bn_i8_tail.hip).
constexpr int kTailG = 4;          // chunks a workgroup holds at a time (models/_lower_i8.py: TAIL_G)
constexpr int kTailWaves = 16;
constexpr int kTailThreads = 64 * kTailWaves;
struct Tail8Layer {
    int H, W, Cin, Cout, S, OH, OW, pt, pl, has_add;
    int zp_in, dw_lo, dw_hi, pw_lo, pw_hi;        // clamp bounds (pointwise: + 128 with the ADD)
    int add_m, add_c1, add_e, add_lo, add_hi;     // output rescale of the ADD (zero point folded into c1) and its clamp
    int g_w, g_dwc, g_pwc, g_lut;                 // word offsets of the block's sections in the constant block
    int x_off, y_off, w_off, dwc_off, pwc_off, lut_off, zp_off;  // LDS byte offsets (x_off < 0: the input map is in global memory)
};
struct Tail8Args {
    const int8_t* x;      // [B][H0][W0][C0]: input map of the first block
    float* scores;        // [B][NC]
    float* logits;        // [B][NC] or null
    const int32_t* cst;   // constant block (models/_lower_i8.py: tail_constants)
    int B, n_layers, NC, P, C;
    int mean_zp_in, mean_mult, mean_shift, mean_zp_out, mean_off;
    int fc_zp_out, fc_lo, fc_hi, g_fcw, g_fcb, g_fcm, g_fcs, g_hlut, head_zp_fc, head_zp_out;
    float s_fc, s_head;
    int lds_bytes;
    int fcw_off;          // LDS copy of the classifier matrix for the head, rows of C / 4 + 1 dwords (-1: read it from memory); overlays the last block's weights
    Tail8Layer L[8];
};
bool tail_plan(const int32_t* desc, int n_words, int n_layers, Tail8Args& a);  // a.NC must be set; false = not a topology / size the kernel takes
long tail_const_words(const Tail8Args& a);
bool launch_i8_tail(Tail8Args a, hipStream_t s);

// The same back half with the DEPTHWISE stage on the matrix cores too (bn_i8_tail2.hip): 8 waves per workgroup, maps in place.
constexpr int kTail2Waves = 8;
constexpr int kTail2Threads = 64 * kTail2Waves;
constexpr int kTail2LayerWords = 32;   // models/_lower_i8.py: TAIL2_LAYER_WORDS
struct Tail2Layer {
    int H, W, Cin, Cout, S, OH, OW, pt, pl, has_add;
    int zp_in, dw_lo, dw_hi, pw_lo, pw_hi;        // clamp bounds (pointwise with the ADD: minus the block's own zero point)
    int add_m, add_c1, add_e, add_lo, add_hi;     // output rescale of the ADD (zero point folded into c1) and its clamp
    int res_m, res_c_lo, res_c_hi, res_k;         // rescale of the residual byte: (((b + 128) << 24) res_m + res_c) >> 32 >> res_k
    int g_cst;                                    // word offset of the block's constants: depthwise part (A fragments | constants), then pointwise part (A fragments | constants)
    int x_off, y_off, dw_off, pw_off, zp_off;     // LDS byte offsets (x_off < 0: the input map is in global memory)
};
constexpr int kMidG = 2;   // chunks a workgroup of i8_mid2_kernel holds at a time
struct Tail2Args {
    const int8_t* x;
    int8_t* y;            // i8_mid2_kernel: where the last map goes, [B][P][C]
    float* scores;
    float* logits;
    const int32_t* cst;   // constant block (models/_lower_i8.py: tail2_constants)
    int B, n_layers, NC, P, C;
    int mean_zp_in, mean_mult, mean_shift, mean_zp_out, mean_off;
    int fc_zp_out, fc_lo, fc_hi, g_fcw, g_fcb, g_fcm, g_fcs, g_hlut, head_zp_fc, head_zp_out;
    float s_fc, s_head;
    int lds_bytes;
    int fcw_off;
    Tail2Layer L[8];
};
bool tail2_plan(const int32_t* desc, int n_words, int n_layers, Tail2Args& a, bool mid = false);  // a.NC must be set (tail); mid: the stage-2 chain, no head
long tail2_const_words(const Tail2Args& a, bool mid = false);
bool launch_i8_tail2(Tail2Args a, hipStream_t s);
bool launch_i8_mid2(Tail2Args a, hipStream_t s);

// INT8 stem 3x3 + depthwise 3x3 stride 2 + pointwise in one kernel (bn_i8_fused.hip)
struct I8FrontParams {
    const int8_t* stem_w; const int32_t* stem_b; const int32_t* stem_mult; const int32_t* stem_shift;
    const int8_t* dw_w; const int32_t* dw_b; const int32_t* dw_mult; const int32_t* dw_shift;
    const int8_t* pw_w; const int32_t* pw_b; const int32_t* pw_mult; const int32_t* pw_shift;
    int H0, W0, C, N, OH, OW;
    int stem_zp_in, stem_zp_out, stem_amin, stem_amax, dw_zp_out, dw_amin, dw_amax, pw_zp_out, pw_amin, pw_amax;
    int rq_right;
};
bool i8_front_supported(int H0, int W0, int C, int N, int OH, int OW);
void launch_i8_front(const I8FrontParams& q, const int8_t* fe, int8_t* y, int B, hipStream_t s);
void launch_i8_dwpw(const DwPw8Args& a, hipStream_t s);

// One per source file with kernels: load that file's device code object now (bn_preload_kernels)
void preload_f32(); void preload_f32_fused(); void preload_f32_pw(); void preload_f32_strip(); void preload_i8(); void preload_i8_fused();
void preload_i8_pw(); void preload_i8_strip(); void preload_i8_tail(); void preload_i8_tail2(); void preload_ingest(); void preload_melspec();
void preload_stft(); void preload_stft_exact(); void preload_sort();

// bn_sort.hip: descending orders of the score matrix for the ranking metrics (per class [C][N] row indices, flattened [N*C] flat indices)
size_t rank_orders_workspace(int N, int C);
bool launch_rank_orders(const float* d_scores, int N, int C, int* d_cols, int* d_flat, void* d_work, size_t work_bytes, hipStream_t s);

}  // namespace bn
