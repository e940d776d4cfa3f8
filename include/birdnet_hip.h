/*
 * birdnet_hip.h — C ABI of libbirdnet_hip.so: the MI355X (gfx950) implementation of the
 * birdnet-stm32 per-chunk inference hot path
 *
 *     3 s audio chunk -> windowed STFT magnitude -> hybrid mel mixer -> PWL/PCEN
 *                     -> DS-CNN (float32 or bit-faithful INT8) -> class scores.
 *
 * This is the drop-in boundary.  Every entry point names the reference interface it
 * replaces (paths relative to the reference repository birdnet-team/birdnet-stm32):
 *
 *   bn_stft_mag      <- birdnet_stm32/audio/spectrogram.py:24-33,61,106-115,133,149
 *   bn_stft_mag_exact   (the same call site, float64 arithmetic like librosa's)
 *                       get_spectrogram_from_audio(audio, n_fft, mel_bins=-1, spec_width)
 *                       as called per chunk by evaluation/metrics.py:55-61
 *   bn_model_load    <- birdnet_stm32/models/runners.py:98-114 load_model_runner(model_path)
 *                       (tf.lite.Interpreter(...)+allocate_tensors / keras load_model)
 *   bn_forward       <- birdnet_stm32/models/runners.py:29-45 KerasRunner.predict and
 *                       :82-95 TFLiteRunner.predict  (x_batch [B,257,W,1] f32 -> [B,C] f32)
 *   bn_infer_audio   <- the two above back to back, i.e. the body of the chunk loop in
 *                       evaluation/metrics.py:55-61 + :129-141, without the host round trip
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Return value 0 = success, negative = error; bn_last_error() gives the message
 *     (thread-local, valid until the next failing call on that thread).
 *   - Every `d_*` pointer is DEVICE memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()).  The library allocates only its own workspace, at bn_model_load time,
 *     sized for the context's max_batch.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued asynchronously on it; the caller synchronises.
 *   - One bn_ctx per device per host thread; a bn_model is not re-entrant (it owns its
 *     activation workspace), like the reference's TFLite interpreter.
 *   - There is no CPU fallback anywhere: without a gfx950 device every compute entry
 *     point fails with BN_ERR_DEVICE.
 */
#ifndef BIRDNET_HIP_H
#define BIRDNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BN_ABI_VERSION 1

#if defined(__GNUC__)
#define BN_API __attribute__((visibility("default")))
#else
#define BN_API
#endif

/* error codes (negative) */
#define BN_OK 0
#define BN_ERR_ARG (-1)      /* bad argument / shape mismatch */
#define BN_ERR_DEVICE (-2)   /* HIP runtime error, or no usable device */
#define BN_ERR_FORMAT (-3)   /* malformed model blob */
#define BN_ERR_UNSUPPORTED (-4)
#define BN_ERR_NOMEM (-5)

typedef struct bn_ctx bn_ctx;
typedef struct bn_model bn_model;

/* arithmetic type a model computes in */
#define BN_DTYPE_F32 0
#define BN_DTYPE_I8 1

/* what bn_forward's d_input holds */
#define BN_INPUT_SPECTROGRAM 0 /* [B, F, W] float32 linear STFT magnitude (hybrid frontend) */
#define BN_INPUT_WAVEFORM 1    /* [B, T] float32 (raw frontend) */
#define BN_INPUT_MEL 2         /* [B, M, W] float32 precomputed (mel / log-mel / MFCC) spectrogram, passed through */

typedef struct bn_model_info {
    int32_t dtype;          /* BN_DTYPE_* */
    int32_t input_kind;     /* BN_INPUT_* */
    int32_t input_elems;    /* float32 elements per chunk at the runner boundary (F*W or T) */
    int32_t fft_bins;       /* F (hybrid) or 0 */
    int32_t spec_width;     /* W */
    int32_t num_classes;    /* C */
    int32_t n_ops;          /* device-plan operators */
    int32_t max_batch;      /* batch the workspace was sized for */
    int64_t workspace_bytes;
    int64_t const_bytes;    /* weights resident in HBM */
} bn_model_info;

BN_API int bn_version(void);
BN_API const char* bn_last_error(void);

/* Number of HIP devices visible to the process (0 if none). */
BN_API int bn_device_count(void);

/* Create a context on `device`; workspaces of models loaded through it are sized for
 * `max_batch` chunks per call. */
BN_API int bn_ctx_create(int device, int max_batch, bn_ctx** out);
BN_API void bn_ctx_destroy(bn_ctx* ctx);

/* Parse a packed model blob (produced by birdnet_stm32.models._pack from a .keras or
 * .tflite file), copy its constants to HBM and allocate the activation workspace.
 * The blob may be freed by the caller after the call returns. */
BN_API int bn_model_load(bn_ctx* ctx, const void* blob, size_t nbytes, bn_model** out);
/* Validate a packed blob without loading it (host only, needs no device): the checks bn_model_load runs before it allocates —
 * tables and payloads inside the blob, operator references in range, every operator's geometry against the slot and tensor
 * sizes it addresses.  BN_ERR_FORMAT + bn_last_error() on the first violation.  (Reference counterpart: the flatbuffer
 * verification inside tf.lite.Interpreter(model_path=...), birdnet_stm32/models/runners.py:57.) */
BN_API int bn_blob_check(const void* blob, size_t nbytes);
BN_API void bn_model_free(bn_model* model);
BN_API int bn_model_get_info(const bn_model* model, bn_model_info* out);

/* Batched linear-magnitude STFT with the evaluate path's framing (centre zero padding of
 * n_fft/2, periodic Hann, frame t = samples [t*hop - n_fft/2, t*hop + n_fft/2), first W
 * frames kept).
 *   d_audio  [B, T] float32
 *   d_spec   [B, n_fft/2+1, W] float32 (frequency-major, like the reference's ndarray)
 *   d_minmax [B, 2] float32 (per-chunk min, max of the magnitudes) — required
 *   normalize != 0: d_spec <- (S - min) / (max - min + 1e-10) per chunk, in place
 * Only n_fft = 512 is implemented (the reference's firmware FFT has the same limit). */
BN_API int bn_stft_mag(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W,
                int normalize, float* d_spec, float* d_minmax, void* stream);

/* The same spectrogram with the reference's arithmetic, value for value: window product and DFT of every bin in float64, the
 * result rounded to complex64, |.| by numpy's float32 formula (birdnet_stm32/audio/spectrogram.py:106-115: np.abs(librosa.stft(...))),
 * float32 min-max normalisation (:12-21).  About ten times slower than bn_stft_mag (a float32 FFT, within 2e-6 of the peak of
 * these values); bn_infer_audio reaches the same INT8 input bytes at full speed by recomputing only the elements in doubt. */
BN_API int bn_stft_mag_exact(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W,
                      int normalize, float* d_spec, float* d_minmax, void* stream);

/* Forward pass from the runner boundary.
 *   d_input   [B, input_elems] float32 (model_info.input_kind says what it is)
 *   d_minmax  NULL, or [B,2]: treat d_input as UN-normalised magnitudes and apply the
 *             min-max normalisation while loading (saves one pass; same float32 arithmetic)
 *   d_scores  [B, C] float32 class scores (sigmoid/softmax output; dequantised for INT8)
 *   d_logits  NULL or [B, C] float32 pre-activation outputs of the classifier
 */
BN_API int bn_forward(bn_model* model, const float* d_input, const float* d_minmax, int B,
               float* d_scores, float* d_logits, void* stream);

/* audio chunks -> scores: bn_stft_mag (into the model's workspace) + bn_forward. */
BN_API int bn_infer_audio(bn_model* model, const float* d_audio, int B, int T, int hop,
                   float* d_scores, float* d_logits, void* stream);

/* ---- the steps either side of the path ------------------------- */

/* sample formats of bn_ingest_resample's interleaved PCM (libsndfile's float scaling: /2^15, /2^23, /2^31) */
#define BN_PCM_S16 0
#define BN_PCM_S24 1 /* packed 3-byte little endian */
#define BN_PCM_S32 2
#define BN_PCM_F32 3

/* Decode + mono mix + polyphase resampling + absolute peak of a batch of audio windows that share one
 * (format, channel count, rate pair) — the device form of load_audio_window's arithmetic
 * (reference: birdnet_stm32/audio/io.py:112-124: y.mean(axis=1), fast_resample -> scipy.signal.resample_poly,
 * np.max(np.abs(y))).  Operation order is numpy's / scipy's, so the result is bit-identical to theirs.
 *   d_pcm       interleaved frames of all windows back to back
 *   d_in_off    [n_windows+1] frame offsets of the windows in d_pcm
 *   d_out_off   [n_windows+1] sample offsets of the resampled windows in d_mono;
 *               d_out_off[i+1]-d_out_off[i] = ceil(n_in * up / down)
 *   d_taps      [up][taps_per_phase] polyphase filter, phase-major, the coefficient for the OLDEST input sample
 *               first (scipy upfirdn's transposed, flipped layout of the zero-padded resample_poly filter);
 *               taps_per_phase = 0 with up == down: no resampling (same rate)
 *   n_pre_remove  leading filter-delay outputs resample_poly drops
 *   d_mono      resampled mono float32, NOT yet peak-normalised
 *   d_peak      [n_windows] max |y| per window
 * Channels 1..8.  Window length * up and output length * down must stay below 2^32. */
BN_API int bn_ingest_resample(bn_ctx* ctx, const void* d_pcm, int sample_format, int channels,
                       const int64_t* d_in_off, const int64_t* d_out_off, int n_windows, int64_t max_in_len,
                       int64_t max_out_len, const float* d_taps, int up, int down, int taps_per_phase,
                       int n_pre_remove, float* d_mono, float* d_peak, void* stream);

/* Fixed-length chunks of peak-normalised audio (reference: audio/io.py:122-124 `y / peak` when peak > 0, then
 * split_audio_into_chunks :133-174; the host computes the start positions, which depend only on lengths).
 *   d_chunk_src    [n_chunks] absolute sample offset of the chunk's first sample in d_mono
 *   d_chunk_valid  [n_chunks] samples to copy (< chunk_len only for a window shorter than one chunk: right zero pad)
 *   d_chunk_window [n_chunks] index into d_peak
 *   d_chunks       [n_chunks, chunk_len] float32 — the [B, T] input of bn_infer_audio / bn_stft_mag */
BN_API int bn_ingest_chunks(bn_ctx* ctx, const float* d_mono, const float* d_peak, const int64_t* d_chunk_src,
                     const int32_t* d_chunk_valid, const int32_t* d_chunk_window, int n_chunks, int chunk_len,
                     float* d_chunks, void* stream);

/* The sorts behind the ranking metrics (reference: sklearn roc_auc_score / average_precision_score, birdnet_stm32/evaluation/metrics.py:155-190,
 * each of which argsorts on the host): descending, stable orders of the [n_rows, n_classes] float32 score matrix, on the device it lives on.
 *   d_cols [n_classes, n_rows] int32 — for class c the row indices by descending score of column c
 *   d_flat [n_rows * n_classes] int32 — flat indices (row * n_classes + class) by descending score
 * Scores must be finite (the caller checks, as the metrics do).  The workspace lives in the context and grows on demand. */
BN_API int bn_rank_orders(bn_ctx* ctx, const float* d_scores, int n_rows, int n_classes, int32_t* d_cols, int32_t* d_flat, void* stream);

/* Per-chunk peak normalisation y = x / (max|x| + eps) of the raw frontend's model input (reference:
 * birdnet_stm32/evaluation/metrics.py:62-69, with eps = 1e-6): d_x, d_y [B, T] float32 (may alias). */
BN_API int bn_chunk_peak_normalize(bn_ctx* ctx, const float* d_x, int B, int T, float eps, float* d_y, void* stream);

/* pooling methods of bn_pool_scores (reference names: 'avg'|'mean'|'average', 'max', 'lme'|'log_mean_exp'|...) */
#define BN_POOL_AVG 0
#define BN_POOL_MAX 1
#define BN_POOL_LME 2

/* File-level pooling of chunk scores (reference: birdnet_stm32/evaluation/pooling.py:6-47 pool_scores /
 * lme_pooling, called once per file by evaluation/metrics.py:143-146), for all files of a batch at once.
 *   d_scores   [n_rows, n_classes] float32, the rows of one file contiguous
 *   d_file_off [n_files+1] row offsets; an empty file pools to zeros
 *   d_pooled   [n_files, n_classes]
 * mean and max are bit-identical to numpy's float32 result; lme = (m + log(mean(exp(beta s - m)) + 1e-12)) / beta. */
BN_API int bn_pool_scores(bn_ctx* ctx, const float* d_scores, const int64_t* d_file_off, int n_files, int n_classes,
                   int method, float beta, float* d_pooled, void* stream);

/* ---- precomputed frontends ---------------------------------------------- */

/* spectrogram modes / magnitude scalings of get_spectrogram_from_audio (reference: audio/spectrogram.py:24-33) */
#define BN_SPEC_MEL 0    /* mode='mel': magnitude mel spectrogram, then mag_scale, then min-max normalise */
#define BN_SPEC_LOGMEL 1 /* mode='log_mel': log1p(magnitude mel), normalise */
#define BN_SPEC_MFCC 2   /* mode='mfcc': power mel -> dB (ref=max, 80 dB floor) -> orthonormal DCT-II, first n_mfcc rows, normalise */
#define BN_MAG_NONE 0
#define BN_MAG_PWL 1
#define BN_MAG_PCEN 2
#define BN_MAG_DB 3

/* Batched get_spectrogram_from_audio(audio, sample_rate, n_fft=512, mel_bins > 0, spec_width, mag_scale, mode, n_mfcc)
 * (reference: birdnet_stm32/audio/spectrogram.py:61-149, as called per chunk by evaluation/metrics.py:49-54 for the
 * 'librosa' frontend and by the data generator for 'log_mel' / 'mfcc'): hop-framed STFT magnitude (same kernel as
 * bn_stft_mag) mixed by the Slaney mel basis while still in LDS, then one finishing pass per chunk.
 *   d_audio     [B, T] float32
 *   d_mel_w / d_mel_bands   band-sparse mel basis (librosa.filters.mel(sr, 512, n_mels, fmin=150, fmax=sr//2)):
 *               d_mel_bands = int32 [3, n_mels] (first bin, band length, offset into d_mel_w), d_mel_w the
 *               non-zero runs back to back — what birdnet_stm32.models._lower_f32.mel_bands() produces
 *   pcen_b      smoothing coefficient of librosa.pcen for (sample_rate, hop): (sqrt(1+4T^2)-1)/(2T^2), T = 0.4 sr / hop
 *               (only read for mag_scale = BN_MAG_PCEN)
 *   d_dct       [n_mfcc, n_mels] orthonormal DCT-II rows (only for mode = BN_SPEC_MFCC), else NULL
 *   d_work      scratch, B * (n_mels * (1 + T / hop) + 2) floats (mfcc takes its dB reference over all frames, like the
 *               reference, and cuts to W afterwards; the other modes use only the first W frames)
 *   d_out       [B, n_mels, W] (mel, log_mel) or [B, n_mfcc, W] (mfcc), values in [0, 1] */
BN_API int bn_mel_spectrogram(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W,
                       const float* d_mel_w, const int32_t* d_mel_bands, int n_mels, int mode, int mag_scale,
                       double pcen_b, const float* d_dct, int n_mfcc, float* d_work, float* d_out, void* stream);

/* Test hook: number of plan operators' outputs and a copy of one of them.
 * `op_index` in [0, n_ops); the element type/shape is what the packer recorded.
 * Valid until the next forward call.  BN_ERR_UNSUPPORTED when the operator's output was not written by that call: a fused kernel
 * kept the map on chip under the current options (front block pairs, expand + depthwise pairs, squeeze-excite gates, the blocks the
 * fused tail covers); bn_set_option switches the fusion off for a per-layer look. */
BN_API int bn_debug_op_output(bn_model* model, int op_index, int B, void* d_dst, size_t dst_bytes,
                       size_t* bytes_per_chunk, void* stream);

/* Test hook: the device's fixed-point requantisation, element-wise on n (accumulator, multiplier, shift) triples — TFLite's
 * MultiplyByQuantizedMultiplier as every INT8 kernel here computes it (csrc/bn_requant.h).  mode 0: the form the generic kernels
 * call; 1: the literal gemmlowp definitions (SaturatingRoundingDoublingHighMul + RoundingDivideByPOT); 2: the branch-free
 * right-shift form (needs multiplier >= 0, shift < 0); 3: the strip kernels' form with rounding offset and zero_point folded into
 * one addend (same preconditions, shift >= -22), zero point subtracted again.  (Reference: the int8 kernels inside
 * tf.lite.Interpreter.invoke, birdnet_stm32/models/runners.py:93.) */
/* Test hook: the int8 bytes the graph's QUANTIZE (op #0) made of the spectrograms of the last bn_infer_audio call on an INT8 plan,
 * d_out [B, 257, W] frequency-major (the production plan never stores them: QUANTIZE is fused into the mel mixer's load).  It is a VIEW:
 * the bytes are recomputed from the spectrogram the call left behind with the graph's exact QUANTIZE chain, not read back from the mixer's
 * tile — tests that check the bytes check the scores (or the pre-sigmoid bytes) as well, which is what the network consumed. */
BN_API int bn_debug_input_bytes(bn_model* model, int B, int8_t* d_out, void* stream);
/* Test hook (synchronises the device): counters of the exactness pass of the last bn_infer_audio call on an INT8 plan (first launch group) —
 * out[0] elements listed as in doubt (sum over the B chunks), out[1] the largest count of one chunk, out[2] (chunk, 64-frame block) pairs
 * whose bytes changed, out[3] / out[4] chunks recomputed as whole float64 spectrograms behind the min / max pass and behind the fix pass,
 * out[5] / out[6] (option stft_audit = 1) elements the audit re-evaluated although the bound did not put them in doubt — the near misses within
 * four bounds of a rounding boundary — and how many of them had a kept byte different from the exact one (violations of the bound: must be 0),
 * out[7] chunks whose minimum the min / max pass enclosed in an interval instead of settling it (option stft_minint; noise-free and flat spectra).
 * `out` holds 8 values. */
BN_API int bn_debug_guard_stats(bn_model* model, int B, int64_t* out);
/* Test hook: which form of the fused INT8 tail operator (BN_OP_I8_TAIL; reference operators #36-#55 of the shipped graph) this model's plan can
 * run — *form = 0 none (per-block operators), 1 = i8_tail_kernel only, 2 = also i8_tail2_kernel (depthwise stage on the matrix cores, the
 * default where available; option i8_tail_mfdw); *lds_bytes = the LDS that form's plan asks for. */
BN_API int bn_debug_tail_form(const bn_model* model, int* form, int* lds_bytes);
/* Test hook: *form = 1 when the plan's fused stage-2 chain (BN_OP_I8_MID: three blocks of the shipped graph as i8_mid2_kernel; option i8_mid)
 * passed the library's LDS plan and runs by default, else 0 (its three strip kernels run instead). */
BN_API int bn_debug_mid_form(const bn_model* model, int* form, int* lds_bytes);

BN_API int bn_debug_requant(bn_ctx* ctx, const int32_t* d_x, const int32_t* d_mult, const int32_t* d_shift, int n, int mode,
                     int zero_point, int32_t* d_out, void* stream);

/* Per-operator timing with HIP events recorded on the launch stream.  While enabled, every plan
 * operator of bn_forward / bn_infer_audio is bracketed by an event pair (index n_ops = the STFT
 * stage of bn_infer_audio).  bn_profile_collect waits for the recorded events, adds the elapsed
 * milliseconds and launch counts per operator into total_ms[n] / launches[n] (n >= n_ops + 1)
 * and forgets them.  INT8 plans from audio have two more entries, n_ops + 1 (exact min / max of the spectrogram) and n_ops + 2 (whole-chunk
 * float64 fallback + second run of the mel mixer): with n >= n_ops + 3 they are reported on their own, else under the STFT stage. */
BN_API int bn_profile_enable(bn_model* model, int enable);
/* Restrict the event pairs to ONE operator (op_index in [0, n_ops]; -1 = every operator again).  A pair per operator costs
 * ~6 % of a 1.5 ms step; bracketing only the kernel under study keeps the timed region undisturbed. */
BN_API int bn_profile_only(bn_model* model, int op_index);
BN_API int bn_profile_collect(bn_model* model, double* total_ms, int64_t* launches, int n);

/* Run-time switches of the kernel launchers, for A/B measurements and tests (process-wide; the defaults are the production
 * choices).  Names: "f32_strip", "f32_strip_th", "f32_front_staged", "f32_front2", "f32_pwdw", "f32_tile_slice", "f32_pw_ws", "i8_pwdw", "i8_pw_lds", "i8_pw_forms", "i8_add_tab", "front_tpw", "wave_dwpw", "i8_strip", "i8_strip_mfdw", "i8_strip_th",
 * "i8_dw_pool", "i8_tail_fclds", "i8_tail", "i8_tail_mfdw", "i8_mid", "i8_mel_generic", "stft_rowmajor", "stft_exact", "stft_flagcap", "stft_guard", "stft_audit", "stft_minint", "ingest_blk", "ingest_generic" (csrc/bn_kernels.h: Options says
 * what each selects).  An environment variable BN_<NAME IN CAPITALS> seeds the value once when the library is loaded; no
 * launch reads the environment.  The reference has no counterpart (tf.lite.Interpreter's delegates / num_threads arguments,
 * birdnet_stm32/models/runners.py:57, are the closest thing).  Unknown name: BN_ERR_ARG. */
BN_API int bn_set_option(const char* name, int value);
BN_API int bn_get_option(const char* name, int* value);
/* The same switches per context: bn_set_option sets the PROCESS DEFAULT, bn_ctx_set_option overrides one switch for the launches made through
 * this context (and the models loaded into it) only, so two models in one process can run under different options; bn_ctx_get_option reads the
 * effective value, bn_ctx_reset_options drops the context's overrides.  (The reference's counterpart is per-interpreter state: every
 * tf.lite.Interpreter of models/runners.py:57 carries its own settings.) */
BN_API int bn_ctx_set_option(bn_ctx* ctx, const char* name, int value);
BN_API int bn_ctx_get_option(bn_ctx* ctx, const char* name, int* value);
BN_API int bn_ctx_reset_options(bn_ctx* ctx);

/* Page-locked host memory for the staging buffers a caller copies audio from (hipHostMalloc / hipHostFree).  Through a foreign-function binding
 * the call runs WITHOUT the host language's interpreter lock: page-locking 256 MiB takes ~17 ms, and an allocation made through PyTorch's
 * pinned allocator holds the GIL for all of it — the reader thread of the evaluate pipeline stood still meanwhile (tools/_cold_trace.py).
 * NULL on failure (bn_last_error).  (Reference counterpart: none — it hands numpy arrays to tf.lite.Interpreter.set_tensor, runners.py:84.) */
BN_API void* bn_host_alloc_pinned(bn_ctx* ctx, size_t bytes);
BN_API int bn_host_free_pinned(void* p);

/* Loads every device code object of the library now.  The HIP runtime loads one at the first launch of any of its kernels (a few ms each; launches
 * and copies of OTHER threads wait meanwhile), which a first batch otherwise pays one file after the other on its critical path; a caller with idle
 * time before that batch (the evaluate pipeline while the first files are being read) calls this instead.  Idempotent.  (Reference counterpart:
 * tf.lite.Interpreter.allocate_tensors(), models/runners.py:58 — the one-off preparation before the first invoke.) */
BN_API int bn_preload_kernels(bn_ctx* ctx);

/* Names of the HIP kernels a forward pass launches, '\n'-separated (for profiling tools). */
BN_API const char* bn_kernel_names(void);

#ifdef __cplusplus
}
#endif
#endif /* BIRDNET_HIP_H */
