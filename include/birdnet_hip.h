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

/* Test hook: number of plan operators' outputs and a copy of one of them.
 * `op_index` in [0, n_ops); the element type/shape is what the packer recorded.
 * Valid until the next forward call. */
BN_API int bn_debug_op_output(bn_model* model, int op_index, int B, void* d_dst, size_t dst_bytes,
                       size_t* bytes_per_chunk, void* stream);

/* Per-operator timing with HIP events recorded on the launch stream.  While enabled, every plan
 * operator of bn_forward / bn_infer_audio is bracketed by an event pair (index n_ops = the STFT
 * stage of bn_infer_audio).  bn_profile_collect waits for the recorded events, adds the elapsed
 * milliseconds and launch counts per operator into total_ms[n] / launches[n] (n >= n_ops + 1)
 * and forgets them. */
BN_API int bn_profile_enable(bn_model* model, int enable);
BN_API int bn_profile_collect(bn_model* model, double* total_ms, int64_t* launches, int n);

/* Names of the HIP kernels a forward pass launches, '\n'-separated (for profiling tools). */
BN_API const char* bn_kernel_names(void);

#ifdef __cplusplus
}
#endif
#endif /* BIRDNET_HIP_H */
