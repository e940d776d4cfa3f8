// bn_blob.h — layout of the packed model blob handed to bn_model_load().
//
// Written by birdnet_stm32/models/_pack.py (which mirrors every constant below) from a
// .keras archive (float32 plan) or a .tflite flatbuffer (INT8 plan).  All integers are
// little-endian; every tensor payload starts on a 256-byte boundary.
//
//   BlobHeader | SlotRec[n_slots] | TensorRec[n_tensors] | OpRec[n_ops] | payloads
//
// A "slot" is an activation buffer; its size is given per chunk and the library allocates
// max_batch times that.  Operators address slots by id; BN_SLOT_INPUT is the caller's input
// tensor, BN_SLOT_SCORES / BN_SLOT_LOGITS the caller's output tensors.
#pragma once
#include <stdint.h>

#define BN_BLOB_MAGIC "BNHIPM01"
#define BN_BLOB_VERSION 5u

#define BN_SLOT_INPUT (-1)
#define BN_SLOT_SCORES (-2)
#define BN_SLOT_LOGITS (-3)
#define BN_SLOT_AUDIO (-4)   // the caller's waveform tensor (operators of the audio path only)
#define BN_SLOT_NONE (-9)

// OpRec.p[BN_OP_PATH]: which entry point runs the operator.  bn_forward starts from the runner-boundary input
// (spectrogram); bn_infer_audio starts from waveforms and may use operators that never materialise the spectrogram.
#define BN_OP_PATH 39
#define BN_PATH_BOTH 0
#define BN_PATH_INPUT 1   // only when starting from BN_SLOT_INPUT
#define BN_PATH_AUDIO 2   // only when starting from BN_SLOT_AUDIO

struct BlobHeader {
    char magic[8];
    uint32_t version;
    uint32_t dtype;        // BN_DTYPE_*
    uint32_t input_kind;   // BN_INPUT_*
    uint32_t input_elems;  // float32 elements per chunk at the runner boundary
    uint32_t fft_bins;
    uint32_t spec_width;
    uint32_t num_classes;
    uint32_t n_slots;
    uint32_t n_tensors;
    uint32_t n_ops;
    uint32_t slots_off;
    uint32_t tensors_off;
    uint32_t ops_off;
    uint32_t reserved;
};
static_assert(sizeof(BlobHeader) == 64, "BlobHeader must be 64 bytes");

struct SlotRec {
    uint64_t bytes_per_chunk;
};

struct TensorRec {
    uint64_t offset;  // from the start of the blob, multiple of 256
    uint64_t nbytes;
};

// OpRec.p[BN_OP_TAIL_TAG]: BN_TAIL_COVERED = the operator is one of the blocks the plan's fused tail operator (BN_OP_I8_TAIL) also
// covers — it is skipped while the tail kernel runs; BN_TAIL_OP marks the tail operator itself, skipped when the option i8_tail is
// off or its maps do not fit the kernel's LDS plan.  Any other value: the operator always runs.
#define BN_OP_TAIL_TAG 38
#define BN_TAIL_COVERED 0x7A110001
#define BN_TAIL_OP 0x7A110002
// BN_FRONT2_HEAD on a BN_OP_F32_FRONT operator: the operator p[BN_OP_FRONT2_DIST] places further on (tagged BN_FRONT2_COVERED) is the
// residual block 32 -> 32 that reads nothing but this operator's output; both may run as one kernel (option f32_front2), which
// writes only the second operator's output.  The packer sets the tags only when no other operator reads the map between them.
#define BN_FRONT2_HEAD 0x7A110003
#define BN_FRONT2_COVERED 0x7A110004
#define BN_OP_FRONT2_DIST 37
// BN_SCALE_HEAD on a BN_OP_I8_SCALE operator: the next operator (tagged BN_SCALE_COVERED) is a plain 1x1 convolution that is the only
// reader of the scaled map; the library may apply the gate while that convolution loads its input (i8_pw_wave_kernel) and skip the MUL.
#define BN_SCALE_HEAD 0x7A110005
#define BN_SCALE_COVERED 0x7A110006
// BN_SEGATE_HEAD on a BN_OP_I8_MEAN operator: the next two operators (tagged BN_SEGATE_COVERED) are the FULLY_CONNECTED layers of a
// squeeze-excite gate, each the only reader of its predecessor: pooling and both layers may run as one kernel per chunk.
#define BN_SEGATE_HEAD 0x7A110007
#define BN_SEGATE_COVERED 0x7A110008
// BN_PWDW_HEAD on a plain 1x1 BN_OP_F32_DWPW operator (the expand convolution of an inverted-residual block): the next operator
// (BN_OP_F32_DW, tagged BN_PWDW_COVERED) is the only reader of its output and may run inside the same kernel (f32_pwdw_kernel)
#define BN_PWDW_HEAD 0x7A110009
#define BN_PWDW_COVERED 0x7A11000A
// BN_PWDW_STEM on a BN_OP_F32_STEM operator: the next two operators are a BN_PWDW_HEAD / BN_PWDW_COVERED pair that is the only reader of the
// stem map; the fused kernel may compute the stem rows itself (the stem map is never written)
#define BN_PWDW_STEM 0x7A11000B
// the INT8 counterpart: BN_PWDW8_HEAD on a plain 1x1 BN_OP_I8_DWPW operator, BN_PWDW8_COVERED on the BN_OP_I8_DW operator behind it (i8_pwdw_kernel)
#define BN_PWDW8_HEAD 0x7A11000C
#define BN_PWDW8_COVERED 0x7A11000D
// BN_MID_COVERED / BN_MID_OP: the same pair of tags as BN_TAIL_COVERED / BN_TAIL_OP for the fused stage-2 chain (BN_OP_I8_MID, option i8_mid)
#define BN_MID_COVERED 0x7A11000E
#define BN_MID_OP 0x7A11000F

#define BN_OP_NP 40
#define BN_OP_NT 16
#define BN_OP_NF 8

struct OpRec {
    int32_t kind;
    int32_t in0;   // slot id
    int32_t in1;   // slot id (residual / gate) or BN_SLOT_NONE
    int32_t out;   // slot id
    int32_t p[BN_OP_NP];
    int32_t t[BN_OP_NT];  // tensor ids, -1 = absent
    float f[BN_OP_NF];
};
static_assert(sizeof(OpRec) == 16 + 4 * BN_OP_NP + 4 * BN_OP_NT + 4 * BN_OP_NF, "OpRec packing");

// ---------------------------------------------------------------------------------------
// Operator kinds.  Activations are NHWC per chunk (C innermost); P = H*W positions.
// act codes: 0 none, 1 relu, 2 relu6.   mag codes: 0 none, 1 pwl, 2 pcen, 3 db.
// ---------------------------------------------------------------------------------------
enum BnOpKind : int32_t {
    // ---- float32 plan --------------------------------------------------------------
    // spec [F][W] -> mel [M][W]   p: F W M mag norm   t: wvals(f32) bands(i32 [3][M]: start,len,off) magp(f32 [NP][M])
    BN_OP_F32_MEL = 1,
    // in place [M][W]: x/(max+1e-6) then magnitude scaling   p: M W mag   t: - - magp
    BN_OP_F32_MAG = 2,
    // wave [T] -> [M][W]  p: T W M stride pad_left mag   t: fb(f32 [16][M] BN-folded) bias magp
    BN_OP_F32_RAWFE = 3,
    // [H][W] (C=1) -> [OH][OW][Cout]  p: H W Cout sh sw act OH OW pad_top pad_left   t: w[3][3][Cout] bias[Cout]
    BN_OP_F32_STEM = 4,
    // [H][W][C] -> [OH][OW][C]        p: H W C sh sw act OH OW pad_top pad_left      t: w[3][3][C] bias[C]
    BN_OP_F32_DW = 5,
    // [P][Cin] -> [P][Cout]  p: P Cin Cout act has_res has_gate   in1: residual slot, p6: gate slot   t: w[Cin][Cout] bias
    BN_OP_F32_PW = 6,
    // [P][C] -> gate [C]  p: P C Cr   t: w1[C][Cr] w2[Cr][C]
    BN_OP_F32_SEGATE = 7,
    // [P][C] * gate(in1)[C] -> [P][C]  p: P C
    BN_OP_F32_SCALE = 8,
    // [P][C] -> [C]   p: P C
    BN_OP_F32_GAP = 9,
    // [Cin] -> scores [Cout] (+ logits)  p: Cin Cout act(0 linear,1 sigmoid,2 softmax)   t: w[Cin][Cout] bias
    BN_OP_F32_DENSE = 10,
    // [P][C] -> [C]  p: P C   t: score[C]
    BN_OP_F32_ATTNPOOL = 11,
    // fused [depthwise 3x3 ->] pointwise 1x1 on the matrix cores; has_dw = 0: plain 1x1 conv of in0
    // p: H W Cin sh sw dw_act OH OW pad_top pad_left | Cout pw_act has_res has_gate gate_slot has_dw TH TW NB
    // in1: residual slot   t: dw_w[3][3][Cin] dw_b[Cin] pw_w(fragment order [Cin/16][Cout/16][64][4]) pw_b[Cout]
    BN_OP_F32_DWPW = 12,
    // audio [T] -> un-normalised mel energies [M][W] (+ min/max of the magnitudes): STFT with the band-sparse mixer fused
    // p: T(0 = runtime) W M   t: wvals bands
    BN_OP_F32_STFTMEL = 13,
    // un-normalised mel energies -> frontend output [M][W]   p: M W mag norm   t: wsum[M] - magp
    BN_OP_F32_MELFIN = 14,
    // frontend output [H0][W0] -> stem 3x3 s(1,2) -> depthwise 3x3 s2 -> pointwise, one kernel
    // p: H0 W0 C N OH OW stem_act dw_act pw_act raw_mel mag   t: stem_w stem_b dw_w dw_b pw_w(fragment order) pw_b wsum magp
    // raw_mel = 1 (audio path): in0 holds un-normalised mel energies, finalised while the patch is loaded
    BN_OP_F32_FRONT = 15,
    // global average pool + Dense + sigmoid/softmax in one kernel: [P][Cin] -> scores [Cout] (+ logits)
    // p: P Cin Cout act   t: w[Cin][Cout] bias
    BN_OP_F32_GAPDENSE = 16,

    // ---- INT8 plan -----------------------------------------------------------------
    // spec f32 [F][W] -> q int8 [W][Kp]   p: F W Kp zp fill   f: scale
    BN_OP_I8_QUANT = 20,
    // [W][Kp] -> [M][W]  p: W Kp M zp_out act_min act_max has_lut   t: w[M][Kp] bias(zp-folded) mult shift lut[M][256]
    BN_OP_I8_MEL = 21,
    // [H][W] -> [OH][OW][Cout]  p: H W Cout sh sw - OH OW pad_top pad_left zp_in zp_out act_min act_max   t: w[3][3][Cout] bias mult shift
    BN_OP_I8_STEM = 22,
    // [H][W][C] -> [OH][OW][C]  p: as STEM with C                                     t: w[3][3][C] bias mult shift
    BN_OP_I8_DW = 23,
    // [P][Cin] -> [P][Cout]  p: P Cin Cout zp_out act_min act_max has_add | z1 m1 s1 m2 s2 mo so zo amin amax   t: w[Cout][Cin] bias(zp-folded) mult shift
    BN_OP_I8_PW = 24,
    // [P][C] -> [C]  p: P C zp_in mult shift zp_out
    BN_OP_I8_MEAN = 25,
    // [Cin] -> [Cout]  p: Cin Cout zp_out act_min act_max has_lut   t: w[Cout][Cin rounded up to 4, zero padded] bias(zp-folded) mult shift lut[256]
    // (lut: the int8 LOGISTIC behind the layer, squeeze-excite gates)
    BN_OP_I8_FC = 26,
    // [C] int8 -> scores f32 (+ logits f32)  p: C zp_fc zp_out has_lut softmax   f: s_fc s_out beta   t: lut[256]
    // softmax = 1: scores = float32 softmax of the dequantised input (DEQUANTIZE -> SOFTMAX graphs of conversion/export.py)
    BN_OP_I8_HEAD = 27,
    // fused [depthwise 3x3 ->] pointwise 1x1 on the int8 matrix cores (has_dw = 0: plain 1x1; transposed = 1: mel mixer)
    // p: H W Cin sh sw - OH OW pad_top pad_left | dw_zp_in dw_zp_out dw_amin dw_amax | Cout pw_zp_out pw_amin pw_amax
    //    | has_add z1 m1 s1 m2 s2 mo so zo amin amax | has_dw transposed TH TW NB has_lut
    // in1: residual slot   t: dw_w dw_b(zp folded) dw_mult dw_shift pw_w(fragment order) pw_b(zp folded) pw_mult pw_shift lut
    BN_OP_I8_DWPW = 28,
    // frontend output [H0][W0] int8 -> stem 3x3 s(1,2) -> depthwise 3x3 s2 -> pointwise, one kernel
    // p: H0 W0 C N OH OW | stem_zp_in stem_zp_out stem_amin stem_amax | dw_zp_out dw_amin dw_amax | pw_zp_out pw_amin pw_amax
    // t: stem_w stem_b stem_mult stem_shift dw_w dw_b(zp folded) dw_mult dw_shift pw_w(fragment order) pw_b(zp folded) pw_mult pw_shift
    BN_OP_I8_FRONT = 29,
    // the back half of the INT8 graph in one kernel: n_layers blocks [DW 3x3 -> PW 1x1 (-> ADD)] with the maps in LDS, then MEAN,
    // FULLY_CONNECTED and the head (bn_i8_tail.hip)
    // p: in_bytes pw_macs dw_macs other_macs n_classes n_layers H0 W0 C0 P_last C_last   f: s_fc s_head
    // t: constant block (int32 words), descriptor table (24 words per block + 16 head words; models/_lower_i8.py: tail_constants)
    BN_OP_I8_TAIL = 30,
    // int8 MUL of a map with a per-chunk gate vector (squeeze-excite): [P][C] * gate(in1)[C] -> [P][C]
    // p: P C zp_x zp_gate mult shift zp_out act_min act_max
    BN_OP_I8_SCALE = 31,
    // per-chunk max normalisation of an int8 map (REDUCE_MAX over the whole map -> ADD epsilon -> DIV by that scalar) followed by an
    // optional per-channel 256-entry table (the PWL behind it): [C][W] -> [C][W].  Everything after the max is a function of bytes:
    // p: C W has_lut   t: denominator byte per max byte (256), DIV table [256 denominators][256 values] (row/column = byte + 128),
    //    per-channel table [C][256] (has_lut)
    BN_OP_I8_MAXNORM = 32,
    // raw frontend of an exported INT8 graph: QUANTIZE of the float32 waveform [T] -> [PAD] -> CONV_2D 1x16 stride s VALID (ReLU6 clamp)
    // -> optional per-channel table (magnitude scaling) -> [M][W] int8
    // p: T W M stride pad_left q_zp zp_out act_min act_max has_lut   f: q_scale
    // t: weights [M][16] int8, bias (zero point of the input folded), multipliers, shifts, table [M][256] (has_lut)
    BN_OP_I8_RAWFE = 33,
    // the three blocks of stage 2 of the shipped INT8 graph in one kernel (a stride-2 block from memory, two residual blocks in LDS, the last
    // map back to memory): bn_i8_tail2.hip, i8_mid2_kernel.  p: in_bytes pw_macs dw_macs 0 0 n_layers H0 W0 C0 P_last C_last
    // t: constant block (int32 words), descriptor table (32 words per block; models/_lower_i8.py: tail2_constants without head)
    BN_OP_I8_MID = 35,
    BN_OP_I8_ATTNPOOL = 34,  // attention pooling of an exported graph: score FC + int8 SOFTMAX over the positions + MUL + SUM (bn_i8.hip)
};
