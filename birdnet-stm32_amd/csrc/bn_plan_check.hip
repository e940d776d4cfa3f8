// bn_plan_check.hip — load-time validation of a packed device plan (host code only).
//
// bn_model_load() hands every operator record of the blob to the kernel launchers unchanged, and the kernels trust the
// geometry in OpRec.p[] for all their addressing.  A truncated, stale or hostile blob must therefore be refused HERE, with
// BN_ERR_FORMAT, instead of becoming an out-of-bounds device access: per operator kind this pass derives the bytes per
// chunk the operator reads from and writes to each slot and the bytes it reads from each constant tensor, and compares
// them with SlotRec.bytes_per_chunk / TensorRec.nbytes.  Slot ids stored in p[] (squeeze-excite gates) are checked like
// in0 / in1 / out.  The reference's counterpart is the flatbuffer verifier inside tf.lite.Interpreter
// (reference: birdnet_stm32/models/runners.py:57).
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/birdnet_hip.h"
#include "bn_blob.h"
#include "bn_kernels.h"

namespace bn {

namespace {

struct Checker {
    const BlobHeader& h;
    const std::vector<SlotRec>& slots;
    const std::vector<TensorRec>& tensors;
    std::string& err;
    size_t oi = 0;
    const OpRec* o = nullptr;
    bool ok = true;

    bool bad(const char* fmt, ...) {
        if (!ok) return false;  // keep the first message
        char buf[384];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = "operator " + std::to_string(oi) + " (kind " + std::to_string(o->kind) + "): " + buf;
        ok = false;
        return false;
    }

    // every dimension positive and below 2^24; byte extents are bounded by slot() (the kernels index with 32-bit integers)
    bool dims(std::initializer_list<int> v, const char* what) {
        for (int x : v)
            if (x <= 0 || x >= (1 << 24)) return bad("%s: dimension %d out of range", what, x);
        return true;
    }

    // slot `id` must be able to hold `need` bytes per chunk
    bool slot(int id, long long need, const char* what) {
        if (!ok) return false;
        if (need < 0 || need >= (1LL << 31)) return bad("%s: %lld bytes per chunk overflow 32-bit addressing", what, need);
        long long cap;
        if (id == BN_SLOT_INPUT) cap = (long long)h.input_elems * 4;
        else if (id == BN_SLOT_SCORES || id == BN_SLOT_LOGITS) cap = (long long)h.num_classes * 4;
        else if (id == BN_SLOT_AUDIO) return true;  // the caller's waveform: its length arrives with the call
        else if (id < 0 || id >= (int)slots.size()) return bad("%s: slot id %d is not a slot of this plan (%zu slots)", what, id, slots.size());
        else if (slots[id].bytes_per_chunk == 0) return bad("%s: slot %d has no storage", what, id);
        else cap = (long long)slots[id].bytes_per_chunk;
        if (need > cap) return bad("%s: needs %lld bytes per chunk, slot %d holds %lld", what, need, id, cap);
        return true;
    }

    // constant tensor t[k] must hold at least `need` bytes
    bool tensor(int k, long long need, const char* what) {
        if (!ok) return false;
        const int id = o->t[k];
        if (id < 0 || id >= (int)tensors.size()) return bad("%s: tensor t[%d] = %d is absent", what, k, id);
        if ((long long)tensors[id].nbytes < need) return bad("%s: tensor t[%d] holds %llu bytes, the operator reads %lld", what, k, (unsigned long long)tensors[id].nbytes, need);
        return true;
    }

    // clamp bounds of a value that is used as a table index (value + 128 into 256 entries): must be int8 bounds
    bool clamp8(int lo, int hi, const char* what) {
        if (lo < -128 || hi > 127 || lo > hi) return bad("%s: clamp [%d, %d] is not an int8 range (the clamped value indexes a 256-entry table)", what, lo, hi);
        return true;
    }

    bool conv_geom(int H, int W, int sh, int sw, int OH, int OW, int pt, int pl) {
        if (sh < 1 || sw < 1 || sh > 2 || sw > 2) return bad("stride %dx%d (1 or 2 expected)", sh, sw);
        if (OH != (H + sh - 1) / sh || OW != (W + sw - 1) / sw) return bad("output %dx%d does not follow from input %dx%d at stride %dx%d (SAME)", OH, OW, H, W, sh, sw);
        if (pt < 0 || pt > 1 || pl < 0 || pl > 1) return bad("padding %d/%d outside a 3x3 window", pt, pl);
        return true;
    }
};

long long up(long long v, long long m) { return (v + m - 1) / m * m; }

}  // namespace

int i8_strip_waves(int Cin, int Cout, int stride, int OW, bool add);  // bn_i8_strip.hip

bool check_plan(const BlobHeader& h, const std::vector<SlotRec>& slots, const std::vector<TensorRec>& tensors,
                const std::vector<OpRec>& ops, std::string& err) {
    Checker c{h, slots, tensors, err};
    if (h.num_classes == 0 || h.num_classes > (1u << 20) || h.input_elems == 0 || h.input_elems >= (1u << 29)) {
        err = "header: implausible num_classes / input_elems";
        return false;
    }
    for (size_t oi = 0; oi < ops.size() && c.ok; ++oi) {
        const OpRec& o = ops[oi];
        const int* p = o.p;
        c.oi = oi;
        c.o = &o;
        if (p[BN_OP_PATH] < BN_PATH_BOTH || p[BN_OP_PATH] > BN_PATH_AUDIO) c.bad("path tag %d", p[BN_OP_PATH]);
        switch (o.kind) {
            case BN_OP_F32_MEL:  // F W M mag norm
                c.dims({p[0], p[1], p[2]}, "mel") && c.slot(o.in0, 4LL * p[0] * p[1], "input") && c.slot(o.out, 4LL * p[2] * p[1], "output") &&
                    c.tensor(0, 4, "band weights") && c.tensor(1, 12LL * p[2], "band table") && (p[3] == 0 || c.tensor(2, 4LL * p[2], "magnitude parameters"));
                break;
            case BN_OP_F32_MAG:  // M W mag
                c.dims({p[0], p[1]}, "mag") && c.slot(o.out, 4LL * p[0] * p[1], "map") && (p[2] == 0 || c.tensor(2, 4LL * p[0], "magnitude parameters"));
                break;
            case BN_OP_F32_RAWFE:  // T W M stride pad_left mag
                c.dims({p[0], p[1], p[2], p[3]}, "raw frontend") && c.slot(o.in0, 4LL * p[0], "waveform") && c.slot(o.out, 4LL * p[2] * p[1], "output") &&
                    c.tensor(0, 64LL * p[2], "filterbank") && c.tensor(1, 4LL * p[2], "bias") && (p[5] == 0 || c.tensor(2, 4LL * p[2], "magnitude parameters"));
                if (c.ok && p[4] < 0) c.bad("negative left padding");
                if (c.ok && p[9]) c.clamp8(p[7], p[8], "raw frontend");
                if (c.ok && p[1] % 4) c.bad("raw frontend width %d is not a multiple of 4 (the kernel stores dwords of one filter)", p[1]);
                break;
            case BN_OP_F32_STEM:
            case BN_OP_F32_DW: {  // H W C sh sw act OH OW pt pl
                const long long cin = o.kind == BN_OP_F32_STEM ? 1 : p[2];
                c.dims({p[0], p[1], p[2], p[6], p[7]}, "conv") && c.conv_geom(p[0], p[1], p[3], p[4], p[6], p[7], p[8], p[9]) &&
                    c.slot(o.in0, 4LL * p[0] * p[1] * cin, "input") && c.slot(o.out, 4LL * p[6] * p[7] * p[2], "output") &&
                    c.tensor(0, 36LL * p[2], "weights") && c.tensor(1, 4LL * p[2], "bias");
                break;
            }
            case BN_OP_F32_PW:  // P Cin Cout act has_res has_gate gate_slot
                c.dims({p[0], p[1], p[2]}, "pointwise") && c.slot(o.in0, 4LL * p[0] * p[1], "input") && c.slot(o.out, 4LL * p[0] * p[2], "output") &&
                    (!p[4] || c.slot(o.in1, 4LL * p[0] * p[2], "residual")) && (!p[5] || c.slot(p[6], 4LL * p[1], "gate")) &&
                    c.tensor(0, 4LL * p[1] * p[2], "weights") && c.tensor(1, 4LL * p[2], "bias");
                break;
            case BN_OP_F32_SEGATE:  // P C Cr
                c.dims({p[0], p[1], p[2]}, "squeeze-excite") && c.slot(o.in0, 4LL * p[0] * p[1], "input") && c.slot(o.out, 4LL * p[1], "gate") &&
                    c.tensor(0, 4LL * p[1] * p[2], "reduce weights") && c.tensor(1, 4LL * p[1] * p[2], "expand weights");
                break;
            case BN_OP_F32_SCALE:  // P C
                c.dims({p[0], p[1]}, "scale") && c.slot(o.in0, 4LL * p[0] * p[1], "input") && c.slot(o.in1, 4LL * p[1], "gate") && c.slot(o.out, 4LL * p[0] * p[1], "output");
                break;
            case BN_OP_F32_GAP:
            case BN_OP_F32_ATTNPOOL:  // P C
                c.dims({p[0], p[1]}, "pool") && c.slot(o.in0, 4LL * p[0] * p[1], "input") && c.slot(o.out, 4LL * p[1], "output") &&
                    (o.kind == BN_OP_F32_GAP || c.tensor(0, 4LL * p[1], "score vector"));
                break;
            case BN_OP_F32_DENSE:  // Cin Cout act
                c.dims({p[0], p[1]}, "dense") && c.slot(o.in0, 4LL * p[0], "input") && c.slot(o.out, 4LL * p[1], "scores") &&
                    c.tensor(0, 4LL * p[0] * p[1], "weights") && c.tensor(1, 4LL * p[1], "bias");
                if (c.ok && p[1] != (int)h.num_classes) c.bad("classifier width %d, header says %u classes", p[1], h.num_classes);
                break;
            case BN_OP_F32_GAPDENSE:  // P Cin Cout act
                c.dims({p[0], p[1], p[2]}, "pool+dense") && c.slot(o.in0, 4LL * p[0] * p[1], "input") && c.slot(o.out, 4LL * p[2], "scores") &&
                    c.tensor(0, 4LL * p[1] * p[2], "weights") && c.tensor(1, 4LL * p[2], "bias");
                if (c.ok && p[2] != (int)h.num_classes) c.bad("classifier width %d, header says %u classes", p[2], h.num_classes);
                break;
            case BN_OP_F32_DWPW: {  // H W Cin sh sw dw_act OH OW pt pl | Cout pw_act has_res has_gate gate_slot has_dw TH TW NB
                const long long out_b = 4LL * p[6] * p[7] * p[10];
                c.dims({p[0], p[1], p[2], p[6], p[7], p[10]}, "fused block") && c.slot(o.in0, 4LL * p[0] * p[1] * p[2], "input") && c.slot(o.out, out_b, "output") &&
                    (!p[12] || c.slot(o.in1, out_b, "residual")) && (!p[13] || c.slot(p[14], 4LL * p[2], "gate")) &&
                    c.tensor(2, 4LL * up(p[2], 16) * p[10], "pointwise weights") && c.tensor(3, 4LL * p[10], "pointwise bias");
                if (c.ok && p[15]) c.conv_geom(p[0], p[1], p[3], p[4], p[6], p[7], p[8], p[9]) && c.tensor(0, 36LL * p[2], "depthwise weights") && c.tensor(1, 4LL * p[2], "depthwise bias");
                if (c.ok && !p[15] && (p[0] != p[6] || p[1] != p[7])) c.bad("plain 1x1 convolution must keep the map size");
                if (c.ok && (p[16] < 1 || p[17] < 1 || p[18] < 1)) c.bad("tile %dx%dx%d", p[16], p[17], p[18]);
                break;
            }
            case BN_OP_F32_STFTMEL:  // T W M
                c.dims({p[1], p[2]}, "stft+mel") && c.slot(o.out, 4LL * p[2] * p[1], "mel energies") && c.tensor(0, 4, "band weights") && c.tensor(1, 12LL * p[2], "band table");
                break;
            case BN_OP_F32_MELFIN:  // M W mag norm
                c.dims({p[0], p[1]}, "mel finish") && c.slot(o.in0, 4LL * p[0] * p[1], "input") && c.slot(o.out, 4LL * p[0] * p[1], "output") &&
                    c.tensor(0, 4LL * p[0], "band sums") && (p[2] == 0 || c.tensor(2, 4LL * p[0], "magnitude parameters"));
                break;
            case BN_OP_F32_FRONT:  // H0 W0 C N OH OW stem_act dw_act pw_act raw_mel mag
                c.dims({p[0], p[1], p[2], p[3], p[4], p[5]}, "front block") && c.slot(o.in0, 4LL * p[0] * p[1], "frontend map") && c.slot(o.out, 4LL * p[4] * p[5] * p[3], "output") &&
                    c.tensor(0, 36LL * p[2], "stem weights") && c.tensor(1, 4LL * p[2], "stem bias") && c.tensor(2, 36LL * p[2], "depthwise weights") &&
                    c.tensor(3, 4LL * p[2], "depthwise bias") && c.tensor(4, 4LL * up(p[2], 16) * p[3], "pointwise weights") && c.tensor(5, 4LL * p[3], "pointwise bias") &&
                    (!p[9] || (c.tensor(6, 4LL * p[0], "band sums") && (p[10] == 0 || c.tensor(7, 4LL * p[0], "magnitude parameters"))));
                if (c.ok && (p[4] != (p[0] + 1) / 2 || p[5] != ((p[1] + 1) / 2 + 1) / 2)) c.bad("front block output %dx%d does not follow from %dx%d", p[4], p[5], p[0], p[1]);
                break;
            case BN_OP_I8_QUANT:  // F W Kp zp fill
                c.dims({p[0], p[1], p[2]}, "quantise") && c.slot(o.in0, 4LL * p[0] * p[1], "spectrogram") && c.slot(o.out, 1LL * p[1] * p[2], "output");
                if (c.ok && p[2] < p[0]) c.bad("padded bin count %d below %d bins", p[2], p[0]);
                break;
            case BN_OP_I8_MEL:  // W Kp M zp_out act_min act_max has_lut
                c.dims({p[0], p[1], p[2]}, "mel") && c.slot(o.in0, 1LL * p[0] * p[1], "input") && c.slot(o.out, 1LL * p[2] * p[0], "output") &&
                    c.tensor(0, 1LL * p[2] * p[1], "weights") && c.tensor(1, 4LL * p[2], "bias") && c.tensor(2, 4LL * p[2], "multipliers") &&
                    c.tensor(3, 4LL * p[2], "shifts") && (!p[6] || c.tensor(4, 256LL * p[2], "table"));
                if (c.ok && p[6]) c.clamp8(p[4], p[5], "mel");  // the clamped value + 128 indexes the 256-entry table
                break;
            case BN_OP_I8_STEM:
            case BN_OP_I8_DW: {  // H W C sh sw - OH OW pt pl ...
                const long long cin = o.kind == BN_OP_I8_STEM ? 1 : p[2];
                c.dims({p[0], p[1], p[2], p[6], p[7]}, "conv") && c.conv_geom(p[0], p[1], p[3], p[4], p[6], p[7], p[8], p[9]) &&
                    c.slot(o.in0, 1LL * p[0] * p[1] * cin, "input") && c.slot(o.out, 1LL * p[6] * p[7] * p[2], "output") && c.tensor(0, 9LL * p[2], "weights") &&
                    c.tensor(1, 4LL * p[2], "bias") && c.tensor(2, 4LL * p[2], "multipliers") && c.tensor(3, 4LL * p[2], "shifts");
                break;
            }
            case BN_OP_I8_PW:  // P Cin Cout zp_out amin amax has_add ...
                c.dims({p[0], p[1], p[2]}, "pointwise") && c.slot(o.in0, 1LL * p[0] * p[1], "input") && c.slot(o.out, 1LL * p[0] * p[2], "output") &&
                    (!p[6] || c.slot(o.in1, 1LL * p[0] * p[2], "residual")) && c.tensor(0, 1LL * p[1] * p[2], "weights") && c.tensor(1, 4LL * p[2], "bias") &&
                    c.tensor(2, 4LL * p[2], "multipliers") && c.tensor(3, 4LL * p[2], "shifts");
                break;
            case BN_OP_I8_MEAN:  // P C ...
                c.dims({p[0], p[1]}, "mean") && c.slot(o.in0, 1LL * p[0] * p[1], "input") && c.slot(o.out, 1LL * p[1], "output");
                break;
            case BN_OP_I8_FC:  // Cin Cout ...
                c.dims({p[0], p[1]}, "fully connected") && c.slot(o.in0, 1LL * p[0], "input") && c.slot(o.out, 1LL * p[1], "output") &&
                    c.tensor(0, up(p[0], 4) * p[1], "weights") && c.tensor(1, 4LL * p[1], "bias") && c.tensor(2, 4LL * p[1], "multipliers") &&
                    c.tensor(3, 4LL * p[1], "shifts") && (!p[5] || c.tensor(4, 256, "table"));
                if (c.ok && p[5]) c.clamp8(p[3], p[4], "fully connected");
                break;
            case BN_OP_I8_SCALE:  // P C zp_x zp_gate mult shift zp_out act_min act_max
                c.dims({p[0], p[1]}, "scale") && c.slot(o.in0, 1LL * p[0] * p[1], "input") && c.slot(o.in1, 1LL * p[1], "gate") && c.slot(o.out, 1LL * p[0] * p[1], "output");
                if (c.ok && p[1] % 4) c.bad("channel count %d is not a multiple of 4", p[1]);
                break;
            case BN_OP_I8_MAXNORM:  // C W has_lut
                c.dims({p[0], p[1]}, "max normalisation") && c.slot(o.in0, 1LL * p[0] * p[1], "input") && c.slot(o.out, 1LL * p[0] * p[1], "output") &&
                    c.tensor(0, 256, "denominator table") && c.tensor(1, 65536, "division table") && (!p[2] || c.tensor(2, 256LL * p[0], "channel table"));
                if (c.ok && p[1] % 4) c.bad("map width %d is not a multiple of 4 (the kernel reads dwords of one channel)", p[1]);
                break;
            case BN_OP_I8_RAWFE:  // T W M stride pad_left q_zp zp_out act_min act_max has_lut
                c.dims({p[0], p[1], p[2], p[3]}, "raw frontend") && c.slot(o.in0, 4LL * p[0], "waveform") && c.slot(o.out, 1LL * p[2] * p[1], "output") &&
                    c.tensor(0, 16LL * p[2], "filterbank") && c.tensor(1, 4LL * p[2], "bias") && c.tensor(2, 4LL * p[2], "multipliers") &&
                    c.tensor(3, 4LL * p[2], "shifts") && (!p[9] || c.tensor(4, 256LL * p[2], "table"));
                if (c.ok && p[4] < 0) c.bad("negative left padding");
                if (c.ok && p[9]) c.clamp8(p[7], p[8], "raw frontend");
                if (c.ok && p[1] % 4) c.bad("raw frontend width %d is not a multiple of 4 (the kernel stores dwords of one filter)", p[1]);
                break;
            case BN_OP_I8_ATTNPOOL:  // P C fc_bias fc_mult fc_shift fc_zo form zx za mul_mult mul_shift mul_zo mul_lo mul_hi sum_mult sum_shift sum_zo
                c.dims({p[0], p[1]}, "attention pooling") && c.slot(o.in0, 1LL * p[0] * p[1], "input map") && c.slot(o.out, 1LL * p[1], "output") &&
                    c.tensor(0, 1LL * p[1], "score vector") && c.tensor(1, p[6] == 0 ? 2048 : 1024, "softmax tables");
                if (c.ok && (p[1] % 4 || p[0] > 4096 || 1LL * p[0] * p[1] + 2LL * p[0] + 32 > 64 * 1024))  // (P <= 4096: the int32 sum of the exponentials)
                    c.bad("attention pooling map %d x %d does not fit the kernel", p[0], p[1]);
                if (c.ok && (p[6] < 0 || p[6] > 1)) c.bad("softmax form %d", p[6]);
                if (c.ok) c.clamp8(p[12], p[13], "attention pooling MUL");
                break;
            case BN_OP_I8_HEAD:  // C zp_fc zp_out has_lut
                c.dims({p[0]}, "head") && c.slot(o.in0, 1LL * p[0], "input") && c.slot(o.out, 4LL * p[0], "scores") && (!p[3] || c.tensor(0, 256, "table"));
                if (c.ok && p[0] != (int)h.num_classes) c.bad("classifier width %d, header says %u classes", p[0], h.num_classes);
                break;
            case BN_OP_I8_DWPW: {
                // H W Cin sh sw F OH OW pt pl | dw q (10..13) | Cout(14) pw q (15..17) | add (18..28) | has_dw(29) transposed(30) TH TW NB has_lut(34)
                // strip(35) quantise_at_load(36) qzp qfill
                const long long out_b = 1LL * p[6] * p[7] * p[14];
                c.dims({p[0], p[1], p[2], p[6], p[7], p[14]}, "fused block");
                if (c.ok && p[36]) c.dims({p[5]}, "spectrogram bins") && c.slot(o.in0, 4LL * p[5] * p[1], "spectrogram");
                else c.slot(o.in0, 1LL * p[0] * p[1] * p[2], "input");
                c.slot(o.out, out_b, "output") && (!p[18] || c.slot(o.in1, out_b, "residual")) && c.tensor(4, up(p[2], 64) * p[14], "pointwise weights") &&
                    c.tensor(5, 4LL * p[14], "pointwise bias") && c.tensor(6, 4LL * p[14], "pointwise multipliers") && c.tensor(7, 4LL * p[14], "pointwise shifts") &&
                    (!p[34] || c.tensor(8, 256LL * p[14], "table"));
                if (c.ok && p[29])
                    c.conv_geom(p[0], p[1], p[3], p[4], p[6], p[7], p[8], p[9]) && c.tensor(0, 9LL * p[2], "depthwise weights") && c.tensor(1, 4LL * p[2], "depthwise bias") &&
                        c.tensor(2, 4LL * p[2], "depthwise multipliers") && c.tensor(3, 4LL * p[2], "depthwise shifts");
                if (c.ok && !p[29] && (p[0] != p[6] || p[1] != p[7])) c.bad("plain 1x1 convolution must keep the map size");
                if (c.ok && !p[29] && p[18] && o.t[10] >= 0) c.tensor(10, 65536, "ADD table");
                if (c.ok && (p[31] < 1 || p[32] < 1 || p[33] < 1)) c.bad("tile %dx%dx%d", p[31], p[32], p[33]);
                // the transposed form is the mel mixer: a plain 1x1 over the frames of one chunk, no residual; only it takes a table or float32 input
                if (c.ok && p[30] && (p[29] || p[18] || p[0] != 1 || p[6] != 1 || p[33] != 1)) c.bad("transposed output on a block that is not a mel mixer");
                if (c.ok && !p[30] && (p[34] || p[36])) c.bad("table / fused QUANTIZE on a block without transposed output");
                if (c.ok && p[34]) c.clamp8(p[16], p[17], "mel mixer");
                if (c.ok && p[18]) c.clamp8(p[16], p[17], "block with ADD") && c.clamp8(p[27], p[28], "ADD");  // both index the 256-entry rescale tables
                if (c.ok && p[35] && o.t[9] >= 0) {  // constant block of the strip kernel (bn_i8_strip.hip: kPWC + NW * nPWC words)
                    const int nw = p[3] == p[4] ? i8_strip_waves(p[2], p[14], p[3], p[7], p[18] != 0) : 0;
                    if (nw) {
                        const long long ql = p[2] / nw / 16, nt = p[14] / nw / 16;
                        const long long words = nw * (4 * ql * 12 + 4 * ql * 4 + 4 * ql * 12 + nt * nw * 64 * ql + 4 * nt * 4 + 4 * nt * 12);
                        c.tensor(9, 4 * words, "strip constants") && (!p[18] || o.t[10] < 0 || c.tensor(10, 65536, "ADD table"));
                    }
                }
                break;
            }
            case BN_OP_I8_FRONT:  // H0 W0 C N OH OW ... strip(16)
                c.dims({p[0], p[1], p[2], p[3], p[4], p[5]}, "front block") && c.slot(o.in0, 1LL * p[0] * p[1], "frontend map") && c.slot(o.out, 1LL * p[4] * p[5] * p[3], "output") &&
                    c.tensor(0, 9LL * p[2], "stem weights") && c.tensor(1, 4LL * p[2], "stem bias") && c.tensor(2, 4LL * p[2], "stem multipliers") && c.tensor(3, 4LL * p[2], "stem shifts") &&
                    c.tensor(4, 9LL * p[2], "depthwise weights") && c.tensor(5, 4LL * p[2], "depthwise bias") && c.tensor(6, 4LL * p[2], "depthwise multipliers") &&
                    c.tensor(7, 4LL * p[2], "depthwise shifts") && c.tensor(8, up(p[2], 64) * p[3], "pointwise weights") && c.tensor(9, 4LL * p[3], "pointwise bias") &&
                    c.tensor(10, 4LL * p[3], "pointwise multipliers") && c.tensor(11, 4LL * p[3], "pointwise shifts") && (!p[16] || o.t[12] < 0 || c.tensor(12, 496 * 4, "strip constants"));
                if (c.ok && (p[4] != (p[0] + 1) / 2 || p[5] != ((p[1] + 1) / 2 + 1) / 2)) c.bad("front block output %dx%d does not follow from %dx%d", p[4], p[5], p[0], p[1]);
                break;
            case BN_OP_I8_TAIL:  // in_bytes pw_macs dw_macs other_macs n_classes n_layers H0 W0 C0 P_last C_last
                c.dims({p[0], p[4], p[5], p[6], p[7], p[8]}, "fused tail") && c.slot(o.in0, 1LL * p[6] * p[7] * p[8], "input map") && c.slot(o.out, 4LL * p[4], "scores") &&
                    c.tensor(0, 16, "constant block") && c.tensor(1, 4LL * (24 * p[5] + 16), "descriptor table");
                if (c.ok && (p[4] != (int)h.num_classes || p[5] > 8 || p[BN_OP_TAIL_TAG] != BN_TAIL_OP)) c.bad("fused tail header");
                break;  // the descriptor table itself is validated by bn::tail_plan at load (bn_api.hip)
            case BN_OP_I8_MID:  // in_bytes pw_macs dw_macs 0 0 n_layers H0 W0 C0 P_last C_last
                c.dims({p[0], p[5], p[6], p[7], p[8], p[9], p[10]}, "fused stage-2 chain") && c.slot(o.in0, 1LL * p[6] * p[7] * p[8], "input map") &&
                    c.slot(o.out, 1LL * p[9] * p[10], "output map") && c.tensor(0, 16, "constant block") && c.tensor(1, 4LL * 32 * p[5], "descriptor table");
                if (c.ok && (p[5] > 8 || p[BN_OP_TAIL_TAG] != BN_MID_OP)) c.bad("fused stage-2 chain header");
                break;  // (descriptor table: bn::tail2_plan at load)
            default:
                c.bad("unknown operator kind");
                break;
        }
    }
    return c.ok;
}

}  // namespace bn
