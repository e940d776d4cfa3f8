// bn_api.hip — the C ABI of libbirdnet_hip.so (include/birdnet_hip.h): context and model
// lifetime, packed-blob parsing, workspace planning and the device-plan executor that turns one
// bn_forward()/bn_infer_audio() call into a sequence of kernel launches on the caller's stream.
#include <hip/hip_runtime.h>

#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/birdnet_hip.h"
#include "bn_blob.h"
#include "bn_kernels.h"
#include "bn_quant_in.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(BN_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int kFft = 512;
constexpr int kMaxGridBatch = 32768;  // chunks per launch group (gridDim.y/z limit is 65535)

struct OptName {
    const char* name;
    int bn::Options::*field;
};
const OptName kOptions[] = {
    {"f32_strip", &bn::Options::f32_strip},       {"f32_strip_th", &bn::Options::f32_strip_th},
    {"f32_front_staged", &bn::Options::f32_front_staged}, {"f32_front2", &bn::Options::f32_front2}, {"f32_pwdw", &bn::Options::f32_pwdw}, {"f32_tile_slice", &bn::Options::f32_tile_slice}, {"f32_pw_ws", &bn::Options::f32_pw_ws}, {"i8_pwdw", &bn::Options::i8_pwdw}, {"i8_pw_lds", &bn::Options::i8_pw_lds}, {"i8_pw_forms", &bn::Options::i8_pw_forms}, {"i8_add_tab", &bn::Options::i8_add_tab}, {"front_tpw", &bn::Options::front_tpw},
    {"wave_dwpw", &bn::Options::wave_dwpw},       {"i8_strip", &bn::Options::i8_strip},
    {"i8_strip_th", &bn::Options::i8_strip_th}, {"i8_dw_pool", &bn::Options::i8_dw_pool}, {"i8_tail_fclds", &bn::Options::i8_tail_fclds},   {"i8_tail", &bn::Options::i8_tail}, {"i8_tail_mfdw", &bn::Options::i8_tail_mfdw}, {"i8_mid", &bn::Options::i8_mid},
    {"i8_mel_generic", &bn::Options::i8_mel_generic}, {"stft_rowmajor", &bn::Options::stft_rowmajor},
    {"i8_strip_mfdw", &bn::Options::i8_strip_mfdw}, {"stft_exact", &bn::Options::stft_exact}, {"stft_flagcap", &bn::Options::stft_flagcap}, {"stft_guard", &bn::Options::stft_guard}, {"stft_audit", &bn::Options::stft_audit}, {"stft_minint", &bn::Options::stft_minint},
    {"ingest_blk", &bn::Options::ingest_blk},
    {"ingest_generic", &bn::Options::ingest_generic},
};

// BN_<NAME> environment variables seed the options once, when the library is loaded (A/B runs of bench.py from a shell)
bn::Options options_from_env() {
    bn::Options o;
    for (const OptName& e : kOptions) {
        std::string var = "BN_";
        for (const char* c = e.name; *c; ++c) var += (char)toupper((unsigned char)*c);
        if (const char* v = getenv(var.c_str())) o.*(e.field) = atoi(v);
    }
    return o;
}

}  // namespace

namespace bn {
thread_local Options g_opt;
}

bn::Options g_opt_default = options_from_env();   // the process default (bn_set_option); copied into bn::g_opt per API call
std::mutex g_opt_mu;

struct bn_ctx {
    std::vector<std::pair<int bn::Options::*, int>> opt_override;   // switches this context sets for itself (bn_ctx_set_option)
    int device = 0;
    int max_batch = 0;
    float* d_window = nullptr;
    float4* d_tw256 = nullptr;
    float4* d_tw512 = nullptr;
    double* d_f64tab = nullptr;      // hann64[512], cs64[512] (bn_stft_exact.hip)
    bn::StftTables tables{};
    float* d_block_peaks = nullptr;  // bn_ingest_resample: per-workgroup maxima, grown on demand
    size_t block_peaks_elems = 0;
    void* d_rank_work = nullptr;     // bn_rank_orders: transposed keys, index arrays, rocPRIM storage; grown on demand
    size_t rank_work_bytes = 0;
};

struct bn_model {
    bn_ctx* ctx = nullptr;
    BlobHeader hdr{};
    std::vector<OpRec> ops;
    std::vector<TensorRec> tensors;
    std::vector<SlotRec> slots;
    char* d_consts = nullptr;            // one allocation, tensors at their blob offsets
    size_t consts_base = 0;              // blob offset of the first payload byte
    size_t consts_bytes = 0;
    std::vector<uint8_t> rq_right;       // per operator: all requantisation multipliers >= 0 and shifts < 0
    std::vector<bn::Tail8Args> tails;    // per operator: arguments of the fused tail kernel (BN_OP_I8_TAIL operators only)
    std::vector<uint8_t> tail_ok;        // per operator: BN_OP_I8_TAIL whose maps fit the kernel's LDS plan
    std::vector<bn::Tail2Args> tails2;   // per operator: the same for i8_tail2_kernel (depthwise stage on the matrix cores), when the plan carries its constants
    std::vector<uint8_t> tail2_ok;
    std::vector<bn::Tail2Args> mids;     // per operator: arguments of the fused stage-2 chain (BN_OP_I8_MID operators only)
    std::vector<uint8_t> mid_ok;
    bool has_mid = false;
    std::vector<uint8_t> out_valid;      // per operator: it wrote its output slot in the last forward call (not when a fused kernel covered it)
    std::vector<uint8_t> slot_valid;     // per slot: some operator wrote it in the last forward call
    bool has_tail = false;               // the plan holds a usable fused tail operator
    bool guard_form_ok = false;          // ... and its QUANTIZE has zero point -128 (the only form the guarded mixer is built for)
    bool spec_tiled_ok = false;          // the plan's first operator reads the spectrogram through i8_mel_mfma_kernel<QIN>: bn_infer_audio
                                         // may hand it the tile-major layout the STFT writes fastest
    bool spec_tiled_now = false;         // set by bn_infer_audio around its bn_forward call
    std::vector<char*> d_slots;          // max_batch * bytes_per_chunk each
    float* d_spec = nullptr;             // [max_batch][F][W] for bn_infer_audio
    float* d_minmax = nullptr;           // [max_batch][2]
    char* d_guard = nullptr;             // buffers of the exactness pass (INT8 plans whose first operator quantises the spectrogram)
    bn::StftGuard guard{};
    bool last_tiled = false;             // layout of d_spec after the last bn_infer_audio call (bn_debug_input_bytes)
    int last_B = 0;
    bool guard_now = false;              // set by bn_infer_audio: the first operator lists doubtful bytes, the float64 pass follows it
    const float* guard_audio = nullptr;
    int* d_audit = nullptr;              // [2] exactness audit: elements audited, violations (option stft_audit; zeroed per bn_infer_audio call)
    int guard_T = 0, guard_hop = 0;
    float* d_smax = nullptr;             // [max_batch] per-sample maxima of the frontend
    float* d_gap_part = nullptr;         // [max_batch][gap_part_elems] channel sums per row block from f32_pwdw_kernel for the squeeze-excite gate behind it
    size_t gap_part_elems = 0;
    int32_t* d_pool8 = nullptr;          // [max_batch][pool8_C] int32 channel sums from i8_dw_stream_kernel for the squeeze-excite gate behind it (zero between uses)
    size_t pool8_C = 0;
    size_t workspace_bytes = 0;
    // per-operator HIP-event timing (bn_profile_*): one (start, stop) pair per launch group
    bool profiling = false;
    int prof_only = -1;                  // >= 0: bracket only this operator (index n_ops = the STFT stage)
    struct EvRec {
        int op;
        hipEvent_t start, stop;
    };
    std::vector<EvRec> ev_used;
    std::vector<hipEvent_t> ev_free;

    hipEvent_t take_event() {
        if (!ev_free.empty()) {
            hipEvent_t e = ev_free.back();
            ev_free.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }

    const void* tensor(int id) const {
        if (id < 0) return nullptr;
        return d_consts + (tensors[id].offset - consts_base);
    }
};

namespace {

int check_device(bn_ctx* ctx) {
    if (!ctx) return fail(BN_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    {   // the switches this call's launchers see: the process default, then the context's own
        std::lock_guard<std::mutex> lock(g_opt_mu);
        bn::g_opt = g_opt_default;
        for (const auto& ov : ctx->opt_override) bn::g_opt.*(ov.first) = ov.second;
    }
    return BN_OK;
}

// Brackets the launches of one plan operator with HIP events on the launch stream when profiling.
struct ProfScope {
    bn_model* m;
    hipStream_t s;
    hipEvent_t stop = nullptr;
    ProfScope(bn_model* m_, int op, hipStream_t s_) : m(m_), s(s_) {
        if (!m->profiling || (m->prof_only >= 0 && m->prof_only != op)) return;
        hipEvent_t start = m->take_event();
        stop = m->take_event();
        (void)hipEventRecord(start, s);
        m->ev_used.push_back({op, start, stop});
    }
    void end() {
        if (stop) (void)hipEventRecord(stop, s);
        stop = nullptr;
    }
    ~ProfScope() { end(); }
};

// A fused kernel runs its head operator and the partner(s) the packer tagged as ONE launch: only when the partner runs wherever the head
// does — it belongs to both entry paths (the usual case: one head per path in front of a shared block) or to the head's own.  A partner of
// the OTHER path would not run at all in this mode: separate launches then.
inline bool same_path(const OpRec& head, const OpRec& partner) {
    return partner.p[BN_OP_PATH] == BN_PATH_BOTH || partner.p[BN_OP_PATH] == head.p[BN_OP_PATH];
}

// The exactness pass's buffers for the chunks from b0 on (the work list is shared: one launch group at a time uses it).
bn::StftGuard guard_slice(const bn_model* m, size_t b0) {
    bn::StftGuard g = m->guard;
    const size_t W = m->hdr.spec_width;
    g.eps += b0 * W;
    g.rec += b0 * ((W + 15) / 16) * bn::kGuardRec;
    g.count += b0;
    g.dirty += b0;
    g.mn_lo += b0;
    g.min_interval = bn::g_opt.stft_minint;
    // every launch group has its own lists and counters (a batch beyond kMaxGridBatch runs the STFT stage of all groups before the plan of the first)
    const size_t group = b0 / kMaxGridBatch;
    g.work += b0 * ((W + 63) / 64);
    g.n_work += group;
    g.hard += b0;
    g.n_hard += 2 * group;
    g.audio = m->guard_audio;  // (already offset to the launch group's first chunk by bn_infer_audio)
    g.T = m->guard_T;
    g.hop = m->guard_hop;
    g.tabs = m->ctx->tables;
    g.flag_cap = bn::g_opt.stft_flagcap;
    // the frame part of the bound (bn_quant_in.h): empirical, proven, or — tests only — far too small
    const int gm = bn::g_opt.stft_guard;
    g.k_l2 = gm == 1 ? bn::kGuardL2Proven : gm == 2 ? bn::kGuardL2 / 1024.0f : bn::kGuardL2;
    g.k_peak = gm == 1 ? 0.0f : gm == 2 ? bn::kGuardPeak / 1024.0f : bn::kGuardPeak;
    g.audit_scale = gm == 2 ? 1024.0f : 1.0f;
    g.slack_scale = gm == 2 ? 0.0f : 1.0f;
    g.audit = bn::g_opt.stft_audit ? m->d_audit : nullptr;
    return g;
}

// Executes the plan for a batch slice.
// `op_begin..op_end` restricts the run to a range of operators, `slot_b0` is the chunk index the slice starts at inside the
// workspace slots (bn_infer_audio runs the first operator per sub-batch, the rest over the whole batch).
int run_plan(bn_model* m, const float* d_input, const float* d_minmax, int B, float* d_scores, float* d_logits,
             hipStream_t s, const float* d_audio = nullptr, int T = 0, int hop = 0, size_t op_begin = 0, size_t op_end = (size_t)-1,
             size_t slot_b0 = 0) {
    const int mode = d_audio ? BN_PATH_AUDIO : BN_PATH_INPUT;
    auto slot_ptr = [&](int id) -> char* {
        if (id == BN_SLOT_INPUT) return (char*)d_input;
        if (id == BN_SLOT_AUDIO) return (char*)d_audio;
        if (id == BN_SLOT_SCORES) return (char*)d_scores;
        if (id == BN_SLOT_LOGITS) return (char*)d_logits;
        if (id < 0 || id >= (int)m->d_slots.size()) return nullptr;
        return m->d_slots[id] + slot_b0 * m->slots[id].bytes_per_chunk;
    };
    const bool tail_on = m->has_tail && bn::g_opt.i8_tail;
    const bool mid_on = m->has_mid && bn::g_opt.i8_mid && bn::g_opt.i8_strip;
    if (op_end > m->ops.size()) op_end = m->ops.size();
    // the pooling scratch is zero between uses (i8_segate_kernel clears what it reads); cleared here as well, so that a call that failed half-way
    // cannot leave sums behind for the next one
    if (m->d_pool8 && op_begin == 0 && bn::g_opt.i8_dw_pool) HIP_TRY(hipMemsetAsync(m->d_pool8, 0, (size_t)B * m->pool8_C * sizeof(int32_t), s));
    size_t gap_for = (size_t)-1;      // squeeze-excite gate whose pooling comes as row-block sums from the fused kernel in front of it
    int gap_R = 0, cand_R = 0;
    size_t cand_for = (size_t)-1;     // (candidate: becomes gap_for once the fused kernel has been launched)
    // the gate right behind a fused (expand, depthwise) pair pools the depthwise map: the fused kernel hands it per-row-block channel sums
    auto gap_target = [&](size_t di) -> float* {
        const OpRec& d = m->ops[di];
        cand_for = (size_t)-1;
        if (bn::g_opt.f32_pwdw < 2 || !m->d_gap_part || di + 1 >= op_end) return nullptr;
        const OpRec& g = m->ops[di + 1];
        const int rb = bn::f32_pwdw_rows(d.p[6]);
        const int R = (d.p[6] + rb - 1) / rb;
        if (g.kind != BN_OP_F32_SEGATE || g.in0 != d.out || g.p[1] != d.p[2] || g.p[0] != d.p[6] * d.p[7] || (size_t)R * d.p[2] > m->gap_part_elems ||
            (g.p[BN_OP_PATH] != BN_PATH_BOTH && g.p[BN_OP_PATH] != mode))
            return nullptr;
        cand_for = di + 1;
        cand_R = R;
        return m->d_gap_part;
    };
    size_t pwdw_head_done = (size_t)-1;  // expand convolution that ran inside the fused kernel of the stem operator in front of it
    size_t pwdw_done = (size_t)-1;    // depthwise stage that ran inside the expand convolution in front of it
    size_t front2_done = (size_t)-1;  // operator that the fused front kernel of this run has already covered
    auto dwpw_args = [&](const OpRec& d) {
        bn::DwPwArgs a{};
        const int* q = d.p;
        a.x = (const float*)slot_ptr(d.in0);
        a.res = q[12] ? (const float*)slot_ptr(d.in1) : nullptr;
        a.gate = q[13] ? (const float*)slot_ptr(q[14]) : nullptr;
        a.y = (float*)slot_ptr(d.out);
        a.dw_w = (const float*)m->tensor(d.t[0]);
        a.dw_b = (const float*)m->tensor(d.t[1]);
        a.pw_w = (const float*)m->tensor(d.t[2]);
        a.pw_b = (const float*)m->tensor(d.t[3]);
        a.B = B; a.H = q[0]; a.W = q[1]; a.Cin = q[2]; a.sh = q[3]; a.sw = q[4]; a.dw_act = q[5];
        a.OH = q[6]; a.OW = q[7]; a.pt = q[8]; a.pl = q[9]; a.Cout = q[10]; a.pw_act = q[11];
        a.has_dw = q[15]; a.TH = q[16]; a.TW = q[17]; a.NB = q[18];
        return a;
    };
    size_t segate_done[2] = {(size_t)-1, (size_t)-1};  // the two dense layers of a squeeze-excite gate that ran inside the pooling kernel
    size_t pool8_for = (size_t)-1;  // MEAN operator whose channel sums the depthwise kernel in front of it has already put into d_pool8
    // does the MEAN operator `mi` run as i8_segate_kernel (MEAN -> FC -> FC in one launch)?
    auto segate_fused = [&](size_t mi) {
        if (mi + 2 >= op_end) return false;
        const OpRec& o = m->ops[mi];
        const OpRec& f1 = m->ops[mi + 1];
        const OpRec& f2 = m->ops[mi + 2];
        const int* p = o.p;
        return o.kind == BN_OP_I8_MEAN && p[BN_OP_TAIL_TAG] == BN_SEGATE_HEAD && bn::g_opt.i8_strip && same_path(o, f1) && same_path(o, f2) &&
               f1.kind == BN_OP_I8_FC && f2.kind == BN_OP_I8_FC && f1.p[BN_OP_TAIL_TAG] == BN_SEGATE_COVERED && f2.p[BN_OP_TAIL_TAG] == BN_SEGATE_COVERED &&
               f1.in0 == o.out && f2.in0 == f1.out && f1.p[0] == p[1] && f2.p[0] == f1.p[1] && f2.p[1] == p[1] && p[1] % 4 == 0 && f2.out != o.in0;
    };
    size_t scale_done = (size_t)-1;  // 1x1 convolution that already ran with the squeeze-excite MUL in front of it applied on load
    auto dwpw8_args = [&](const OpRec& d, size_t di) {
        bn::DwPw8Args a{};
        const int* q = d.p;
        a.x = (const int8_t*)slot_ptr(d.in0);
        a.res = q[18] ? (const int8_t*)slot_ptr(d.in1) : nullptr;
        a.y = (int8_t*)slot_ptr(d.out);
        a.dw_w = (const int8_t*)m->tensor(d.t[0]);
        a.dw_b = (const int32_t*)m->tensor(d.t[1]);
        a.dw_mult = (const int32_t*)m->tensor(d.t[2]);
        a.dw_shift = (const int32_t*)m->tensor(d.t[3]);
        a.pw_w = (const int8_t*)m->tensor(d.t[4]);
        a.pw_b = (const int32_t*)m->tensor(d.t[5]);
        a.pw_mult = (const int32_t*)m->tensor(d.t[6]);
        a.pw_shift = (const int32_t*)m->tensor(d.t[7]);
        a.lut = q[34] ? (const int8_t*)m->tensor(d.t[8]) : nullptr;
        a.B = B; a.H = q[0]; a.W = q[1]; a.Cin = q[2]; a.sh = q[3]; a.sw = q[4]; a.OH = q[6]; a.OW = q[7];
        a.pt = q[8]; a.pl = q[9]; a.dw_zp_in = q[10]; a.dw_zp_out = q[11]; a.dw_amin = q[12]; a.dw_amax = q[13];
        a.Cout = q[14]; a.pw_zp_out = q[15]; a.pw_amin = q[16]; a.pw_amax = q[17];
        a.add = bn::I8AddParams{q[18], q[19], q[20], q[21], q[22], q[23], q[24], q[25], q[26], q[27], q[28]};
        a.has_dw = q[29]; a.transposed = q[30]; a.TH = q[31]; a.TW = q[32]; a.NB = q[33];
        a.rq_right = m->rq_right[di];
        a.add_tab = (q[18] && !q[29] && d.t[10] >= 0) ? (const int8_t*)m->tensor(d.t[10]) : nullptr;
        return a;
    };
    m->out_valid.resize(m->ops.size());
    for (size_t oi = op_begin; oi < op_end; ++oi) m->out_valid[oi] = 0;
    if (op_begin == 0) m->slot_valid.assign(m->d_slots.size(), 0);
    for (size_t oi = op_begin; oi < op_end; ++oi) {
        const OpRec& o = m->ops[oi];
        const int* p = o.p;
        if (p[BN_OP_PATH] != BN_PATH_BOTH && p[BN_OP_PATH] != mode) continue;
        if (oi == front2_done || oi == pwdw_done || oi == pwdw_head_done || oi == scale_done || oi == segate_done[0] || oi == segate_done[1]) continue;  // ran inside a preceding operator's kernel
        if (p[BN_OP_TAIL_TAG] == BN_MID_COVERED && mid_on) continue;    // the fused stage-2 chain runs these blocks
        if (p[BN_OP_TAIL_TAG] == BN_MID_OP && !(mid_on && m->mid_ok[oi])) continue;
        if (p[BN_OP_TAIL_TAG] == BN_TAIL_COVERED && tail_on) continue;  // the fused tail operator runs these blocks
        if (p[BN_OP_TAIL_TAG] == BN_TAIL_OP && !(tail_on && m->tail_ok[oi])) continue;
        ProfScope prof(m, (int)oi, s);
        m->out_valid[oi] = 1;  // (a fused kernel that keeps this operator's map on chip clears it again and marks the partner it wrote)
        auto mark_slot = [&](int sid, int v) {
            if (sid >= 0 && (size_t)sid < m->slot_valid.size()) m->slot_valid[sid] = (uint8_t)v;
        };
        mark_slot(o.out, 1);
        auto fused_into = [&](size_t partner) {
            m->out_valid[oi] = 0;
            mark_slot(o.out, 0);
            m->out_valid[partner] = 1;
            mark_slot(m->ops[partner].out, 1);
        };
        char* in0 = slot_ptr(o.in0);
        char* in1 = slot_ptr(o.in1);
        char* out = slot_ptr(o.out);
        const float* mm = (o.in0 == BN_SLOT_INPUT) ? d_minmax : nullptr;
        switch (o.kind) {
            case BN_OP_F32_MEL: {
                if (p[4]) bn::launch_u32_fill((uint32_t*)m->d_smax, 0u, B, s);
                bn::launch_f32_mel((const float*)in0, mm, (float*)out, m->d_smax, B, p[0], p[1], p[2],
                                   (const float*)m->tensor(o.t[0]), (const int*)m->tensor(o.t[1]),
                                   (const float*)m->tensor(o.t[2]), p[3], p[4], s);
                break;
            }
            case BN_OP_F32_STFTMEL: {
                bn::launch_minmax_init(m->d_minmax, B, s);
                if (!bn::launch_stft512_mel(m->ctx->tables, d_audio, B, T, hop, p[1], (float*)out, p[2],
                                            (const float*)m->tensor(o.t[0]), (const int*)m->tensor(o.t[1]), m->d_minmax, s))
                    return fail(BN_ERR_UNSUPPORTED, "the fused STFT+mel kernel takes at most 128 mel bins (got %d)", p[2]);
                break;
            }
            case BN_OP_F32_MELFIN:
                bn::launch_f32_melfin((const float*)in0, m->d_minmax, (float*)out, B, p[0], p[1], (const float*)m->tensor(o.t[0]),
                                      (const float*)m->tensor(o.t[2]), p[2], p[3], s);
                break;
            case BN_OP_F32_RAWFE:
                bn::launch_f32_rawfe((const float*)in0, (float*)out, B, p[0], p[1], p[2], p[3], p[4], (const float*)m->tensor(o.t[0]),
                                     (const float*)m->tensor(o.t[1]), (const float*)m->tensor(o.t[2]), p[5], s);
                break;
            case BN_OP_F32_MAG:
                bn::launch_f32_mag((float*)out, m->d_smax, B, p[0], p[1], (const float*)m->tensor(o.t[2]), p[2], s);
                break;
            case BN_OP_F32_STEM:
                if (p[BN_OP_TAIL_TAG] == BN_PWDW_STEM && bn::g_opt.f32_pwdw && bn::g_opt.f32_strip && oi + 2 < op_end) {
                    // stem -> expand 1x1 -> depthwise 3x3 as ONE kernel: neither the stem map nor the expanded map is written
                    const OpRec& e = m->ops[oi + 1];
                    const OpRec& d = m->ops[oi + 2];
                    const int* q = d.p;
                    if (same_path(o, e) && same_path(o, d) &&
                        e.kind == BN_OP_F32_DWPW && e.p[BN_OP_TAIL_TAG] == BN_PWDW_HEAD && d.kind == BN_OP_F32_DW && q[BN_OP_TAIL_TAG] == BN_PWDW_COVERED &&
                        e.in0 == o.out && d.in0 == e.out && d.out != o.in0 && d.out != o.out && d.out != e.out && e.p[0] == p[6] && e.p[1] == p[7] &&
                        e.p[2] == p[2]) {
                        const bn::DwPwArgs ea = dwpw_args(e);
                        const bn::F32StemIn st{(const float*)in0, (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]), p[0], p[1], p[3], p[4],
                                               p[8], p[9], p[5]};
                        if (ea.Cin <= 32 && bn::f32_pwdw_supported(ea, q[0], q[1], q[2], q[3], q[4], q[6], q[7]) &&  // (the stem on the matrix cores feeds at most two channel tiles)
                            bn::launch_f32_pwdw(ea, (const float*)m->tensor(d.t[0]), (const float*)m->tensor(d.t[1]), (float*)slot_ptr(d.out), q[3], q[6], q[7],
                                                q[8], q[9], q[5], &st, gap_target(oi + 2), s)) {
                            pwdw_head_done = oi + 1;
                            pwdw_done = oi + 2;
                            fused_into(oi + 2);
                            gap_for = cand_for;
                            gap_R = cand_R;
                            break;
                        }
                    }
                }
                bn::launch_f32_stem((const float*)in0, (float*)out, B, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7],
                                    p[8], p[9], (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]), s);
                break;
            case BN_OP_F32_DW: {
                // a squeeze-excite gate right behind the stage pools per-strip channel sums written by the depthwise kernel (as behind fused pairs)
                float* gp = nullptr;
                int R = 0;
                if (bn::g_opt.f32_pwdw >= 2 && m->d_gap_part && oi + 1 < op_end) {
                    const OpRec& g = m->ops[oi + 1];
                    R = bn::f32_dw_stream_strips(B, p[2], p[6], p[7]);
                    if (g.kind == BN_OP_F32_SEGATE && g.in0 == o.out && g.p[1] == p[2] && g.p[0] == p[6] * p[7] && (size_t)R * p[2] <= m->gap_part_elems &&
                        (g.p[BN_OP_PATH] == BN_PATH_BOTH || g.p[BN_OP_PATH] == mode))
                        gp = m->d_gap_part;
                }
                if (bn::launch_f32_dw((const float*)in0, (float*)out, B, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9],
                                      (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]), gp, s) && gp) {
                    gap_for = oi + 1;
                    gap_R = R;
                }
                break;
            }
            case BN_OP_F32_PW:
                bn::launch_f32_pw((const float*)in0, p[4] ? (const float*)in1 : nullptr,
                                  p[5] ? (const float*)slot_ptr(p[6]) : nullptr, (float*)out, B, p[0], p[1], p[2], p[3],
                                  (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]), s);
                break;
            case BN_OP_F32_DWPW: {
                const bn::DwPwArgs a = dwpw_args(o);
                if (p[BN_OP_TAIL_TAG] == BN_PWDW_HEAD && bn::g_opt.f32_pwdw && bn::g_opt.f32_strip && oi + 1 < op_end) {
                    // inverted-residual block: the expand convolution runs inside the depthwise kernel behind it (the expanded map stays in LDS)
                    const OpRec& d = m->ops[oi + 1];
                    const int* q = d.p;
                    if (same_path(o, d) && d.kind == BN_OP_F32_DW && q[BN_OP_TAIL_TAG] == BN_PWDW_COVERED && d.in0 == o.out && d.out != o.in0 && d.out != o.out &&
                        bn::f32_pwdw_supported(a, q[0], q[1], q[2], q[3], q[4], q[6], q[7]) &&
                        bn::launch_f32_pwdw(a, (const float*)m->tensor(d.t[0]), (const float*)m->tensor(d.t[1]), (float*)slot_ptr(d.out), q[3], q[6], q[7], q[8],
                                            q[9], q[5], nullptr, gap_target(oi + 1), s)) {
                        pwdw_done = oi + 1;
                        fused_into(oi + 1);
                        gap_for = cand_for;
                        gap_R = cand_R;
                        break;
                    }
                }
                if (!bn::f32_dwpw_supported(a.Cin, a.Cout) || (a.has_dw && a.Cin % 16) || a.TH * a.TW * a.NB != 64 || a.OH % a.TH || a.OW % a.TW)
                    return fail(BN_ERR_FORMAT, "operator %zu: unsupported fused block geometry", oi);
                bn::launch_f32_dwpw(a, s);
                break;
            }
            case BN_OP_F32_FRONT:
                if (!bn::f32_front_supported(p[0], p[1], p[2], p[3], p[4], p[5]))
                    return fail(BN_ERR_FORMAT, "operator %zu: unsupported front-block geometry", oi);
                if (p[BN_OP_TAIL_TAG] == BN_FRONT2_HEAD && bn::g_opt.f32_front2 && bn::g_opt.f32_strip && bn::g_opt.f32_front_staged &&
                    p[BN_OP_FRONT2_DIST] > 0 && oi + (size_t)p[BN_OP_FRONT2_DIST] < op_end) {
                    // front block + the residual block behind it as one kernel: the 32-channel map between them stays in LDS
                    const OpRec& d = m->ops[oi + (size_t)p[BN_OP_FRONT2_DIST]];
                    if (same_path(o, d) && d.kind == BN_OP_F32_DWPW && d.p[BN_OP_TAIL_TAG] == BN_FRONT2_COVERED && d.in0 == o.out && d.out != o.in0) {  // (never in place)
                        const bn::F32FrontStripArgs f{(const float*)in0, nullptr,
                                                      (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]),
                                                      (const float*)m->tensor(o.t[2]), (const float*)m->tensor(o.t[3]),
                                                      (const float*)m->tensor(o.t[4]), (const float*)m->tensor(o.t[5]),
                                                      p[9] ? m->d_minmax : nullptr, (const float*)m->tensor(o.t[6]),
                                                      (const float*)m->tensor(o.t[7]), B, p[0], p[1], p[4], p[5], 0, p[6], p[7], p[8], p[10]};
                        const bn::DwPwArgs da = dwpw_args(d);
                        if (p[2] == 16 && p[3] == 32 && bn::f32_front2_supported(f, da) && bn::launch_f32_front2(f, da, s)) {
                            front2_done = oi + (size_t)p[BN_OP_FRONT2_DIST];
                            fused_into(front2_done);
                            break;
                        }
                    }
                }
                bn::launch_f32_front((const float*)in0, (float*)out, B, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8],
                                     (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]),
                                     (const float*)m->tensor(o.t[2]), (const float*)m->tensor(o.t[3]),
                                     (const float*)m->tensor(o.t[4]), (const float*)m->tensor(o.t[5]), p[9] ? m->d_minmax : nullptr,
                                     (const float*)m->tensor(o.t[6]), (const float*)m->tensor(o.t[7]), p[10], s);
                break;
            case BN_OP_F32_GAPDENSE:
                bn::launch_f32_gap_dense((const float*)in0, (float*)out, d_logits, B, p[0], p[1], p[2], p[3],
                                         (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]), s);
                break;
            case BN_OP_F32_SEGATE:
                bn::launch_f32_segate((const float*)in0, (float*)out, B, p[0], p[1], p[2], (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]),
                                      oi == gap_for ? m->d_gap_part : nullptr, gap_R, s);
                break;
            case BN_OP_F32_SCALE:
                bn::launch_f32_scale((const float*)in0, (const float*)in1, (float*)out, B, p[0], p[1], s);
                break;
            case BN_OP_F32_GAP:
                bn::launch_f32_gap((const float*)in0, (float*)out, B, p[0], p[1], s);
                break;
            case BN_OP_F32_ATTNPOOL:
                bn::launch_f32_attnpool((const float*)in0, (float*)out, B, p[0], p[1], (const float*)m->tensor(o.t[0]),
                                        s);
                break;
            case BN_OP_F32_DENSE:
                bn::launch_f32_dense((const float*)in0, (float*)out, d_logits, B, p[0], p[1], p[2],
                                     (const float*)m->tensor(o.t[0]), (const float*)m->tensor(o.t[1]), s);
                break;
            case BN_OP_I8_QUANT:
                bn::launch_i8_quant((const float*)in0, mm, (int8_t*)out, B, p[0], p[1], p[2], p[3], p[4], o.f[0], s);
                break;
            case BN_OP_I8_MEL:
                bn::launch_i8_mel((const int8_t*)in0, (int8_t*)out, B, p[0], p[1], p[2], p[3], p[4], p[5],
                                  (const int8_t*)m->tensor(o.t[0]), (const int32_t*)m->tensor(o.t[1]),
                                  (const int32_t*)m->tensor(o.t[2]), (const int32_t*)m->tensor(o.t[3]),
                                  p[6] ? (const int8_t*)m->tensor(o.t[4]) : nullptr, s);
                break;
            case BN_OP_I8_STEM:
            case BN_OP_I8_DW: {
                bn::I8ConvGeom g{p[0], p[1], p[2], p[3], p[4], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13]};
                g.rq_right = m->rq_right[oi];
                if (o.kind == BN_OP_I8_DW) {  // row-streaming form (three loads per input row instead of nine per output) where the shape allows
                    // ... which also adds up what it stores when the squeeze-excite gate's MEAN is the next operator (integer sums: bit-identical)
                    int32_t* pool = nullptr;
                    if (bn::g_opt.i8_dw_pool && m->d_pool8 && oi + 1 < op_end && segate_fused(oi + 1)) {
                        const OpRec& mo = m->ops[oi + 1];
                        if (same_path(o, mo) && mo.in0 == o.out && mo.p[1] == p[2] && mo.p[0] == p[6] * p[7] && (size_t)p[2] <= m->pool8_C) pool = m->d_pool8;
                    }
                    if (bn::launch_i8_dw_stream((const int8_t*)in0, (int8_t*)out, B, g, (const int8_t*)m->tensor(o.t[0]), (const int32_t*)m->tensor(o.t[1]),
                                                (const int32_t*)m->tensor(o.t[2]), (const int32_t*)m->tensor(o.t[3]), s, pool)) {
                        if (pool) pool8_for = oi + 1;
                        break;
                    }
                }
                if (o.kind == BN_OP_I8_STEM &&
                    bn::launch_i8_stem_stream((const int8_t*)in0, (int8_t*)out, B, g, (const int8_t*)m->tensor(o.t[0]), (const int32_t*)m->tensor(o.t[1]),
                                              (const int32_t*)m->tensor(o.t[2]), (const int32_t*)m->tensor(o.t[3]), s))
                    break;
                auto fn = o.kind == BN_OP_I8_STEM ? bn::launch_i8_stem : bn::launch_i8_dw;
                fn((const int8_t*)in0, (int8_t*)out, B, g, (const int8_t*)m->tensor(o.t[0]),
                   (const int32_t*)m->tensor(o.t[1]), (const int32_t*)m->tensor(o.t[2]),
                   (const int32_t*)m->tensor(o.t[3]), s);
                break;
            }
            case BN_OP_I8_PW: {
                bn::I8AddParams add{p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15], p[16]};
                bn::launch_i8_pw((const int8_t*)in0, (const int8_t*)in1, (int8_t*)out, B, p[0], p[1], p[2], p[3], p[4],
                                 p[5], add, (const int8_t*)m->tensor(o.t[0]), (const int32_t*)m->tensor(o.t[1]),
                                 (const int32_t*)m->tensor(o.t[2]), (const int32_t*)m->tensor(o.t[3]), s);
                break;
            }
            case BN_OP_I8_DWPW: {
                bn::DwPw8Args a = dwpw8_args(o, oi);
                if (p[BN_OP_TAIL_TAG] == BN_PWDW8_HEAD && bn::g_opt.i8_pwdw && bn::g_opt.i8_strip && oi + 1 < op_end) {
                    // inverted-residual block of an exported graph: expand convolution + depthwise stage as one kernel (the expanded map stays in LDS)
                    const OpRec& d = m->ops[oi + 1];
                    const int* q = d.p;
                    if (same_path(o, d) && d.kind == BN_OP_I8_DW && q[BN_OP_TAIL_TAG] == BN_PWDW8_COVERED && d.in0 == o.out && d.out != o.in0 && d.out != o.out) {
                        const bn::I8ConvGeom g{q[0], q[1], q[2], q[3], q[4], q[6], q[7], q[8], q[9], q[10], q[11], q[12], q[13]};
                        if (bn::i8_pwdw_supported(a, g) &&
                            bn::launch_i8_pwdw(a, g, (const int8_t*)m->tensor(d.t[0]), (const int32_t*)m->tensor(d.t[1]), (const int32_t*)m->tensor(d.t[2]),
                                               (const int32_t*)m->tensor(d.t[3]), (int8_t*)slot_ptr(d.out), s)) {
                            pwdw_done = oi + 1;
                            fused_into(oi + 1);
                            break;
                        }
                    }
                }
                if (a.transposed && p[36]) {  // QUANTIZE fused into the mel mixer: the input slot holds the float32 spectrogram
                    a.qx = (const float*)in0;
                    a.qminmax = mm;
                    a.qscale = o.f[0];
                    a.qzp = p[37];
                    a.qfill = p[38];
                    a.qF = p[5];
                    a.qtiled = (m->spec_tiled_now && o.in0 == BN_SLOT_INPUT) ? 1 : 0;
                    a.x = nullptr;
                    if (!bn::i8_mel_mfma_supported(a)) return fail(BN_ERR_FORMAT, "operator %zu: fused QUANTIZE needs the mel-mixer kernel's geometry", oi);
                    if (m->guard_now && a.qtiled && mm) {
                        // audio path: list the bytes the float32 STFT leaves in doubt, recompute those elements in float64, run the
                        // blocks whose bytes changed once more (bn_stft_exact.hip)
                        a.qguard = guard_slice(m, slot_b0);
                        a.qmode = 1;
                        bn::launch_i8_dwpw(a, s);
                        prof.end();  // (the operator's own launch; the float64 pass has its own profiling entry)
                        ProfScope fix(m, (int)m->ops.size() + 2, s);
                        bn::launch_stft_fix(m->ctx->tables, m->guard_audio, B, m->guard_T, m->guard_hop, a.W, (float*)in0, true, a.qguard, mm, a.qscale,
                                            a.qzp, s);
                        a.qmode = 2;
                        bn::launch_i8_dwpw(a, s);
                        break;
                    }
                }
                // wide early layers: wave-autonomous strip kernel when the packer prepared its constant block
                if (p[35] && o.t[9] >= 0 && bn::g_opt.i8_strip && a.has_dw && !a.transposed && a.sh == a.sw &&
                    bn::i8_strip_supported(a.Cin, a.Cout, a.sh, a.OW, a.add.enabled != 0) &&
                    (!a.add.enabled || (a.res == a.x && o.t[10] >= 0))) {
                    const int off = a.add.enabled ? 128 : 0;
                    bn::Strip8Args q{a.x, a.y, (const int32_t*)m->tensor(o.t[9]), B, a.H, a.W, a.OH, a.OW, 0, a.pt, a.pl,
                                     a.dw_zp_in, a.dw_amin, a.dw_amax, a.pw_amin + off, a.pw_amax + off, a.pw_zp_out, a.add,
                                     a.add.enabled ? (const int8_t*)m->tensor(o.t[10]) : nullptr};
                    bn::launch_i8_strip(q, a.Cin, a.Cout, a.sh, s);
                    break;
                }
                if (!bn::i8_dwpw_supported(a.Cin, a.Cout) || a.TH * a.TW * a.NB != 64 || a.OH % a.TH || a.OW % a.TW)
                    return fail(BN_ERR_FORMAT, "operator %zu: unsupported fused INT8 block geometry", oi);
                bn::launch_i8_dwpw(a, s);
                break;
            }
            case BN_OP_I8_FRONT: {
                bn::I8FrontParams q{};
                q.stem_w = (const int8_t*)m->tensor(o.t[0]); q.stem_b = (const int32_t*)m->tensor(o.t[1]);
                q.stem_mult = (const int32_t*)m->tensor(o.t[2]); q.stem_shift = (const int32_t*)m->tensor(o.t[3]);
                q.dw_w = (const int8_t*)m->tensor(o.t[4]); q.dw_b = (const int32_t*)m->tensor(o.t[5]);
                q.dw_mult = (const int32_t*)m->tensor(o.t[6]); q.dw_shift = (const int32_t*)m->tensor(o.t[7]);
                q.pw_w = (const int8_t*)m->tensor(o.t[8]); q.pw_b = (const int32_t*)m->tensor(o.t[9]);
                q.pw_mult = (const int32_t*)m->tensor(o.t[10]); q.pw_shift = (const int32_t*)m->tensor(o.t[11]);
                q.H0 = p[0]; q.W0 = p[1]; q.C = p[2]; q.N = p[3]; q.OH = p[4]; q.OW = p[5];
                q.stem_zp_in = p[6]; q.stem_zp_out = p[7]; q.stem_amin = p[8]; q.stem_amax = p[9];
                q.dw_zp_out = p[10]; q.dw_amin = p[11]; q.dw_amax = p[12]; q.pw_zp_out = p[13]; q.pw_amin = p[14]; q.pw_amax = p[15];
                q.rq_right = m->rq_right[oi];
                if (p[16] && o.t[12] >= 0 && bn::g_opt.i8_strip && bn::i8_front_strip_supported(q.H0, q.W0, q.C, q.N, q.OH, q.OW)) {
                    bn::FrontStrip8Args fa{(const int8_t*)in0, (int8_t*)out, (const int32_t*)m->tensor(o.t[12]), B, q.H0, q.W0, q.OH, q.OW, 0,
                                           q.stem_zp_in, q.stem_amin, q.stem_amax, q.stem_zp_out, q.dw_amin, q.dw_amax, q.pw_amin, q.pw_amax};
                    bn::launch_i8_front_strip(fa, s);
                    break;
                }
                if (!bn::i8_front_supported(q.H0, q.W0, q.C, q.N, q.OH, q.OW))
                    return fail(BN_ERR_FORMAT, "operator %zu: unsupported INT8 front-block geometry", oi);
                bn::launch_i8_front(q, (const int8_t*)in0, (int8_t*)out, B, s);
                break;
            }
            case BN_OP_I8_MID: {
                bn::Tail2Args ma = m->mids[oi];
                ma.x = (const int8_t*)in0;
                ma.y = (int8_t*)out;
                ma.cst = (const int32_t*)m->tensor(o.t[0]);
                ma.B = B;
                if (!bn::launch_i8_mid2(ma, s)) return fail(BN_ERR_DEVICE, "could not raise the LDS limit of the fused stage-2 kernel");
                break;
            }
            case BN_OP_I8_TAIL: {
                if (bn::g_opt.i8_tail_mfdw && m->tail2_ok[oi]) {
                    bn::Tail2Args t2 = m->tails2[oi];
                    t2.x = (const int8_t*)in0;
                    t2.scores = d_scores;
                    t2.logits = d_logits;
                    t2.cst = (const int32_t*)m->tensor(o.t[2]);
                    t2.B = B;
                    if (!bn::launch_i8_tail2(t2, s)) return fail(BN_ERR_DEVICE, "could not raise the LDS limit of the fused tail kernel");
                    break;
                }
                bn::Tail8Args ta = m->tails[oi];
                ta.x = (const int8_t*)in0;
                ta.scores = d_scores;
                ta.logits = d_logits;
                ta.cst = (const int32_t*)m->tensor(o.t[0]);
                ta.B = B;
                if (!bn::launch_i8_tail(ta, s)) return fail(BN_ERR_DEVICE, "could not raise the LDS limit of the fused tail kernel");
                break;
            }
            case BN_OP_I8_MEAN:
                if (segate_fused(oi)) {
                    const OpRec& f1 = m->ops[oi + 1];
                    const OpRec& f2 = m->ops[oi + 2];
                    bn::launch_i8_segate((const int8_t*)in0, (int8_t*)slot_ptr(f2.out), B, p[0], p[1], p[2], p[3], p[4], p[5], f1.p[1], f1.p[2], f1.p[3], f1.p[4],
                                         (const int8_t*)m->tensor(f1.t[0]), (const int32_t*)m->tensor(f1.t[1]), (const int32_t*)m->tensor(f1.t[2]),
                                         (const int32_t*)m->tensor(f1.t[3]), f1.p[5] ? (const int8_t*)m->tensor(f1.t[4]) : nullptr, f2.p[2], f2.p[3], f2.p[4],
                                         (const int8_t*)m->tensor(f2.t[0]), (const int32_t*)m->tensor(f2.t[1]), (const int32_t*)m->tensor(f2.t[2]),
                                         (const int32_t*)m->tensor(f2.t[3]), f2.p[5] ? (const int8_t*)m->tensor(f2.t[4]) : nullptr, s,
                                         oi == pool8_for ? m->d_pool8 : nullptr);
                    segate_done[0] = oi + 1;
                    segate_done[1] = oi + 2;
                    fused_into(oi + 2);
                    break;
                }
                bn::launch_i8_mean((const int8_t*)in0, (int8_t*)out, B, p[0], p[1], p[2], p[3], p[4], p[5], s);
                break;
            case BN_OP_I8_FC:
                bn::launch_i8_fc((const int8_t*)in0, (int8_t*)out, B, p[0], p[1], p[2], p[3], p[4],
                                 (const int8_t*)m->tensor(o.t[0]), (const int32_t*)m->tensor(o.t[1]),
                                 (const int32_t*)m->tensor(o.t[2]), (const int32_t*)m->tensor(o.t[3]),
                                 p[5] ? (const int8_t*)m->tensor(o.t[4]) : nullptr, s);
                break;
            case BN_OP_I8_ATTNPOOL:
                if (!bn::launch_i8_attnpool((const int8_t*)in0, (int8_t*)out, B, p, (const int8_t*)m->tensor(o.t[0]), (const int32_t*)m->tensor(o.t[1]), s))
                    return fail(BN_ERR_FORMAT, "operator %zu: attention pooling geometry", oi);
                break;
            case BN_OP_I8_SCALE:
                if (p[BN_OP_TAIL_TAG] == BN_SCALE_HEAD && bn::g_opt.i8_strip && oi + 1 < op_end) {
                    // the projection convolution behind the gate applies it while loading (the scaled map is never written)
                    const OpRec& d = m->ops[oi + 1];
                    if (same_path(o, d) && d.kind == BN_OP_I8_DWPW && d.p[BN_OP_TAIL_TAG] == BN_SCALE_COVERED && d.in0 == o.out && d.out != o.in0 && d.out != o.in1 &&
                        d.p[2] == p[1] && d.p[6] * d.p[7] == p[0]) {
                        bn::DwPw8Args a2 = dwpw8_args(d, oi + 1);
                        a2.x = (const int8_t*)in0;
                        a2.gate = (const int8_t*)in1;
                        a2.g_zx = p[2]; a2.g_zg = p[3]; a2.g_mult = p[4]; a2.g_shift = p[5]; a2.g_zo = p[6]; a2.g_amin = p[7]; a2.g_amax = p[8];
                        if (!a2.has_dw && !a2.transposed && bn::i8_pw_wave_takes(a2)) {
                            bn::launch_i8_dwpw(a2, s);
                            scale_done = oi + 1;
                            fused_into(oi + 1);
                            break;
                        }
                    }
                }
                bn::launch_i8_scale((const int8_t*)in0, (const int8_t*)in1, (int8_t*)out, B, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], s);
                break;
            case BN_OP_I8_MAXNORM:
                bn::launch_i8_maxnorm((const int8_t*)in0, (int8_t*)out, B, p[0], p[1], (const int8_t*)m->tensor(o.t[0]), (const int8_t*)m->tensor(o.t[1]),
                                      p[2] ? (const int8_t*)m->tensor(o.t[2]) : nullptr, s);
                break;
            case BN_OP_I8_RAWFE:
                bn::launch_i8_rawfe((const float*)in0, (int8_t*)out, B, p[0], p[1], p[2], p[3], p[4], o.f[0], p[5], p[6], p[7], p[8],
                                    (const int8_t*)m->tensor(o.t[0]), (const int32_t*)m->tensor(o.t[1]), (const int32_t*)m->tensor(o.t[2]),
                                    (const int32_t*)m->tensor(o.t[3]), p[9] ? (const int8_t*)m->tensor(o.t[4]) : nullptr, s);
                break;
            case BN_OP_I8_HEAD:
                if (p[4])  // float32 softmax behind DEQUANTIZE
                    bn::launch_i8_head_softmax((const int8_t*)in0, d_scores, d_logits, B, p[0], p[1], o.f[0], o.f[2], s);
                else
                    bn::launch_i8_head((const int8_t*)in0, d_scores, d_logits, B, p[0], p[1], p[2], o.f[0], o.f[1],
                                       p[3] ? (const int8_t*)m->tensor(o.t[0]) : nullptr, s);
                break;
            default:
                return fail(BN_ERR_UNSUPPORTED, "plan operator %zu has unknown kind %d", oi, o.kind);
        }
    }
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

}  // namespace

extern "C" {

int bn_version(void) { return BN_ABI_VERSION; }

const char* bn_last_error(void) { return g_err.c_str(); }

int bn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bn_ctx_create(int device, int max_batch, bn_ctx** out) {
    if (!out) return fail(BN_ERR_ARG, "out is null");
    *out = nullptr;
    if (max_batch <= 0) return fail(BN_ERR_ARG, "max_batch must be positive, got %d", max_batch);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(BN_ERR_DEVICE, "no HIP device is visible; libbirdnet_hip has no CPU fallback");
    if (device < 0 || device >= n) return fail(BN_ERR_ARG, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BN_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

    bn_ctx* c = new bn_ctx();
    struct Guard {  // a failing HIP call below returns early: release what has been allocated so far
        bn_ctx* c;
        ~Guard() {
            if (c) bn_ctx_destroy(c);
        }
    } guard{c};
    c->device = device;
    c->max_batch = max_batch;
    // STFT tables in double, rounded once to float32
    std::vector<float> win(kFft);
    std::vector<float4> t256(256), t512(257);
    const double two_pi = 6.283185307179586476925286766559;
    // per-lane base angles; the kernel rebuilds window and split-pass twiddles from them (see bn_stft.hip)
    win.assign(64, 0.0f);
    for (int j = 0; j < 16; ++j) {
        const double t0 = two_pi * (2 * j) / 512.0, t1 = two_pi * (2 * j + 1) / 512.0;
        win[4 * j + 0] = (float)(-0.25 * cos(t0));
        win[4 * j + 1] = (float)(-0.25 * cos(t1));
        win[4 * j + 2] = (float)(0.25 * sin(t0));
        win[4 * j + 3] = (float)(0.25 * sin(t1));
    }
    for (int i = 0; i < 256; ++i) {  // w = exp(-2 pi i p / 256), stored with its rotation (-w.y, w.x)
        const float c = (float)cos(two_pi * i / 256.0), sn = (float)sin(two_pi * i / 256.0);
        t256[i] = make_float4(c, -sn, sn, c);
    }
    t512.assign(16, make_float4(0, 0, 0, 0));
    for (int j = 0; j < 16; ++j) {
        const float c = (float)cos(two_pi * j / 512.0), sn = (float)sin(two_pi * j / 512.0);
        t512[j] = make_float4(-sn, -c, -c, sn);
    }
    HIP_TRY(hipMalloc(&c->d_window, win.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&c->d_tw256, t256.size() * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->d_tw512, t512.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(c->d_window, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_tw256, t256.data(), t256.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_tw512, t512.data(), t512.size() * sizeof(float4), hipMemcpyHostToDevice));
    // float64 tables of the exactness pass: cos(2 pi j / 512) with the symmetries exact (cs[128] = 0, cs[j] = -cs[256 - j], ...),
    // the reference's periodic Hann window 0.5 - 0.5 cos(2 pi n / 512) from it
    std::vector<double> f64tab(2048);
    for (int j = 0; j < 512; ++j) {
        const int a2 = j <= 256 ? j : 512 - j;                 // cos is even about pi
        const int a3 = a2 <= 128 ? a2 : 256 - a2;              // and odd about pi / 2
        const double v = a3 <= 64 ? cos(two_pi * a3 / 512.0) : sin(two_pi * (128 - a3) / 512.0);
        f64tab[512 + j] = a2 <= 128 ? v : -v;
        if (a3 == 128) f64tab[512 + j] = 0.0;
    }
    for (int n = 0; n < 512; ++n) f64tab[n] = 0.5 - 0.5 * f64tab[512 + n];
    for (int j = 0; j < 512; ++j) {  // (cos, sin) pairs: sin(a) = cos(a - pi / 2)
        f64tab[1024 + 2 * j] = f64tab[512 + j];
        f64tab[1024 + 2 * j + 1] = f64tab[512 + ((j + 384) & 511)];
    }
    HIP_TRY(hipMalloc(&c->d_f64tab, f64tab.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(c->d_f64tab, f64tab.data(), f64tab.size() * sizeof(double), hipMemcpyHostToDevice));
    c->tables = bn::StftTables{c->d_window, c->d_tw256, c->d_tw512, c->d_f64tab, c->d_f64tab + 512, reinterpret_cast<const double2*>(c->d_f64tab + 1024)};
    // bn_ingest_resample's per-workgroup peak scratch (one float per >= 1024 resampled samples): sized here for windows that
    // yield max_batch 3 s chunks at 24 kHz (72 blocks per chunk) with headroom, so that the ingest call itself does not allocate
    // (no hidden device sync on that path); a call that needs more still grows it, once.
    c->block_peaks_elems = (size_t)max_batch * 128 + 65536;
    HIP_TRY(hipMalloc(&c->d_block_peaks, c->block_peaks_elems * sizeof(float)));
    guard.c = nullptr;
    *out = c;
    return BN_OK;
}

void bn_ctx_destroy(bn_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->d_window);
    (void)hipFree(c->d_tw256);
    (void)hipFree(c->d_tw512);
    (void)hipFree(c->d_f64tab);
    (void)hipFree(c->d_block_peaks);
    if (c->d_rank_work) (void)hipFree(c->d_rank_work);
    delete c;
}

// Parse and validate a packed blob (host only): tables inside the blob, payloads aligned and in range, operator references in
// range, every operator's geometry against the slot and tensor sizes (bn_plan_check.hip).
static int parse_blob(const void* blob, size_t nbytes, BlobHeader& h, std::vector<SlotRec>& slots, std::vector<TensorRec>& tensors,
                      std::vector<OpRec>& ops) {
    if (!blob || nbytes < sizeof(BlobHeader)) return fail(BN_ERR_FORMAT, "blob too small (%zu bytes)", nbytes);
    const char* base = (const char*)blob;
    memcpy(&h, base, sizeof h);
    if (memcmp(h.magic, BN_BLOB_MAGIC, 8) != 0) return fail(BN_ERR_FORMAT, "bad blob magic");
    if (h.version != BN_BLOB_VERSION)
        return fail(BN_ERR_FORMAT, "blob version %u, library expects %u", h.version, BN_BLOB_VERSION);
    auto in_range = [&](uint64_t off, uint64_t len) { return off <= nbytes && len <= nbytes - off; };
    if (!in_range(h.slots_off, (uint64_t)h.n_slots * sizeof(SlotRec)) ||
        !in_range(h.tensors_off, (uint64_t)h.n_tensors * sizeof(TensorRec)) ||
        !in_range(h.ops_off, (uint64_t)h.n_ops * sizeof(OpRec)))
        return fail(BN_ERR_FORMAT, "blob tables exceed the blob size");
    slots.resize(h.n_slots);
    tensors.resize(h.n_tensors);
    ops.resize(h.n_ops);
    if (h.n_slots) memcpy(slots.data(), base + h.slots_off, h.n_slots * sizeof(SlotRec));
    if (h.n_tensors) memcpy(tensors.data(), base + h.tensors_off, h.n_tensors * sizeof(TensorRec));
    if (h.n_ops) memcpy(ops.data(), base + h.ops_off, h.n_ops * sizeof(OpRec));
    for (const TensorRec& t : tensors)
        if (!in_range(t.offset, t.nbytes) || (t.offset & 255)) return fail(BN_ERR_FORMAT, "tensor payload out of range or misaligned");
    for (const SlotRec& sl : slots)
        if (sl.bytes_per_chunk > (1ull << 32)) return fail(BN_ERR_FORMAT, "slot of %llu bytes per chunk", (unsigned long long)sl.bytes_per_chunk);
    for (const OpRec& o : ops) {
        for (int k = 0; k < BN_OP_NT; ++k)
            if (o.t[k] >= (int)h.n_tensors) return fail(BN_ERR_FORMAT, "operator references tensor %d of %u", o.t[k], h.n_tensors);
        const int ids[3] = {o.in0, o.in1, o.out};
        for (int id : ids)
            if (id >= (int)h.n_slots || (id < 0 && id != BN_SLOT_INPUT && id != BN_SLOT_SCORES && id != BN_SLOT_LOGITS && id != BN_SLOT_AUDIO && id != BN_SLOT_NONE))
                return fail(BN_ERR_FORMAT, "operator references slot %d of %u", id, h.n_slots);
    }
    std::string why;
    if (!bn::check_plan(h, slots, tensors, ops, why)) return fail(BN_ERR_FORMAT, "%s", why.c_str());
    return BN_OK;
}

int bn_blob_check(const void* blob, size_t nbytes) {
    BlobHeader h;
    std::vector<SlotRec> slots;
    std::vector<TensorRec> tensors;
    std::vector<OpRec> ops;
    return parse_blob(blob, nbytes, h, slots, tensors, ops);
}

int bn_model_load(bn_ctx* ctx, const void* blob, size_t nbytes, bn_model** out) {
    if (!out) return fail(BN_ERR_ARG, "out is null");
    *out = nullptr;
    if (int rc = check_device(ctx)) return rc;
    bn_model* m = new bn_model();
    m->ctx = ctx;
    if (int rc = parse_blob(blob, nbytes, m->hdr, m->slots, m->tensors, m->ops)) {
        delete m;
        return rc;
    }
    const BlobHeader& h = m->hdr;
    const char* base = (const char*)blob;
    size_t lo = nbytes, hi = 0;
    for (const TensorRec& t : m->tensors)
        if (t.nbytes) {
            lo = t.offset < lo ? (size_t)t.offset : lo;
            hi = t.offset + t.nbytes > hi ? (size_t)(t.offset + t.nbytes) : hi;
        }
    // fused tail operators: build the kernel arguments and the LDS plan from the descriptor table
    m->tails.resize(h.n_ops);
    m->tail_ok.assign(h.n_ops, 0);
    m->tails2.resize(h.n_ops);
    m->tail2_ok.assign(h.n_ops, 0);
    m->mids.resize(h.n_ops);
    m->mid_ok.assign(h.n_ops, 0);
    for (size_t oi = 0; oi < m->ops.size(); ++oi) {
        const OpRec& o = m->ops[oi];
        if (o.kind != BN_OP_I8_MID) continue;
        bn::Tail2Args& ma = m->mids[oi];
        ma = bn::Tail2Args{};
        const TensorRec& td = m->tensors[o.t[1]];
        const TensorRec& tc = m->tensors[o.t[0]];
        const bool ok = (td.nbytes & 3) == 0 && bn::tail2_plan((const int32_t*)(base + td.offset), (int)(td.nbytes / 4), o.p[5], ma, true) &&
                        ma.L[0].H == o.p[6] && ma.L[0].W == o.p[7] && ma.L[0].Cin == o.p[8] && bn::tail2_const_words(ma, true) * 4 <= (long)tc.nbytes &&
                        ma.L[o.p[5] - 1].OH * ma.L[o.p[5] - 1].OW == o.p[9] && ma.L[o.p[5] - 1].Cout == o.p[10];
        m->mid_ok[oi] = ok;
        m->has_mid = m->has_mid || ok;
    }
    for (size_t oi = 0; oi < m->ops.size(); ++oi) {
        const OpRec& o = m->ops[oi];
        if (o.kind != BN_OP_I8_TAIL) continue;
        bn::Tail8Args& ta = m->tails[oi];
        ta = bn::Tail8Args{};
        ta.NC = o.p[4];
        ta.s_fc = o.f[0];
        ta.s_head = o.f[1];
        const TensorRec& td = m->tensors[o.t[1]];
        const TensorRec& tc = m->tensors[o.t[0]];
        const bool ok = (td.nbytes & 3) == 0 && bn::tail_plan((const int32_t*)(base + td.offset), (int)(td.nbytes / 4), o.p[5], ta) &&
                        ta.L[0].H == o.p[6] && ta.L[0].W == o.p[7] && ta.L[0].Cin == o.p[8] && bn::tail_const_words(ta) * 4 <= (long)tc.nbytes;
        m->tail_ok[oi] = ok;
        m->has_tail = m->has_tail || ok;
        // the second form's constants (t[2], t[3]) are optional; it only ever runs where the first form could (same coverage, same fallback)
        if (ok && o.t[2] >= 0 && o.t[3] >= 0 && (size_t)o.t[2] < m->tensors.size() && (size_t)o.t[3] < m->tensors.size()) {
            bn::Tail2Args& t2 = m->tails2[oi];
            t2 = bn::Tail2Args{};
            t2.NC = o.p[4];
            t2.s_fc = o.f[0];
            t2.s_head = o.f[1];
            const TensorRec& td2 = m->tensors[o.t[3]];
            const TensorRec& tc2 = m->tensors[o.t[2]];
            m->tail2_ok[oi] = (td2.nbytes & 3) == 0 && bn::tail2_plan((const int32_t*)(base + td2.offset), (int)(td2.nbytes / 4), o.p[5], t2) &&
                              t2.L[0].H == o.p[6] && t2.L[0].W == o.p[7] && t2.L[0].Cin == o.p[8] && bn::tail2_const_words(t2) * 4 <= (long)tc2.nbytes;
        }
    }
    // INT8 blocks: can every requantisation of the operator take the branch-free right-shift form?
    m->rq_right.assign(h.n_ops, 0);
    for (const OpRec& o : m->ops)
        if (o.in0 == BN_SLOT_INPUT) {
            m->spec_tiled_ok = o.kind == BN_OP_I8_DWPW && o.p[36] && o.p[30] && o.p[1] % 64 == 0;
            m->guard_form_ok = m->spec_tiled_ok && o.p[37] == -128;
        }
    for (size_t oi = 0; oi < m->ops.size(); ++oi) {
        const OpRec& o = m->ops[oi];
        auto all_right = [&](int t_mult, int t_shift) {
            if (t_mult < 0 || t_shift < 0) return false;
            const TensorRec& tm = m->tensors[t_mult];
            const TensorRec& ts = m->tensors[t_shift];
            const int32_t* pm = (const int32_t*)(base + tm.offset);
            const int32_t* ps = (const int32_t*)(base + ts.offset);
            if ((tm.nbytes | ts.nbytes) & 3) return false;
            for (size_t i = 0; i < tm.nbytes / 4; ++i)
                if (pm[i] < 0) return false;
            for (size_t i = 0; i < ts.nbytes / 4; ++i)
                if (ps[i] >= 0) return false;
            return true;
        };
        if (o.kind == BN_OP_I8_DWPW) {
            const int* p = o.p;
            const bool pw_ok = all_right(o.t[6], o.t[7]);  // (bit 2: the pointwise stage alone, whatever the ADD behind it looks like)
            bool ok = pw_ok && (!p[29] || all_right(o.t[2], o.t[3]));
            if (p[18]) ok = ok && p[20] >= 0 && p[21] < 0 && p[22] >= 0 && p[23] < 0 && p[24] >= 0 && p[25] < 0;  // ADD: m1 s1 m2 s2 mo so
            // bit 1: every pointwise shift lies in [-20, -1] — the 64-bit addend of the one-multiply-add requantisation (i8_pw_lds_kernel) cannot overflow
            bool narrow = pw_ok && o.t[7] >= 0;
            if (narrow) {
                const TensorRec& ts = m->tensors[o.t[7]];
                const int32_t* ps = (const int32_t*)(base + ts.offset);
                for (size_t i = 0; i < ts.nbytes / 4; ++i) narrow = narrow && ps[i] >= -20;
            }
            m->rq_right[oi] = (ok ? 1 : 0) | (narrow ? 2 : 0) | (pw_ok ? 4 : 0);
        } else if (o.kind == BN_OP_I8_DW || o.kind == BN_OP_I8_STEM) {
            // bit 0: multipliers >= 0, right shifts; bit 1: every shift >= -20 (the 64-bit addend of the one-multiply-add form cannot overflow)
            bool narrow = o.t[3] >= 0;
            if (narrow) {
                const TensorRec& ts = m->tensors[o.t[3]];
                const int32_t* ps = (const int32_t*)(base + ts.offset);
                for (size_t i = 0; i < ts.nbytes / 4; ++i) narrow = narrow && ps[i] >= -20;
            }
            m->rq_right[oi] = (all_right(o.t[2], o.t[3]) ? 1 : 0) | (narrow ? 2 : 0);
        } else if (o.kind == BN_OP_I8_FRONT) {
            m->rq_right[oi] = all_right(o.t[2], o.t[3]) && all_right(o.t[6], o.t[7]) && all_right(o.t[10], o.t[11]);
        }
    }
    auto cleanup_fail = [&](int code) {
        bn_model_free(m);
        return code;
    };
    if (hi > lo) {
        m->consts_base = lo;
        m->consts_bytes = hi - lo;
        if (hipMalloc(&m->d_consts, m->consts_bytes) != hipSuccess)
            return cleanup_fail(fail(BN_ERR_NOMEM, "hipMalloc of %zu constant bytes failed", m->consts_bytes));
        if (hipMemcpy(m->d_consts, base + lo, m->consts_bytes, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup_fail(fail(BN_ERR_DEVICE, "copying constants to the device failed"));
    }
    const size_t mb = (size_t)ctx->max_batch;
    m->d_slots.assign(h.n_slots, nullptr);
    for (uint32_t i = 0; i < h.n_slots; ++i) {
        const size_t bytes = mb * m->slots[i].bytes_per_chunk;
        if (bytes == 0) continue;
        if (hipMalloc(&m->d_slots[i], bytes) != hipSuccess)
            return cleanup_fail(fail(BN_ERR_NOMEM, "hipMalloc of %zu workspace bytes (slot %u) failed", bytes, i));
        m->workspace_bytes += bytes;
    }
    if (h.input_kind == BN_INPUT_SPECTROGRAM) {
        const size_t bytes = mb * h.input_elems * sizeof(float);
        if (hipMalloc(&m->d_spec, bytes) != hipSuccess)
            return cleanup_fail(fail(BN_ERR_NOMEM, "hipMalloc of %zu spectrogram workspace bytes failed", bytes));
        m->workspace_bytes += bytes;
    }
    if (m->guard_form_ok && h.dtype == BN_DTYPE_I8) {
        // exactness pass of the audio path (plans with another zero point take the float64 STFT for every bin): per chunk W bounds, W / 16 tile records, a list of flagged elements, counters
        const size_t W = h.spec_width, n_tiles = (W + 15) / 16, t64 = (W + 63) / 64;
        const int cap = 1 << 20;  // a chunk's count beyond this = one of its mel-mixer workgroups gave up (more in doubt than it keeps)
        size_t off = 0;
        auto take = [&](size_t bytes) {
            const size_t o = off;
            off += (bytes + 255) & ~(size_t)255;
            return o;
        };
        const size_t o_eps = take(mb * W * 4), o_rec = take(mb * n_tiles * bn::kGuardRec * 4), o_list = take(mb * 4), o_cnt = take(mb * 4),
                     o_dirty = take(mb * 4), o_work = take(mb * t64 * 4), o_nw = take(4 * (mb / kMaxGridBatch + 1)), o_hard = take(2 * mb * 4),
                     o_nh = take(8 * (mb / kMaxGridBatch + 1)), o_audit = take(8);
        if (hipMalloc(&m->d_guard, off) != hipSuccess) return cleanup_fail(fail(BN_ERR_NOMEM, "hipMalloc of %zu exactness-pass bytes failed", off));
        m->workspace_bytes += off;
        char* g = m->d_guard;
        m->guard = bn::StftGuard{(float*)(g + o_eps), (int*)(g + o_rec), (float*)(g + o_list), (int*)(g + o_cnt), cap, (int*)(g + o_dirty),
                                 (int*)(g + o_work), (int*)(g + o_nw), (int*)(g + o_hard), (int*)(g + o_nh), (int)mb};
        m->d_audit = (int*)(g + o_audit);
    }
    if (hipMalloc(&m->d_minmax, mb * 2 * sizeof(float)) != hipSuccess ||
        hipMalloc(&m->d_smax, mb * sizeof(float)) != hipSuccess)
        return cleanup_fail(fail(BN_ERR_NOMEM, "hipMalloc of reduction scratch failed"));
    m->workspace_bytes += mb * 3 * sizeof(float);
    for (size_t i = 0; i + 1 < m->ops.size(); ++i)  // row-block channel sums of fused inverted-residual pairs (largest pair decides)
        if (m->ops[i].kind == BN_OP_F32_DWPW && m->ops[i].p[BN_OP_TAIL_TAG] == BN_PWDW_HEAD && m->ops[i + 1].kind == BN_OP_F32_DW) {
            const int* q = m->ops[i + 1].p;
            const int rb = bn::f32_pwdw_rows(q[6]);
            const size_t need = (size_t)((q[6] + rb - 1) / rb) * (size_t)q[2];
            if (need > m->gap_part_elems) m->gap_part_elems = need;
        }
    for (size_t i = 0; i + 1 < m->ops.size(); ++i)  // stand-alone depthwise stage -> gate: one partial sum per strip (at most OH / 4 row blocks)
        if (m->ops[i].kind == BN_OP_F32_DW && m->ops[i + 1].kind == BN_OP_F32_SEGATE && m->ops[i + 1].in0 == m->ops[i].out) {
            const int* q = m->ops[i].p;
            int cq = 16;
            while ((q[2] / 4) % cq) cq >>= 1;
            const int ncol = 64 / cq;
            const size_t need = (size_t)((q[7] + ncol - 1) / ncol) * (size_t)((q[6] + 3) / 4) * (size_t)q[2];
            if (need > m->gap_part_elems) m->gap_part_elems = need;
        }
    for (size_t i = 0; i + 1 < m->ops.size(); ++i)  // INT8 depthwise stage -> MEAN of a squeeze-excite gate: channel sums taken on the way out
        if (m->ops[i].kind == BN_OP_I8_DW && m->ops[i + 1].kind == BN_OP_I8_MEAN && m->ops[i + 1].in0 == m->ops[i].out && m->ops[i + 1].p[1] == m->ops[i].p[2])
            if ((size_t)m->ops[i].p[2] > m->pool8_C) m->pool8_C = (size_t)m->ops[i].p[2];
    if (m->pool8_C) {
        if (hipMalloc(&m->d_pool8, mb * m->pool8_C * sizeof(int32_t)) != hipSuccess)
            return cleanup_fail(fail(BN_ERR_NOMEM, "hipMalloc of pooling scratch failed"));
        if (hipMemset(m->d_pool8, 0, mb * m->pool8_C * sizeof(int32_t)) != hipSuccess) return cleanup_fail(fail(BN_ERR_DEVICE, "clearing the pooling scratch failed"));
        m->workspace_bytes += mb * m->pool8_C * sizeof(int32_t);
    }
    if (m->gap_part_elems) {
        if (hipMalloc(&m->d_gap_part, mb * m->gap_part_elems * sizeof(float)) != hipSuccess)
            return cleanup_fail(fail(BN_ERR_NOMEM, "hipMalloc of pooling scratch failed"));
        m->workspace_bytes += mb * m->gap_part_elems * sizeof(float);
    }
    *out = m;
    return BN_OK;
}

void bn_model_free(bn_model* m) {
    if (!m) return;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    (void)hipFree(m->d_consts);
    for (char* p : m->d_slots) (void)hipFree(p);
    (void)hipFree(m->d_spec);
    (void)hipFree(m->d_minmax);
    (void)hipFree(m->d_guard);
    (void)hipFree(m->d_smax);
    (void)hipFree(m->d_gap_part);
    (void)hipFree(m->d_pool8);
    for (auto& r : m->ev_used) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    for (hipEvent_t e : m->ev_free) (void)hipEventDestroy(e);
    delete m;
}

int bn_model_get_info(const bn_model* m, bn_model_info* out) {
    if (!m || !out) return fail(BN_ERR_ARG, "null argument");
    out->dtype = (int32_t)m->hdr.dtype;
    out->input_kind = (int32_t)m->hdr.input_kind;
    out->input_elems = (int32_t)m->hdr.input_elems;
    out->fft_bins = (int32_t)m->hdr.fft_bins;
    out->spec_width = (int32_t)m->hdr.spec_width;
    out->num_classes = (int32_t)m->hdr.num_classes;
    out->n_ops = (int32_t)m->hdr.n_ops;
    out->max_batch = m->ctx->max_batch;
    out->workspace_bytes = (int64_t)m->workspace_bytes;
    out->const_bytes = (int64_t)m->consts_bytes;
    return BN_OK;
}

static int stft_mag_impl(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W, int normalize, float* d_spec,
                         float* d_minmax, void* stream, bool tile_major, bool exact = false);

int bn_stft_mag(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W, int normalize,
                float* d_spec, float* d_minmax, void* stream) {
    return stft_mag_impl(ctx, d_audio, B, T, n_fft, hop, W, normalize, d_spec, d_minmax, stream, false);
}

int bn_stft_mag_exact(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W, int normalize,
                      float* d_spec, float* d_minmax, void* stream) {
    return stft_mag_impl(ctx, d_audio, B, T, n_fft, hop, W, normalize, d_spec, d_minmax, stream, false, true);
}

// tile_major: spectrogram as [W/16][257][16] per chunk (private to bn_infer_audio; the public entry point keeps [257][W])
static int stft_mag_impl(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W, int normalize, float* d_spec,
                         float* d_minmax, void* stream, bool tile_major, bool exact) {
    if (int rc = check_device(ctx)) return rc;
    if (!d_audio || !d_spec || !d_minmax) return fail(BN_ERR_ARG, "null device pointer");
    if (n_fft != kFft) return fail(BN_ERR_UNSUPPORTED, "n_fft=%d: only 512 is implemented", n_fft);
    if (B < 0 || T <= 0 || W <= 0 || hop <= 0) return fail(BN_ERR_ARG, "bad shape B=%d T=%d hop=%d W=%d", B, T, hop, W);
    if (1 + T / hop < W)
        return fail(BN_ERR_ARG, "T=%d hop=%d gives %d frames, fewer than spec_width=%d", T, hop, 1 + T / hop, W);
    if (B == 0) return BN_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t per_chunk = (size_t)(kFft / 2 + 1) * W;
    for (int b0 = 0; b0 < B; b0 += kMaxGridBatch) {
        const int nb = B - b0 < kMaxGridBatch ? B - b0 : kMaxGridBatch;
        bn::launch_minmax_init(d_minmax + 2 * (size_t)b0, nb, s);
        if (exact)  // every bin as a float64 DFT: the reference's values (bn_stft_exact.hip)
            bn::launch_stft512_f64(ctx->tables, d_audio + (size_t)b0 * T, nb, T, hop, W, d_spec + b0 * per_chunk, d_minmax + 2 * (size_t)b0, s,
                                   tile_major);
        else
            bn::launch_stft512(ctx->tables, d_audio + (size_t)b0 * T, nb, T, hop, W, d_spec + b0 * per_chunk,
                               d_minmax + 2 * (size_t)b0, s, tile_major);
        if (normalize)
            bn::launch_spec_normalize(d_spec + b0 * per_chunk, d_minmax + 2 * (size_t)b0, nb, (int)per_chunk, s);
    }
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_mel_spectrogram(bn_ctx* ctx, const float* d_audio, int B, int T, int n_fft, int hop, int W, const float* d_mel_w,
                       const int32_t* d_mel_bands, int n_mels, int mode, int mag_scale, double pcen_b, const float* d_dct,
                       int n_mfcc, float* d_work, float* d_out, void* stream) {
    if (int rc = check_device(ctx)) return rc;
    if (!d_audio || !d_mel_w || !d_mel_bands || !d_work || !d_out) return fail(BN_ERR_ARG, "null device pointer");
    if (n_fft != kFft) return fail(BN_ERR_UNSUPPORTED, "n_fft=%d: only 512 is implemented", n_fft);
    if (B < 0 || T <= 0 || W <= 0 || hop <= 0 || n_mels <= 0) return fail(BN_ERR_ARG, "bad shape B=%d T=%d hop=%d W=%d mels=%d", B, T, hop, W, n_mels);
    if (1 + T / hop < W)
        return fail(BN_ERR_ARG, "T=%d hop=%d gives %d frames, fewer than spec_width=%d", T, hop, 1 + T / hop, W);
    if (mode < BN_SPEC_MEL || mode > BN_SPEC_MFCC) return fail(BN_ERR_ARG, "unknown spectrogram mode %d", mode);
    if (mag_scale < BN_MAG_NONE || mag_scale > BN_MAG_DB) return fail(BN_ERR_ARG, "unknown mag_scale %d", mag_scale);
    if (mode != BN_SPEC_MEL) mag_scale = BN_MAG_NONE;  // the reference applies mag_scale only in 'mel' / 'linear' mode
    if (mode == BN_SPEC_MFCC && (!d_dct || n_mfcc <= 0 || n_mfcc > n_mels)) return fail(BN_ERR_ARG, "mfcc needs 0 < n_mfcc <= n_mels and a DCT matrix");
    // mfcc: the reference takes the dB reference and floor over ALL 1 + T / hop frames and cuts to W after the DCT
    const int Wall = mode == BN_SPEC_MFCC ? 1 + T / hop : W;
    if ((n_mels * Wall) % 4) return fail(BN_ERR_UNSUPPORTED, "n_mels * frames must be a multiple of 4");
    if (bn::melspec_finish_lds_bytes(n_mels, Wall, W, mode, mag_scale, n_mfcc) + 64 > 160 * 1024)
        return fail(BN_ERR_UNSUPPORTED, "a %d x %d mel map (mode %d) does not fit one workgroup's LDS", n_mels, Wall, mode);
    if (B == 0) return BN_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t per = (size_t)n_mels * Wall;
    const size_t per_out = mode == BN_SPEC_MFCC ? (size_t)n_mfcc * W : per;
    for (int b0 = 0; b0 < B; b0 += kMaxGridBatch) {
        const int nb = B - b0 < kMaxGridBatch ? B - b0 : kMaxGridBatch;
        float* mel = d_work + (size_t)b0 * per;
        float* minmax = d_work + (size_t)B * per + 2 * (size_t)b0;
        bn::launch_minmax_init(minmax, nb, s);
        if (!bn::launch_stft512_mel(ctx->tables, d_audio + (size_t)b0 * T, nb, T, hop, Wall, mel, n_mels, d_mel_w, d_mel_bands, minmax, s,
                                    mode == BN_SPEC_MFCC))
            return fail(BN_ERR_UNSUPPORTED, "n_mels=%d: the fused STFT+mel kernel takes at most 128 mel bins", n_mels);
        if (!bn::launch_melspec_finish(mel, d_out + (size_t)b0 * per_out, d_dct, nb, n_mels, Wall, W, mode, mag_scale, n_mfcc, pcen_b, s))
            return fail(BN_ERR_DEVICE, "could not raise the LDS limit of the spectrogram finishing kernel");
    }
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_forward(bn_model* m, const float* d_input, const float* d_minmax, int B, float* d_scores, float* d_logits,
               void* stream) {
    if (!m) return fail(BN_ERR_ARG, "null model");
    if (int rc = check_device(m->ctx)) return rc;
    if (!d_input || !d_scores) return fail(BN_ERR_ARG, "null device pointer");
    if (B < 0 || B > m->ctx->max_batch)
        return fail(BN_ERR_ARG, "batch %d exceeds the context's max_batch %d", B, m->ctx->max_batch);
    if (B == 0) return BN_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t in_stride = m->hdr.input_elems, C = m->hdr.num_classes;
    for (int b0 = 0; b0 < B; b0 += kMaxGridBatch) {
        const int nb = B - b0 < kMaxGridBatch ? B - b0 : kMaxGridBatch;
        if (int rc = run_plan(m, d_input + b0 * in_stride, d_minmax ? d_minmax + 2 * (size_t)b0 : nullptr, nb,
                              d_scores + b0 * C, d_logits ? d_logits + b0 * C : nullptr, s))
            return rc;
    }
    return BN_OK;
}

int bn_infer_audio(bn_model* m, const float* d_audio, int B, int T, int hop, float* d_scores, float* d_logits,
                   void* stream) {
    if (!m) return fail(BN_ERR_ARG, "null model");
    if (m->hdr.input_kind != BN_INPUT_SPECTROGRAM)
        return fail(BN_ERR_UNSUPPORTED, "bn_infer_audio needs a hybrid-frontend model; feed waveforms to bn_forward");
    if (B < 0 || B > m->ctx->max_batch)
        return fail(BN_ERR_ARG, "batch %d exceeds the context's max_batch %d", B, m->ctx->max_batch);
    const int F = (int)m->hdr.fft_bins, W = (int)m->hdr.spec_width;
    if (F != kFft / 2 + 1) return fail(BN_ERR_UNSUPPORTED, "model expects %d frequency bins; the STFT kernel gives 257", F);
    bool audio_plan = false;
    for (const OpRec& o : m->ops) audio_plan |= o.p[BN_OP_PATH] == BN_PATH_AUDIO;
    if (audio_plan) {
        // the plan holds operators that start from the waveform (fused STFT + mel mixer): no spectrogram in HBM
        if (int rc = check_device(m->ctx)) return rc;
        if (!d_audio || !d_scores) return fail(BN_ERR_ARG, "null device pointer");
        if (T <= 0 || hop <= 0 || 1 + T / hop < W) return fail(BN_ERR_ARG, "bad audio geometry T=%d hop=%d W=%d", T, hop, W);
        if (B == 0) return BN_OK;
        const size_t C = m->hdr.num_classes;
        for (int b0 = 0; b0 < B; b0 += kMaxGridBatch) {
            const int nb = B - b0 < kMaxGridBatch ? B - b0 : kMaxGridBatch;
            if (int rc = run_plan(m, nullptr, nullptr, nb, d_scores + b0 * C, d_logits ? d_logits + b0 * C : nullptr,
                                  (hipStream_t)stream, d_audio + (size_t)b0 * T, T, hop))
                return rc;
        }
        return BN_OK;
    }
    // un-normalised magnitudes + per-chunk min/max; the plan's first operator normalises while loading
    if (int rc = check_device(m->ctx)) return rc;
    if (!d_audio || !d_scores) return fail(BN_ERR_ARG, "null device pointer");
    if (B == 0) return BN_OK;
    const bool tiled = m->spec_tiled_ok && !bn::g_opt.stft_rowmajor;  // option stft_rowmajor: keep the reference layout (A/B)
    hipStream_t s = (hipStream_t)stream;
    const size_t in_stride = m->hdr.input_elems, C = m->hdr.num_classes;
    // INT8 plans: the quantised input bytes must be the reference's (float64 STFT).  Production form: float32 STFT with error bounds,
    // exact min / max, doubtful bytes listed by the first operator and recomputed in float64 behind it (bn_stft_exact.hip);
    // plans / options outside that form (debug plans, row-major layout, generic mel kernel) take the float64 STFT for every bin.
    const bool i8 = m->hdr.dtype == BN_DTYPE_I8;
    const int exact_opt = i8 ? bn::g_opt.stft_exact : 0;
    const bool guarded = exact_opt == 2 && tiled && m->d_guard && !bn::g_opt.i8_mel_generic && W % 16 == 0 && W <= 1024;
    // (Sub-batching the STFT -> first operator pair for the Infinity Cache and a two-stream skewed schedule were measured and removed:
    // slower / no gain)
    m->spec_tiled_now = tiled;
    int rc = BN_OK;
    if (guarded) {
        // profiling entries: n_ops = the float32 STFT kernel, n_ops + 1 = exact min / max, n_ops + 2 = the float64 pass behind the first operator
        if (T <= 0 || hop <= 0 || 1 + T / hop < W) rc = fail(BN_ERR_ARG, "T=%d hop=%d gives %d frames, fewer than spec_width=%d", T, hop, hop > 0 ? 1 + T / hop : 0, W);
        if (m->d_audit) HIP_TRY(hipMemsetAsync(m->d_audit, 0, 2 * sizeof(int), s));
        for (int b0 = 0; rc == BN_OK && b0 < B; b0 += kMaxGridBatch) {
            const int nb = B - b0 < kMaxGridBatch ? B - b0 : kMaxGridBatch;
            bn::StftGuard g = guard_slice(m, (size_t)b0);
            {
                ProfScope prof(m, (int)m->ops.size(), s);
                bn::launch_stft512(m->ctx->tables, d_audio + (size_t)b0 * T, nb, T, hop, W, m->d_spec + b0 * in_stride, m->d_minmax + 2 * (size_t)b0, s,
                                   true, &g);
            }
            ProfScope prof(m, (int)m->ops.size() + 1, s);
            bn::launch_stft_minmax_exact(m->ctx->tables, d_audio + (size_t)b0 * T, nb, T, hop, W, m->d_spec + b0 * in_stride, true, g,
                                         m->d_minmax + 2 * (size_t)b0, s);
        }
    } else {
        ProfScope prof(m, (int)m->ops.size(), s);
        rc = stft_mag_impl(m->ctx, d_audio, B, T, kFft, hop, W, /*normalize=*/0, m->d_spec, m->d_minmax, stream, tiled, exact_opt != 0);
    }
    if (rc == BN_OK) {
        for (int b0 = 0; b0 < B; b0 += kMaxGridBatch) {
            const int nb = B - b0 < kMaxGridBatch ? B - b0 : kMaxGridBatch;
            m->guard_now = guarded;
            m->guard_audio = d_audio + (size_t)b0 * T;
            m->guard_T = T;
            m->guard_hop = hop;
            rc = run_plan(m, m->d_spec + b0 * in_stride, m->d_minmax + 2 * (size_t)b0, nb, d_scores + b0 * C, d_logits ? d_logits + b0 * C : nullptr, s,
                          nullptr, 0, 0, 0, (size_t)-1, (size_t)b0);
            if (rc != BN_OK) break;
        }
    }
    m->guard_now = false;
    m->spec_tiled_now = false;
    m->last_tiled = tiled;
    m->last_B = rc == BN_OK ? B : 0;
    return rc;
}

int bn_debug_tail_form(const bn_model* m, int* form, int* lds_bytes) {
    if (!m || !form || !lds_bytes) return fail(BN_ERR_ARG, "null argument");
    *form = 0;
    *lds_bytes = 0;
    for (size_t oi = 0; oi < m->ops.size(); ++oi) {
        if (m->ops[oi].kind != BN_OP_I8_TAIL || !m->tail_ok[oi]) continue;
        *form = m->tail2_ok[oi] ? 2 : 1;
        *lds_bytes = m->tail2_ok[oi] ? m->tails2[oi].lds_bytes : m->tails[oi].lds_bytes;
    }
    return BN_OK;
}

int bn_debug_mid_form(const bn_model* m, int* form, int* lds_bytes) {
    if (!m || !form || !lds_bytes) return fail(BN_ERR_ARG, "null argument");
    *form = 0;
    *lds_bytes = 0;
    for (size_t oi = 0; oi < m->ops.size(); ++oi)
        if (m->ops[oi].kind == BN_OP_I8_MID && m->mid_ok[oi]) {
            *form = 1;
            *lds_bytes = m->mids[oi].lds_bytes;
        }
    return BN_OK;
}

int bn_debug_guard_stats(bn_model* m, int B, int64_t* out) {
    if (!m || !out) return fail(BN_ERR_ARG, "null argument");
    if (int rc = check_device(m->ctx)) return rc;
    if (!m->d_guard) return fail(BN_ERR_UNSUPPORTED, "the plan has no exactness pass");
    if (B <= 0 || B > m->last_B || B > kMaxGridBatch) return fail(BN_ERR_ARG, "B=%d: the last bn_infer_audio call left %d spectrograms", B, m->last_B);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<int> cnt((size_t)B);
    int nw = 0, nh[2] = {0, 0};
    HIP_TRY(hipMemcpy(cnt.data(), m->guard.count, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&nw, m->guard.n_work, sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nh, m->guard.n_hard, 2 * sizeof(int), hipMemcpyDeviceToHost));
    int64_t total = 0, mx = 0;
    for (int c : cnt) {
        c %= m->guard.cap + 1;   // (every workgroup of the mixer that hands its chunk over adds cap + 1: a marker, not elements)
        total += c;
        if (c > mx) mx = c;
    }
    out[0] = total;
    out[1] = mx;
    out[2] = nw;
    out[3] = nh[0];
    out[4] = nh[1];
    int au[2] = {0, 0};
    if (m->d_audit) HIP_TRY(hipMemcpy(au, m->d_audit, sizeof au, hipMemcpyDeviceToHost));
    out[5] = au[0];
    out[6] = au[1];
    std::vector<float> lo((size_t)B);   // chunks whose minimum was enclosed in an interval instead of settled (option stft_minint)
    HIP_TRY(hipMemcpy(lo.data(), m->guard.mn_lo, (size_t)B * sizeof(float), hipMemcpyDeviceToHost));
    int64_t wide = 0;
    for (float v : lo) wide += v >= 0.0f;
    out[7] = wide;
    return BN_OK;
}

int bn_debug_input_bytes(bn_model* m, int B, int8_t* d_out, void* stream) {
    if (!m || !d_out) return fail(BN_ERR_ARG, "null argument");
    if (int rc = check_device(m->ctx)) return rc;
    const OpRec* first = nullptr;
    for (const OpRec& o : m->ops)
        if (o.in0 == BN_SLOT_INPUT && o.kind == BN_OP_I8_DWPW && o.p[36] && o.p[30]) first = &o;
    if (!first || !m->d_spec) return fail(BN_ERR_UNSUPPORTED, "the plan's first operator is not the mel mixer with QUANTIZE fused into its load");
    if (B <= 0 || B > m->last_B || B > kMaxGridBatch) return fail(BN_ERR_ARG, "B=%d: the last bn_infer_audio call left %d spectrograms", B, m->last_B);
    bn::launch_spec_bytes(m->d_spec, m->d_minmax, B, (int)m->hdr.spec_width, m->last_tiled, first->f[0], first->p[37], d_out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_ingest_resample(bn_ctx* ctx, const void* d_pcm, int sample_format, int channels, const int64_t* d_in_off,
                       const int64_t* d_out_off, int n_windows, int64_t max_in_len, int64_t max_out_len, const float* d_taps,
                       int up, int down, int taps_per_phase, int n_pre_remove, float* d_mono, float* d_peak, void* stream) {
    if (int rc = check_device(ctx)) return rc;
    if (!d_pcm || !d_in_off || !d_out_off || !d_mono || !d_peak) return fail(BN_ERR_ARG, "null device pointer");
    if (sample_format < BN_PCM_S16 || sample_format > BN_PCM_F32) return fail(BN_ERR_ARG, "unknown sample format %d", sample_format);
    if (channels < 1 || channels > 8)
        return fail(BN_ERR_UNSUPPORTED, "%d channels: the channel mean is implemented for 1..8 channels", channels);
    if (n_windows < 0 || max_in_len < 0 || max_out_len < 0 || up < 1 || down < 1 || taps_per_phase < 0 || n_pre_remove < 0)
        return fail(BN_ERR_ARG, "bad ingest geometry");
    if (taps_per_phase > 0 && !d_taps) return fail(BN_ERR_ARG, "null filter");
    if (taps_per_phase == 0 && up != down) return fail(BN_ERR_ARG, "up=%d down=%d needs a filter", up, down);
    if (n_windows == 0 || max_out_len == 0) return BN_OK;
    if (n_windows > 65535) return fail(BN_ERR_ARG, "at most 65535 windows per call");
    if ((double)(max_out_len + n_pre_remove + 4096) * down >= 4294967296.0 || (double)max_in_len * up >= 4294967296.0)
        return fail(BN_ERR_UNSUPPORTED, "window too long for the 32-bit polyphase index (%lld samples, up=%d, down=%d)",
                    (long long)max_in_len, up, down);
    const size_t lds = bn::ingest_resample_lds_bytes(up, down, taps_per_phase, bn::ingest_resample_block(up, down, taps_per_phase));
    const bool fast_kernel = (up == 1 && (down == 2 || down == 4)) || (up > 1 && up <= 256 && (taps_per_phase == 21 || taps_per_phase == 29 || taps_per_phase == 39));
    if (lds > (fast_kernel ? 64 : 156) * 1024)  // (only the generic kernel's limit is raised to the CU's 160 KB)
        return fail(BN_ERR_UNSUPPORTED, "resampling ratio %d/%d needs %zu bytes of LDS per workgroup (limit %d)", up, down, lds, (fast_kernel ? 64 : 156) * 1024);
    hipStream_t s = (hipStream_t)stream;
    const size_t need = bn::ingest_partial_elems(n_windows, (long)max_out_len, up, down, taps_per_phase);
    if (need > ctx->block_peaks_elems) {  // growing frees the old buffer, which waits for launches still using it
        if (ctx->d_block_peaks) HIP_TRY(hipFree(ctx->d_block_peaks));
        ctx->d_block_peaks = nullptr;
        ctx->block_peaks_elems = 0;
        HIP_TRY(hipMalloc(&ctx->d_block_peaks, need * sizeof(float)));
        ctx->block_peaks_elems = need;
    }
    bn::launch_ingest_resample(d_pcm, sample_format, channels, (const long*)d_in_off, (const long*)d_out_off, n_windows,
                               (long)max_out_len, d_taps, up, down, taps_per_phase, n_pre_remove, d_mono, ctx->d_block_peaks,
                               d_peak, s);
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_ingest_chunks(bn_ctx* ctx, const float* d_mono, const float* d_peak, const int64_t* d_chunk_src,
                     const int32_t* d_chunk_valid, const int32_t* d_chunk_window, int n_chunks, int chunk_len,
                     float* d_chunks, void* stream) {
    if (int rc = check_device(ctx)) return rc;
    if (n_chunks < 0 || chunk_len <= 0) return fail(BN_ERR_ARG, "bad chunk geometry n=%d T=%d", n_chunks, chunk_len);
    if (n_chunks == 0) return BN_OK;
    if (!d_mono || !d_peak || !d_chunk_src || !d_chunk_valid || !d_chunk_window || !d_chunks)
        return fail(BN_ERR_ARG, "null device pointer");
    hipStream_t s = (hipStream_t)stream;
    for (int c0 = 0; c0 < n_chunks; c0 += kMaxGridBatch) {
        const int nc = n_chunks - c0 < kMaxGridBatch ? n_chunks - c0 : kMaxGridBatch;
        bn::launch_ingest_chunks(d_mono, d_peak, (const long*)d_chunk_src + c0, d_chunk_valid + c0, d_chunk_window + c0, nc,
                                 chunk_len, d_chunks + (size_t)c0 * chunk_len, s);
    }
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_chunk_peak_normalize(bn_ctx* ctx, const float* d_x, int B, int T, float eps, float* d_y, void* stream) {
    if (int rc = check_device(ctx)) return rc;
    if (B < 0 || T <= 0) return fail(BN_ERR_ARG, "bad shape B=%d T=%d", B, T);
    if (B == 0) return BN_OK;
    if (!d_x || !d_y) return fail(BN_ERR_ARG, "null device pointer");
    hipStream_t s = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += kMaxGridBatch) {
        const int nb = B - b0 < kMaxGridBatch ? B - b0 : kMaxGridBatch;
        bn::launch_chunk_peaknorm(d_x + (size_t)b0 * T, d_y + (size_t)b0 * T, nb, T, eps, s);
    }
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_pool_scores(bn_ctx* ctx, const float* d_scores, const int64_t* d_file_off, int n_files, int n_classes, int method,
                   float beta, float* d_pooled, void* stream) {
    if (int rc = check_device(ctx)) return rc;
    if (n_files < 0 || n_classes <= 0) return fail(BN_ERR_ARG, "bad pooling geometry files=%d classes=%d", n_files, n_classes);
    if (method < BN_POOL_AVG || method > BN_POOL_LME) return fail(BN_ERR_ARG, "Unsupported pooling method: %d", method);
    if (n_files == 0) return BN_OK;
    if (!d_scores || !d_file_off || !d_pooled) return fail(BN_ERR_ARG, "null device pointer");
    if ((int64_t)n_files * n_classes > 0x7fffffffLL) return fail(BN_ERR_ARG, "files x classes exceeds 2^31");
    bn::launch_pool_scores(d_scores, (const long*)d_file_off, n_files, n_classes, method, beta, d_pooled, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_rank_orders(bn_ctx* ctx, const float* d_scores, int n_rows, int n_classes, int32_t* d_cols, int32_t* d_flat, void* stream) {
    if (int rc = check_device(ctx)) return rc;
    if (n_rows < 0 || n_classes <= 0) return fail(BN_ERR_ARG, "bad score matrix %d x %d", n_rows, n_classes);
    if (n_rows == 0) return BN_OK;
    if (!d_scores || !d_cols || !d_flat) return fail(BN_ERR_ARG, "null device pointer");
    if ((int64_t)n_rows * n_classes > 0x3fffffffLL) return fail(BN_ERR_ARG, "rows x classes exceeds 2^30");
    const size_t need = bn::rank_orders_workspace(n_rows, n_classes);
    if (need > ctx->rank_work_bytes) {   // (grown outside any capture: the call is made once per evaluation, behind the last inference)
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        if (ctx->d_rank_work) (void)hipFree(ctx->d_rank_work);
        ctx->d_rank_work = nullptr;
        ctx->rank_work_bytes = 0;
        if (hipMalloc(&ctx->d_rank_work, need + need / 4) != hipSuccess) {
            (void)hipGetLastError();
            return fail(BN_ERR_DEVICE, "hipMalloc of %zu bytes for the sort workspace failed", need + need / 4);
        }
        ctx->rank_work_bytes = need + need / 4;
    }
    if (!bn::launch_rank_orders(d_scores, n_rows, n_classes, d_cols, d_flat, ctx->d_rank_work, ctx->rank_work_bytes, (hipStream_t)stream)) {
        (void)hipGetLastError();
        return fail(BN_ERR_DEVICE, "the device sort failed");
    }
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_debug_op_output(bn_model* m, int op_index, int B, void* d_dst, size_t dst_bytes, size_t* bytes_per_chunk,
                       void* stream) {
    if (!m) return fail(BN_ERR_ARG, "null model");
    if (op_index < 0 || op_index >= (int)m->ops.size()) return fail(BN_ERR_ARG, "op_index %d out of range", op_index);
    const int sid = m->ops[op_index].out;
    if (sid < 0) return fail(BN_ERR_ARG, "operator %d writes a caller buffer, not a workspace slot", op_index);
    // valid: the operator ran, or its twin of the other entry path wrote the same slot (plans keep one operator per path for the first stages)
    bool valid = (size_t)op_index < m->out_valid.size() && m->out_valid[op_index];
    if (!valid && (size_t)sid < m->slot_valid.size() && m->slot_valid[sid]) valid = true;
    if (d_dst && !valid)
        return fail(BN_ERR_UNSUPPORTED, "operator %d did not write its output in the last forward call: a fused kernel keeps that map on chip under the "
                                        "current options", op_index);
    const size_t per = m->slots[sid].bytes_per_chunk;
    if (bytes_per_chunk) *bytes_per_chunk = per;
    if (!d_dst) return BN_OK;
    if (B < 0 || B > m->ctx->max_batch || dst_bytes < per * (size_t)B) return fail(BN_ERR_ARG, "destination too small");
    if (int rc = check_device(m->ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, m->d_slots[sid], per * (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BN_OK;
}

int bn_debug_requant(bn_ctx* ctx, const int32_t* d_x, const int32_t* d_mult, const int32_t* d_shift, int n, int mode, int zero_point,
                     int32_t* d_out, void* stream) {
    if (int rc = check_device(ctx)) return rc;
    if (n < 0 || mode < 0 || mode > 3) return fail(BN_ERR_ARG, "bad n / mode");
    if (n == 0) return BN_OK;
    if (!d_x || !d_mult || !d_shift || !d_out) return fail(BN_ERR_ARG, "null device pointer");
    bn::launch_debug_requant(d_x, d_mult, d_shift, n, mode, zero_point, d_out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return BN_OK;
}

int bn_profile_enable(bn_model* m, int enable) {
    if (!m) return fail(BN_ERR_ARG, "null model");
    m->profiling = enable != 0;
    return BN_OK;
}

int bn_profile_only(bn_model* m, int op_index) {
    if (!m) return fail(BN_ERR_ARG, "null model");
    if (op_index < -1 || op_index > (int)m->ops.size() + 2) return fail(BN_ERR_ARG, "op_index %d out of range", op_index);
    m->prof_only = op_index;
    return BN_OK;
}

int bn_profile_collect(bn_model* m, double* total_ms, int64_t* launches, int n) {
    if (!m || !total_ms || !launches) return fail(BN_ERR_ARG, "null argument");
    if (n < (int)m->ops.size() + 1) return fail(BN_ERR_ARG, "need room for n_ops + 1 entries");
    if (int rc = check_device(m->ctx)) return rc;
    for (int i = 0; i < n; ++i) {
        total_ms[i] = 0.0;
        launches[i] = 0;
    }
    for (auto& r : m->ev_used) {
        HIP_TRY(hipEventSynchronize(r.stop));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
        const int slot = r.op < n ? r.op : (int)m->ops.size();  // a caller without room for the exactness-pass entries gets them under the STFT stage
        total_ms[slot] += ms;
        launches[slot] += 1;
        m->ev_free.push_back(r.start);
        m->ev_free.push_back(r.stop);
    }
    m->ev_used.clear();
    return BN_OK;
}

int bn_set_option(const char* name, int value) {
    if (!name) return fail(BN_ERR_ARG, "null option name");
    for (const OptName& e : kOptions)
        if (strcmp(name, e.name) == 0) {
            std::lock_guard<std::mutex> lock(g_opt_mu);
            g_opt_default.*(e.field) = value;
            return BN_OK;
        }
    return fail(BN_ERR_ARG, "unknown option '%s'", name);
}

int bn_ctx_set_option(bn_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return fail(BN_ERR_ARG, "null argument");
    for (const OptName& e : kOptions)
        if (strcmp(name, e.name) == 0) {
            std::lock_guard<std::mutex> lock(g_opt_mu);
            for (auto& ov : ctx->opt_override)
                if (ov.first == e.field) {
                    ov.second = value;
                    return BN_OK;
                }
            ctx->opt_override.emplace_back(e.field, value);
            return BN_OK;
        }
    return fail(BN_ERR_ARG, "unknown option '%s'", name);
}

int bn_ctx_get_option(bn_ctx* ctx, const char* name, int* value) {
    if (!ctx || !name || !value) return fail(BN_ERR_ARG, "null argument");
    for (const OptName& e : kOptions)
        if (strcmp(name, e.name) == 0) {
            std::lock_guard<std::mutex> lock(g_opt_mu);
            *value = g_opt_default.*(e.field);
            for (const auto& ov : ctx->opt_override)
                if (ov.first == e.field) *value = ov.second;
            return BN_OK;
        }
    return fail(BN_ERR_ARG, "unknown option '%s'", name);
}

int bn_ctx_reset_options(bn_ctx* ctx) {
    if (!ctx) return fail(BN_ERR_ARG, "null context");
    std::lock_guard<std::mutex> lock(g_opt_mu);
    ctx->opt_override.clear();
    return BN_OK;
}

void* bn_host_alloc_pinned(bn_ctx* ctx, size_t bytes) {
    if (!ctx || bytes == 0) {
        fail(BN_ERR_ARG, "null context or empty allocation");
        return nullptr;
    }
    if (check_device(ctx) != BN_OK) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(BN_ERR_DEVICE, "hipHostMalloc of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}

int bn_host_free_pinned(void* p) {
    if (p && hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(BN_ERR_DEVICE, "hipHostFree failed");
    }
    return BN_OK;
}

int bn_preload_kernels(bn_ctx* ctx) {
    if (!ctx) return fail(BN_ERR_ARG, "null context");
    if (int rc = check_device(ctx)) return rc;
    // The runtime loads a device code object at the first launch of one of its kernels (a few ms each, and every other thread's launches and
    // copies wait meanwhile).  A caller with idle time before its first batch — the evaluate pipeline while the first files are read — asks here.
    bn::preload_ingest(); bn::preload_stft(); bn::preload_stft_exact(); bn::preload_i8_fused(); bn::preload_i8_strip(); bn::preload_i8_tail2();
    bn::preload_i8_tail(); bn::preload_i8(); bn::preload_i8_pw(); bn::preload_f32(); bn::preload_f32_fused(); bn::preload_f32_strip();
    bn::preload_f32_pw(); bn::preload_melspec(); bn::preload_sort();
    return BN_OK;
}

int bn_get_option(const char* name, int* value) {
    if (!name || !value) return fail(BN_ERR_ARG, "null argument");
    for (const OptName& e : kOptions)
        if (strcmp(name, e.name) == 0) {
            std::lock_guard<std::mutex> lock(g_opt_mu);
            *value = g_opt_default.*(e.field);
            return BN_OK;
        }
    return fail(BN_ERR_ARG, "unknown option '%s'", name);
}

const char* bn_kernel_names(void) {
    return "ingest_resample_kernel\ningest_decimate_kernel\ningest_peak_kernel\ningest_chunks_kernel\nchunk_peaknorm_kernel\npool_scores_kernel\nstft512_mag_kernel\nspec_normalize_kernel\nmelspec_finish_kernel\nf32_mel_kernel\nf32_melfin_kernel\nf32_mag_kernel\nf32_rawfe_kernel\nf32_stem_kernel\nf32_dw_kernel\n"
           "f32_pw_kernel\nf32_pw_ws_kernel\nf32_dwpw_kernel\nf32_dwpw_wave_kernel\nf32_strip_kernel\nf32_front_strip_kernel\nf32_front2_kernel\nf32_pwdw_kernel\nf32_dw_stream_kernel\nf32_front_kernel\nf32_gap_kernel\nf32_gap_dense_kernel\nf32_dense_kernel\nf32_segate_kernel\nf32_scale_kernel\nf32_attnpool_kernel\n"
           "i8_quant_kernel\ni8_mel_kernel\ni8_stem_kernel\ni8_dw_kernel\ni8_pw_kernel\ni8_dwpw_kernel\ni8_mel_mfma_kernel\ni8_strip_kernel\ni8_strip_mf_kernel\ni8_front_strip_kernel\ni8_front_kernel\ni8_tail_kernel\ni8_tail2_kernel\ni8_mid2_kernel\ni8_mean_kernel\ni8_fc_kernel\ni8_scale_kernel\ni8_maxnorm_kernel\ni8_rawfe_kernel\ni8_pwdw_kernel\ni8_dw_stream_kernel\ni8_stem_stream_kernel\ni8_segate_kernel\ni8_pw_wave_kernel\ni8_pw_lds_kernel\ni8_attnpool_kernel\n"
           "i8_head_kernel\ni8_head_softmax_kernel";
}

}  // extern "C"
