"""Packed model blob writer — the Python side of ``csrc/bn_blob.h``.

``bn_model_load`` (include/birdnet_hip.h) takes one self-contained byte string: a header,
the activation-slot table, the tensor table, the operator table and 256-byte-aligned tensor
payloads.  The lowering passes (``_lower_f32`` for `.keras`, ``_lower_i8`` for `.tflite`)
describe a device plan through :class:`PlanBuilder`; this module lays it out.  Every constant
here mirrors ``bn_blob.h`` and is asserted against the library at load time through
``bn_version``/the blob version field.
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

BLOB_MAGIC = b"BNHIPM01"
BLOB_VERSION = 5

DTYPE_F32, DTYPE_I8 = 0, 1
INPUT_SPECTROGRAM, INPUT_WAVEFORM, INPUT_MEL = 0, 1, 2

SLOT_INPUT, SLOT_SCORES, SLOT_LOGITS, SLOT_AUDIO, SLOT_NONE = -1, -2, -3, -4, -9
OP_PATH, PATH_BOTH, PATH_INPUT, PATH_AUDIO = 39, 0, 1, 2

OP_NP, OP_NT, OP_NF = 40, 16, 8

# operator kinds (enum BnOpKind)
F32_MEL, F32_MAG, F32_RAWFE, F32_STEM, F32_DW, F32_PW = 1, 2, 3, 4, 5, 6
F32_SEGATE, F32_SCALE, F32_GAP, F32_DENSE, F32_ATTNPOOL, F32_DWPW, F32_STFTMEL, F32_MELFIN, F32_FRONT, F32_GAPDENSE = 7, 8, 9, 10, 11, 12, 13, 14, 15, 16
I8_QUANT, I8_MEL, I8_STEM, I8_DW, I8_PW, I8_MEAN, I8_FC, I8_HEAD, I8_DWPW, I8_FRONT, I8_TAIL, I8_SCALE, I8_MAXNORM, I8_RAWFE, I8_ATTNPOOL, I8_MID = 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35
TAIL_TAG = 38  # OpRec.p[TAIL_TAG] = TAIL_COVERED: the operator is covered by the plan's fused tail operator; TAIL_OP: it is that operator
TAIL_COVERED, TAIL_OP = 0x7A110001, 0x7A110002  # (bn_blob.h; values no other use of p[38] can take)
MID_COVERED, MID_OP = 0x7A11000E, 0x7A11000F    # the same for the fused stage-2 chain (i8_mid2_kernel)
FRONT2_HEAD, FRONT2_COVERED = 0x7A110003, 0x7A110004  # front block + the residual block FRONT2_DIST operators further on may run as one kernel
FRONT2_DIST = 37
PWDW8_HEAD, PWDW8_COVERED = 0x7A11000C, 0x7A11000D  # the INT8 counterpart (i8_pwdw_kernel)
PWDW_STEM = 0x7A11000B  # a stem convolution whose only reader is a PWDW pair: the fused kernel may compute the stem rows itself
PWDW_HEAD, PWDW_COVERED = 0x7A110009, 0x7A11000A  # expand 1x1 + the depthwise 3x3 behind it (inverted-residual blocks) may run as one kernel
SEGATE_HEAD, SEGATE_COVERED = 0x7A110007, 0x7A110008  # an I8_MEAN operator and the two I8_FC operators of a squeeze-excite gate behind it: may run as one kernel
SCALE_HEAD, SCALE_COVERED = 0x7A110005, 0x7A110006  # an I8_SCALE operator and the plain 1x1 convolution right behind it, its only reader: may run as one kernel

KIND_NAMES = {
    F32_MEL: "f32_mel", F32_MAG: "f32_mag", F32_RAWFE: "f32_rawfe", F32_STEM: "f32_stem", F32_DW: "f32_dw",
    F32_PW: "f32_pw", F32_SEGATE: "f32_segate", F32_SCALE: "f32_scale", F32_GAP: "f32_gap", F32_DENSE: "f32_dense",
    F32_ATTNPOOL: "f32_attnpool", F32_DWPW: "f32_dwpw", F32_STFTMEL: "f32_stftmel", F32_MELFIN: "f32_melfin", F32_FRONT: "f32_front", F32_GAPDENSE: "f32_gapdense", I8_QUANT: "i8_quant", I8_MEL: "i8_mel", I8_STEM: "i8_stem", I8_DW: "i8_dw",
    I8_PW: "i8_pw", I8_DWPW: "i8_dwpw", I8_FRONT: "i8_front", I8_MEAN: "i8_mean", I8_FC: "i8_fc", I8_HEAD: "i8_head", I8_TAIL: "i8_tail", I8_SCALE: "i8_scale", I8_MAXNORM: "i8_maxnorm", I8_RAWFE: "i8_rawfe", I8_ATTNPOOL: "i8_attnpool", I8_MID: "i8_mid",
}  # fmt: skip

ACT_CODES = {"none": 0, "linear": 0, "relu": 1, "relu6": 2}
MAG_CODES = {"none": 0, "pwl": 1, "pcen": 2, "db": 3}


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


@dataclass
class PlanOp:
    """One device-plan operator plus what tests need to interpret its output."""

    kind: int
    in0: int
    in1: int
    out: int
    p: list[int]
    t: list[int]
    f: list[float]
    name: str = ""  # reference layer / tflite tensor this output corresponds to
    out_shape: tuple = ()  # per chunk
    out_dtype: str = "float32"


@dataclass
class Plan:
    """A lowered model: operators, constant tensors, slot sizes and header facts."""

    dtype: int
    input_kind: int
    input_elems: int
    fft_bins: int
    spec_width: int
    num_classes: int
    ops: list[PlanOp] = field(default_factory=list)
    tensors: list[np.ndarray] = field(default_factory=list)
    slot_bytes: list[int] = field(default_factory=list)
    meta: dict = field(default_factory=dict)

    def to_blob(self) -> bytes:
        return pack_plan(self)


class PlanBuilder:
    """Collects operators over symbolic activation values, then maps values to slots.

    Lowering passes call :meth:`value` for every operator output and refer to values by id.
    :meth:`finalize` runs a linear scan over the operator list: with ``reuse=True`` a slot is
    recycled once the last reader of its value has run (an output never aliases a value that
    is still live, so no kernel runs in place unless its operator says so; ``_extra_uses`` keeps
    the inputs of an operator alive up to a later operator that may run fused with it); with
    ``reuse=False`` every value keeps its own slot so that tests can read each intermediate
    activation back through ``bn_debug_op_output``.
    """

    def __init__(self, plan: Plan):
        self.plan = plan
        self._value_bytes: list[int] = []
        self._gate_refs: list[tuple[int, int]] = []  # (op index, p index) holding a value id
        self._extra_uses: list[tuple[int, int]] = []  # (op index, value id): the value must stay live until that operator has run

    def tensor(self, arr: np.ndarray, dtype) -> int:
        a = np.ascontiguousarray(np.asarray(arr).astype(dtype))
        self.plan.tensors.append(a)
        return len(self.plan.tensors) - 1

    def value(self, nbytes: int) -> int:
        self._value_bytes.append(_align(int(nbytes), 256))
        return len(self._value_bytes) - 1

    def op(self, kind, in0, out, p=(), t=(), f=(), in1=SLOT_NONE, name="", out_shape=(), out_dtype="float32",
           value_params=(), path=PATH_BOTH) -> PlanOp:
        pp = [int(v) for v in p] + [0] * (OP_NP - len(p))
        pp[OP_PATH] = int(path)
        tt = [int(v) for v in t] + [-1] * (OP_NT - len(t))
        ff = [float(v) for v in f] + [0.0] * (OP_NF - len(f))
        if len(pp) != OP_NP or len(tt) != OP_NT or len(ff) != OP_NF:
            raise ValueError("operator record overflow")
        o = PlanOp(kind, int(in0), int(in1), int(out), pp, tt, ff, name, tuple(out_shape), out_dtype)
        for pi in value_params:
            self._gate_refs.append((len(self.plan.ops), pi))
        self.plan.ops.append(o)
        return o

    def finalize(self, reuse: bool = True) -> Plan:
        ops = self.plan.ops
        refs: dict[int, list[int]] = {}
        for oi, pi in self._gate_refs:
            refs.setdefault(oi, []).append(pi)
        last_use: dict[int, int] = {}
        for oi, o in enumerate(ops):
            for v in [o.in0, o.in1, o.out] + [o.p[pi] for pi in refs.get(oi, [])]:
                if v >= 0:
                    last_use[v] = oi
        extra: dict[int, list[int]] = {}
        for oi, v in self._extra_uses:
            last_use[v] = max(last_use.get(v, oi), oi)
            extra.setdefault(oi, []).append(v)
        slot_of: dict[int, int] = {}
        slot_bytes: list[int] = []
        free: list[int] = []
        for oi, o in enumerate(ops):
            if o.out >= 0 and o.out not in slot_of:
                need = self._value_bytes[o.out]
                if reuse and free:
                    sid = min(free, key=lambda s: (slot_bytes[s] < need, abs(slot_bytes[s] - need)))
                    free.remove(sid)
                    slot_bytes[sid] = max(slot_bytes[sid], need)
                else:
                    slot_bytes.append(need)
                    sid = len(slot_bytes) - 1
                slot_of[o.out] = sid
            if reuse:
                for v in {o.in0, o.in1, o.out, *[o.p[pi] for pi in refs.get(oi, [])], *extra.get(oi, [])}:
                    if v >= 0 and last_use[v] == oi and v in slot_of and slot_of[v] not in free:
                        free.append(slot_of[v])
        for oi, o in enumerate(ops):
            o.in0 = slot_of[o.in0] if o.in0 >= 0 else o.in0
            o.in1 = slot_of[o.in1] if o.in1 >= 0 else o.in1
            o.out = slot_of[o.out] if o.out >= 0 else o.out
            for pi in refs.get(oi, []):
                o.p[pi] = slot_of[o.p[pi]]
        self.plan.slot_bytes = slot_bytes
        return self.plan


def pack_plan(plan: Plan) -> bytes:
    """Serialise ``plan`` into the byte layout ``bn_model_load`` parses."""
    n_slots, n_tensors, n_ops = len(plan.slot_bytes), len(plan.tensors), len(plan.ops)
    op_size = 16 + 4 * (OP_NP + OP_NT + OP_NF)
    slots_off = 64
    tensors_off = slots_off + 8 * n_slots
    ops_off = tensors_off + 16 * n_tensors
    data_off = _align(ops_off + op_size * n_ops)

    tensor_recs = []
    payload = bytearray()
    for arr in plan.tensors:
        off = data_off + len(payload)
        raw = arr.tobytes()
        tensor_recs.append((off, len(raw)))
        payload += raw
        payload += b"\x00" * (_align(len(payload)) - len(payload))

    out = bytearray()
    out += struct.pack(
        "<8s14I", BLOB_MAGIC, BLOB_VERSION, plan.dtype, plan.input_kind, plan.input_elems, plan.fft_bins,
        plan.spec_width, plan.num_classes, n_slots, n_tensors, n_ops, slots_off, tensors_off, ops_off, 0,
    )  # fmt: skip
    assert len(out) == 64
    for nb in plan.slot_bytes:
        out += struct.pack("<Q", nb)
    for off, nb in tensor_recs:
        out += struct.pack("<QQ", off, nb)
    for o in plan.ops:
        out += struct.pack(f"<4i{OP_NP}i{OP_NT}i{OP_NF}f", o.kind, o.in0, o.in1, o.out, *o.p, *o.t, *o.f)
    out += b"\x00" * (data_off - len(out))
    out += payload
    return bytes(out)
