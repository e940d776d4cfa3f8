"""Dependency-free writer for `.tflite` FlatBuffers (schema v3, identifier ``TFL3``) — the inverse of ``_tflite_reader``.

The reference gets its `.tflite` bytes from ``tf.lite.TFLiteConverter.convert()`` and writes them to disk
(reference: birdnet_stm32/conversion/quantize.py:131-168).  The own post-training quantiser of this build
(``birdnet_stm32.conversion``) produces a :class:`TfliteModel` in memory; this module serialises it so that the result is
an ordinary `.tflite` file: ``load_model_runner`` reads it back like any other, and the tables follow the public schema
(``Model{version, operator_codes, subgraphs, description, buffers}``, ``SubGraph{tensors, inputs, outputs, operators,
name}``, ``Tensor{shape, type, buffer, name, quantization}``, ``QuantizationParameters{scale, zero_point,
quantized_dimension}``, ``Operator{opcode_index, inputs, outputs, builtin_options_type, builtin_options}``,
``OperatorCode{deprecated_builtin_code, version, builtin_code}``, ``Buffer{data}``), so other TFLite tooling can open it.

The FlatBuffer is built back to front the way the format intends: children first, every table preceded by its vtable,
``uoffset`` fields pointing forward, scalars aligned to their size, buffer payloads to 16 bytes.
"""

from __future__ import annotations

import struct

import numpy as np

from birdnet_stm32.models._tflite_reader import _ACTIVATIONS, _TENSOR_DTYPES, BUILTIN_NAMES, TfliteModel

_TYPE_OF = {np.dtype(v): k for k, v in _TENSOR_DTYPES.items()}
_CODE_OF = {v: k for k, v in BUILTIN_NAMES.items()}
_ACT_OF = {v: k for k, v in _ACTIVATIONS.items()}

# BuiltinOptions union tags of the schema (checked against the shipped file for the operators it holds)
_OPTIONS_TYPE = {"CONV_2D": 1, "DEPTHWISE_CONV_2D": 2, "FULLY_CONNECTED": 8, "SOFTMAX": 9, "CONCATENATION": 10, "ADD": 11, "RESHAPE": 17,
                 "MUL": 21, "PAD": 22, "TRANSPOSE": 0, "MEAN": 27, "REDUCE_MAX": 27, "SUM": 27, "DIV": 29, "STRIDED_SLICE": 32, "SHAPE": 55, "PACK": 59}
# operator versions the converter writes for int8 graphs (read off the shipped file where present)
_VERSION = {"QUANTIZE": 1, "TRANSPOSE": 2, "STRIDED_SLICE": 2, "SHAPE": 1, "PACK": 1, "FILL": 3, "CONCATENATION": 2, "CONV_2D": 3,
            "DEPTHWISE_CONV_2D": 3, "ADD": 2, "MEAN": 2, "FULLY_CONNECTED": 4, "LOGISTIC": 2, "DEQUANTIZE": 2, "MUL": 2, "SOFTMAX": 1,
            "REDUCE_MAX": 2, "DIV": 2, "RESHAPE": 1, "PAD": 2, "SUM": 2}


class _Builder:
    """Minimal FlatBuffer builder; offsets are distances from the END of the finished buffer."""

    def __init__(self):
        self.b = bytearray()
        self.minalign = 1

    def off(self) -> int:
        return len(self.b)

    def _prepend(self, raw: bytes) -> None:
        self.b[0:0] = raw

    def prep(self, size: int, additional: int = 0) -> None:
        self.minalign = max(self.minalign, size)
        pad = (-(len(self.b) + additional)) % size
        if pad:
            self._prepend(b"\x00" * pad)

    def scalar(self, fmt: str, value) -> int:
        raw = struct.pack("<" + fmt, value)
        self.prep(len(raw))
        self._prepend(raw)
        return self.off()

    def uoffset(self, target: int) -> int:
        """A forward offset to the object at ``target`` (distance from end), written here."""
        self.prep(4)
        self._prepend(struct.pack("<I", self.off() - target + 4))
        return self.off()

    def string(self, text: str) -> int:
        raw = text.encode("utf-8")
        self.prep(4, len(raw) + 1)
        self._prepend(raw + b"\x00")
        self._prepend(struct.pack("<I", len(raw)))
        return self.off()

    def vector(self, arr: np.ndarray, align: int | None = None) -> int:
        a = np.ascontiguousarray(arr)
        raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
        align = align or max(a.dtype.itemsize, 1)
        self.prep(4, len(raw))
        self.prep(align, len(raw))
        self._prepend(raw)
        self._prepend(struct.pack("<I", a.size))
        return self.off()

    def offset_vector(self, targets: list[int]) -> int:
        self.prep(4, 4 * len(targets))
        for t in reversed(targets):
            self._prepend(struct.pack("<I", self.off() - t + 4))
        self._prepend(struct.pack("<I", len(targets)))
        return self.off()

    def table(self, fields: list) -> int:
        """``fields[i]`` = None (absent) | ("s", fmt, value) scalar | ("o", target) offset to a finished child."""
        end = self.off()
        at = [0] * len(fields)
        # larger scalars first keeps the padding small; order inside a table is free
        order = sorted(range(len(fields)), key=lambda i: -(struct.calcsize("<" + fields[i][1]) if fields[i] and fields[i][0] == "s" else 4))
        for i in order:
            f = fields[i]
            if f is None:
                continue
            at[i] = self.scalar(f[1], f[2]) if f[0] == "s" else self.uoffset(f[1])
        self.prep(4)
        self._prepend(b"\x00\x00\x00\x00")  # soffset to the vtable, patched below
        tab = self.off()
        n = len(fields)
        while n and at[n - 1] == 0:
            n -= 1
        vt = struct.pack(f"<{2 + n}H", 4 + 2 * n, tab - end, *[(tab - at[i]) if at[i] else 0 for i in range(n)])
        if len(vt) % 4:
            vt = b"\x00\x00" + vt  # keep the table 4-byte aligned (pad in front of the vtable)
            self._prepend(vt)
            vt_off = self.off() - 2
        else:
            self._prepend(vt)
            vt_off = self.off()
        struct.pack_into("<i", self.b, len(self.b) - tab, vt_off - tab)
        return tab

    def finish(self, root: int, ident: bytes) -> bytes:
        self.prep(max(self.minalign, 4), 8)
        self._prepend(ident)
        self._prepend(struct.pack("<I", self.off() - root + 4))
        return bytes(self.b)


def _options(b: _Builder, op) -> tuple[int, int]:
    """(union tag, table offset or 0) of an operator's option table."""
    o, n = op.options, op.name
    act = lambda: ("s", "b", _ACT_OF[o.get("activation", "none")])  # noqa: E731
    tag = _OPTIONS_TYPE.get(n, 0)
    if n == "CONV_2D":
        return tag, b.table([("s", "b", 0 if o["padding"] == "SAME" else 1), ("s", "i", o["stride_w"]), ("s", "i", o["stride_h"]), act(),
                             ("s", "i", o.get("dilation_w", 1)), ("s", "i", o.get("dilation_h", 1))])
    if n == "DEPTHWISE_CONV_2D":
        return tag, b.table([("s", "b", 0 if o["padding"] == "SAME" else 1), ("s", "i", o["stride_w"]), ("s", "i", o["stride_h"]),
                             ("s", "i", o.get("depth_multiplier", 1)), act(), ("s", "i", o.get("dilation_w", 1)), ("s", "i", o.get("dilation_h", 1))])
    if n in ("ADD", "MUL", "DIV"):
        return tag, b.table([act()])
    if n == "FULLY_CONNECTED":
        return tag, b.table([act(), None, ("s", "b", int(bool(o.get("keep_num_dims"))))])
    if n in ("MEAN", "REDUCE_MAX", "SUM"):
        return tag, b.table([("s", "b", int(bool(o.get("keep_dims"))))])
    if n == "CONCATENATION":
        return tag, b.table([("s", "i", o["axis"]), act()])
    if n == "STRIDED_SLICE":
        return tag, b.table([("s", "i", o["begin_mask"]), ("s", "i", o["end_mask"]), ("s", "i", o["ellipsis_mask"]), ("s", "i", o["new_axis_mask"]),
                             ("s", "i", o["shrink_axis_mask"]), ("s", "b", int(bool(o.get("offset"))))])
    if n == "PACK":
        return tag, b.table([("s", "i", o["values_count"]), ("s", "i", o["axis"])])
    if n == "SOFTMAX":
        return tag, b.table([("s", "f", float(o.get("beta", 1.0)))])
    if n == "SHAPE":
        return tag, b.table([("s", "b", 2)])  # out_type INT32
    return 0, 0


def write_tflite(model: TfliteModel) -> bytes:
    """Serialise ``model`` (subgraph 0 of a TFLite file as :func:`parse_tflite` decodes it) into `.tflite` bytes."""
    b = _Builder()
    # buffers: 0 is the empty sentinel, one per constant tensor
    buf_index: dict[int, int] = {}
    buf_tables = []
    payloads = []
    for t in model.tensors:
        if t.data is not None:
            buf_index[t.index] = len(payloads) + 1
            payloads.append(np.ascontiguousarray(np.asarray(t.data).astype(t.dtype)).reshape(-1).view(np.uint8))
    for p in reversed(payloads):
        buf_tables.append(b.table([("o", b.vector(p, align=16))]))
    buf_tables.append(b.table([]))
    buf_tables.reverse()
    buffers_vec = b.offset_vector(buf_tables)

    # operator codes in order of first use
    codes: list[tuple[int, int]] = []
    for op in model.ops:
        key = (_CODE_OF[op.name], op.version or _VERSION.get(op.name, 1))
        if key not in codes:
            codes.append(key)
    code_tabs = [b.table([("s", "b", min(c, 127)), None, ("s", "i", v), ("s", "i", c)]) for c, v in codes]
    codes_vec = b.offset_vector(code_tabs)

    op_tabs = []
    for op in model.ops:
        tag, opt = _options(b, op)
        ins = b.vector(np.asarray(op.inputs, np.int32))
        outs = b.vector(np.asarray(op.outputs, np.int32))
        idx = codes.index((_CODE_OF[op.name], op.version or _VERSION.get(op.name, 1)))
        op_tabs.append(b.table([("s", "I", idx), ("o", ins), ("o", outs), ("s", "B", tag) if tag else None, ("o", opt) if opt else None]))
    ops_vec = b.offset_vector(op_tabs)

    t_tabs = []
    for t in model.tensors:
        q = 0
        if t.scale.size:
            sc = b.vector(np.asarray(t.scale, np.float32))
            zp = b.vector(np.asarray(t.zero_point, np.int64))
            q = b.table([None, None, ("o", sc), ("o", zp), None, None, ("s", "i", int(t.quantized_dimension)) if t.quantized_dimension else None])
        name = b.string(t.name)
        shape = b.vector(np.asarray(t.shape, np.int32))
        t_tabs.append(b.table([("o", shape), ("s", "b", _TYPE_OF[np.dtype(t.dtype)]), ("s", "I", buf_index.get(t.index, 0)), ("o", name),
                               ("o", q) if q else None]))
    tensors_vec = b.offset_vector(t_tabs)
    sg_name = b.string("main")
    outputs = b.vector(np.asarray(model.outputs, np.int32))
    inputs = b.vector(np.asarray(model.inputs, np.int32))
    sg = b.table([("o", tensors_vec), ("o", inputs), ("o", outputs), ("o", ops_vec), ("o", sg_name)])
    sgs = b.offset_vector([sg])
    desc = b.string(model.description or "birdnet_stm32.conversion (MI355X build)")
    root = b.table([("s", "I", int(model.version or 3)), ("o", codes_vec), ("o", sgs), ("o", desc), ("o", buffers_vec)])
    return b.finish(root, b"TFL3")
