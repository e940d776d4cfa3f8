"""Dependency-free reader for `.tflite` FlatBuffers (schema v3, identifier ``TFL3``).

The reference hands a `.tflite` path to ``tf.lite.Interpreter``
(reference: birdnet_stm32/models/runners.py:51-68).  TensorFlow and the
``flatbuffers`` package are not available on the MI355X box, so this module walks
the FlatBuffer wire format directly: a table is an ``int32`` back-pointer to its
vtable, the vtable is ``[u16 vtable_bytes, u16 table_bytes, u16 field_offset...]``,
vectors and strings are ``u32 length`` + payload reached through a ``u32`` forward
offset.  Only the tables the shipped INT8 graph uses are decoded (SURVEY.md
Appendix B); an operator outside that set raises ``ValueError`` so nothing is
silently skipped.
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

# TensorType enum of the TFLite schema -> numpy dtype
_TENSOR_DTYPES = {0: np.float32, 1: np.float16, 2: np.int32, 3: np.uint8, 4: np.int64, 6: np.bool_, 7: np.int16, 9: np.int8}

# BuiltinOperator enum values that appear in DS-CNN graphs produced by the reference converter
BUILTIN_NAMES = {
    0: "ADD",
    2: "CONCATENATION",
    3: "CONV_2D",
    4: "DEPTHWISE_CONV_2D",
    6: "DEQUANTIZE",
    9: "FULLY_CONNECTED",
    14: "LOGISTIC",
    18: "MUL",
    22: "RESHAPE",
    25: "SOFTMAX",
    34: "PAD",
    39: "TRANSPOSE",
    40: "MEAN",
    42: "DIV",
    45: "STRIDED_SLICE",
    77: "SHAPE",
    74: "SUM",
    82: "REDUCE_MAX",
    83: "PACK",
    94: "FILL",
    114: "QUANTIZE",
}

_ACTIVATIONS = {0: "none", 1: "relu", 2: "relu_n1_to_1", 3: "relu6", 4: "tanh", 5: "sign_bit"}


class _Buf:
    """Random-access little-endian view of the flatbuffer bytes."""

    __slots__ = ("b",)

    def __init__(self, raw: bytes):
        self.b = raw

    def u8(self, p: int) -> int:
        return self.b[p]

    def i8(self, p: int) -> int:
        return struct.unpack_from("<b", self.b, p)[0]

    def u16(self, p: int) -> int:
        return struct.unpack_from("<H", self.b, p)[0]

    def i32(self, p: int) -> int:
        return struct.unpack_from("<i", self.b, p)[0]

    def u32(self, p: int) -> int:
        return struct.unpack_from("<I", self.b, p)[0]

    def u64(self, p: int) -> int:
        return struct.unpack_from("<Q", self.b, p)[0]


class _Table:
    """One FlatBuffer table; ``slot(i)`` gives the absolute byte position of field *i* or 0."""

    __slots__ = ("buf", "pos", "vt", "vt_len")

    def __init__(self, buf: _Buf, pos: int):
        self.buf = buf
        self.pos = pos
        self.vt = pos - buf.i32(pos)
        self.vt_len = buf.u16(self.vt)

    def slot(self, idx: int) -> int:
        entry = 4 + 2 * idx
        if entry + 2 > self.vt_len:
            return 0
        off = self.buf.u16(self.vt + entry)
        return self.pos + off if off else 0

    # scalar fields -------------------------------------------------------
    def scalar(self, idx: int, kind: str, default=0):
        p = self.slot(idx)
        if not p:
            return default
        return struct.unpack_from("<" + kind, self.buf.b, p)[0]

    # offset fields -------------------------------------------------------
    def _indirect(self, idx: int) -> int:
        p = self.slot(idx)
        return p + self.buf.u32(p) if p else 0

    def table(self, idx: int) -> "_Table | None":
        p = self._indirect(idx)
        return _Table(self.buf, p) if p else None

    def string(self, idx: int) -> str:
        p = self._indirect(idx)
        if not p:
            return ""
        n = self.buf.u32(p)
        return self.buf.b[p + 4 : p + 4 + n].decode("utf-8", "replace")

    def vector(self, idx: int, dtype) -> np.ndarray:
        p = self._indirect(idx)
        if not p:
            return np.zeros((0,), dtype=dtype)
        n = self.buf.u32(p)
        return np.frombuffer(self.buf.b, dtype=np.dtype(dtype).newbyteorder("<"), count=n, offset=p + 4).copy()

    def tables(self, idx: int) -> list["_Table"]:
        p = self._indirect(idx)
        if not p:
            return []
        n = self.buf.u32(p)
        out = []
        for k in range(n):
            e = p + 4 + 4 * k
            out.append(_Table(self.buf, e + self.buf.u32(e)))
        return out


@dataclass
class TfliteTensor:
    """A tensor record: static shape, dtype, quantisation and (for constants) its data."""

    index: int
    name: str
    shape: tuple[int, ...]
    dtype: np.dtype
    scale: np.ndarray  # float32 [0|1|C]
    zero_point: np.ndarray  # int64 [0|1|C]
    quantized_dimension: int
    data: np.ndarray | None = None

    @property
    def is_quantized(self) -> bool:
        return self.scale.size > 0


@dataclass
class TfliteOp:
    """One operator: builtin name, tensor indices and its decoded option table."""

    index: int
    code: int
    name: str
    version: int
    inputs: list[int]
    outputs: list[int]
    options: dict = field(default_factory=dict)


@dataclass
class TfliteModel:
    """Decoded subgraph 0 of a `.tflite` file."""

    version: int
    description: str
    tensors: list[TfliteTensor]
    ops: list[TfliteOp]
    inputs: list[int]
    outputs: list[int]

    def constant_bytes(self) -> int:
        """Total bytes of constant tensors (weights, biases, shape constants)."""
        return int(sum(t.data.nbytes for t in self.tensors if t.data is not None))


def _decode_options(name: str, t: _Table | None) -> dict:
    if t is None:
        return {}
    act = lambda i: _ACTIVATIONS.get(t.scalar(i, "b", 0), "unknown")  # noqa: E731
    if name == "CONV_2D":
        return {
            "padding": "SAME" if t.scalar(0, "b", 0) == 0 else "VALID",
            "stride_w": t.scalar(1, "i", 0),
            "stride_h": t.scalar(2, "i", 0),
            "activation": act(3),
            "dilation_w": t.scalar(4, "i", 1),
            "dilation_h": t.scalar(5, "i", 1),
        }
    if name == "DEPTHWISE_CONV_2D":
        return {
            "padding": "SAME" if t.scalar(0, "b", 0) == 0 else "VALID",
            "stride_w": t.scalar(1, "i", 0),
            "stride_h": t.scalar(2, "i", 0),
            "depth_multiplier": t.scalar(3, "i", 0),
            "activation": act(4),
            "dilation_w": t.scalar(5, "i", 1),
            "dilation_h": t.scalar(6, "i", 1),
        }
    if name in ("ADD", "MUL", "DIV"):
        return {"activation": act(0)}
    if name == "FULLY_CONNECTED":
        return {"activation": act(0), "keep_num_dims": bool(t.scalar(2, "b", 0))}
    if name in ("MEAN", "REDUCE_MAX", "SUM"):
        return {"keep_dims": bool(t.scalar(0, "b", 0))}
    if name == "CONCATENATION":
        return {"axis": t.scalar(0, "i", 0), "activation": act(1)}
    if name == "STRIDED_SLICE":
        return {
            "begin_mask": t.scalar(0, "i", 0),
            "end_mask": t.scalar(1, "i", 0),
            "ellipsis_mask": t.scalar(2, "i", 0),
            "new_axis_mask": t.scalar(3, "i", 0),
            "shrink_axis_mask": t.scalar(4, "i", 0),
            "offset": bool(t.scalar(5, "b", 0)),
        }
    if name == "PACK":
        return {"values_count": t.scalar(0, "i", 0), "axis": t.scalar(1, "i", 0)}
    if name == "SOFTMAX":
        return {"beta": t.scalar(0, "f", 0.0)}
    return {}


def parse_tflite(raw: bytes) -> TfliteModel:
    """Decode a `.tflite` byte string into tensors (with constant data) and operators."""
    if len(raw) < 8 or raw[4:8] != b"TFL3":
        raise ValueError("not a TFLite flatbuffer (missing 'TFL3' identifier)")
    buf = _Buf(raw)
    root = _Table(buf, buf.u32(0))
    version = root.scalar(0, "I", 0)

    opcodes = []
    for oc in root.tables(1):
        legacy = oc.scalar(0, "b", 0)
        full = oc.scalar(3, "i", 0)
        opcodes.append((max(legacy, full), oc.scalar(2, "i", 1), oc.string(1)))

    buffers = root.tables(4)

    def buffer_bytes(i: int) -> bytes | None:
        if i <= 0 or i >= len(buffers):
            return None
        bt = buffers[i]
        p = bt._indirect(0)
        if p:
            n = buf.u32(p)
            return bytes(buf.b[p + 4 : p + 4 + n]) if n else None
        off, size = bt.scalar(1, "Q", 0), bt.scalar(2, "Q", 0)
        if off > 1 and size:
            return bytes(buf.b[off : off + size])
        return None

    subgraphs = root.tables(2)
    if not subgraphs:
        raise ValueError("tflite file holds no subgraph")
    sg = subgraphs[0]

    tensors: list[TfliteTensor] = []
    for ti, tt in enumerate(sg.tables(0)):
        shape = tuple(int(v) for v in tt.vector(0, np.int32))
        ttype = tt.scalar(1, "b", 0)
        if ttype not in _TENSOR_DTYPES:
            raise ValueError(f"tensor {ti}: unsupported TensorType {ttype}")
        dtype = np.dtype(_TENSOR_DTYPES[ttype])
        q = tt.table(4)
        if q is not None:
            scale = q.vector(2, np.float32)
            zp = q.vector(3, np.int64)
            qdim = q.scalar(6, "i", 0)
        else:
            scale, zp, qdim = np.zeros(0, np.float32), np.zeros(0, np.int64), 0
        data = None
        rawdata = buffer_bytes(tt.scalar(2, "I", 0))
        if rawdata is not None:
            arr = np.frombuffer(rawdata, dtype=dtype.newbyteorder("<")).astype(dtype)
            n_expected = int(np.prod(shape)) if shape else 1
            if arr.size != n_expected:
                raise ValueError(f"tensor {ti}: buffer holds {arr.size} elements, shape {shape} needs {n_expected}")
            data = arr.reshape(shape)
        tensors.append(TfliteTensor(ti, tt.string(3), shape, dtype, scale, zp, qdim, data))

    ops: list[TfliteOp] = []
    for oi, ot in enumerate(sg.tables(3)):
        code, ver, custom = opcodes[ot.scalar(0, "I", 0)]
        if code not in BUILTIN_NAMES:
            raise ValueError(f"operator {oi}: builtin code {code} ({custom!r}) is not supported by this reader")
        name = BUILTIN_NAMES[code]
        ops.append(
            TfliteOp(
                index=oi,
                code=code,
                name=name,
                version=ver,
                inputs=[int(v) for v in ot.vector(1, np.int32)],
                outputs=[int(v) for v in ot.vector(2, np.int32)],
                options=_decode_options(name, ot.table(4)),
            )
        )

    return TfliteModel(
        version=version,
        description=root.string(3),
        tensors=tensors,
        ops=ops,
        inputs=[int(v) for v in sg.vector(1, np.int32)],
        outputs=[int(v) for v in sg.vector(2, np.int32)],
    )


def load_tflite(path: str) -> TfliteModel:
    """Read and decode a `.tflite` file."""
    with open(path, "rb") as fh:
        return parse_tflite(fh.read())


def patch_tflite(template_raw: bytes, new: TfliteModel) -> bytes:
    """Write ``new`` as a `.tflite` file by overwriting the constant payloads and quantisation vectors of ``template_raw``.

    ``new`` must have the template's structure (same tensors, shapes, dtypes, operators, per-tensor number of scales) —
    what :func:`birdnet_stm32.conversion.quantize.requantize_like` produces.  Every changed value has a fixed-size slot in
    the FlatBuffer (buffer bytes, ``scale`` float32 vector, ``zero_point`` int64 vector, ``quantized_dimension``), so no
    general FlatBuffer writer is needed and everything else (operator codes, options, names, metadata) stays byte-identical.
    """
    if len(template_raw) < 8 or template_raw[4:8] != b"TFL3":
        raise ValueError("not a TFLite flatbuffer (missing 'TFL3' identifier)")
    out = bytearray(template_raw)
    buf = _Buf(template_raw)
    root = _Table(buf, buf.u32(0))
    buffers = root.tables(4)
    sg = root.tables(2)[0]
    ttabs = sg.tables(0)
    if len(ttabs) != len(new.tensors):
        raise ValueError(f"template has {len(ttabs)} tensors, the model {len(new.tensors)}")

    def put_vector(table: _Table, idx: int, arr: np.ndarray, what: str):
        p = table._indirect(idx)
        n = buf.u32(p) if p else 0
        if n != arr.size:
            raise ValueError(f"{what}: template holds {n} values, the model {arr.size}")
        if n:
            out[p + 4 : p + 4 + arr.nbytes] = arr.tobytes()

    for tt, t in zip(ttabs, new.tensors):
        shape = tuple(int(v) for v in tt.vector(0, np.int32))
        if shape != tuple(t.shape) or _TENSOR_DTYPES.get(tt.scalar(1, "b", 0)) != t.dtype.type:
            raise ValueError(f"tensor {t.index}: shape/dtype differ from the template")
        q = tt.table(4)
        if q is not None:
            put_vector(q, 2, np.ascontiguousarray(t.scale, "<f4"), f"tensor {t.index} scale")
            put_vector(q, 3, np.ascontiguousarray(t.zero_point, "<i8"), f"tensor {t.index} zero_point")
            p = q.slot(6)
            if p:
                struct.pack_into("<i", out, p, int(t.quantized_dimension))
            elif t.quantized_dimension != 0 and t.scale.size > 1:
                raise ValueError(f"tensor {t.index}: template stores no quantized_dimension (default 0), the model needs {t.quantized_dimension}")
        elif t.scale.size:
            raise ValueError(f"tensor {t.index}: quantised in the model, not in the template")
        bi = tt.scalar(2, "I", 0)
        if t.data is None or bi <= 0 or bi >= len(buffers):
            continue
        payload = np.ascontiguousarray(t.data.astype(t.dtype.newbyteorder("<"), copy=False)).tobytes()
        bt = buffers[bi]
        p = bt._indirect(0)
        if p:
            n, at = buf.u32(p), p + 4
        else:
            at, n = bt.scalar(1, "Q", 0), bt.scalar(2, "Q", 0)
        if n != len(payload):
            raise ValueError(f"tensor {t.index}: buffer holds {n} bytes, the model {len(payload)}")
        out[at : at + n] = payload
    return bytes(out)
