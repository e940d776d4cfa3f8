"""ctypes glue for ``oracle/_build/liboracle_cpu.so`` (plain-C float path) and ``oracle/_ref/libfw_ref.so``.

ORACLE — test infrastructure only (see oracle/__init__.py).  ``CpuFloatPath`` runs the shipped float32 graph
on host cores with OpenMP: it is the ``cpu_baseline`` of bench.py and is itself checked against the numpy
oracle in tests.  ``FirmwareRef`` wraps the reference firmware's own C (FFT, STFT, mel) compiled in place.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CPU_LIB = os.path.join(_HERE, "_build", "liboracle_cpu.so")
I8_LIB = os.path.join(_HERE, "_build", "liboracle_i8.so")
NATIVE_DIR = os.path.join(_HERE, "_build", "native")  # `make -C oracle native`: the same sources built -march=native on the host they are timed on


def cpu_share() -> int:
    """CPUs this process may actually use at once: its affinity mask, capped by the cgroup CPU quota (a container on a 256-thread host with a
    quota of 16 CPUs runs 256 OpenMP threads sixteen times slower than sixteen)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, -(-q // per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def build_native() -> bool:
    """Build the -march=native variants of the two C ports on THIS host (bench.py's cpu_baseline leg; seconds).  False if gcc / make fail."""
    import subprocess

    try:
        subprocess.run(["make", "-C", _HERE, "native"], check=True, capture_output=True, timeout=300)
    except Exception:  # noqa: BLE001
        return False
    return os.path.isfile(os.path.join(NATIVE_DIR, "liboracle_i8.so")) and os.path.isfile(os.path.join(NATIVE_DIR, "liboracle_cpu.so"))
FW_LIB = os.path.join(_HERE, "_ref", "libfw_ref.so")

_f = ctypes.POINTER(ctypes.c_float)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f)


class CpuFloatPath:
    """audio -> scores for a NetSpec of plain DS blocks (the shipped checkpoint), in C with OpenMP."""

    def __init__(self, spec, native: bool = False):
        self.lib = ctypes.CDLL(os.path.join(NATIVE_DIR, "liboracle_cpu.so") if native else CPU_LIB)
        self.lib.oc_max_threads.restype = ctypes.c_int
        self.lib.oc_set_threads(min(int(self.lib.oc_max_threads()), cpu_share()))
        self.threads = int(self.lib.oc_max_threads())
        self.spec = spec
        self.steps = self._fold(spec)

    @staticmethod
    def _fold(spec):
        """(kind, params...) per fused step; BatchNorm folded in float64 like any inference converter does."""
        L = {ly.name: ly for ly in spec.layers}
        cons = {}
        for ly in spec.layers:
            for s in ly.inputs:
                cons.setdefault(s, []).append(ly)
        steps = []
        for ly in spec.layers:
            if ly.kind not in ("conv2d", "dwconv2d"):
                continue
            bn = cons[ly.name][0]
            assert bn.kind == "batchnorm"
            s = bn.weights["gamma"].astype(np.float64) / np.sqrt(bn.weights["var"].astype(np.float64) + bn.attrs["eps"])
            w = (ly.weights["kernel"].astype(np.float64) * s).astype(np.float32)
            b = (bn.weights["beta"].astype(np.float64) - bn.weights["mean"].astype(np.float64) * s).astype(np.float32)
            cur, res, act = bn, None, 0
            while True:
                nxt = cons.get(cur.name, [])
                if len(nxt) != 1:
                    break
                n = nxt[0]
                if n.kind == "identity":
                    cur = n
                elif n.kind == "add":
                    res = [i for i in n.inputs if i != cur.name][0]
                    cur = n
                elif n.kind == "relu":
                    act, cur = 2, n
                    break
                else:
                    break
            steps.append({"layer": ly, "w": np.ascontiguousarray(w), "b": b, "res": res, "act": act, "out": cur.name, "src": ly.inputs[0]})
        return steps

    def __call__(self, audio: np.ndarray, hop: int | None = None):
        lib, spec = self.lib, self.spec
        x = np.ascontiguousarray(audio, np.float32)
        B, T = x.shape
        fa = spec.frontend.attrs
        W, M = fa["spec_width"], fa["mel_bins"]
        hop = hop or T // W
        S = np.empty((B, 257, W), np.float32)
        lib.oc_stft_norm(_p(x), B, T, hop, W, _p(S))
        fw = spec.frontend.weights
        mel = np.ascontiguousarray(fw["mel"][:257], np.float32)
        pwl = np.ascontiguousarray(np.stack([fw["pwl_k0"], *fw["pwl_k"], *fw["pwl_w"], *fw["pwl_b"]]), np.float32)
        y = np.empty((B, M, W), np.float32)
        lib.oc_mel_pwl(_p(S), B, 257, W, M, _p(mel), _p(pwl), int(bool(fa.get("norm"))), _p(y))
        vals = {spec.frontend.name: (y, (M, W, 1))}
        for st in self.steps:
            ly = st["layer"]
            src, (H, Wd, C) = vals[st["src"]]
            kh, kw = ly.attrs["kernel"]
            sh, sw = ly.attrs["strides"]
            OH, OW = -(-H // sh), -(-Wd // sw)
            pt = max((OH - 1) * sh + kh - H, 0) // 2
            pl = max((OW - 1) * sw + kw - Wd, 0) // 2
            if ly.kind == "dwconv2d":
                out = np.empty((B, OH, OW, C), np.float32)
                lib.oc_dw3x3(_p(src), _p(out), B, H, Wd, C, sh, sw, OH, OW, pt, pl, _p(st["w"]), _p(st["b"]), st["act"])
                shp = (OH, OW, C)
            elif (kh, kw) == (3, 3):
                cout = st["w"].shape[-1]
                out = np.empty((B, OH, OW, cout), np.float32)
                w = np.ascontiguousarray(st["w"][:, :, 0, :])
                lib.oc_conv3x3_c1(_p(src), _p(out), B, H, Wd, cout, sh, sw, OH, OW, pt, pl, _p(w), _p(st["b"]), st["act"])
                shp = (OH, OW, cout)
            else:
                cout = st["w"].shape[-1]
                out = np.empty((B, H, Wd, cout), np.float32)
                w = np.ascontiguousarray(st["w"][0, 0])
                res = vals[st["res"]][0] if st["res"] else None
                lib.oc_pw(_p(src), _p(res) if res is not None else None, _p(out), ctypes.c_long(B * H * Wd), C, cout, _p(w), _p(st["b"]), st["act"])
                shp = (H, Wd, cout)
            vals[st["out"]] = (out, shp)
            vals[ly.name] = (out, shp)
        last_conv = self.steps[-1]["out"]
        feat, (H, Wd, C) = vals[last_conv]
        head = spec.layers[-1]
        N = head.attrs["units"]
        act = {"linear": 0, "sigmoid": 1, "softmax": 2}[head.attrs["activation"]]
        logits = np.empty((B, N), np.float32)
        scores = np.empty((B, N), np.float32)
        wd = np.ascontiguousarray(head.weights["kernel"], np.float32)
        bd = np.ascontiguousarray(head.weights["bias"], np.float32)
        lib.oc_gap_dense(_p(feat), B, H * Wd, C, N, _p(wd), _p(bd), act, _p(logits), _p(scores))
        return scores, logits, S


def _i8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int8))


def _i32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if a is not None else None


class CpuInt8Path:
    """The numpy TFLite-semantics interpreter (oracle/int8_graph.py) with its heavy operators in plain C + OpenMP
    (oracle/c/oracle_i8.c): CONV_2D, DEPTHWISE_CONV_2D, ADD, MEAN, FULLY_CONNECTED.  Quantisation parameters, multipliers
    and activation ranges come from the numpy interpreter's own preparation code, so both paths share one definition;
    tests check that their tensors are identical.  ``spectrogram()`` is the C STFT of the float port."""

    def __init__(self, model, native: bool = False, **interp_kwargs):
        from oracle import int8_graph as ig

        self.ig = ig
        self.native = bool(native)
        self.lib = ctypes.CDLL(os.path.join(NATIVE_DIR, "liboracle_i8.so") if native else I8_LIB)
        self.lib.oi_max_threads.restype = ctypes.c_int
        self.lib.oi_set_threads(min(int(self.lib.oi_max_threads()), cpu_share()))
        self.threads = int(self.lib.oi_max_threads())
        self.vectorised = bool(self.lib.oi_vectorised())  # the AVX-512 / VNNI paths of oracle_i8.c are compiled in
        outer = self

        class _Interp(ig.Int8Interpreter):
            def _conv(self, op, env, depthwise):
                x = np.ascontiguousarray(self._value(env, op.inputs[0]), np.int8)
                wt = self.model.tensors[op.inputs[1]]
                w = np.ascontiguousarray(wt.data, np.int8)
                bias = (np.ascontiguousarray(self.model.tensors[op.inputs[2]].data, np.int32)
                        if len(op.inputs) > 2 and op.inputs[2] >= 0 else None)
                s_in, zp_in = self._q(op.inputs[0])
                s_out, zp_out = self._q(op.outputs[0])
                o = op.options
                if o["padding"] != "SAME" or o.get("dilation_w", 1) != 1 or o.get("dilation_h", 1) != 1:
                    raise ValueError("only SAME, undilated convolutions occur in the reference graphs")
                sh, sw = o["stride_h"], o["stride_w"]
                B, H, W, Cin = x.shape
                if depthwise:
                    _, kh, kw, cout = w.shape
                else:
                    cout, kh, kw, _ = w.shape
                if op.index not in self._prep:
                    mult, shift = ig._per_channel_multipliers(s_in, wt.scale, s_out, cout)
                    self._prep[op.index] = {"mult": np.ascontiguousarray(mult, np.int32), "shift": np.ascontiguousarray(shift, np.int32),
                                            "act": ig.activation_range(o["activation"], s_out, zp_out)}
                p = self._prep[op.index]
                oh, pt, _ = ig._same(H, kh, sh)
                ow, pl, _ = ig._same(W, kw, sw)
                y = np.empty((B, oh, ow, cout), np.int8)
                lo, hi = (int(v) for v in p["act"])
                if depthwise:
                    outer.lib.oi_dwconv(_i8(x), _i8(y), B, H, W, Cin, kh, kw, sh, sw, oh, ow, pt, pl, _i8(w), _i32(bias), zp_in, zp_out,
                                        _i32(p["mult"]), _i32(p["shift"]), lo, hi)
                else:
                    outer.lib.oi_conv(_i8(x), _i8(y), B, H, W, Cin, kh, kw, cout, sh, sw, oh, ow, pt, pl, _i8(w), _i32(bias), zp_in, zp_out,
                                      _i32(p["mult"]), _i32(p["shift"]), lo, hi)
                return y

            def _add(self, op, env):
                a = np.ascontiguousarray(self._value(env, op.inputs[0]), np.int8)
                b = np.ascontiguousarray(self._value(env, op.inputs[1]), np.int8)
                if b.shape != a.shape:  # broadcast operand: only a trailing-axes match maps onto the C kernel's period
                    if a.size < b.size or b.size == 0 or a.size % b.size or tuple(a.shape[a.ndim - b.ndim:]) != tuple(b.shape):
                        return super()._add(op, env)
                s1, z1 = self._q(op.inputs[0])
                s2, z2 = self._q(op.inputs[1])
                so, zo = self._q(op.outputs[0])
                if op.index not in self._prep:
                    twice_max = 2.0 * max(float(np.float32(s1)), float(np.float32(s2)))
                    self._prep[op.index] = {"m1": ig.quantize_multiplier(float(np.float32(s1)) / twice_max),
                                            "m2": ig.quantize_multiplier(float(np.float32(s2)) / twice_max),
                                            "mo": ig.quantize_multiplier(twice_max / ((1 << 20) * float(np.float32(so)))),
                                            "act": ig.activation_range(op.options["activation"], so, zo)}
                p = self._prep[op.index]
                y = np.empty(a.shape, np.int8)
                lo, hi = (int(v) for v in p["act"])
                outer.lib.oi_add(_i8(a), _i8(b), _i8(y), ctypes.c_long(a.size), ctypes.c_long(b.size), z1, int(p["m1"][0]), int(p["m1"][1]), z2,
                                 int(p["m2"][0]), int(p["m2"][1]), int(p["mo"][0]), int(p["mo"][1]), zo, lo, hi)
                return y

            def _mean(self, op, env):
                x = self._value(env, op.inputs[0])
                axes = sorted(int(a) % x.ndim for a in np.atleast_1d(self._value(env, op.inputs[1])))
                if x.ndim != 4 or axes != [1, 2] or self.mean_form != "int":  # (the C kernel is the integer form)
                    return super()._mean(op, env)
                x = np.ascontiguousarray(x, np.int8)
                B, H, W, C = x.shape
                s_in, zp_in = self._q(op.inputs[0])
                s_out, zp_out = self._q(op.outputs[0])
                n = H * W
                mult, shift = ig.quantize_multiplier(float(np.float32(s_in)) / float(np.float32(s_out)))
                fold = min(n.bit_length() - 1, 32, 31 + shift)
                mult = int((mult << fold) // n)
                shift -= fold
                y = np.empty((B, C), np.int8)
                outer.lib.oi_mean(_i8(x), _i8(y), B, n, C, zp_in, mult, shift, zp_out)
                return y.reshape((B, 1, 1, C)) if op.options.get("keep_dims") else y

            def _fully_connected(self, op, env):
                x = np.ascontiguousarray(self._value(env, op.inputs[0]), np.int8)
                wt = self.model.tensors[op.inputs[1]]
                w = np.ascontiguousarray(wt.data, np.int8)
                bias = (np.ascontiguousarray(self.model.tensors[op.inputs[2]].data, np.int32)
                        if len(op.inputs) > 2 and op.inputs[2] >= 0 else None)
                s_in, zp_in = self._q(op.inputs[0])
                s_out, zp_out = self._q(op.outputs[0])
                if op.index not in self._prep:
                    mult, shift = ig._per_channel_multipliers(s_in, wt.scale, s_out, w.shape[0])
                    self._prep[op.index] = {"mult": np.ascontiguousarray(mult, np.int32), "shift": np.ascontiguousarray(shift, np.int32),
                                            "act": ig.activation_range(op.options["activation"], s_out, zp_out)}
                p = self._prep[op.index]
                x2 = x.reshape(-1, w.shape[1])
                y = np.empty((x2.shape[0], w.shape[0]), np.int8)
                lo, hi = (int(v) for v in p["act"])
                outer.lib.oi_fc(_i8(x2), _i8(y), x2.shape[0], w.shape[1], w.shape[0], _i8(w), _i32(bias), zp_in, zp_out, _i32(p["mult"]),
                                _i32(p["shift"]), lo, hi)
                return y

        self.interp = _Interp(model, **interp_kwargs)

    def invoke(self, x, return_all: bool = False):
        return self.interp.invoke(x, return_all=return_all)

    def spectrogram(self, audio: np.ndarray, hop: int, width: int, out: np.ndarray | None = None) -> np.ndarray:
        """Normalised |STFT| [B, 257, width, 1] through the float port's C STFT (OpenMP over chunks).  ``out``: a buffer of that shape to
        reuse (a fresh 270 MB array per 1024 chunks is 66 k page faults before the first butterfly)."""
        if getattr(self, "_stft_lib", None) is None:
            self._stft_lib = ctypes.CDLL(os.path.join(NATIVE_DIR, "liboracle_cpu.so") if getattr(self, "native", False) else CPU_LIB)
            self._stft_lib.oc_max_threads.restype = ctypes.c_int
            self._stft_lib.oc_set_threads(min(int(self._stft_lib.oc_max_threads()), cpu_share()))
        x = np.ascontiguousarray(audio, np.float32)
        if out is not None and out.shape == (x.shape[0], 257, width, 1) and out.dtype == np.float32 and out.flags.c_contiguous:
            S = out.reshape(x.shape[0], 257, width)
        else:
            S = np.empty((x.shape[0], 257, width), np.float32)
        self._stft_lib.oc_stft_norm(_p(x), x.shape[0], x.shape[1], hop, width, _p(S))
        return S[..., None]


class FirmwareRef:
    """The reference firmware's FFT / STFT / mel code (firmware/Src/{fft,audio_stft,audio_mel}.c), built in place."""

    def __init__(self):
        self.lib = ctypes.CDLL(FW_LIB)

    def fft_512_real(self, x: np.ndarray) -> np.ndarray:
        """Packed spectrum -> complex [257] (reference: firmware/Inc/fft.h:19-28)."""
        buf = np.ascontiguousarray(x, np.float32).copy()
        self.lib.fft_512_real(_p(buf))
        out = np.empty(257, np.complex64)
        out[0], out[256] = buf[0], buf[1]
        out[1:256] = buf[2::2] + 1j * buf[3::2]
        return out

    def stft_magnitude(self, audio: np.ndarray, hop: int, width: int) -> np.ndarray:
        """No-centre, symmetric-Hann STFT magnitude [257, width] (reference: firmware/Src/audio_stft.c:24-71)."""
        a = np.ascontiguousarray(audio, np.float32)
        out = np.empty((257, width), np.float32)
        self.lib.stft_magnitude(_p(a), ctypes.c_uint32(a.size), ctypes.c_uint32(512), ctypes.c_uint32(hop), ctypes.c_uint32(width), _p(out))
        return out

    def mel_matrix(self, n_mels: int, sample_rate: int, fmin: float, fmax: float) -> np.ndarray:
        """[n_mels, 257] weights, read out by pushing an identity 'spectrogram' through mel_filterbank."""
        self.lib.mel_init(ctypes.c_uint32(257), ctypes.c_uint32(n_mels), ctypes.c_uint32(sample_rate), ctypes.c_float(fmin), ctypes.c_float(fmax))
        eye = np.ascontiguousarray(np.eye(257, dtype=np.float32))
        out = np.empty((n_mels, 257), np.float32)
        self.lib.mel_filterbank(_p(eye), ctypes.c_uint32(257), ctypes.c_uint32(257), ctypes.c_uint32(n_mels), _p(out))
        return out


class CpuInt8Program:
    """The INT8 graph as a flat per-chunk program run entirely inside ``oracle_i8.c`` (``oi_program_run``): the CPU baseline without an
    interpreter between the operators.  ORACLE — test infrastructure and ``bench.py``'s ``cpu_baseline`` leg only.

    Built by tracing ``CpuInt8Path``'s interpreter ONCE on a batch of one: every operator becomes a record (shapes, quantisation parameters,
    constants prepared by the interpreter's own code, so there is one definition of the arithmetic); TRANSPOSE / STRIDED_SLICE / RESHAPE /
    CONCATENATION with constants — pure data movement — become gather maps (found by pushing element indices through the numpy operator);
    SHAPE / PACK / FILL fold into constants.  ``invoke`` then deals the chunks to the OpenMP threads, each walking the whole program on its own
    activations.  Graphs with operators outside this set (squeeze-excite MUL, SOFTMAX heads, ...) raise ``NotImplementedError``: callers keep
    ``CpuInt8Path`` for those."""

    class _Op(ctypes.Structure):
        _fields_ = [("kind", ctypes.c_int32), ("in0", ctypes.c_int32), ("in1", ctypes.c_int32), ("out", ctypes.c_int32), ("n", ctypes.c_int64),
                    ("p", ctypes.c_int32 * 24), ("f", ctypes.c_float * 4), ("ptr", ctypes.c_void_p * 6)]

    QUANT, GATHER, CONV, DWCONV, ADD, MEAN, FC, LUT, DEQUANT = range(1, 10)

    def __init__(self, model, native: bool = False):
        from oracle import int8_graph as ig

        self.path = CpuInt8Path(model, native=native)
        self.lib = self.path.lib
        self.threads, self.vectorised = self.path.threads, self.path.vectorised
        assert self.lib.oi_op_bytes() == ctypes.sizeof(self._Op), "oi_op layout"
        interp = self.path.interp
        m = model
        in_shape = [1] + [int(v) for v in m.tensors[m.inputs[0]].shape[1:]]
        x0 = np.random.default_rng(0).random(in_shape, dtype=np.float32)
        _, env = interp.invoke(x0, return_all=True)
        self._keep: list = []       # arrays the records point into
        self.in_elems = int(np.prod(in_shape[1:]))
        ops, off, pos = [], {}, 0
        const_val: dict[int, np.ndarray] = {}   # tensors whose value does not depend on the input (folded)
        src_map: dict[int, tuple[int, np.ndarray, np.ndarray]] = {}  # movement results not yet materialised: tensor -> (source tensor, idx, fill)

        def place(t, nbytes):
            nonlocal pos
            off[t] = pos
            pos += (int(nbytes) + 63) & ~63

        def is_const(t):
            return t in const_val or m.tensors[t].data is not None

        def value(t):
            return const_val[t] if t in const_val else m.tensors[t].data

        def materialise(t):
            """Make sure tensor t exists in the arena (emit the pending gather)."""
            if t in src_map:
                s, idx, fill = src_map.pop(t)
                materialise(s)
                idx = np.ascontiguousarray(idx.reshape(-1), np.int32)
                fill = np.ascontiguousarray(fill.reshape(-1), np.int8)
                self._keep += [idx, fill]
                place(t, idx.size)
                ops.append(self._rec(self.GATHER, s, -1, t, idx.size, ptr=[idx, fill]))
            elif t not in off:
                raise NotImplementedError(f"tensor {t} has no producer in the program")

        def movement(op, fn):
            """out = fn(inputs) is pure data movement: compose the index map of every dynamic input, constants become fill bytes."""
            dyn = [t for t in op.inputs if not is_const(t) and np.asarray(env.get(t, 0)).dtype == np.int8]
            if len(set(dyn)) != 1:
                raise NotImplementedError(f"{op.name} with {len(set(dyn))} dynamic inputs")
            d = dyn[0]
            base_idx = src_map[d][1] if d in src_map else np.arange(int(np.prod(env[d].shape)), dtype=np.int64).reshape(env[d].shape)
            base_fill = src_map[d][2] if d in src_map else np.zeros(env[d].shape, np.int8)
            root = src_map[d][0] if d in src_map else d
            args_i, args_f = [], []
            for t in op.inputs:
                if t == d:
                    args_i.append(base_idx.reshape(env[d].shape))
                    args_f.append(base_fill.reshape(env[d].shape))
                elif is_const(t) and np.asarray(value(t)).dtype == np.int8 and np.asarray(env.get(t, value(t))).shape == np.asarray(value(t)).shape:
                    args_i.append(np.full(np.asarray(value(t)).shape, -1, np.int64))
                    args_f.append(np.asarray(value(t), np.int8))
                else:
                    args_i.append(None)
                    args_f.append(None)
            src_map[op.outputs[0]] = (root, fn(args_i), fn(args_f))

        for op in m.ops:
            n, o0 = op.name, op.outputs[0]
            y = env[o0]
            if all(is_const(t) for t in op.inputs) and n not in ("QUANTIZE",):
                const_val[o0] = np.asarray(y)   # SHAPE / PACK / FILL and anything computed from constants only (batch 1: shapes are constants)
                continue
            if n == "SHAPE":
                const_val[o0] = np.asarray(y)
                continue
            if n == "QUANTIZE":
                s, zp = interp._q(o0)
                place(o0, y.size)
                ops.append(self._rec(self.QUANT, -1, -1, o0, y.size, p=[zp], f=[np.float32(s)]))
            elif n == "DEQUANTIZE":
                s, zp = interp._q(op.inputs[0])
                materialise(op.inputs[0])
                ops.append(self._rec(self.DEQUANT, op.inputs[0], -1, op.inputs[0], y.size, p=[zp], f=[np.float32(s)]))
                self.out_elems = int(y.size)
            elif n == "TRANSPOSE":
                perm = [int(v) for v in value(op.inputs[1])]
                movement(op, lambda a, perm=perm: np.transpose(a[0], perm))
            elif n == "RESHAPE":
                movement(op, lambda a, shp=y.shape: a[0].reshape(shp))
            elif n == "STRIDED_SLICE":
                class _E(dict):
                    pass
                def ss(a, op=op):
                    e = {op.inputs[0]: a[0]}
                    for t in op.inputs[1:]:
                        e[t] = value(t)
                    return interp._strided_slice(op, e)
                movement(op, ss)
            elif n == "CONCATENATION":
                movement(op, lambda a, ax=op.options["axis"]: np.concatenate(a, axis=ax))
            elif n in ("CONV_2D", "DEPTHWISE_CONV_2D"):
                materialise(op.inputs[0])
                dw = n == "DEPTHWISE_CONV_2D"
                x = env[op.inputs[0]]
                wt = m.tensors[op.inputs[1]]
                w = np.ascontiguousarray(wt.data, np.int8)
                bias = np.ascontiguousarray(m.tensors[op.inputs[2]].data, np.int32) if len(op.inputs) > 2 and op.inputs[2] >= 0 else None
                s_in, zp_in = interp._q(op.inputs[0])
                s_out, zp_out = interp._q(o0)
                o = op.options
                sh, sw = o["stride_h"], o["stride_w"]
                _, H, W, Cin = x.shape
                if dw:
                    _, kh, kw, cout = w.shape
                else:
                    cout, kh, kw, _ = w.shape
                pr = interp._prep[op.index]
                oh, pt, _ = ig._same(H, kh, sh)
                ow, pl, _ = ig._same(W, kw, sw)
                lo, hi = (int(v) for v in pr["act"])
                ptrs = [w, bias, pr["mult"], pr["shift"]]
                packed = 0
                if not dw and self.vectorised and (kh, kw, sh, sw, pt, pl) == (1, 1, 1, 1, 0, 0):
                    K4, N16 = (Cin + 3) // 4, (cout + 15) // 16
                    wp = _aligned(K4 * N16 * 64, np.int8)
                    cst = _aligned(N16 * 16 * 3, np.int32)
                    self.lib.oi_pack_1x1(_i8(w), _i32(bias), _i32(pr["mult"]), _i32(pr["shift"]), Cin, cout, zp_in, _i8(wp), _i32(cst))
                    ptrs += [wp, cst]
                    packed = 1
                self._keep += [a for a in ptrs if a is not None]
                place(o0, y.size)
                ops.append(self._rec(self.DWCONV if dw else self.CONV, op.inputs[0], -1, o0, y.size,
                                     p=[H, W, Cin, kh, kw, cout, sh, sw, oh, ow, pt, pl, zp_in, zp_out, lo, hi, packed], ptr=ptrs))
            elif n == "ADD":
                a_t, b_t = op.inputs
                if is_const(a_t):
                    raise NotImplementedError("ADD with a constant first operand")
                materialise(a_t)
                bconst = is_const(b_t)
                if not bconst:
                    materialise(b_t)
                a, b = env[a_t], (np.asarray(value(b_t)) if bconst else env[b_t])
                if b.shape != a.shape and (a.size % b.size or tuple(a.shape[a.ndim - b.ndim:]) != tuple(b.shape)):
                    raise NotImplementedError("ADD broadcast that is not a trailing-axes period")
                pr = interp._prep[op.index]
                _, z1 = interp._q(a_t)
                _, z2 = interp._q(b_t)
                _, zo = interp._q(o0)
                lo, hi = (int(v) for v in pr["act"])
                bc = np.ascontiguousarray(b, np.int8) if bconst else None
                if bc is not None:
                    self._keep.append(bc)
                place(o0, y.size)
                ops.append(self._rec(self.ADD, a_t, -1 if bconst else b_t, o0, y.size,
                                     p=[b.size, z1, int(pr["m1"][0]), int(pr["m1"][1]), z2, int(pr["m2"][0]), int(pr["m2"][1]), int(pr["mo"][0]), int(pr["mo"][1]), zo, lo, hi,
                                        int(bconst)], ptr=[bc]))
            elif n == "MEAN":
                x = env[op.inputs[0]]
                axes = sorted(int(a) % x.ndim for a in np.atleast_1d(value(op.inputs[1])))
                if x.ndim != 4 or axes != [1, 2] or interp.mean_form != "int":
                    raise NotImplementedError("MEAN other than the integer form over H, W")
                materialise(op.inputs[0])
                s_in, zp_in = interp._q(op.inputs[0])
                s_out, zp_out = interp._q(o0)
                npos = x.shape[1] * x.shape[2]
                mult, shift = ig.quantize_multiplier(float(np.float32(s_in)) / float(np.float32(s_out)))
                fold = min(npos.bit_length() - 1, 32, 31 + shift)
                place(o0, y.size)
                ops.append(self._rec(self.MEAN, op.inputs[0], -1, o0, y.size, p=[npos, x.shape[3], zp_in, int((mult << fold) // npos), shift - fold, zp_out]))
            elif n == "FULLY_CONNECTED":
                materialise(op.inputs[0])
                wt = m.tensors[op.inputs[1]]
                w = np.ascontiguousarray(wt.data, np.int8)
                bias = np.ascontiguousarray(m.tensors[op.inputs[2]].data, np.int32) if len(op.inputs) > 2 and op.inputs[2] >= 0 else None
                _, zp_in = interp._q(op.inputs[0])
                _, zp_out = interp._q(o0)
                pr = interp._prep[op.index]
                lo, hi = (int(v) for v in pr["act"])
                self._keep += [a for a in (w, bias, pr["mult"], pr["shift"]) if a is not None]
                place(o0, y.size)
                ops.append(self._rec(self.FC, op.inputs[0], -1, o0, y.size, p=[w.shape[1], w.shape[0], zp_in, zp_out, lo, hi], ptr=[w, bias, pr["mult"], pr["shift"]]))
            elif n == "LOGISTIC":
                if interp.logistic_form != "lut":
                    raise NotImplementedError("fixed-point LOGISTIC")
                materialise(op.inputs[0])
                lut = np.ascontiguousarray(interp.logistic_lut(op), np.int8)
                self._keep.append(lut)
                place(o0, y.size)
                ops.append(self._rec(self.LUT, op.inputs[0], -1, o0, y.size, ptr=[lut]))
            else:
                raise NotImplementedError(f"operator {n} is not part of the C program (use CpuInt8Path)")
        if not hasattr(self, "out_elems"):
            raise NotImplementedError("the graph does not end in DEQUANTIZE")
        self.tensor_off = off
        self.arena_bytes = pos
        n_t = max(off) + 1
        self._off = np.full(n_t, -1, np.int64)
        for t, v in off.items():
            self._off[t] = v
        self._ops = (self._Op * len(ops))(*ops)
        self.n_ops = len(ops)
        self.lib.oi_program_run.restype = ctypes.c_int
        self.shapes = {t: env[t].shape[1:] for t in off}

    def _rec(self, kind, in0, in1, out, n, p=(), f=(), ptr=()):
        r = self._Op()
        r.kind, r.in0, r.in1, r.out, r.n = kind, in0, in1, out, int(n)
        for i, v in enumerate(p):
            r.p[i] = int(v)
        for i, v in enumerate(f):
            r.f[i] = float(v)
        for i, a in enumerate(ptr):
            r.ptr[i] = None if a is None else a.ctypes.data
        return r

    def invoke(self, x: np.ndarray, return_all: bool = False):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.in_elems)
        B = x.shape[0]
        out = np.empty((B, self.out_elems), np.float32)
        keep = np.empty((B, self.arena_bytes), np.int8) if return_all else None
        rc = self.lib.oi_program_run(self._ops, self.n_ops, _p(x), B, ctypes.c_int64(self.in_elems), _p(out), ctypes.c_int64(self.out_elems),
                                     self._off.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(self.arena_bytes), _i8(keep) if keep is not None else None)
        if rc != 0:
            raise MemoryError("oi_program_run could not allocate a thread's arena")
        if not return_all:
            return out
        env = {t: keep[:, o:o + int(np.prod(self.shapes[t]))].reshape((B,) + tuple(self.shapes[t])) for t, o in self.tensor_off.items()}
        return out, env

    def spectrogram(self, audio, hop, width, out=None):
        return self.path.spectrogram(audio, hop, width, out=out)


def _aligned(n: int, dtype):
    """Zeroed array of n elements whose data is 64-byte aligned (vector loads of the packed weights)."""
    item = np.dtype(dtype).itemsize
    raw = np.zeros(n * item + 64, np.uint8)
    o = (-raw.ctypes.data) % 64
    return raw[o:o + n * item].view(dtype)
