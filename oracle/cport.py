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

    def spectrogram(self, audio: np.ndarray, hop: int, width: int) -> np.ndarray:
        """Normalised |STFT| [B, 257, width, 1] through the float port's C STFT (OpenMP over chunks)."""
        lib = ctypes.CDLL(os.path.join(NATIVE_DIR, "liboracle_cpu.so") if getattr(self, "native", False) else CPU_LIB)
        x = np.ascontiguousarray(audio, np.float32)
        S = np.empty((x.shape[0], 257, width), np.float32)
        lib.oc_stft_norm(_p(x), x.shape[0], x.shape[1], hop, width, _p(S))
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
