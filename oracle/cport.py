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
FW_LIB = os.path.join(_HERE, "_ref", "libfw_ref.so")

_f = ctypes.POINTER(ctypes.c_float)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f)


class CpuFloatPath:
    """audio -> scores for a NetSpec of plain DS blocks (the shipped checkpoint), in C with OpenMP."""

    def __init__(self, spec):
        self.lib = ctypes.CDLL(CPU_LIB)
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
