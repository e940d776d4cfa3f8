"""Inference runners with the reference's ``predict(x_batch) -> [B, C]`` interface, on MI355X.

Drop-in for birdnet_stm32/models/runners.py (reference :14-114): ``load_model_runner(path)``
dispatches on the file suffix exactly as the reference does (``.tflite`` -> INT8 graph,
anything else -> Keras archive) and returns an object whose ``predict`` takes the same float32
batches (hybrid frontend: ``[B, 257, W, 1]`` normalised linear spectrograms) and returns fresh
float32 ``[B, C]`` arrays.  Instead of ``tf.lite.Interpreter`` / ``tf.keras`` the runner
lowers the file to a device plan and executes it through ``libbirdnet_hip.so``.

Beyond the reference interface the runner exposes device-resident entry points used by the
batched evaluator and the throughput benchmark:

* ``predict_device(x)``  — torch tensor in HBM -> torch tensor in HBM, no host copies;
* ``infer_audio_device(audio)`` — ``[B, T]`` waveform chunks in HBM -> scores, the STFT, frontend
  and network all on the GPU.

Like the reference's interpreter (``resize_tensor_input``), any batch size is accepted; batches
larger than the workspace are processed in slices.
"""

from __future__ import annotations

import ctypes

import numpy as np

from birdnet_stm32 import _hip
from birdnet_stm32.models import _pack as pk


class HipRunner:
    """Executes a lowered plan on one MI355X."""

    def __init__(self, plan: pk.Plan, device: int = 0, max_batch: int = 1024, ctx: "_hip.Context | None" = None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible: HipRunner has no CPU fallback")
        self._torch = torch
        self.plan = plan
        self.device = torch.device("cuda", device)
        self.ctx = ctx if ctx is not None else _hip.Context(device, max_batch)   # (load_model_runner creates it early when it prepares the pipeline)
        self.model = _hip.Model(self.ctx, plan.to_blob())
        self.lib = self.ctx.lib
        info = self.model.info
        self.max_batch = int(max_batch)
        self.num_classes = int(info.num_classes)
        self.input_elems = int(info.input_elems)
        self.fft_bins, self.spec_width = int(info.fft_bins), int(info.spec_width)
        self.dtype = "i8" if info.dtype == pk.DTYPE_I8 else "f32"
        self.input_kind = int(info.input_kind)
        self._precomputed: dict | None = None  # spectrogram settings of a precomputed-frontend model (configure_precomputed)

    # -- helpers ----------------------------------------------------------------------------
    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _check_dev(self, t, name: str):
        torch = self._torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous float32 CUDA tensor")
        if t.device != self.device:
            raise ValueError(f"{name} lives on {t.device}, runner on {self.device}")

    # -- reference interface ------------------------------------------------------------------
    def predict(self, x_batch: np.ndarray) -> np.ndarray:
        """Forward pass on a host batch; same contract as the reference's runners (:29-45, :82-95)."""
        torch = self._torch
        x = np.ascontiguousarray(np.asarray(x_batch).astype(np.float32, copy=False))
        if x.ndim < 2 or int(np.prod(x.shape[1:])) != self.input_elems:
            want = f"[B, {self.fft_bins}, {self.spec_width}, 1]" if self.fft_bins else f"[B, {self.input_elems}, 1]"
            raise ValueError(f"expected input of shape {want}, got {x.shape}")
        B = x.shape[0]
        out = np.empty((B, self.num_classes), np.float32)
        for b0 in range(0, B, self.max_batch):
            xb = torch.from_numpy(x[b0 : b0 + self.max_batch].reshape(-1, self.input_elems)).to(self.device)
            out[b0 : b0 + xb.shape[0]] = self.predict_device(xb).cpu().numpy()
        return out

    # -- device-resident interface --------------------------------------------------------------
    def predict_device(self, x, minmax=None, return_logits: bool = False):
        """``x``: CUDA float32 ``[B, input_elems]`` (any trailing shape with that many elements)."""
        torch = self._torch
        self._check_dev(x, "x")
        B = x.shape[0]
        if x.numel() != B * self.input_elems:
            raise ValueError(f"input has {x.numel() // max(B, 1)} elements per chunk, model expects {self.input_elems}")
        scores = torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        logits = torch.empty_like(scores) if return_logits else None
        if minmax is not None:
            self._check_dev(minmax, "minmax")
        with torch.cuda.device(self.device):
            for b0 in range(0, B, self.max_batch):
                nb = min(self.max_batch, B - b0)
                _hip.check(
                    self.lib.bn_forward(
                        self.model.handle,
                        x[b0:].data_ptr(),
                        minmax[b0:].data_ptr() if minmax is not None else None,
                        nb,
                        scores[b0:].data_ptr(),
                        logits[b0:].data_ptr() if logits is not None else None,
                        self._stream(),
                    )
                )
        return (scores, logits) if return_logits else scores

    def configure_precomputed(self, audio_frontend: str, sample_rate: int, mag_scale: str = "none", n_fft: int = 512,
                              num_mels: int | None = None, n_mfcc: int = 20) -> None:
        """Tell a precomputed-frontend runner ('librosa' | 'log_mel' | 'mfcc') how its input spectrograms are made, so that
        ``infer_audio_device`` can start from waveforms.  The graph itself only passes the map through (reference:
        models/frontend.py:296-297); which host transform feeds it is a property of the training config
        (``audio_frontend``, ``mag_scale``, ``num_mels``, ``n_mfcc``: reference evaluation/metrics.py:49-54, data/generator)."""
        from birdnet_stm32.models.frontend import normalize_frontend_name

        name = normalize_frontend_name(audio_frontend)
        if name not in ("librosa", "log_mel", "mfcc"):
            raise ValueError(f"'{audio_frontend}' is not a precomputed frontend")
        if self.input_kind != pk.INPUT_MEL:
            raise ValueError("this model does not take precomputed spectrograms")
        rows = self.input_elems // self.spec_width
        mels = int(num_mels) if num_mels is not None else rows
        if (n_mfcc if name == "mfcc" else mels) != rows:
            raise ValueError(f"model input has {rows} rows; frontend '{name}' with num_mels={mels}, n_mfcc={n_mfcc} does not produce that")
        self._precomputed = {"mode": {"librosa": "mel", "log_mel": "log_mel", "mfcc": "mfcc"}[name], "sample_rate": int(sample_rate),
                             "mag_scale": mag_scale if name == "librosa" else "none", "n_fft": int(n_fft), "mel_bins": mels, "n_mfcc": int(n_mfcc)}

    def infer_audio_device(self, audio, hop: int | None = None, return_logits: bool = False, out=None):
        """``audio``: CUDA float32 ``[B, T]`` chunks -> scores ``[B, C]`` (STFT + frontend + network on the GPU)."""
        torch = self._torch
        self._check_dev(audio, "audio")
        if self.input_kind == pk.INPUT_MEL:
            if self._precomputed is None:
                raise ValueError("precomputed-frontend model: call configure_precomputed(audio_frontend, sample_rate, mag_scale, ...) first")
            from birdnet_stm32.audio.spectrogram import mel_spectrograms_device

            c = self._precomputed
            spec = mel_spectrograms_device(self.ctx, audio, c["sample_rate"], c["n_fft"], c["mel_bins"], self.spec_width, c["mag_scale"],
                                           c["mode"], c["n_mfcc"])
            res = self.predict_device(spec.view(spec.shape[0], -1), return_logits=return_logits)
            if out is not None:
                out.copy_(res[0] if return_logits else res)
            return res
        if self.input_kind == pk.INPUT_WAVEFORM:
            # raw frontend: the model input is the chunk divided by (its peak + 1e-6) (reference: evaluation/metrics.py:62-69)
            if audio.shape[1] != self.input_elems:
                raise ValueError(f"raw-frontend model expects {self.input_elems} samples per chunk, got {audio.shape[1]}")
            x = torch.empty_like(audio)
            with torch.cuda.device(self.device):
                _hip.check(self.lib.bn_chunk_peak_normalize(self.ctx.handle, audio.data_ptr(), audio.shape[0], audio.shape[1], 1e-6,
                                                            x.data_ptr(), self._stream()))
            res = self.predict_device(x, return_logits=return_logits)
            if out is not None:
                out.copy_(res[0] if return_logits else res)
            return res
        B, T = audio.shape
        hop = int(hop) if hop is not None else T // self.spec_width
        scores = out if out is not None else torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        logits = torch.empty_like(scores) if return_logits else None
        with torch.cuda.device(self.device):
            for b0 in range(0, B, self.max_batch):
                nb = min(self.max_batch, B - b0)
                _hip.check(
                    self.lib.bn_infer_audio(
                        self.model.handle, audio[b0:].data_ptr(), nb, T, hop, scores[b0:].data_ptr(),
                        logits[b0:].data_ptr() if logits is not None else None, self._stream(),
                    )
                )  # fmt: skip
        return (scores, logits) if return_logits else scores

    def stft_device(self, audio, n_fft: int = 512, hop: int | None = None, spec_width: int | None = None, normalize: bool = True,
                    exact: bool = False):
        """Batched ``get_spectrogram_from_audio(mel_bins=-1)`` on the GPU: ``[B, T]`` -> ``[B, n_fft//2+1, W]``."""
        return stft_device(self.ctx, audio, n_fft, hop, spec_width or self.spec_width, normalize, exact=exact)

    def op_output(self, op_index: int, B: int) -> np.ndarray:
        """Test hook: activation written by plan operator ``op_index`` in the last forward call."""
        torch = self._torch
        op = self.plan.ops[op_index]
        per = ctypes.c_size_t()
        _hip.check(self.lib.bn_debug_op_output(self.model.handle, op_index, B, None, 0, ctypes.byref(per), None))
        buf = torch.empty((B, per.value), dtype=torch.uint8, device=self.device)
        _hip.check(self.lib.bn_debug_op_output(self.model.handle, op_index, B, buf.data_ptr(), buf.numel(), ctypes.byref(per), self._stream()))
        torch.cuda.synchronize(self.device)
        # activations are densely packed [B][elements]; the slot's bytes_per_chunk is only its capacity
        raw = buf.cpu().numpy().reshape(-1)
        n = int(np.prod(op.out_shape))
        dt = np.dtype(op.out_dtype)
        return raw[: B * n * dt.itemsize].copy().view(dt).reshape(B, *op.out_shape)

    def input_bytes(self, B: int) -> np.ndarray:
        """Test hook: the int8 bytes QUANTIZE made of the spectrograms of the last ``infer_audio_device`` call, ``[B, 257, W]``."""
        torch = self._torch
        out = torch.empty((B, self.fft_bins, self.spec_width), dtype=torch.int8, device=self.device)
        _hip.check(self.lib.bn_debug_input_bytes(self.model.handle, B, out.data_ptr(), self._stream()))
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy()

    def guard_stats(self, B: int) -> dict:
        """Test hook: counters of the exactness pass of the last ``infer_audio_device`` call (INT8 plans)."""
        out = (ctypes.c_int64 * 8)()
        _hip.check(self.lib.bn_debug_guard_stats(self.model.handle, B, out))
        return {"listed": int(out[0]), "listed_max": int(out[1]), "dirty_blocks": int(out[2]), "whole_minmax": int(out[3]), "whole_fix": int(out[4]),
                "audited": int(out[5]), "audit_violations": int(out[6]), "interval_min": int(out[7])}

    def tail_form(self) -> tuple[int, int]:
        """Test hook: (form, LDS bytes) of the fused INT8 tail operator this plan can run — 0 none, 1 ``i8_tail_kernel``, 2 also
        ``i8_tail2_kernel`` (depthwise stage on the matrix cores)."""
        form, lds = ctypes.c_int(0), ctypes.c_int(0)
        _hip.check(self.lib.bn_debug_tail_form(self.model.handle, ctypes.byref(form), ctypes.byref(lds)))
        return form.value, lds.value

    def mid_form(self) -> tuple[int, int]:
        """Test hook: (1, LDS bytes) when the plan's fused stage-2 chain (``i8_mid2_kernel``) passed the library's LDS plan, else (0, 0)."""
        form, lds = ctypes.c_int(0), ctypes.c_int(0)
        _hip.check(self.lib.bn_debug_mid_form(self.model.handle, ctypes.byref(form), ctypes.byref(lds)))
        return form.value, lds.value

    # -- per-operator timing (HIP events on the launch stream) ------------------------------------
    def profile(self, enable: bool) -> None:
        _hip.check(self.lib.bn_profile_enable(self.model.handle, int(bool(enable))))

    def profile_only(self, op_index: int) -> None:
        """Bracket only operator ``op_index`` with events (-1: all operators again)."""
        _hip.check(self.lib.bn_profile_only(self.model.handle, int(op_index)))

    def profile_collect(self) -> list[dict]:
        """Elapsed time per plan operator since the last collect; behind the operators: the STFT stage and (INT8 plans from audio)
        the two float64 passes of the exactness pass (csrc/bn_stft_exact.hip)."""
        n = len(self.plan.ops) + 3
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int64 * n)()
        _hip.check(self.lib.bn_profile_collect(self.model.handle, ms, cnt, n))
        rows = []
        for i in range(n):
            if i < len(self.plan.ops):
                op = self.plan.ops[i]
                rows.append({"op": i, "kind": pk.KIND_NAMES[op.kind], "name": op.name, "ms": ms[i], "launches": int(cnt[i]), "p": list(op.p)})
            else:
                kind, name = (("stft512", "stft"), ("stft_minmax_exact", "exact min/max"), ("stft512_f64_list", "float64 pass + redo"))[i - len(self.plan.ops)]
                rows.append({"op": i, "kind": kind, "name": name, "ms": ms[i], "launches": int(cnt[i]), "p": []})
        return rows

    def close(self):
        import sys

        pl = sys.modules.get("birdnet_stm32.audio.pipeline")
        if pl is not None:   # (a helper thread of prepare_for_evaluate may still be page-locking through this context)
            while pl._PREPARE:
                pl._PREPARE.pop().join()
        self.model.close()
        self.ctx.close()


def stft_device(ctx: _hip.Context, audio, n_fft: int = 512, hop: int | None = None, spec_width: int = 256, normalize: bool = True,
                return_minmax: bool = False, exact: bool = False):
    """``bn_stft_mag`` on CUDA tensors: float32 ``[B, T]`` -> float32 ``[B, n_fft//2+1, spec_width]``.

    ``exact=True`` calls ``bn_stft_mag_exact``: the reference's float64 arithmetic value for value (about ten times slower)."""
    import torch

    if not (audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous() and audio.dim() == 2):
        raise ValueError("audio must be a contiguous float32 CUDA tensor [B, T]")
    B, T = audio.shape
    hop = int(hop) if hop is not None else T // spec_width
    spec = torch.empty((B, n_fft // 2 + 1, spec_width), dtype=torch.float32, device=audio.device)
    minmax = torch.empty((B, 2), dtype=torch.float32, device=audio.device)
    with torch.cuda.device(audio.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(audio.device).cuda_stream)
        fn = ctx.lib.bn_stft_mag_exact if exact else ctx.lib.bn_stft_mag
        _hip.check(fn(ctx.handle, audio.data_ptr(), B, T, n_fft, hop, spec_width, int(normalize), spec.data_ptr(), minmax.data_ptr(), stream))
    return (spec, minmax) if return_minmax else spec


def lower_model_file(model_path: str, keep_all: bool = False, frontend_norm: bool | None = None, fuse: bool = True) -> pk.Plan:
    """Read a `.tflite` or `.keras` file and lower it to a device plan (no GPU needed)."""
    if model_path.lower().endswith(".tflite"):
        from birdnet_stm32.models._lower_i8 import lower_i8
        from birdnet_stm32.models._tflite_reader import load_tflite

        return lower_i8(load_tflite(model_path), keep_all=keep_all, fuse=fuse)
    from birdnet_stm32.models._keras_loader import load_keras_archive
    from birdnet_stm32.models._lower_f32 import lower_f32

    return lower_f32(load_keras_archive(model_path, frontend_norm=frontend_norm), keep_all=keep_all, fuse=fuse)


def load_model_runner(model_path: str, device: int = 0, max_batch: int = 1024, keep_all: bool = False,
                      frontend_norm: bool | None = None, fuse: bool = True, prepare_pipeline: bool = False) -> HipRunner:
    """Load a `.keras` or `.tflite` model and return a runner with ``predict()`` (reference :98-114).

    ``prepare_pipeline``: the caller is going to ``evaluate`` files with this runner (the CLI does) — the context is created first and, while the
    model file is parsed and lowered, a helper thread loads the library's code objects and page-locks the evaluate pipeline's staging slabs
    (``audio.pipeline.prepare_for_evaluate``: 768 MiB of page-locked memory, kept until ``release_pinned_slabs()``)."""
    ctx = None
    if prepare_pipeline:
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible: HipRunner has no CPU fallback")
        from birdnet_stm32.audio.pipeline import prepare_for_evaluate

        ctx = _hip.Context(device, max_batch)
        prepare_for_evaluate(ctx, torch)
    plan = lower_model_file(model_path, keep_all=keep_all, frontend_norm=frontend_norm, fuse=fuse)
    return HipRunner(plan, device=device, max_batch=max_batch, ctx=ctx)
