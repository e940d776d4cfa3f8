"""ctypes binding of ``libbirdnet_hip.so`` (C ABI: ``include/birdnet_hip.h``).

The reference binds its accelerated path through ``tf.lite.Interpreter`` (reference:
birdnet_stm32/models/runners.py:57); here the same role is played by a plain C ABI loaded
with ``ctypes`` (``cffi`` is not installed on the target image).  PyTorch is imported first on
purpose: it loads the process's HIP runtime (``libamdhip64.so``), and ``libbirdnet_hip.so``
must bind to that same runtime instance so that device pointers and ``hipStream_t`` handles
coming from torch tensors are valid inside the library.

There is no fallback: if the shared object is missing or no gfx950 device is present, loading
or context creation raises.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # .../birdnet-stm32_amd
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libbirdnet_hip.so")

ABI_VERSION = 1

# every symbol include/birdnet_hip.h declares
EXPORTS = (
    "bn_version", "bn_last_error", "bn_device_count", "bn_ctx_create", "bn_ctx_destroy", "bn_model_load",
    "bn_model_free", "bn_model_get_info", "bn_stft_mag", "bn_forward", "bn_infer_audio", "bn_debug_op_output",
    "bn_kernel_names", "bn_profile_enable", "bn_profile_collect", "bn_ingest_resample", "bn_ingest_chunks",
    "bn_pool_scores", "bn_mel_spectrogram", "bn_profile_only", "bn_chunk_peak_normalize", "bn_set_option", "bn_get_option", "bn_ctx_set_option", "bn_ctx_get_option", "bn_ctx_reset_options", "bn_preload_kernels", "bn_host_alloc_pinned", "bn_host_free_pinned", "bn_rank_orders",
    "bn_blob_check", "bn_debug_requant", "bn_stft_mag_exact", "bn_debug_input_bytes", "bn_debug_guard_stats", "bn_debug_tail_form", "bn_debug_mid_form",
)  # fmt: skip

# launcher switches of bn_set_option (include/birdnet_hip.h); the production defaults are what a fresh process has
OPTION_NAMES = ("f32_strip", "f32_strip_th", "f32_front_staged", "f32_front2", "f32_pwdw", "f32_tile_slice", "f32_pw_ws", "i8_pwdw", "i8_pw_lds", "i8_pw_forms", "i8_add_tab", "front_tpw", "wave_dwpw", "i8_strip", "i8_strip_mfdw", "i8_strip_th", "i8_dw_pool", "i8_tail_fclds", "i8_tail", "i8_tail_mfdw", "i8_mid",
                "i8_mel_generic", "stft_rowmajor", "stft_exact", "stft_flagcap", "stft_guard", "stft_audit", "stft_minint", "ingest_blk", "ingest_generic")


class BnModelInfo(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int32), ("input_kind", c_int32), ("input_elems", c_int32), ("fft_bins", c_int32),
        ("spec_width", c_int32), ("num_classes", c_int32), ("n_ops", c_int32), ("max_batch", c_int32),
        ("workspace_bytes", c_int64), ("const_bytes", c_int64),
    ]  # fmt: skip


class HipError(RuntimeError):
    """A libbirdnet_hip call returned a negative status."""


_lib = None


def load_library(path: str | None = None):
    """Load (once) and type the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("BIRDNET_HIP_LIB", LIB_PATH)
    if not os.path.isfile(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C birdnet-stm32_amd/csrc`). There is no CPU fallback for this path."
        )
    import torch  # noqa: F401  (brings in the HIP runtime this library must share)

    lib = ctypes.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f"{path} does not export {name}")
    lib.bn_version.restype = c_int
    lib.bn_last_error.restype = c_char_p
    lib.bn_device_count.restype = c_int
    lib.bn_ctx_create.argtypes = [c_int, c_int, POINTER(c_void_p)]
    lib.bn_ctx_destroy.argtypes = [c_void_p]
    lib.bn_ctx_destroy.restype = None
    lib.bn_model_load.argtypes = [c_void_p, c_char_p, c_size_t, POINTER(c_void_p)]
    lib.bn_model_free.argtypes = [c_void_p]
    lib.bn_model_free.restype = None
    lib.bn_model_get_info.argtypes = [c_void_p, POINTER(BnModelInfo)]
    lib.bn_stft_mag.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.bn_stft_mag_exact.argtypes = lib.bn_stft_mag.argtypes
    lib.bn_debug_input_bytes.argtypes = [c_void_p, c_int, c_void_p, c_void_p]
    lib.bn_debug_guard_stats.argtypes = [c_void_p, c_int, POINTER(c_int64)]
    lib.bn_debug_tail_form.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.bn_debug_mid_form.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.bn_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
    lib.bn_infer_audio.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.bn_debug_op_output.argtypes = [c_void_p, c_int, c_int, c_void_p, c_size_t, POINTER(c_size_t), c_void_p]
    lib.bn_kernel_names.restype = c_char_p
    lib.bn_profile_enable.argtypes = [c_void_p, c_int]
    lib.bn_profile_only.argtypes = [c_void_p, c_int]
    lib.bn_chunk_peak_normalize.argtypes = [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]
    lib.bn_profile_collect.argtypes = [c_void_p, POINTER(ctypes.c_double), POINTER(c_int64), c_int]
    lib.bn_ingest_resample.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                       c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.bn_ingest_chunks.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                     c_void_p]
    lib.bn_pool_scores.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]
    lib.bn_mel_spectrogram.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                       ctypes.c_double, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
    lib.bn_blob_check.argtypes = [c_char_p, c_size_t]
    lib.bn_debug_requant.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.bn_set_option.argtypes = [c_char_p, c_int]
    lib.bn_get_option.argtypes = [c_char_p, POINTER(c_int)]
    lib.bn_ctx_set_option.argtypes = [c_void_p, c_char_p, c_int]
    lib.bn_ctx_get_option.argtypes = [c_void_p, c_char_p, POINTER(c_int)]
    lib.bn_ctx_reset_options.argtypes = [c_void_p]
    lib.bn_preload_kernels.argtypes = [c_void_p]
    lib.bn_rank_orders.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.bn_host_alloc_pinned.argtypes = [c_void_p, ctypes.c_size_t]
    lib.bn_host_alloc_pinned.restype = c_void_p
    lib.bn_host_free_pinned.argtypes = [c_void_p]
    if lib.bn_version() != ABI_VERSION:
        raise RuntimeError(f"libbirdnet_hip ABI {lib.bn_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().bn_last_error()
        raise HipError(f"libbirdnet_hip error {rc}: {msg.decode('utf-8', 'replace') if msg else ''}")


def blob_check(blob: bytes) -> None:
    """``bn_blob_check``: raise :class:`HipError` unless ``blob`` is a well-formed plan (host only, no device needed)."""
    check(load_library().bn_blob_check(blob, len(blob)))


def set_option(name: str, value: int) -> None:
    """``bn_set_option``: process-wide launcher switch (A/B runs, tests)."""
    check(load_library().bn_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = c_int()
    check(load_library().bn_get_option(name.encode(), byref(v)))
    return int(v.value)


class options:
    """``with options(i8_strip=0, i8_strip_th=3): ...`` — set launcher switches, restore the previous values on exit."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def loaded_hip_runtimes() -> list[str]:
    """Paths of every libamdhip64 mapped in this process (must be exactly one)."""
    seen = set()
    with open("/proc/self/maps") as fh:
        for line in fh:
            if "libamdhip64" in line:
                seen.add(line.split()[-1])
    return sorted(seen)


class Context:
    """Owns a ``bn_ctx`` on one device."""

    def __init__(self, device: int = 0, max_batch: int = 1024):
        self.lib = load_library()
        self.device, self.max_batch = int(device), int(max_batch)
        h = c_void_p()
        check(self.lib.bn_ctx_create(self.device, self.max_batch, byref(h)))
        self.handle = h
        rts = loaded_hip_runtimes()
        if len(rts) != 1:
            raise RuntimeError(f"expected one HIP runtime in the process, found {rts}")

    def set_option(self, name: str, value: int) -> None:
        """``bn_ctx_set_option``: a launcher switch for this context only (``set_option`` of the module sets the process default)."""
        check(self.lib.bn_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        """The value the launches through this context see: its own override, else the process default."""
        v = c_int()
        check(self.lib.bn_ctx_get_option(self.handle, name.encode(), byref(v)))
        return v.value

    def reset_options(self) -> None:
        check(self.lib.bn_ctx_reset_options(self.handle))

    def alloc_pinned(self, nbytes: int) -> int:
        """Address of ``nbytes`` of page-locked host memory (``bn_host_alloc_pinned``; the GIL is released for the call); free with ``free_pinned``."""
        p = self.lib.bn_host_alloc_pinned(self.handle, int(nbytes))
        if not p:
            raise HipError(f"bn_host_alloc_pinned({nbytes}): {self.lib.bn_last_error().decode()}")
        return int(p)

    def free_pinned(self, ptr: int) -> None:
        check(self.lib.bn_host_free_pinned(c_void_p(int(ptr))))

    def preload_kernels(self) -> None:
        """Load every device code object of the library now instead of at each kernel's first launch (``bn_preload_kernels``)."""
        check(self.lib.bn_preload_kernels(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.bn_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


class Model:
    """Owns a ``bn_model`` (constants in HBM + activation workspace)."""

    def __init__(self, ctx: Context, blob: bytes):
        self.ctx = ctx
        self.lib = ctx.lib
        h = c_void_p()
        check(self.lib.bn_model_load(ctx.handle, blob, len(blob), byref(h)))
        self.handle = h
        info = BnModelInfo()
        check(self.lib.bn_model_get_info(h, byref(info)))
        self.info = info

    def close(self):
        if getattr(self, "handle", None):
            self.lib.bn_model_free(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


__all__ = ["load_library", "check", "Context", "Model", "HipError", "BnModelInfo", "EXPORTS", "LIB_PATH", "c_float", "set_option", "get_option",
           "options", "OPTION_NAMES"]
