"""Audio-frontend registry: name -> static facts about a frontend.

API of the reference's birdnet_stm32/models/registry.py:12-140 (``FrontendInfo``,
``register_frontend``, ``get_frontend_info``, ``list_frontends``, ``is_precomputed``,
``is_n6_compatible``) with the same five built-in entries and the same error behaviour:
re-registering a name raises ``ValueError("... already registered")``, looking up an unknown
name raises ``KeyError("... not registered ...")``.

``hip_path`` is an addition: whether this build has a GPU path for the frontend (the
registry is how callers discover that without importing kernels).
"""

from __future__ import annotations

from dataclasses import dataclass

__all__ = ["FrontendInfo", "register_frontend", "get_frontend_info", "list_frontends", "is_precomputed", "is_n6_compatible"]


@dataclass(frozen=True)
class FrontendInfo:
    """Static description of one frontend (fields as in the reference, :12-29)."""

    name: str
    mode: str  # AudioFrontendLayer mode: 'precomputed' | 'hybrid' | 'raw'
    precomputed: bool  # spectrogram computed outside the model graph
    n6_compatible: bool
    description: str = ""
    hip_path: bool = False  # MI355X kernels exist for this frontend in this build


_TABLE: dict[str, FrontendInfo] = {}


def register_frontend(info: FrontendInfo) -> None:
    if info.name in _TABLE:
        raise ValueError(f"Frontend '{info.name}' is already registered.")
    _TABLE[info.name] = info


def list_frontends() -> list[str]:
    return sorted(_TABLE)


def get_frontend_info(name: str) -> FrontendInfo:
    try:
        return _TABLE[name]
    except KeyError:
        raise KeyError(f"Frontend '{name}' is not registered. Available: {list_frontends()}") from None


def is_precomputed(name: str) -> bool:
    return get_frontend_info(name).precomputed


def is_n6_compatible(name: str) -> bool:
    return get_frontend_info(name).n6_compatible


for _row in (
    # name       mode           precomputed  n6    hip    description
    ("librosa", "precomputed", True, True, True, "Host-side mel spectrogram; the model passes it through."),
    ("hybrid", "hybrid", False, True, True, "Linear STFT magnitude outside the model, 1x1 mel mixer + magnitude scaling inside."),
    ("raw", "raw", False, True, True, "Waveform in, learned strided filterbank inside the model (T < 65536 on the STM32N6)."),
    ("mfcc", "precomputed", True, True, True, "Host-side MFCC (mel -> dB -> DCT -> truncate); passed through."),
    ("log_mel", "precomputed", True, True, True, "Host-side log1p mel spectrogram; passed through."),
):
    register_frontend(FrontendInfo(*_row[:4], description=_row[5], hip_path=_row[4]))
del _row
