"""Model registry and runners (reference: birdnet_stm32/models/__init__.py:17-65).

``register_model(name)`` decorates a builder, ``build_model(name, **kw)`` calls it, ``list_models()``
lists the names; duplicate registration raises ``ValueError``, an unknown name raises
``KeyError("Unknown model ...")``.  Builders return a :class:`NetSpec` (a plain topology + weights
record) instead of a ``tf.keras.Model``.
"""

from __future__ import annotations

from collections.abc import Callable
from typing import Any

_BUILDERS: dict[str, Callable[..., Any]] = {}


def register_model(name: str):
    def wrap(fn: Callable[..., Any]) -> Callable[..., Any]:
        if name in _BUILDERS:
            raise ValueError(f"Model '{name}' is already registered.")
        _BUILDERS[name] = fn
        return fn

    return wrap


def list_models() -> list[str]:
    return sorted(_BUILDERS)


def build_model(name: str, **kwargs: Any):
    if name not in _BUILDERS:
        raise KeyError(f"Unknown model: '{name}'. Available: {list_models()}")
    return _BUILDERS[name](**kwargs)


def _register_builtin() -> None:
    from birdnet_stm32.models.dscnn import build_dscnn_model

    _BUILDERS.setdefault("dscnn", build_dscnn_model)


_register_builtin()
