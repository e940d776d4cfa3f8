"""Frontend names and the hybrid-frontend description.

Mirrors the name handling of the reference's birdnet_stm32/models/frontend.py:24-53:
``VALID_FRONTENDS``, the deprecated aliases ``precomputed -> librosa`` and ``tf -> raw`` (with a
``DeprecationWarning``), and ``ValueError("Invalid audio frontend ...")`` for anything else.
The Keras layer itself (frontend.py:59-384) has no counterpart object here: its computation is
the ``frontend`` record of a NetSpec (``_netspec.py``), lowered to the mel/PWL kernels.
"""

from __future__ import annotations

import warnings

VALID_FRONTENDS = ("librosa", "hybrid", "raw", "mfcc", "log_mel")
_FRONTEND_ALIASES = {"precomputed": "librosa", "tf": "raw"}


def normalize_frontend_name(name: str) -> str:
    """Canonical frontend name; aliases warn, unknown names raise ``ValueError``."""
    if name in VALID_FRONTENDS:
        return name
    if name in _FRONTEND_ALIASES:
        target = _FRONTEND_ALIASES[name]
        warnings.warn(f"Frontend name '{name}' is deprecated, use '{target}' instead.", DeprecationWarning, stacklevel=2)
        return target
    raise ValueError(f"Invalid audio frontend: '{name}'. Valid options: {VALID_FRONTENDS}")
