"""Channel arithmetic shared by the model builder (reference: birdnet_stm32/models/blocks.py:13-24)."""

from birdnet_stm32.models._netspec import make_divisible as _make_divisible

__all__ = ["_make_divisible"]
