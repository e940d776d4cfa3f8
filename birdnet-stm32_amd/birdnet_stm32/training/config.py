"""``ModelConfig``: the JSON sidecar every checkpoint carries (``<model>_model_config.json``).

Same schema, defaults, validation messages and legacy tolerance as the reference's
birdnet_stm32/training/config.py:14-148: unknown keys are dropped on load, missing keys take the
defaults below, ``class_names`` must match ``num_classes`` when given.  Note the consequence the
reference shares: a legacy file without ``use_se`` / ``use_inverted_residual`` reads back as
``True`` for both although the shipped network has neither — the architecture therefore always
comes from the weights file, never from this JSON.
"""

from __future__ import annotations

import dataclasses
import json
from dataclasses import dataclass, field
from pathlib import Path

_FRONTENDS = ("hybrid", "librosa", "log_mel", "mfcc", "raw")
_MAG_SCALES = ("db", "none", "pcen", "pwl")


@dataclass
class ModelConfig:
    # audio
    sample_rate: int = 24000
    num_mels: int = 64
    spec_width: int = 256
    fft_length: int = 512
    chunk_duration: float = 3.0
    hop_length: int = 281
    audio_frontend: str = "hybrid"
    mag_scale: str = "pwl"
    n_mfcc: int = 20
    # architecture
    embeddings_size: int = 256
    alpha: float = 1.0
    depth_multiplier: int = 1
    use_se: bool = True
    se_reduction: int = 8
    use_inverted_residual: bool = True
    expansion_factor: int = 2
    use_attention_pooling: bool = False
    dropout_rate: float = 0.5
    frontend_trainable: bool = False
    # classes
    num_classes: int = 0
    class_names: list[str] = field(default_factory=list)

    def __post_init__(self) -> None:
        for name in ("sample_rate", "num_mels", "spec_width", "fft_length", "chunk_duration", "alpha"):
            value = getattr(self, name)
            if value <= 0:
                raise ValueError(f"{name} must be positive, got {value}")
        if self.audio_frontend not in _FRONTENDS:
            raise ValueError(f"audio_frontend '{self.audio_frontend}' not in {sorted(_FRONTENDS)}")
        if self.mag_scale not in _MAG_SCALES:
            raise ValueError(f"mag_scale '{self.mag_scale}' not in {sorted(_MAG_SCALES)}")
        if self.depth_multiplier < 1:
            raise ValueError(f"depth_multiplier must be >= 1, got {self.depth_multiplier}")
        if not 0 <= self.dropout_rate < 1:
            raise ValueError(f"dropout_rate must be in [0, 1), got {self.dropout_rate}")
        if self.num_classes < 0:
            raise ValueError(f"num_classes must be >= 0, got {self.num_classes}")
        if self.class_names and len(self.class_names) != self.num_classes:
            raise ValueError(f"class_names length ({len(self.class_names)}) != num_classes ({self.num_classes})")

    def to_dict(self) -> dict:
        return dataclasses.asdict(self)

    def save(self, path: str | Path) -> None:
        target = Path(path)
        target.parent.mkdir(parents=True, exist_ok=True)
        target.write_text(json.dumps(self.to_dict(), indent=2) + "\n")

    @classmethod
    def from_dict(cls, data: dict) -> "ModelConfig":
        known = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in data.items() if k in known})

    @classmethod
    def load(cls, path: str | Path) -> "ModelConfig":
        return cls.from_dict(json.loads(Path(path).read_text()))
