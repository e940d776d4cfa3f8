"""birdnet_stm32 — MI355X-native host package for the per-chunk inference hot path.

Same module paths as the reference package for everything on that path
(``models.runners``, ``models.registry``, ``models.frontend``, ``audio.spectrogram``,
``audio.io``, ``evaluation.{metrics,pooling}``, ``training.config``, ``cli.evaluate``); the
compute is done by ``libbirdnet_hip.so`` (see ``include/birdnet_hip.h``).  Training, conversion
and STM32 deployment are out of scope.
"""

__version__ = "0.1.0+mi355x"
