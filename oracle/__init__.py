"""CPU oracle for the birdnet-stm32 per-chunk inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or executed by
the product path (``birdnet-stm32_amd/``).  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

Each module restates one slice of the reference algorithm in numpy (or plain C under
``oracle/c``) and cites the reference file:line it follows.  Where the arithmetic lives in
a third-party dependency that is not vendored in the reference (librosa 0.11.0,
tensorflow 2.19.0 / TFLite builtin kernels — reference: requirements.txt:1-2) the
published algorithm is restated and anchored on the reference's call sites.

Pinning status (see DESIGN.md "Oracle"):
* mel filterbank: pinned against the shipped checkpoint's ``mel_mixer`` weights (<=2e-9).
* PWL constants, folded-BN <-> INT8 weight/bias consistency: pinned across the shipped
  ``.keras`` and ``.tflite`` artefacts.
* pooling / ModelConfig / chunking: pinned against outputs of the importable reference
  modules (fixtures under tests/golden, generator script committed).
* FFT butterflies and mel formulas: pinned against the reference's own firmware C
  (``oracle/_ref``; different framing than the evaluate path, so butterfly-level only).
* STFT framing (librosa), Keras float logits, TFLite INT8 logits: **parity unpinned** —
  TensorFlow and librosa cannot be imported here and the reference's tests hold no
  golden vectors for them.
"""
