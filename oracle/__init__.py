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
* STFT: ``stft_magnitude`` with the firmware's framing switches is pinned against the
  reference's ``audio_stft.c`` (same function, two switches); the evaluate path's framing
  is cross-checked with ``scipy.signal.ShortTimeFFT``; ``np.abs(complex64)`` is pinned
  to the installed numpy.  The librosa call itself stays **unpinned**.
* gemmlowp / TFLite primitives (SRDHM, RoundingDivideByPOT, MultiplyByQuantizedMultiplier,
  QuantizeMultiplier): pinned against the published definitions written out on Python
  integers (tests/conftest.py), on the CPU and on the device.
* Operators with more than one published form are ALL restated, behind switches of
  ``Int8Interpreter``, and the choice is counted on the shipped checkpoint
  (tests/test_oracle_pinning.py): int8 SOFTMAX (``softmax_form``: fixed point / float
  table), int8 LOGISTIC (``logistic_form``: float table / gemmlowp fixed point —
  IDENTICAL on all 256 inputs of the shipped head, so the choice cannot matter there),
  int8 MEAN (``mean_form``: count folded into the integer multiplier / float-arithmetic
  ``QuantizedMeanOrSum`` — 3-4 % of the pooled bytes differ by one step, so the device
  plan carries the same switch: ``lower_i8(mean_form=...)``).
* The C ports (``oracle/c``) are pinned to the numpy modules tensor for tensor, in both
  builds (portable, and ``-march=native`` with the AVX-512 VNNI paths).
* ``tests/golden/oracle_vectors.npz``: stage outputs and per-tensor checksums of the shipped
  graph on the reference's test signals, configs[4] in float32 and through the INT8
  exporter; the GPU suite checks the HIP path against these COMMITTED values
  (tests/test_gpu_golden.py), not only against the live oracle.
* Keras float logits, TFLite INT8 logits against a real interpreter: **parity unpinned** —
  TensorFlow and librosa cannot be imported here and the reference's tests hold no
  golden vectors for them.
"""
