"""Test-set discovery for the evaluate CLI (reference: birdnet_stm32/data/dataset.py:49-99).

``load_file_paths_from_directory(root, classes, max_samples, exts)`` walks ``root/<class>/*.<ext>``,
keeps files whose parent directory is in ``classes`` (when given), caps each class at ``max_samples``
by a uniform random draw, shuffles the result with ``numpy.random`` and returns ``(paths, classes)`` —
noise-like class names ('noise', 'silence', 'background', 'other') are dropped from the class list but
their files stay.  ``os.walk`` replaces ``tf.io.gfile.walk``.
"""

from __future__ import annotations

import os

import numpy as np

SUPPORTED_AUDIO_EXTS = (".wav", ".mp3", ".flac", ".ogg", ".m4a")
_NOISE = {"noise", "silence", "background", "other"}


def load_file_paths_from_directory(directory: str, classes: list[str] | None = None, max_samples: int | None = None,
                                   exts: tuple = SUPPORTED_AUDIO_EXTS) -> tuple[list[str], list[str]]:
    by_class: dict[str, list[str]] = {}
    for root, _dirs, names in os.walk(directory):
        label = os.path.basename(root)
        if classes is not None and label not in classes:
            continue
        for name in names:
            if name.lower().endswith(exts):
                by_class.setdefault(label, []).append(os.path.join(root, name))
    paths: list[str] = []
    for files in by_class.values():
        if max_samples is not None and 0 < max_samples < len(files):
            files = [files[i] for i in np.random.permutation(len(files))[:max_samples]]
        paths.extend(files)
    np.random.shuffle(paths)
    return paths, sorted(c for c in by_class if c.lower() not in _NOISE)
