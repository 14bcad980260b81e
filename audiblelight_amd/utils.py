"""Small host-side helpers with the reference's names and contracts (audiblelight/utils.py)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Union

import numpy as np


def tiny(x) -> float:
    """Smallest positive normal of x's float dtype; float32 for non-float input (utils.py:691-706)."""
    dt = np.asarray(x).dtype
    if not (np.issubdtype(dt, np.floating) or np.issubdtype(dt, np.complexfloating)):
        dt = np.dtype(np.float32)
    return np.finfo(dt).tiny


def pad_or_truncate_audio(audio: np.ndarray, desired_samples: int, pad_mode: str = "constant") -> np.ndarray:
    """(C, L) -> (C, desired_samples) by right padding or truncation (utils.py:667-688)."""
    have = audio.shape[1]
    if have < desired_samples:
        return np.pad(audio, ((0, 0), (0, int(desired_samples) - have)), mode=pad_mode)
    if have > desired_samples:
        return audio[:, : int(desired_samples)]
    return audio


def validate_shape(shape_a: tuple, shape_b: tuple) -> None:
    """Raise ValueError when two shapes differ in a dimension both define (utils.py:483-503)."""
    n = max(len(shape_a), len(shape_b))
    pa = tuple(shape_a) + (None,) * (n - len(shape_a))
    pb = tuple(shape_b) + (None,) * (n - len(shape_b))
    for i, (a, b) in enumerate(zip(pa, pb)):
        if a is not None and b is not None and a != b:
            raise ValueError(f"Incompatible shapes at index {i}: {a} != {b} (full shapes: {pa} vs {pb})")


def valid_audio(y) -> bool:
    """The checks librosa.util.valid_audio performs on the path (synthesize.py:398,552,603)."""
    if not isinstance(y, np.ndarray):
        raise ValueError("Audio data must be of type numpy.ndarray")
    if not np.issubdtype(y.dtype, np.floating):
        raise ValueError("Audio data must be floating-point")
    if y.ndim == 0:
        raise ValueError("Audio data must be at least one-dimensional")
    if not np.isfinite(y).all():
        raise ValueError("Audio buffer is not finite everywhere")
    return True


class LazyAudioDict(OrderedDict):
    """``event.spatial_audio``-style mapping whose values may live in HBM.

    A value is either an ndarray or a zero-argument callable that downloads (and post-processes)
    the array on first access; reads always hand back ``np.ndarray`` as the reference does
    (tests/test_synthesize.py:85,110,202 check isinstance(..., np.ndarray)).
    """

    def _resolve(self, key, val):
        if callable(val):
            val = val()
            super().__setitem__(key, val)
        return val

    def __getitem__(self, key):
        return self._resolve(key, super().__getitem__(key))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def device_source(self, key):
        """The (RenderResult, event index) behind a value that is STILL the device-backed render, else None (the value
        was read into a host array and possibly edited, replaced by the user, or never came from this package)."""
        if not self.is_resident(key):
            return None
        return getattr(super().__getitem__(key), "al_device_source", None)

    def is_resident(self, key) -> bool:
        """True while the value has not been copied to the host yet."""
        return key in self and callable(super().__getitem__(key))


def as_lazy(mapping) -> LazyAudioDict:
    if isinstance(mapping, LazyAudioDict):
        return mapping
    out = LazyAudioDict()
    if mapping:
        for k, v in dict(mapping).items():
            OrderedDict.__setitem__(out, k, v)
    return out

