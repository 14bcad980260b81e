"""Sample-wise event augmentations on the GPU, with the reference's class names, ``params`` and
``to_dict`` layout (audiblelight/augmentation.py).

Covered: Gain, Invert, Reverse, Fade, Clipping, Distortion, Bitcrush, Preemphasis, Deemphasis,
TimeWarpSilence / Duplicate / Remove / Reverse and the peak normalisation of ``Event.load_audio``.
The stateful pedalboard effects (filters, dynamics, modulation, codecs, time-stretch) stay on
the host with the reference implementation: out of scope.

Definitions for effects whose reference arithmetic lives in un-vendored third-party wheels
(parity unpinned): Gain = x*10^(dB/20); Clipping = clamp at +-10^(dB/20);
Distortion = tanh(x*10^(dB/20)); Bitcrush = rint(x*2^bits)/2^bits (pedalboard 0.9.17);
Preemphasis / Deemphasis = librosa 0.11 ``effects.preemphasis`` / ``deemphasis`` including their
linear-extrapolation initial state.
"""
from __future__ import annotations

import ctypes as ct
import random as _random
from typing import Any, Optional

import numpy as np

from . import _hip, config
from .utils import tiny


def _renderer():
    from . import synthesize

    return synthesize.get_renderer()


def _sample(override, lo: float, hi: float) -> float:
    """Numeric override, object with ``rvs()``, or uniform(lo, hi) (Augmentation.sample_value, :62-89)."""
    if override is None:
        return float(np.random.uniform(lo, hi))
    if isinstance(override, (int, float, np.integer, np.floating)):
        return override
    if hasattr(override, "rvs"):
        return float(override.rvs())
    raise TypeError(f"Cannot handle type {type(override)}")


def _positive(x, cast=float):
    if not isinstance(x, (int, float, np.integer, np.floating)) or isinstance(x, bool):
        raise TypeError(f"Expected a numeric input, but got {type(x)}")
    if x < 0:
        raise ValueError(f"Expected a positive numeric input, but got {x}")
    return cast(x)


class DeviceClip:
    """A mono clip resident in HBM plus its length; FX ping-pong between two buffers.  A whole FX chain and the peak
    normalisation run on ONE DeviceClip: one upload, N kernel launches, at most one download.  ``device``: the samples are
    already in HBM (a view of the scene's staging arena, ``stage_clips``): nothing is uploaded here."""

    def __init__(self, renderer, host: Optional[np.ndarray] = None, device=None, n: Optional[int] = None):
        self.r = renderer
        if device is not None:
            self.n, self.buf = int(n), device
        else:
            self.n = int(host.shape[-1])
            self.buf = renderer.mem.upload(np.ascontiguousarray(host, dtype=np.float32))
        self.alt = None
        self.uploads, self.downloads = 1, 0   # PCIe crossings of the samples (tests assert the chain stays in HBM)

    def other(self, n: Optional[int] = None):
        n = self.n if n is None else n
        if self.alt is None or len(self.alt) < n:
            self.alt = self.r.mem.empty(n)
        return self.alt

    def swap(self, n: Optional[int] = None):
        self.buf, self.alt = self.alt, self.buf
        if n is not None:
            self.n = n

    def host(self, normalize: bool = False) -> np.ndarray:
        """The clip on the host; ``normalize``: peak-normalised (event.py:535-536) in a device COPY, so the resident clip
        stays as the chain left it (the renderer folds the same scalar into the clip spectra instead, ``al_clip_scales``)."""
        self.downloads += 1
        if not normalize:
            return self.r.mem.download(self.buf)[: self.n].astype(np.float32)
        r, dst = self.r, self.other()
        dst[: self.n] = self.buf[: self.n]
        scale = r.mem.empty(1)
        r.lib.call("al_peak_scale", r.mem.ptr(dst), self.n, ct.c_float(1.0), r.mem.ptr(scale), r.mem.stream())
        r.lib.call("al_scale_rows", r.mem.ptr(dst), self.n, r.mem.ptr(scale), r.mem.stream())
        return r.mem.download(dst)[: self.n].astype(np.float32)

    def peak_normalize(self) -> None:
        """``a / max(|a| + tiny(a))`` (event.py:535-536): the peak is reduced into a DEVICE scalar and applied from it
        (al_peak_scale + al_scale_rows), nothing crosses PCIe."""
        r = self.r
        scale = r.mem.empty(1)
        r.lib.call("al_peak_scale", r.mem.ptr(self.buf), self.n, ct.c_float(1.0), r.mem.ptr(scale), r.mem.stream())
        r.lib.call("al_scale_rows", r.mem.ptr(self.buf), self.n, r.mem.ptr(scale), r.mem.stream())


def stage_clips(renderer, raws) -> list:
    """Raw clips of one scene -> HBM through ONE page-locked arena and ONE asynchronous DMA (instead of a blocking pageable
    copy per event); returns one DeviceClip per clip, each a 4-float aligned view of the arena."""
    mem = renderer.mem
    if not raws:
        return []
    if not hasattr(mem, "upload_staged"):
        return [DeviceClip(renderer, raw) for raw in raws]
    offs, total = [], 0
    for raw in raws:
        offs.append(total)
        total += (len(raw) + 3) // 4 * 4

    def fill(view):
        for off, raw in zip(offs, raws):
            view[off: off + len(raw)] = raw

    arena = mem.upload_staged(total, fill)
    return [DeviceClip(renderer, device=arena[off: off + (len(raw) + 3) // 4 * 4], n=len(raw)) for off, raw in zip(offs, raws)]


def _fx(clip: DeviceClip, op: int, p0: float = 0.0, iparams=None, out_of_place: bool = False) -> None:
    r = clip.r
    pa = (ct.c_float * 1)(p0)  # host scalars: the C ABI reads them while building the launch
    ip = (ct.c_int32 * 4)(*iparams) if iparams is not None else None
    dst = clip.other() if out_of_place else clip.buf
    r.lib.call("al_fx_apply", op, r.mem.ptr(clip.buf), r.mem.ptr(dst), clip.n, ct.cast(pa, ct.c_void_p),
               ct.cast(ip, ct.c_void_p) if ip is not None else None, r.mem.stream())
    if out_of_place:
        clip.swap()


def peak_normalize(audio: np.ndarray) -> np.ndarray:
    """``a / max(|a| + tiny(a))`` (event.py:535-536) with the peak reduced and the scale applied on the GPU."""
    clip = DeviceClip(_renderer(), audio)
    clip.peak_normalize()
    return clip.host()


def run_chain(clip: DeviceClip, augmentations, normalize: bool = True) -> DeviceClip:
    """``for aug: a = aug(a)`` then the peak normalisation (event.py:529-536), entirely on one device-resident clip.
    Every augmentation keeps its own pad/truncate(wrap)-to-input-length contract (augmentation.py:117-123)."""
    for aug in augmentations:
        aug.process_device(clip)
    if normalize:
        clip.peak_normalize()
    return clip


def fold_scalars(augmentations) -> Optional[float]:
    """Product of a chain of PURE scalar FX (Gain, Invert), or None when the chain has anything else.  Such a chain
    followed by the peak normalisation is one scalar on the raw clip, ``s / (|s| max|raw| + tiny)``, which the renderer
    evaluates on the device (al_clip_scales) and folds into the clip spectra: the FX kernels are not even launched."""
    s = 1.0
    for aug in augmentations:
        k = getattr(aug, "scalar", None)
        if k is None:
            return None
        s *= float(np.float32(k))
    return s


class Augmentation:
    """Base class: callable object with ``params`` and the pad/truncate(wrap)-to-input-length contract
    of the reference's ``Augmentation.process`` (augmentation.py:91-130)."""

    def __init__(self, sample_rate: Optional[int] = config.SAMPLE_RATE):
        self.sample_rate = _positive(sample_rate, int)
        self.params: dict = dict()

    # subclasses implement this on a DeviceClip (may change clip.n)
    def apply_device(self, clip: DeviceClip) -> None:
        return None

    def process_device(self, clip: DeviceClip) -> None:
        """The FX plus the wrap-pad / truncate back to the input length, on a clip that stays in HBM."""
        n_in = clip.n
        self.apply_device(clip)
        if clip.n != n_in:
            r = clip.r
            dst = clip.other(n_in)
            r.lib.call("al_wrap_copy", r.mem.ptr(clip.buf), clip.n, r.mem.ptr(dst), n_in, r.mem.stream())
            clip.swap(n_in)

    def host_dtype(self, in_dtype: np.dtype) -> np.dtype:
        """The dtype the reference's numpy code leaves a host array of ``in_dtype`` in (most FX keep it; Fade multiplies by a
        float64 envelope, TimeWarpSilence splices float64 zeros in: augmentation.py:1554,1719)."""
        return in_dtype

    def process(self, input_array: np.ndarray) -> np.ndarray:
        arr = np.asarray(input_array)
        if arr.ndim == 2:
            return np.stack([self.process(row) for row in arr])
        clip = DeviceClip(_renderer(), arr)
        self.process_device(clip)
        return clip.host().astype(self.host_dtype(arr.dtype if np.issubdtype(arr.dtype, np.floating) else np.dtype(np.float32)))

    def __call__(self, input_array: np.ndarray) -> np.ndarray:
        return self.process(input_array)

    @property
    def name(self) -> str:
        return type(self).__name__

    def to_dict(self) -> dict[str, Any]:
        return dict(name=self.name, sample_rate=self.sample_rate, **self.params)

    @classmethod
    def from_dict(cls, input_dict: dict[str, Any]):
        d = dict(input_dict)
        name = d.pop("name", cls.__name__)
        target = globals().get(name)
        if target is None or not isinstance(target, type) or not issubclass(target, Augmentation):
            raise ValueError(f"Augmentation class {name} not found")
        return target(**d)

    def __eq__(self, other) -> bool:
        return isinstance(other, Augmentation) and self.to_dict() == other.to_dict()

    def __repr__(self) -> str:
        return f"{self.name}({self.params})"


class EventAugmentation(Augmentation):
    AUGMENTATION_TYPE = "event"


class Gain(EventAugmentation):
    MIN_GAIN, MAX_GAIN = -10, 10

    def __init__(self, sample_rate=config.SAMPLE_RATE, gain_db=None):
        super().__init__(sample_rate)
        self.gain_db = _sample(gain_db, self.MIN_GAIN, self.MAX_GAIN)
        self.params = dict(gain_db=self.gain_db)

    @property
    def linear(self) -> float:
        return float(10.0 ** (self.gain_db / 20.0))

    @property
    def scalar(self) -> float:   # pure scalar FX: foldable (fold_scalars)
        return self.linear

    def apply_device(self, clip):
        _fx(clip, _hip.FX_GAIN, self.linear)


class Invert(EventAugmentation):
    scalar = -1.0

    def apply_device(self, clip):
        _fx(clip, _hip.FX_INVERT)


class Reverse(EventAugmentation):
    def apply_device(self, clip):
        _fx(clip, _hip.FX_REVERSE, out_of_place=True)


class Clipping(EventAugmentation):
    MIN_THRESHOLD_DB, MAX_THRESHOLD_DB = -10, -1

    def __init__(self, sample_rate=config.SAMPLE_RATE, threshold_db=None):
        super().__init__(sample_rate)
        self.threshold_db = -abs(int(_sample(threshold_db, self.MIN_THRESHOLD_DB, self.MAX_THRESHOLD_DB)))
        self.params = dict(threshold_db=self.threshold_db)

    def apply_device(self, clip):
        _fx(clip, _hip.FX_CLIP, float(10.0 ** (self.threshold_db / 20.0)))


class Distortion(EventAugmentation):
    MIN_DRIVE, MAX_DRIVE = 10, 30

    def __init__(self, sample_rate=config.SAMPLE_RATE, drive_db=None):
        super().__init__(sample_rate)
        self.drive_db = _positive(_sample(drive_db, self.MIN_DRIVE, self.MAX_DRIVE))
        self.params = dict(drive_db=self.drive_db)

    def apply_device(self, clip):
        _fx(clip, _hip.FX_TANH, float(10.0 ** (self.drive_db / 20.0)))


class Bitcrush(EventAugmentation):
    MIN_DEPTH, MAX_DEPTH = 8, 32

    def __init__(self, sample_rate=config.SAMPLE_RATE, bit_depth=None):
        super().__init__(sample_rate)
        self.bit_depth = _positive(_sample(bit_depth, self.MIN_DEPTH, self.MAX_DEPTH))
        self.params = dict(bit_depth=self.bit_depth)

    def apply_device(self, clip):
        _fx(clip, _hip.FX_BITCRUSH, float(2.0 ** self.bit_depth))


class Preemphasis(EventAugmentation):
    MIN_COEF, MAX_COEF = 0.0, 1.0
    _OP = _hip.FX_PREEMPH

    def __init__(self, sample_rate=config.SAMPLE_RATE, coef=None):
        super().__init__(sample_rate)
        self.coef = _positive(_sample(coef, self.MIN_COEF, self.MAX_COEF))
        self.params = dict(coef=self.coef)

    def apply_device(self, clip):
        _fx(clip, self._OP, float(self.coef), out_of_place=True)


class Deemphasis(Preemphasis):
    _OP = _hip.FX_DEEMPH


class Fade(EventAugmentation):
    MIN_FADE, MAX_FADE = 0.0, 1.0
    FADE_SHAPES = ["linear", "exponential", "logarithmic", "quarter_sine", "half_sine", "none"]

    def __init__(self, sample_rate=config.SAMPLE_RATE, fade_in_len=None, fade_out_len=None, fade_in_shape=None,
                 fade_out_shape=None):
        super().__init__(sample_rate)
        self.fade_in_len = _positive(_sample(fade_in_len, self.MIN_FADE, self.MAX_FADE))
        self.fade_out_len = _positive(_sample(fade_out_len, self.MIN_FADE, self.MAX_FADE))
        self.fade_in_shape = self._shape(fade_in_shape)
        self.fade_out_shape = self._shape(fade_out_shape)
        self.params = dict(fade_in_len=self.fade_in_len, fade_out_len=self.fade_out_len,
                           fade_in_shape=self.fade_in_shape, fade_out_shape=self.fade_out_shape)

    def _shape(self, given):
        given = str(np.random.choice(self.FADE_SHAPES)) if given is None else given
        if given not in self.FADE_SHAPES:
            raise ValueError(f"Expected `shape` to be one of {', '.join(self.FADE_SHAPES)} but got {given}")
        return given

    def host_dtype(self, in_dtype):
        return np.result_type(in_dtype, np.float64)

    def apply_device(self, clip):
        n_in = min(int(round(self.fade_in_len * self.sample_rate)), clip.n)
        n_out = min(int(round(self.fade_out_len * self.sample_rate)), clip.n)
        _fx(clip, _hip.FX_FADE, iparams=[n_in, n_out, _hip.FADE_SHAPES[self.fade_in_shape],
                                         _hip.FADE_SHAPES[self.fade_out_shape]])


class TimeWarp(EventAugmentation):
    """Frame-shuffle family (augmentation.py:1604-1790).  The per-row coin flips use Python's ``random()``
    exactly like the reference; the shuffle itself is one gather on the GPU."""

    MIN_PROB, MAX_PROB = 0.05, 0.15
    MIN_FPS, MAX_FPS = 2, 10.0
    MODE = None  # None: identity, 1 silence, 2 reverse, "dup", "rm"

    def __init__(self, sample_rate=config.SAMPLE_RATE, fps=None, prob=None):
        super().__init__(sample_rate)
        self.fps = _positive(_sample(fps, self.MIN_FPS, self.MAX_FPS))
        if self.fps == 0.0:
            raise ValueError(f"Expected fps to be greater than 0 but got {fps}")
        self.prob = _positive(_sample(prob, self.MIN_PROB, self.MAX_PROB))
        self.params = dict(fps=self.fps, prob=self.prob)

    def row_plan(self, n: int):
        """(frame_len, row_len, [(src_row, mode)]): the reference frames with librosa.util.frame, whose
        (frame_len, n_frames) result it iterates BY ROWS (augmentation.py:1684-1692)."""
        fl = round(self.sample_rate / self.fps)
        if fl > n:
            fl_eff, row_len, n_src = n, n, 1      # a single "frame": the whole clip, contiguous
            stride = 1
        else:
            row_len, n_src, stride = 1 + (n - fl) // fl, fl, fl
            fl_eff = fl
        rows = []
        for r in range(n_src):
            hit = _random.random() < self.prob
            if self.MODE == "dup":
                rows.extend([(r, 0)] * (2 if hit else 1))
            elif self.MODE == "rm":
                if not hit:
                    rows.append((r, 0))
            else:
                rows.append((r, self.MODE if (hit and self.MODE) else 0))
        self._spliced_zeros = self.MODE == 1 and any(mode == 1 for _, mode in rows)
        return (1 if fl > n else stride), row_len, rows

    def host_dtype(self, in_dtype):
        # TimeWarpSilence replaces a hit frame by np.zeros(len(frame)) -- float64 -- and np.concatenate widens the rest to it
        return np.result_type(in_dtype, np.float64) if getattr(self, "_spliced_zeros", False) else in_dtype

    def apply_device(self, clip):
        self._spliced_zeros = False
        if self.prob == 0:
            return
        stride, row_len, rows = self.row_plan(clip.n)
        if not rows:
            return  # every row removed: the reference falls back to the input (augmentation.py:1698-1701)
        r = clip.r
        table = r.mem.upload(np.array(rows, dtype=np.int32).reshape(-1))
        n_out = len(rows) * row_len
        dst = clip.other(max(n_out, clip.n))
        r.lib.call("al_fx_frame_shuffle", r.mem.ptr(clip.buf), r.mem.ptr(dst), n_out, stride, row_len,
                   r.mem.ptr(table), len(rows), r.mem.stream())
        clip.swap(n_out)   # `table` may go: the allocator orders its reuse behind the launch (same stream)


class TimeWarpSilence(TimeWarp):
    MODE = 1


class TimeWarpReverse(TimeWarp):
    MODE = 2


class TimeWarpDuplicate(TimeWarp):
    MODE = "dup"


class TimeWarpRemove(TimeWarp):
    MODE = "rm"


ALL_EVENT_AUGMENTATIONS = [Gain, Invert, Reverse, Fade, Clipping, Distortion, Bitcrush, Preemphasis, Deemphasis,
                           TimeWarpSilence, TimeWarpDuplicate, TimeWarpRemove, TimeWarpReverse]
