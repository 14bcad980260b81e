"""Background ambience on the GPU with the reference's ``Ambience`` / ``powerlaw_psd_gaussian`` API
(audiblelight/ambience.py).

Two sources for the random draws, chosen per Ambience (``rng=``) or by ``AL_AMBIENCE_RNG``:

* ``"host"`` (default): numpy on the host, exactly where the reference takes them (default_rng(seed) for coloured
  noise, the global RNG for "gaussian").  Its PCG64 + ziggurat stream is sequential and data-dependent, so for a fixed
  seed the noise equals the reference's up to float32 rounding -- the mode the golden vectors pin.  One host thread:
  2 x 32 x 1.44 M draws for a cfg2-size ambience take about a second.
* ``"device"``: Philox-4x32-10 counters + Box-Muller in HIP kernels (csrc/al_rng.h).  Nothing is drawn, cast, uploaded or
  read back by the host; a seed still fixes the realisation (a different one than numpy's), and the reference's own
  acceptance tests of this function -- variance, spectral slope, cumulative scaling, fixed-seed reproducibility, unit
  per-channel peak (tests/test_ambience.py:30-76,107-138) -- are the parity bar (tests/test_gpu_ambience_device.py).
  White noise (beta = 0) is drawn directly in time: with a flat spectrum the Timmer-Koenig synthesis is iid Gaussian.

Everything after the draws runs in HIP kernels in both modes: spectral shaping, the DC / Nyquist fix-ups, an
arbitrary-length inverse real FFT, the 1/sigma scale, per-channel peak normalisation and the tiling of file-based ambience.
"""
from __future__ import annotations

import ctypes as ct
import functools
import os
import random
from pathlib import Path
from typing import Any, Iterable, Optional, Union

import numpy as np

from . import config
from .utils import tiny, valid_audio

# popular names -> beta (higher beta = more low-frequency energy), reference ambience.py:23
NOISE_MAPPING = dict(pink=1, brown=2, red=2, blue=-1, white=0, violet=-2)


def _renderer():
    from . import synthesize

    return synthesize.get_renderer()


def _parse_beta(noise: Any) -> Union[float, str]:
    """Colour name / "gaussian" / numeric exponent -> beta (reference ambience.py:378-400)."""
    if isinstance(noise, str):
        if noise in NOISE_MAPPING:
            return NOISE_MAPPING[noise]
        if noise.lower() == "gaussian":
            return "gaussian"
        raise KeyError(f"Expected a string in {', '.join(NOISE_MAPPING)} but got {noise}.")
    if isinstance(noise, (int, float, np.integer, np.floating)) and not isinstance(noise, bool):
        return noise
    raise TypeError(f"Expected either a string or numeric input, but got {type(noise)}.")


def _check_fmin(fmin) -> None:
    if not isinstance(fmin, (int, float, np.integer, np.floating)) or fmin < 0:
        raise ValueError(f"Expected a positive numeric input, but got {fmin}")
    if not 0 <= fmin <= 0.5:
        raise ValueError(f"Argument `fmin` must be chosen between 0 and 0.5 but got {fmin:.2f}.")


def _spectral_shape(beta: float, samples: int, fmin: float):
    """Scaling vector s[f] and theoretical output std sigma (reference ambience.py:319-340). Host side:
    one pass over samples/2+1 frequencies, no audio involved."""
    _check_fmin(fmin)
    return _spectral_shape_cached(float(beta), int(samples), float(fmin))


@functools.lru_cache(maxsize=4)      # the scenes of a dataset share (beta, length): one pass over 1.4 M bins per dataset, not per scene
def _spectral_shape_cached(beta: float, samples: int, fmin: float):
    f = np.fft.rfftfreq(samples)
    fmin = max(fmin, 1.0 / (samples + tiny(samples)))
    cut = int(np.sum(f < fmin))
    if cut and cut < len(f):
        f[:cut] = f[cut]
    s = f ** (-beta / 2.0)
    w = s[1:].copy()
    w[-1] *= (1 + (samples % 2)) / 2.0
    sigma = 2 * np.sqrt(np.sum(w ** 2)) / (samples + tiny(samples))
    s = s.astype(np.float32)
    s.setflags(write=False)
    return s, sigma


def _flat_sigma(samples: int) -> float:
    """sigma of ``_spectral_shape`` for beta = 0 (s = 1 everywhere) without the vector."""
    bins = samples // 2 + 1
    w2 = (bins - 2) + ((1 + samples % 2) / 2.0) ** 2 if bins >= 2 else 0.0
    return 2 * np.sqrt(w2) / (samples + tiny(samples))


def default_rng_mode() -> str:
    """AL_AMBIENCE_RNG ("host" / "device"), parsed once with the other switches (switches.py)."""
    from . import switches

    return switches.current().ambience_rng


def _clone(buf):
    """A copy of a device buffer in the memory provider's own type (torch tensor or, under host emulation, ndarray)."""
    return buf.clone() if hasattr(buf, "clone") else np.array(buf)


def _seed64(seed) -> int:
    """Key of the device generator.  ``seed=None`` means fresh entropy, as ``np.random.default_rng(None)`` does for the host draws."""
    if seed is None:
        return int.from_bytes(os.urandom(8), "little")
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def powerlaw_noise_device(renderer, beta: float, rows: int, samples: int, fmin: float = 0.0,
                          seed: Optional[int] = config.SEED, rng: str = "host"):
    """Device buffer (rows*samples float32) of Gaussian (1/f)^beta noise.  ``rng="device"``: stream-ordered, no host
    synchronisation (the draws are made by the kernels)."""
    r = renderer
    out = r.mem.empty(rows * samples)
    if rng == "device":
        _check_fmin(fmin)
        if float(beta) == 0.0:
            # flat spectrum: irfft of iid complex Gaussians (|S[k]|^2 = 2 at every bin incl. the sqrt(2)-scaled real DC /
            # Nyquist) is iid N(0, 2/n) in time, so the samples are drawn where they are needed
            scale = float(np.sqrt(2.0 / samples) / _flat_sigma(samples))
            r.lib.call("al_normal_fill", r.mem.ptr(out), rows * samples, ct.c_uint64(_seed64(seed)), 2, scale, r.mem.stream())
            return out
        s, sigma = _spectral_shape(beta, samples, fmin)
        d_s = r.mem.upload(s)
        work = r.mem.empty(r.lib.call("al_noise_workspace_floats", rows, samples))
        r.lib.call("al_noise_irfft_seeded", ct.c_uint64(_seed64(seed)), r.mem.ptr(d_s), rows, samples, float(1.0 / sigma),
                   r.mem.ptr(out), r.mem.ptr(work), r.mem.stream())
        if not hasattr(r.mem, "torch"):
            r.mem.synchronize()
        return out
    s, sigma = _spectral_shape(beta, samples, fmin)
    gen = np.random.default_rng(seed)
    bins = samples // 2 + 1
    zr = gen.normal(size=(rows, bins)).astype(np.float32)   # same draw order as ambience.py:355-356
    zi = gen.normal(size=(rows, bins)).astype(np.float32)
    d_zr, d_zi, d_s = r.mem.upload(zr.reshape(-1)), r.mem.upload(zi.reshape(-1)), r.mem.upload(s)
    work = r.mem.empty(r.lib.call("al_noise_workspace_floats", rows, samples))
    r.lib.call("al_noise_irfft", r.mem.ptr(d_zr), r.mem.ptr(d_zi), r.mem.ptr(d_s), rows, samples,
               float(1.0 / sigma), r.mem.ptr(out), r.mem.ptr(work), r.mem.stream())
    r.mem.synchronize()
    return out


def powerlaw_psd_gaussian(beta, shape: Union[int, Iterable[int]], fmin: Optional[float] = 0.0,
                          seed: Optional[int] = config.SEED, rng: Optional[str] = None) -> np.ndarray:
    """Gaussian (1/f)**beta noise, last axis = time (reference ambience.py:271-375; Timmer & Koenig 1995).
    ``rng``: "host" (numpy draws, the reference's realisation) or "device" (module docstring); default AL_AMBIENCE_RNG."""
    if isinstance(shape, (np.integer, int)):
        size = [int(shape)]
    elif isinstance(shape, Iterable):
        size = [int(v) for v in shape]
    else:
        raise ValueError(f"Argument `shape` must be of type int or Iterable[int] but got {type(shape)}")
    samples = size[-1]
    rows = int(np.prod(size[:-1])) if len(size) > 1 else 1
    r = _renderer()
    dev = powerlaw_noise_device(r, beta, rows, samples, fmin, seed, rng or default_rng_mode())
    return r.mem.download(dev)[: rows * samples].reshape(size).astype(np.float64)


def peak_normalize_rows_device(renderer, dev, rows: int, cols: int) -> None:
    """Per-channel ``ch / max(|ch| + tiny)`` in place (reference ambience.py:211-214)."""
    r = renderer
    stats = r.mem.download(r.row_stats(dev, rows, cols)).reshape(-1, 4)[:rows]
    # a silent channel: 1 / tiny does not fit float32 and inf * 0 is NaN where the reference keeps zeros; saturated, FLT_MAX * 0 = 0
    scale = np.minimum(1.0 / (stats[:, 1] + tiny(np.float64(0))), np.finfo(np.float32).max).astype(np.float32)
    r.lib.call("al_scale_matrix_rows", r.mem.ptr(dev), rows, cols, r.mem.ptr(r.mem.upload(scale)), r.mem.stream())
    r.mem.synchronize()


class Ambience:
    """Persistent background noise of a Scene (reference ambience.py:26-267): coloured noise or a tiled clip."""

    def __init__(self, channels: int, duration: float, alias: str, filepath: Optional[Union[str, Path]] = None,
                 noise: Optional[Union[str, float]] = None, ref_db: Optional[float] = config.DEFAULT_REF_DB,
                 sample_rate: Optional[int] = config.SAMPLE_RATE, clip: Optional[np.ndarray] = None,
                 rng: Optional[str] = None, device_seed: Optional[int] = None, **kwargs):
        for name, val in (("channels", channels), ("sample_rate", sample_rate), ("duration", duration)):
            if not isinstance(val, (int, float, np.integer, np.floating)) or isinstance(val, bool):
                raise TypeError(f"Expected a numeric input for {name}, but got {type(val)}")
            if val < 0:
                raise ValueError(f"Expected a positive numeric input for {name}, but got {val}")
        self.channels, self.sample_rate, self.duration, self.alias = int(channels), int(sample_rate), float(duration), alias
        self.clip = None if clip is None else np.atleast_2d(np.asarray(clip, dtype=np.float32))
        have_file = filepath is not None or clip is not None
        if noise is None and have_file:
            self.filepath, self.beta = (Path(filepath) if filepath is not None else None), None
        elif noise is not None and not have_file:
            self.filepath, self.beta = None, _parse_beta(noise)
        elif noise is not None and have_file:
            raise AttributeError("Only one of `noise` or `filepath` should be provided.")
        else:
            raise AttributeError("One of `noise` or `filepath` must be provided")
        bad = set(kwargs) - {"fmin", "seed"}
        if bad:
            raise AttributeError(f"`{sorted(bad)[0]}` is not a valid keyword argument for `powerlaw_psd_gaussian`")
        self.noise_kwargs = kwargs
        if rng not in (None, "host", "device"):
            raise ValueError(f"`rng` must be 'host' or 'device', got {rng!r}")
        self.rng = rng or default_rng_mode()   # where the normal draws are made (module docstring)
        if not isinstance(ref_db, (int, float, np.integer, np.floating)) or -ref_db < 0:
            raise ValueError(f"Expected a negative noise floor in dB, but got {ref_db}")
        self.ref_db = ref_db
        self.audio: Optional[np.ndarray] = None
        self._device = None
        self._scaled = None
        self.device_shape = None
        # device-drawn "gaussian" noise without a seed= keyword: the key of this object's realisation is drawn HERE, once, so
        # that to_dict() is the same before and after the first draw (``device_seed``: a key of its own that __eq__ ignores --
        # two Ambiences built from the same arguments stay equal, as in the reference, and each still reproduces its own noise)
        self._entropy = None
        if self.rng == "device" and self.beta == "gaussian" and self.noise_kwargs.get("seed") is None:
            self._entropy = _seed64(device_seed)

    @property
    def is_audio_loaded(self) -> bool:
        return self.audio is not None and valid_audio(self.audio)

    def _decode(self) -> np.ndarray:
        """(file channels, samples) float32 at ``self.sample_rate`` (``librosa.load(path, sr=..., mono=False)``,
        ambience.py:171-174).  WAV only (decoding other containers is the host decoder's business); integer PCM scaled like
        libsndfile, another sample rate resampled on the device (ingest.resample_irs: polyphase FIR pinned to
        scipy.signal.resample_poly)."""
        if self.clip is not None:
            return self.clip
        from scipy.io import wavfile

        from . import ingest

        sr, data = wavfile.read(self.filepath)
        if data.dtype == np.uint8:
            data = (data.astype(np.float32) - 128.0) / 128.0
        elif np.issubdtype(data.dtype, np.integer):
            data = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
        data = np.atleast_2d(np.asarray(data, dtype=np.float32).T if data.ndim == 2 else np.asarray(data, dtype=np.float32))
        if int(sr) != self.sample_rate:
            n_out = ingest.resampled_length(data.shape[1], sr, self.sample_rate)   # librosa.resample: ceil(n * ratio)
            res = ingest.resample_irs(_renderer(), data, int(sr), self.sample_rate)
            data = np.zeros((data.shape[0], n_out), dtype=np.float32)
            data[:, : min(n_out, res.shape[1])] = res[:, :n_out]
        return data

    def _generate_device(self, r, total: int):
        """(channels * total) float32 in HBM, not yet peak-normalised."""
        if self.beta is not None:
            if self.beta == "gaussian":
                if self.rng == "device":   # ambience.py:160-165 draws from the GLOBAL numpy RNG: every Ambience gets fresh noise.
                    # Mirrored here: no seed kwarg = fresh entropy per draw (a fixed default would put the SAME noise into every
                    # scene of a dataset); only an explicit seed= makes the realisation reproducible.
                    dev = r.mem.empty(self.channels * total)
                    r.lib.call("al_normal_fill", r.mem.ptr(dev), self.channels * total,
                               ct.c_uint64(self._gaussian_seed()), 3, 1.0, r.mem.stream())
                    return dev
                return r.mem.upload(np.random.normal(0, 1, (self.channels, total)).astype(np.float32).reshape(-1))
            return powerlaw_noise_device(r, self.beta, self.channels, total, rng=self.rng, **self.noise_kwargs)
        src = self._decode()
        if src.shape[0] != self.channels:
            row = 0 if src.shape[0] == 1 else random.choice(range(src.shape[0]))  # Python's global RNG, ambience.py:183-192
            src = np.repeat(src[row: row + 1], self.channels, axis=0)
        dev = r.mem.empty(self.channels * total)
        for c in range(self.channels):  # np.tile along time (ambience.py:204-208)
            one = r.mem.upload(np.ascontiguousarray(src[c]))
            r.lib.call("al_wrap_copy", r.mem.ptr(one), src.shape[1], r.mem.ptr(dev) + 4 * c * total, total, r.mem.stream())
        r.mem.synchronize()
        return dev

    def _gaussian_seed(self) -> int:
        """Key of this object's device-drawn "gaussian" noise: the seed= keyword, else entropy drawn ONCE per object (so the
        object reproduces its own realisation, and its dictionary records it), never a fixed default."""
        seed = self.noise_kwargs.get("seed", None)
        if seed is not None:
            return _seed64(seed)
        if getattr(self, "_entropy", None) is None:
            self._entropy = _seed64(None)
        return self._entropy

    def load_ambience_device(self, renderer=None, ignore_cache: bool = False, normalize: bool = True):
        """(channels*samples) float32 device buffer of the ambience; cached."""
        if self._device is not None and not ignore_cache:
            return self._device
        r = renderer or _renderer()
        total = round(self.duration * self.sample_rate)
        # the realisation a scene has already MIXED (noise_and_scales_device) is the one this Ambience hands out afterwards,
        # as the reference caches the array it mixed (ambience.py:142-148): derived from that buffer, not drawn again
        mixed = None if ignore_cache else getattr(self, "_scaled", None)
        dev = _clone(mixed[0]) if mixed is not None else self._generate_device(r, total)
        if normalize and self.rng == "device" and self.channels <= 1024:
            # per-channel 1 / (peak + tiny) from device statistics, applied on the device: no host round trip
            scales = r.mem.empty(self.channels)
            r.lib.call("al_ambience_scales", r.mem.ptr(r.row_stats(dev, self.channels, total)), self.channels, total, 0.0, 2,
                       r.mem.ptr(scales), r.mem.stream())
            r.lib.call("al_scale_matrix_rows", r.mem.ptr(dev), self.channels, total, r.mem.ptr(scales), r.mem.stream())
        elif normalize:
            peak_normalize_rows_device(r, dev, self.channels, total)
        self._device, self.device_shape = dev, (self.channels, total)
        if mixed is None:
            self._scaled = None
        return dev

    def noise_and_scales_device(self, renderer, shape):
        """What the mixdown needs of a device-drawn ambience, with NO host synchronisation: the un-normalised noise and one
        multiplier per channel = db_to_multiplier(ref_db, mean|normalised|) / (peak_c + tiny) (al_ambience_scales:
        ambience.py:211-214 and synthesize.py:350-356 folded into the scalar the mixdown applies anyway).  None for
        ambiences this does not cover (host draws, file mode, a cached host array): the caller takes the general route."""
        if self.rng != "device" or self.beta is None or self.audio is not None or self.channels > 1024:
            return None
        r = renderer
        total = round(self.duration * self.sample_rate)
        if (self.channels, total) != tuple(shape):
            raise ValueError(f"Scene ambient noise does not match expected shape. Expected {tuple(shape)}, but got {(self.channels, total)}.")
        if getattr(self, "_scaled", None) is not None:
            return self._scaled
        cached = self._device if tuple(getattr(self, "device_shape", ()) or ()) == (self.channels, total) else None
        dev = cached if cached is not None else self._generate_device(r, total)   # a buffer load_ambience_device made is the one mixed
        scales = r.mem.empty(self.channels)
        r.lib.call("al_ambience_scales", r.mem.ptr(r.row_stats(dev, self.channels, total)), self.channels, total,
                   float(self.ref_db), 1, r.mem.ptr(scales), r.mem.stream())   # 1 / (peak + tiny) of an already normalised row is 1
        self._scaled = (dev, scales)
        return self._scaled

    def load_ambience(self, ignore_cache: Optional[bool] = False, normalize: Optional[bool] = True) -> np.ndarray:
        """(channels, samples) array; computed on the GPU, cached in ``self.audio`` (reference ambience.py:142-217)."""
        if self.is_audio_loaded and not ignore_cache:
            return self.audio
        r = _renderer()
        self._device = None
        dev = self.load_ambience_device(r, ignore_cache=bool(ignore_cache), normalize=bool(normalize))
        c, n = self.device_shape
        self.audio = r.mem.download(dev)[: c * n].reshape(c, n).astype(np.float64)
        return self.audio

    def to_dict(self) -> dict:
        d = dict(alias=self.alias, beta=self.beta, filepath=str(self.filepath) if self.filepath is not None else None,
                 channels=self.channels, sample_rate=self.sample_rate, duration=self.duration, ref_db=self.ref_db,
                 noise_kwargs=self.noise_kwargs)
        # "rng" is beyond the reference's layout (from_dict tolerates its absence), so the plain case -- host draws under the
        # default environment -- keeps the reference's exact dictionary; every other case records it: a device Ambience
        # always (the realisation depends on it, whatever AL_AMBIENCE_RNG says when the file is read back), and an explicit
        # rng="host" made under AL_AMBIENCE_RNG=device
        if self.rng != "host" or default_rng_mode() != "host":
            d["rng"] = self.rng
        if self._entropy is not None:
            d["device_seed"] = self._entropy     # the realisation this object draws, reproducible from its dictionary
        return d

    @classmethod
    def from_dict(cls, input_dict: dict[str, Any]):
        for k in ["alias", "filepath", "duration", "ref_db", "beta", "channels"]:
            if k not in input_dict:
                raise KeyError(f"Missing key: '{k}'")
        return cls(channels=input_dict["channels"], sample_rate=input_dict["sample_rate"], alias=input_dict["alias"],
                   filepath=input_dict["filepath"], duration=input_dict["duration"], noise=input_dict["beta"],
                   ref_db=input_dict["ref_db"], rng=input_dict.get("rng"), device_seed=input_dict.get("device_seed"),
                   **input_dict["noise_kwargs"])

    def __eq__(self, other: Any) -> bool:
        if not isinstance(other, Ambience):
            return False
        mine, theirs = self.to_dict(), other.to_dict()
        mine.pop("device_seed", None)
        theirs.pop("device_seed", None)
        return mine == theirs

    def __str__(self) -> str:
        return f"'Ambience' with alias '{self.alias}' (currently {'loaded' if self.audio is not None else 'unloaded'})."
