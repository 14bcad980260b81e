"""IR ingest: the step in front of the synthesis path, on the device.

* ``pack_ragged_irs``: what ``WorldStateRLR.get_irs()`` does with a four-deep Python loop executed twice
  (worldstate.py:2183-2255): the ray tracer's per-(capsule, source) IRs of different lengths become one zero-padded
  (C, N, maxlen) tensor -- here directly in the float32 HBM layout the kernels read, never as a float64 host tensor.
* ``resample_irs``: SOFA IRs stored at another sample rate (worldstate.py:2995-3006, ``librosa.resample``) resampled by
  a polyphase FIR on the device.  librosa's default resampler (soxr_hq) is an un-vendored third-party algorithm, so
  parity with it is UNPINNED by definition; this implementation is pinned to
  ``scipy.signal.resample_poly(x, up, down)`` (default Kaiser-5 design), whose taps it uses.

* ``resample_clip`` / ``read_wav_excerpt``: the source side (``librosa.load(path, sr, mono=True,
  offset, duration)``, event.py:520-527): the excerpt is read from a WAV file on the host (decoding is host work in the
  reference too), down-mixed, and resampled to the scene rate ON THE DEVICE as the first step of the event's FX chain.

Host code here is index arithmetic and filter DESIGN (a few thousand taps); samples are touched by kernels only.
"""
from __future__ import annotations

import ctypes as ct
from fractions import Fraction
from typing import Sequence, Tuple

import numpy as np


def pack_ragged_irs(renderer, irs: Sequence[Sequence[np.ndarray]]):
    """``irs[c][n]`` = 1-D IR of capsule c and source n (any lengths, float32 or float64; an empty array is a silent
    path).  Returns ``(device buffer, (stride_c, stride_n), maxlen)`` ready for ``Renderer.prepare(plan, clips, buf,
    strides)``: rows zero-padded to the longest IR (rounded up to 4 floats), like the reference's ``zero_arr``."""
    c = len(irs)
    n = len(irs[0]) if c else 0
    if c == 0 or n == 0 or any(len(row) != n for row in irs):
        raise ValueError("irs must be a non-empty (capsules x sources) nest of 1-D arrays")
    flat_list = [np.asarray(irs[ci][ni]).reshape(-1) for ci in range(c) for ni in range(n)]
    lens = np.array([len(a) for a in flat_list], dtype=np.int32)
    maxlen = int(lens.max())
    if maxlen == 0:
        raise ValueError("all impulse responses are empty")
    f64 = any(a.dtype == np.float64 for a in flat_list)
    flat = np.concatenate([a.astype(np.float64 if f64 else np.float32, copy=False) for a in flat_list])
    offsets = np.zeros(len(flat_list), dtype=np.int64)
    offsets[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    mem, lib = renderer.mem, renderer.lib
    pitch = (maxlen + 3) // 4 * 4
    src, off_d, len_d = mem.upload(flat), mem.upload(offsets), mem.upload(lens)
    dst = mem.empty(c * n * pitch)
    lib.call("al_pack_ragged_irs", mem.ptr(src), 1 if f64 else 0, mem.ptr(off_d), mem.ptr(len_d), c * n, pitch, mem.ptr(dst),
             mem.stream())
    return dst, (n * pitch, pitch), maxlen


def resample_taps(up: int, down: int) -> Tuple[np.ndarray, int]:
    """The FIR scipy.signal.resample_poly designs by default: Kaiser(5.0) windowed sinc, cut-off 1/max(up, down),
    half length 10*max(up, down), gain ``up``."""
    from scipy.signal import firwin

    max_rate = max(up, down)
    half = 10 * max_rate
    h = firwin(2 * half + 1, 1.0 / max_rate, window=("kaiser", 5.0)) * up
    return h.astype(np.float32), half


def resample_irs(renderer, irs: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    """(..., L) IRs at ``orig_sr`` -> (..., round(L * target_sr / orig_sr)) at ``target_sr`` (the length the reference
    allocates, worldstate.py:2985), float32, resampled on the device."""
    irs = np.asarray(irs)
    if int(orig_sr) == int(target_sr):
        return irs.astype(np.float32)
    ratio = Fraction(int(target_sr), int(orig_sr))
    up, down = ratio.numerator, ratio.denominator
    lead, n_in = irs.shape[:-1], irs.shape[-1]
    rows = int(np.prod(lead)) if lead else 1
    n_poly = -(-n_in * up // down)                       # resample_poly's own output length
    n_out = int(round(n_in * (target_sr / orig_sr)))     # what the reference's buffer holds
    taps, half = resample_taps(up, down)
    mem, lib = renderer.mem, renderer.lib
    x = mem.upload(np.ascontiguousarray(irs, dtype=np.float32).reshape(-1))
    h = mem.upload(taps)
    out = mem.empty(rows * n_out)
    for r0 in range(0, rows, 65535):
        nr = min(65535, rows - r0)
        lib.call("al_resample_poly", mem.ptr(x) + 4 * r0 * n_in, nr, n_in, mem.ptr(h), half, up, down,
                 mem.ptr(out) + 4 * r0 * n_out, min(n_poly, n_out), n_out, mem.stream())
    return mem.download(out)[: rows * n_out].reshape(lead + (n_out,))


def resampled_length(n_in: int, orig_sr: int, target_sr: int) -> int:
    """``ceil(n * target_sr / orig_sr)``: the length librosa.resample(fix=True) returns (what librosa.load hands the
    reference, event.py:520-527)."""
    return int(np.ceil(n_in * (float(target_sr) / float(orig_sr))))


def resample_clip(clip, orig_sr: int, target_sr: int) -> None:
    """Resample an ``augmentation.DeviceClip`` in place (ping-pong buffer) from ``orig_sr`` to ``target_sr``: polyphase
    FIR pinned to scipy.signal.resample_poly, output length ``ceil(n * ratio)`` (samples past resample_poly's own length
    are zero, as librosa's fix_length pads)."""
    if int(orig_sr) == int(target_sr):
        return
    ratio = Fraction(int(target_sr), int(orig_sr))
    up, down = ratio.numerator, ratio.denominator
    n_in = clip.n
    n_poly = -(-n_in * up // down)
    n_out = resampled_length(n_in, orig_sr, target_sr)
    taps, half = resample_taps(up, down)
    mem, lib = clip.r.mem, clip.r.lib
    h = mem.upload(taps)
    dst = clip.other(n_out)
    lib.call("al_resample_poly", mem.ptr(clip.buf), 1, n_in, mem.ptr(h), half, up, down, mem.ptr(dst), min(n_poly, n_out), n_out,
             mem.stream())
    clip.swap(n_out)


def read_wav_excerpt(path: str, offset: float = 0.0, duration: float = None) -> Tuple[np.ndarray, int]:
    """``[offset, offset + duration)`` seconds of a RIFF/WAVE file as mono float32 at the FILE's sample rate (PCM 8/16/24/32
    or float; other containers are the host decoder's business and raise).  Integer PCM is scaled like libsndfile
    (``x / 2**(bits-1)``), channels are averaged like ``librosa.to_mono``; frames are cut the way ``librosa.load`` cuts
    them (``int(offset * sr)`` frames in, ``int(duration * sr)`` frames long)."""
    from scipy.io import wavfile

    sr, data = wavfile.read(path, mmap=True)
    start = int(offset * sr)
    stop = data.shape[0] if duration is None else min(data.shape[0], start + int(duration * sr))
    part = np.asarray(data[start:stop])
    if part.dtype == np.uint8:
        part = (part.astype(np.float32) - 128.0) / 128.0
    elif part.dtype.kind == "i":
        part = part.astype(np.float32) / float(2 ** (8 * part.dtype.itemsize - 1))   # 24-bit arrives left-justified in int32
    elif part.dtype.kind == "f":
        part = part.astype(np.float32)
    else:
        raise ValueError(f"unsupported WAV sample type {part.dtype}")
    if part.ndim == 2:
        part = part.mean(axis=1, dtype=np.float32)
    return np.ascontiguousarray(part, dtype=np.float32), int(sr)
