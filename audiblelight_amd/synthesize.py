"""MI355X implementation of AudibleLight's synthesis functions, same names and contracts.

Mirror of ``audiblelight/synthesize.py`` (reference lines cited per function).  Every function
that touches samples runs on the GPU through the C ABI of include/audiblelight_hip.h; there is
no CPU fallback (a missing extension or GPU raises).  Results are written onto the Scene/Event
objects exactly where the reference writes them; arrays that stay in HBM are exposed through
``utils.LazyAudioDict`` and are copied to the host as ``np.ndarray`` on first access.

Drop-in use with the real AudibleLight package: ``audiblelight_amd.dropin.install()``.
"""
from __future__ import annotations

import logging
from time import time
from typing import Dict, List, Optional

import numpy as np

from . import config, engine
from . import plan as planning
from .plan import generate_interpolation_matrix  # noqa: F401  (same public name as the reference)
from .utils import LazyAudioDict, as_lazy, pad_or_truncate_audio, tiny, valid_audio, validate_shape

logger = logging.getLogger("audiblelight_amd")

_renderer: Optional[engine.Renderer] = None


def get_renderer() -> engine.Renderer:
    """Process-wide renderer on the current device (created on first use; raises without a GPU)."""
    global _renderer
    if _renderer is None:
        _renderer = engine.Renderer()
    return _renderer


def set_renderer(renderer: Optional[engine.Renderer]) -> None:
    global _renderer
    _renderer = renderer


# ----------------------------------------------------------------------------- scalar helpers
def apply_snr(x: np.ndarray, snr) -> np.ndarray:
    """Scale to a maximum SNR: ``x * snr / max(|x|, 1e-15)`` (reference synthesize.py:40-49).

    Host utility kept for API compatibility; the render path fuses it into the level law on device.
    """
    return x * snr / np.abs(x).max(initial=1e-15)


def db_to_multiplier(db, x) -> float:
    """``10^(db/20) / (x + tiny(x))`` (reference synthesize.py:52-68)."""
    return 10 ** (db / 20) / (x + tiny(x))


# ----------------------------------------------------------------------------- convolutions
def time_invariant_convolution(audio: np.ndarray, ir: np.ndarray) -> np.ndarray:
    """Full linear convolution of a mono clip with each IR channel -> (C, La + Lir - 1).

    Reference synthesize.py:71-106 (scipy.signal.fftconvolve along axis 0).  Here: partitioned
    overlap-save FFT convolution on the GPU.
    """
    if audio.ndim != 1:
        raise ValueError(f"Only mono input is supported, but got {audio.ndim} dimensions!")
    if ir.ndim != 2:
        raise ValueError(f"Expected shape of IR should be (n_samples, n_channels), but got ({ir.shape}) instead")
    n_ir, n_ch = ir.shape
    n_out = audio.shape[0] + n_ir - 1
    clip = np.zeros(n_out, dtype=np.float32)
    clip[: audio.shape[0]] = audio
    r = get_renderer()
    pl = planning.plan_batch([planning.EventSpec(n_samples=n_out, n_emitters=1, snr=1.0)], n_ch, n_ir, 1.0, lib=r.lib)
    res = r.render(pl, [clip], np.ascontiguousarray(ir.T)[:, None, :], normalize_irs=False)
    return res.raw_spatial(0).astype(np.float64)


def _envelope_geometry(fft_size: int, win_size: int, hop_size: int) -> bool:
    """True where the envelope form of the STFT-domain algorithm holds: the sin^2 window at 50 % overlap
    is a partition of unity and an fft_size-point frame holds the linear convolution of two win_size-sample frames."""
    return win_size == 2 * hop_size and fft_size >= 2 * win_size - 1


def _check_geometry(fft_size: int, win_size: int, hop_size: int) -> None:
    """What the reference itself refuses (both ValueErrors there too): ``stft`` pads by win - hop on the left (np.pad raises for
    a negative width, synthesize.py:124-127) and ``istft_overlap_synthesis`` adds fft_size-sample frames into a buffer of
    (n_frames + 1) * hop + win samples (the last frame does not fit when fft_size > 2*hop + win: numpy's broadcast error,
    synthesize.py:268-272)."""
    if min(fft_size, win_size, hop_size) <= 0:
        raise ValueError("fft_size, win_size and hop_size must be positive")
    if win_size < hop_size:
        raise ValueError(f"index can't contain negative values: win_size ({win_size}) must be at least hop_size ({hop_size})")
    if fft_size > 2 * hop_size + win_size:
        raise ValueError(f"operands could not be broadcast together: frames of fft_size = {fft_size} samples do not fit the "
                         f"overlap-add buffer, fft_size must be at most 2*hop_size + win_size = {2 * hop_size + win_size}")


def _permuted(buf, shape, axes):
    """A contiguous device copy of ``buf`` viewed as ``shape`` with its axes permuted (a copy, no arithmetic)."""
    n = int(np.prod(shape))
    if hasattr(buf, "permute"):
        return buf[:n].view(*shape).permute(*axes).contiguous().view(-1)
    return np.ascontiguousarray(buf[:n].reshape(shape).transpose(axes)).reshape(-1)


def _compact_rows(buf, rows: int, pitch: int, length: int):
    """rows x pitch -> contiguous rows x length (the first ``length`` elements of every row); a copy, no arithmetic."""
    if hasattr(buf, "permute"):
        return buf[: rows * pitch].view(rows, pitch)[:, :length].contiguous().view(-1)
    return np.ascontiguousarray(buf[: rows * pitch].reshape(rows, pitch)[:, :length]).reshape(-1)


def _tv_chain_on_device(r, ir_dev, n_ch: int, n_irs: int, ir_len: int, audio_dev, n_audio: int, duration: float, sr: float,
                        fft_size: int, win_size: int, hop_size: int, ir_pitch: Optional[int] = None):
    """The reference's literal STFT-domain chain (synthesize.py:298-310) on device buffers, for STFT geometries outside the
    envelope form: stft of the IRs and of the clip (al_stft), the frame-domain convolution with the cross-fade weights
    (al_tv_stft_mac), inverse transforms + overlap-add (al_istft_ola).  ``ir_dev``: (C, N, ir_pitch) float32 rows of which the
    first ``ir_len`` samples are the IR.  The frame count of the IR spectrogram comes from ``ir_len`` exactly as in the reference
    (2 * ceil(L / (2 hop)) + 1): frames past it would not be all-zero -- they would window the IR's tail once more -- so rows
    with a pitch are compacted first (a copy).  Returns (device (n_out, C) float32, n_out); nothing is synchronised or downloaded."""
    mem, lib, st = r.mem, r.lib, r.mem.stream()
    n_freq = fft_size // 2 + 1
    if ir_pitch is not None and ir_pitch != ir_len:
        ir_dev = _compact_rows(ir_dev, n_ch * n_irs, ir_pitch, ir_len)
    f_ir = planning.stft_frame_count(ir_len, hop_size, lib=lib)
    f_a = planning.stft_frame_count(n_audio, hop_size, lib=lib)
    w = generate_interpolation_matrix(np.linspace(0, duration, n_irs), sr, hop_size, lib=lib)
    n_frames = min(f_a, w.shape[0])
    n_out = n_frames * hop_size - win_size
    if n_frames <= 0 or n_out <= 0:
        return mem.zeros(1), 0
    rows = n_ch * n_irs
    h_spec = mem.empty(2 * rows * f_ir * n_freq)
    # ONE workspace for the three transforms (stream-ordered: each is done with it before the next starts)
    work = mem.empty(max(lib.call("al_stft_workspace_floats", rows * f_ir, fft_size), lib.call("al_stft_workspace_floats", f_a, fft_size),
                         lib.call("al_istft_workspace_floats", n_frames, n_ch, fft_size)))
    lib.call("al_stft", mem.ptr(ir_dev), rows, ir_len, fft_size, win_size, hop_size, mem.ptr(h_spec), mem.ptr(work), st)
    s_ir = _permuted(h_spec, (n_ch, n_irs, f_ir, n_freq, 2), (2, 3, 0, 1, 4))      # (F_ir, freq, C, N): the reference's layout
    del h_spec
    a_spec = mem.empty(2 * f_a * n_freq)
    lib.call("al_stft", mem.ptr(audio_dev), 1, n_audio, fft_size, win_size, hop_size, mem.ptr(a_spec), mem.ptr(work), st)
    w_dev = mem.upload(np.ascontiguousarray(w, dtype=np.float32).reshape(-1))
    y = mem.empty(2 * n_frames * n_freq * n_ch)
    lib.call("al_tv_stft_mac", mem.ptr(a_spec), mem.ptr(s_ir), mem.ptr(w_dev), n_frames, f_ir, n_freq, n_ch, n_irs, mem.ptr(y), st)
    out = mem.empty(n_out * n_ch)
    lib.call("al_istft_ola", mem.ptr(y), n_frames, n_freq, n_ch, fft_size, win_size, hop_size, mem.ptr(out), mem.ptr(work), st)
    return out, n_out


def time_variant_convolution(irs: np.ndarray, event, fft_size=config.FFT_SIZE, win_size=config.WIN_SIZE,
                             hop_size=config.HOP_SIZE) -> np.ndarray:
    """Time-variant convolution of a moving event: (C, N, L) IRs -> (C, n_frames*hop - win).

    Reference synthesize.py:277-310 (STFT-domain convolution with linear IR cross-fades).  Computed
    here in the mathematically identical envelope form ("Moving events") where that form
    holds (win == 2*hop, fft >= 2*win - 1: the defaults), else by the literal chain on the device.
    """
    audio = np.asarray(event.load_audio(), dtype=np.float32)
    fft_size, win_size, hop_size = int(fft_size), int(win_size), int(hop_size)
    _check_geometry(fft_size, win_size, hop_size)
    n_ch, n_irs, n_ir = irs.shape
    if not _envelope_geometry(fft_size, win_size, hop_size):
        r = get_renderer()
        ir_dev = r.mem.upload(np.ascontiguousarray(irs, dtype=np.float32).reshape(-1))
        out, n_out = _tv_chain_on_device(r, ir_dev, n_ch, n_irs, n_ir, r.mem.upload(np.ascontiguousarray(audio)), len(audio),
                                         float(event.duration), float(event.sample_rate), fft_size, win_size, hop_size)
        return np.ascontiguousarray(r.mem.download(out)[: n_out * n_ch].reshape(n_out, n_ch).T).astype(np.float64)
    # the reference returns n_frames*hop - win samples, up to `hop` more than the clip: render into a
    # clip extended by win zeros while counting STFT frames on the original length
    clip = np.zeros(len(audio) + win_size, dtype=np.float32)
    clip[: len(audio)] = audio
    spec = planning.EventSpec(n_samples=len(clip), n_emitters=n_irs, snr=1.0, is_moving=True,
                              duration=event.duration, stft_len=len(audio))
    r = get_renderer()
    pl = planning.plan_batch([spec], n_ch, n_ir, event.sample_rate, hop=hop_size, win=win_size, fft_size=fft_size, lib=r.lib)
    res = r.render(pl, [clip], irs, normalize_irs=False)
    return res.raw_spatial(0)[:, : int(pl.events["valid_len"][0])].astype(np.float64)


# ----------------------------------------------------------------------------- STFT-domain helpers (A7 intermediates)
def _complex_in(r, arr: np.ndarray):
    """complex array -> device buffer of interleaved float32 (re, im)."""
    return r.mem.upload(np.ascontiguousarray(arr, dtype=np.complex64).view(np.float32).reshape(-1))


def _complex_out(r, dev, shape, like=np.complex128) -> np.ndarray:
    n = int(np.prod(shape))
    return r.mem.download(dev)[: 2 * n].view(np.complex64).reshape(shape).astype(like)


def stft(y: np.ndarray, fft_size=config.FFT_SIZE, win_size=config.WIN_SIZE, hop_size=config.HOP_SIZE,
         stft_dims_first: Optional[bool] = True) -> np.ndarray:
    """STFT with a sin^2 window, left pad ``win - hop`` (reference synthesize.py:109-145); computed on the GPU
    (al_stft: windowing + batched FFT).  ``y`` is (..., samples); returns (frames, freq, ...) when
    ``stft_dims_first`` else (..., freq, frames), like the reference."""
    y = np.asarray(y)
    fft_size, win_size, hop_size = int(fft_size), int(win_size), int(hop_size)
    lead, n = y.shape[:-1], y.shape[-1]
    rows = int(np.prod(lead)) if lead else 1
    n_frames = 2 * int(np.ceil(n / (2.0 * hop_size))) + 1
    n_freq = fft_size // 2 + 1
    r = get_renderer()
    dev = r.mem.upload(np.ascontiguousarray(y, dtype=np.float32).reshape(-1))
    spec = r.mem.empty(2 * rows * n_frames * n_freq)
    work = r.mem.empty(r.lib.call("al_stft_workspace_floats", rows * n_frames, fft_size))
    r.lib.call("al_stft", r.mem.ptr(dev), rows, n, fft_size, win_size, hop_size, r.mem.ptr(spec), r.mem.ptr(work), r.mem.stream())
    r.mem.synchronize()
    like = np.complex128 if y.dtype == np.float64 else np.complex64
    out = _complex_out(r, spec, lead + (n_frames, n_freq), like)          # (..., frames, freq)
    if stft_dims_first:
        out = np.moveaxis(np.moveaxis(out, -1, 0), -1, 0)                  # (frames, freq, ...): synthesize.py:141-142
    else:
        out = np.swapaxes(out, -1, -2)                                     # (..., freq, frames)
    return np.ascontiguousarray(out)


def perform_time_variant_convolution(s_audio: np.ndarray, s_ir: np.ndarray, w_ir: np.ndarray, ir_slice_min=0,
                                     ir_relevant_ratio_max=0.5) -> np.ndarray:
    """Convolve a bank of cross-faded IR spectrograms with an audio spectrogram (reference synthesize.py:184-252):
    ``out[i,f,c] = sum_k S[i-k,f] sum_l W[i-k,l] H[k,f,c,l]`` on the GPU (al_tv_stft_mac).  ``ir_slice_min`` /
    ``ir_relevant_ratio_max`` only steer a copy optimisation in the reference (zero-weight IRs are dropped, which does
    not change the sum) and are accepted for signature compatibility."""
    n_frames_ir, n_freq, n_ch, n_irs = s_ir.shape
    n_frames = min(s_audio.shape[0], w_ir.shape[0])
    r = get_renderer()
    sa, si = _complex_in(r, s_audio), _complex_in(r, s_ir)
    w = r.mem.upload(np.ascontiguousarray(w_ir, dtype=np.float32).reshape(-1))
    out = r.mem.empty(2 * n_frames * n_freq * n_ch)
    r.lib.call("al_tv_stft_mac", r.mem.ptr(sa), r.mem.ptr(si), r.mem.ptr(w), n_frames, n_frames_ir, n_freq, n_ch, n_irs,
               r.mem.ptr(out), r.mem.stream())
    r.mem.synchronize()
    return _complex_out(r, out, (n_frames, n_freq, n_ch))


def istft_overlap_synthesis(spatial_stft: np.ndarray, fft_size=config.FFT_SIZE, win_size=config.WIN_SIZE,
                            hop_size=config.HOP_SIZE) -> np.ndarray:
    """Inverse FFT of every frame + overlap-add synthesis -> (n_frames*hop - win, channels) (reference
    synthesize.py:255-274), on the GPU (al_istft_ola)."""
    n_frames, n_freq, n_ch = spatial_stft.shape
    fft_size, win_size, hop_size = int(fft_size), int(win_size), int(hop_size)
    r = get_renderer()
    spec = _complex_in(r, spatial_stft)
    n_out = n_frames * hop_size - win_size
    out = r.mem.empty(max(n_out, 1) * n_ch)
    work = r.mem.empty(r.lib.call("al_istft_workspace_floats", n_frames, n_ch, fft_size))
    r.lib.call("al_istft_ola", r.mem.ptr(spec), n_frames, n_freq, n_ch, fft_size, win_size, hop_size, r.mem.ptr(out),
               r.mem.ptr(work), r.mem.stream())
    r.mem.synchronize()
    return r.mem.download(out)[: n_out * n_ch].reshape(n_out, n_ch).astype(np.float64)


def normalize_irs(irs: np.ndarray) -> np.ndarray:
    """Divide by the mean (axis -2) of the L2 norms (axis -1) (reference synthesize.py:404-428).

    Row energies are reduced on the GPU (al_row_stats) and each row is rescaled on the GPU.
    """
    arr = np.asarray(irs)
    if arr.ndim < 2:
        raise ValueError("normalize_irs expects at least a 2-D array")
    cols, rows = arr.shape[-1], arr.shape[-2]
    flat = np.ascontiguousarray(arr.reshape(-1, cols), dtype=np.float32)
    r = get_renderer()
    dev = r.mem.upload(flat.reshape(-1))
    stats = r.mem.download(r.row_stats(dev, flat.shape[0], cols)).reshape(-1, 4)[: flat.shape[0]]
    e = np.sqrt(stats[:, 3]).reshape(-1, rows)
    e = e + tiny(e)
    # as float32 a 1 / tiny (all-zero rows: the reference keeps zeros) would be inf and inf * 0 = NaN: saturate, FLT_MAX * 0 = 0
    scale = np.repeat(np.minimum(1.0 / e.mean(axis=1), np.finfo(np.float32).max), rows).astype(np.float32)
    s_dev = r.mem.upload(scale)
    for r0 in range(0, flat.shape[0], 65535):  # one launch per 65535 rows (grid.y limit)
        nr = min(65535, flat.shape[0] - r0)
        r.lib.call("al_scale_matrix_rows", r.mem.ptr(dev) + 4 * r0 * cols, nr, cols, r.mem.ptr(s_dev) + 4 * r0, r.mem.stream())
    out = r.mem.download(dev)[: flat.size].reshape(arr.shape)
    return out.astype(arr.dtype) if np.issubdtype(arr.dtype, np.floating) else out.astype(np.float64)


# ----------------------------------------------------------------------------- event rendering
def _clip_of(event, ignore_cache: bool, chain_is_fresh: bool = False):
    """The event's clip for the renderer: ``engine.ClipSource`` from events that can hand it over without a host
    round trip (core.Event.clip_source: device FX chain, folded Gain/Invert + peak normalisation), else the array
    ``event.load_audio()`` returns (reference Events: event.py:496-539)."""
    return _clip_and_dtype(event, ignore_cache, chain_is_fresh)[0]


def _clip_and_dtype(event, ignore_cache: bool, chain_is_fresh: bool = False):
    """(``_clip_of``'s clip, the dtype ``event.load_audio()`` hands the reference's arithmetic: float32 for this package's Events
    and for the reference's, librosa.load's dtype; whatever a duck-typed event returns otherwise)."""
    if hasattr(event, "clip_source"):
        src = event.clip_source(ignore_cache=ignore_cache, chain_is_fresh=chain_is_fresh)
        loaded = getattr(event, "is_audio_loaded", False) and not ignore_cache and isinstance(getattr(event, "audio", None), np.ndarray)
        return src, (event.audio.dtype if loaded else event.chain_dtype() if hasattr(event, "chain_dtype") else np.dtype(np.float32))
    audio = event.load_audio(ignore_cache=ignore_cache, normalize=True)
    valid_audio(audio)
    return np.ascontiguousarray(audio, dtype=np.float32), audio.dtype


def _spec_of(event, clip: np.ndarray, n_emitters: int, emitter0: int, ref_db: float) -> planning.EventSpec:
    if n_emitters == 1 and event.is_moving:
        raise ValueError("Moving Event has only one emitter!")
    if n_emitters > 1 and not event.is_moving:
        raise ValueError("Expected a moving event!")
    if n_emitters == 0:
        logger.warning(f"No IRs were found for Event with alias {event.alias}. Audio is being tiled along the "
                       f"channel dimension to match the expected shape.")
    return planning.EventSpec(n_samples=len(clip), n_emitters=n_emitters, snr=float(event.snr), emitter0=emitter0,
                              is_moving=bool(event.is_moving), duration=getattr(event, "duration", None),
                              ref_db=float(ref_db))


def _reference_dtype(a_dt, n_emitters: int, irs) -> np.dtype:
    """The dtype the reference's arithmetic leaves ``event.spatial_audio[mic]`` (and the dry render) in (synthesize.py:560-599 under
    NumPy 2's weak Python scalars): a clip tiled over the capsules keeps the CLIP's dtype (float32 from Event.load_audio), a static
    event has fftconvolve's result type of clip and IRs (float32 only if both are), a moving event passes through complex128 STFTs and
    is float64 whatever went in."""
    a_dt = np.dtype(a_dt)
    if n_emitters == 0:
        return a_dt
    if n_emitters > 1:
        return np.dtype(np.float64)
    i_dt = getattr(irs, "dtype", np.float64)
    try:
        i_dt = np.dtype(i_dt)
    except TypeError:                                   # a device tensor's dtype ("torch.float32"): by name, nothing is copied
        i_dt = np.dtype(str(i_dt).rpartition(".")[2])
    return np.result_type(a_dt, i_dt if np.issubdtype(i_dt, np.floating) else np.float64)


def _publish(event, mic_alias: str, res: engine.RenderResult, index: int, dtype=np.float64) -> None:
    """event.spatial_audio[mic] = scaled (C, La) render, kept in HBM until read (synthesize.py:606), handed back in ``dtype``."""
    event.spatial_audio = as_lazy(getattr(event, "spatial_audio", None))
    n_ch, n_samp = res.plan.n_capsules, int(res.plan.events["len"][index])

    def fetch():
        res.check_finite()      # librosa.util.valid_audio of synthesize.py:603: raised when the render first meets the host
        out = res.spatial_audio(index, dtype)
        validate_shape(out.shape, (n_ch, n_samp))
        return out

    # the mixdown reads the render straight from HBM for as long as THIS entry is what event.spatial_audio[mic] holds:
    # an assignment by the user or clear_audio() replaces / drops the entry and with it the device source
    fetch.al_device_source = (res, index)
    LazyAudioDict.__setitem__(event.spatial_audio, mic_alias, fetch)


def compute_dry_audio(event, irs: np.ndarray, event_scale: float, mic_alias: str) -> None:
    """Direct-path ("dry") render on the reference capsule (reference synthesize.py:432-504).

    ``irs`` are the NORMALISED IRs (C, N, L); the windowed IR of ``ref_ir_channel`` is convolved
    with the clip on the GPU and scaled by ``event_scale``.
    """
    ref, window = getattr(event, "ref_ir_channel", None), getattr(event, "direct_path_time_ms", None)
    if ref is None and window is None:
        return
    if ref is None or window is None:
        logger.warning("Only one of `ref_ir_channel` or `direct_path_time` were specified when creating the Event. "
                       "Dry audio will not be computed for this Event. Pass both variables to compute dry audio.")
        return
    if ref > irs.shape[0]:
        raise ValueError(f"Reference channel index out of range for IRs with {irs.shape[0]} channels")
    lo = int(window[0] * event.sample_rate / 1000)
    hi = int(window[1] * event.sample_rate / 1000)
    ir = np.array(irs[ref, 0, :], dtype=np.float64)
    peak = int(np.argmax(ir))
    if peak + hi < ir.shape[0]:
        ir[peak + hi:] = 0
    if peak - lo > 0:
        ir[: peak - lo] = 0
    dry = time_invariant_convolution(np.asarray(event.load_audio(ignore_cache=False)), ir[:, None])[0]
    if not hasattr(event, "_spatial_audio_dry") or event._spatial_audio_dry is None:
        event._spatial_audio_dry = {}
    event._spatial_audio_dry[mic_alias] = dry * event_scale


def _wants_dry(event) -> bool:
    return getattr(event, "ref_ir_channel", None) is not None and getattr(event, "direct_path_time_ms", None) is not None


def _dry_batch(r: engine.Renderer, items, mic_alias: str, res: engine.RenderResult) -> None:
    """The direct-path renders of ALL events of one microphone that ask for one (synthesize.py:432-504), as ONE batch of
    one-capsule static events behind the main render instead of an upload / render / download round trip per event:
    clip zero-padded to La + Lir - 1 (the dry render keeps the whole convolution), IR = the windowed row of the reference
    capsule, UN-normalised; the emitter's normalisation gain and the event's noise-floor multiplier -- both results of the
    main render -- are applied when ``event._spatial_audio_dry[mic]`` is first read, so nothing here waits for the GPU.
    ``items``: [(event, mic_ir slice (C, N, L), index in ``res``, first IR column)]."""
    todo = []
    for event, irs, index, em0 in items:
        ref, window = getattr(event, "ref_ir_channel", None), getattr(event, "direct_path_time_ms", None)
        if ref is None and window is None:
            continue
        if ref is None or window is None:
            logger.warning("Only one of `ref_ir_channel` or `direct_path_time` were specified when creating the Event. "
                           "Dry audio will not be computed for this Event. Pass both variables to compute dry audio.")
            continue
        if ref > irs.shape[0]:
            raise ValueError(f"Reference channel index out of range for IRs with {irs.shape[0]} channels")
        if irs.shape[1] == 0:
            continue
        lo, hi = int(window[0] * event.sample_rate / 1000), int(window[1] * event.sample_rate / 1000)
        ir = np.array(irs[ref, 0, :], dtype=np.float32)
        peak = int(np.argmax(ir))     # the normalisation gain is positive: the peak of the normalised row is this one
        if peak + hi < ir.shape[0]:
            ir[peak + hi:] = 0
        if peak - lo > 0:
            ir[: peak - lo] = 0
        audio = event.load_audio(ignore_cache=False)
        clip = np.asarray(audio, dtype=np.float32)
        todo.append((event, ir, clip, index, em0, _reference_dtype(np.asarray(audio).dtype, 1, irs)))
    if not todo:
        return
    n_ir = todo[0][1].shape[0]
    specs = [planning.EventSpec(n_samples=len(clip) + n_ir - 1, n_emitters=1, snr=1.0, emitter0=j)
             for j, (_, _, clip, _, _, _) in enumerate(todo)]
    clips = [np.concatenate([clip, np.zeros(n_ir - 1, np.float32)]) for _, _, clip, _, _, _ in todo]
    pl = planning.plan_batch(specs, 1, n_ir, todo[0][0].sample_rate, lib=r.lib)
    dry = r.prepare(pl, clips, np.stack([ir for _, ir, _, _, _, _ in todo])[None, :, :], normalize_irs=False)
    dry_res = dry.run(("al_forward_spectra", "al_emitter_gains", "al_spectral_mac", "al_block_synthesis"))
    for j, (event, _, _, index, em0, dtype) in enumerate(todo):
        def fetch(j=j, index=index, em0=em0, dtype=dtype):
            # scaled on the device by the two results of the MAIN render it depends on -- emitter_gain[em0] (normalize_irs) and
            # event_stats[index][3], the noise-floor multiplier the reference calls event_scale (synthesize.py:598,608) --
            # and widened to the reference's dtype (float64 unless clip AND IRs are float32) after the copy
            ev = dry_res.plan.events[j]
            mem = res.memory
            dev = dry_res.scaled_copy(int(ev["out_off"]), int(ev["len"]), [mem.ptr(res.emitter_gain) + 4 * em0],
                                      [mem.ptr(res.event_stats) + 8 * (4 * index + 3)])
            return mem.download(dev)[: int(ev["len"])].astype(dtype)

        event._spatial_audio_dry = as_lazy(getattr(event, "_spatial_audio_dry", None))
        LazyAudioDict.__setitem__(event._spatial_audio_dry, mic_alias, fetch)


def _render_moving_general(r: engine.Renderer, spec, clip, irs: np.ndarray, fft_size: int, win_size: int, hop_size: int,
                           sample_rate: float) -> engine.RenderResult:
    """One moving event whose STFT geometry is outside the envelope form (win != 2*hop or fft < 2*win - 1), all on the device:
    the batch machinery gives the buffers, the IR energies and normalize_irs' scalar per emitter (al_forward_spectra +
    al_emitter_gains on a plan of this event; A1), the IR rows are scaled by those gains, the literal STFT chain convolves
    (``_tv_chain_on_device``), the result lands truncated / zero-padded to the clip length in the batch's ``spatial`` buffer (A3),
    al_row_stats + al_event_levels_from_stats evaluate the level law (A4/A5/A9).  The RenderResult is the one every other
    event gets, so the lazy download, the dry render and the mixdown see no difference."""
    import ctypes as ct

    mem, lib, st = r.mem, r.lib, r.mem.stream()
    n_ch, n_irs, n_ir = irs.shape
    host_clip = engine.as_clip_source(clip)
    pl = planning.plan_batch([spec], n_ch, n_ir, sample_rate, lib=lib)          # layout + tables only (default geometry)
    # emitter_parts = 0 everywhere: the IR pass only takes the energies (normalize_irs); no partition spectrum, signal spectrum or
    # output spectrum is ever made on this path, so those workspaces are one block each
    batch = r.prepare(pl, [clip], irs, emitter_parts=np.zeros(n_irs, dtype=np.int32), spectra_workspaces=False)
    desc = batch.descs[0]
    for name in ("al_ir_spectra", "al_emitter_gains"):
        lib.call(name, ct.byref(desc), st)
    bufs = batch.bufs
    pitch = int(desc.ir_stride_n)
    gains = bufs["emitter_gain"][:n_irs]
    if int(desc.ir_stride_c) == n_irs * pitch:     # the (C * N) rows are evenly spaced: h[c, n, :] *= g[n] as ONE launch per 65 535 rows
        tiled = gains.repeat(n_ch) if hasattr(gains, "repeat") and not isinstance(gains, np.ndarray) else np.tile(gains, n_ch)
        for r0 in range(0, n_ch * n_irs, 65535):
            nr = min(65535, n_ch * n_irs - r0)
            lib.call("al_scale_matrix_rows", mem.ptr(bufs["ir"]) + 4 * r0 * pitch, nr, pitch, mem.ptr(tiled) + 4 * r0, st)
    else:
        for c in range(n_ch):      # a capsule pitch of its own: one launch per capsule over its (N, pitch) rows
            lib.call("al_scale_matrix_rows", mem.ptr(bufs["ir"]) + 4 * c * int(desc.ir_stride_c), n_irs, pitch, mem.ptr(gains), st)
    n_audio = len(host_clip)
    audio_dev = bufs["audio"]      # the clip as uploaded (a finished clip: event.load_audio()'s peak-normalised array)
    if bufs.get("clip_scale") is not None:   # folded scalar FX / device-side peak normalisation: applied to a copy of the clip
        audio_dev = audio_dev[:n_audio].clone() if hasattr(audio_dev, "clone") else np.array(audio_dev[:n_audio])
        lib.call("al_scale_rows", mem.ptr(audio_dev), n_audio, mem.ptr(bufs["clip_scale"]), st)
    out, n_out = _tv_chain_on_device(r, bufs["ir"], n_ch, n_irs, n_ir, audio_dev, n_audio, float(spec.duration), sample_rate,
                                     fft_size, win_size, hop_size, ir_pitch=pitch)
    # (n_out, C) -> the event's (C, La) block of `spatial`, cut or zero-padded to the clip length (synthesize.py:590)
    lo, keep = int(pl.events["out_off"][0]), min(n_out, n_audio)
    block = bufs["spatial"][lo: lo + n_ch * n_audio]
    block[:] = 0
    if keep > 0:
        if hasattr(block, "permute"):      # torch (HBM); else numpy (host emulation)
            block.view(n_ch, n_audio)[:, :keep] = out[: n_out * n_ch].view(n_out, n_ch)[:keep].t()
        else:
            block.reshape(n_ch, n_audio)[:, :keep] = out[: n_out * n_ch].reshape(n_out, n_ch)[:keep].T
    partials = mem.empty(lib.call("al_row_stats_partials", 1, n_ch * n_audio))
    lib.call("al_row_stats", mem.ptr(block), 1, n_ch * n_audio, mem.ptr(partials), mem.ptr(bufs["event_stats"]), st)
    lib.call("al_event_levels_from_stats", ct.byref(desc), n_ch, st)
    return batch.result()


def render_event_audio(event, irs: np.ndarray, mic_alias: str, ref_db=config.DEFAULT_REF_DB,
                       ignore_cache: Optional[bool] = True, fft_size=config.FFT_SIZE, win_size=config.WIN_SIZE,
                       hop_size=config.HOP_SIZE) -> None:
    """Render one Event at one microphone (reference synthesize.py:507-608).

    IR energy normalisation, static / tiled / time-variant convolution, truncation to the clip
    length, SNR + noise-floor scaling and the finite check all run on the GPU; the result is
    stored in ``event.spatial_audio[mic_alias]`` (and the dry render when the event asks for it).
    """
    if mic_alias in getattr(event, "spatial_audio", {}).keys() and not ignore_cache:
        return
    n_ch, n_emitters, n_ir = irs.shape
    fft_size, win_size, hop_size = int(fft_size), int(win_size), int(hop_size)
    clip, clip_dtype = _clip_and_dtype(event, bool(ignore_cache))
    spec = _spec_of(event, clip, n_emitters, 0, ref_db)
    r = get_renderer()
    if n_emitters > 1:
        _check_geometry(fft_size, win_size, hop_size)
    if n_emitters > 1 and not _envelope_geometry(fft_size, win_size, hop_size):
        res = _render_moving_general(r, spec, clip, irs, fft_size, win_size, hop_size, float(event.sample_rate))
    else:
        geometry = dict(hop=hop_size, win=win_size, fft_size=fft_size) if n_emitters > 1 else {}   # static / tiled events never frame
        pl = planning.plan_batch([spec], n_ch, max(n_ir, 1), event.sample_rate, lib=r.lib, **geometry)
        res = r.render(pl, [clip], irs)
    res.check_finite()
    _publish(event, mic_alias, res, 0, _reference_dtype(clip_dtype, n_emitters, irs))
    _dry_batch(r, [(event, irs, 0, 0)], mic_alias, res)


def render_audio_for_all_scene_events(scene, ignore_cache: Optional[bool] = False) -> None:
    """Render every Event of a Scene at every microphone (reference synthesize.py:613-677).

    One batched launch sequence per microphone: all events of the scene are convolved together.
    """
    if ignore_cache:
        scene.state.simulate()
    else:
        try:
            _ = scene.state.irs
        except AttributeError:
            scene.state.simulate()
    validate_scene(scene)
    irs = scene.state.get_irs()
    start = time()
    r = get_renderer()
    from .core import stage_event_chains

    def has_work(mic_alias):
        return ignore_cache or any(mic_alias not in getattr(ev, "spatial_audio", {}).keys() for ev in scene.events.values())

    # The IR tensors start on their way to HBM first, on a helper thread (the one long blocking call of a scene: 14 ms for
    # cfg2's 0.8 GB of pageable memory); this thread stages the clips and plans the batches meanwhile.
    ir_on_device = {}
    if hasattr(r, "upload_irs_beside"):
        for mic_alias, mic_ir in irs.items():
            started = r.upload_irs_beside(mic_ir) if has_work(mic_alias) else None
            if started is not None:
                ir_on_device[mic_alias] = started
    for n_mic, (mic_alias, mic_ir) in enumerate(irs.items()):
        # FX chains that have to run on the device: raw clips through one staging arena + one DMA (once per scene; with
        # ignore_cache every microphone draws a fresh realisation, as the reference's per-microphone load_audio does)
        if n_mic == 0 or ignore_cache:
            fresh = stage_event_chains([ev for ev in scene.events.values()
                                        if ignore_cache or mic_alias not in getattr(ev, "spatial_audio", {}).keys()],
                                       bool(ignore_cache))
        specs, clips, todo = [], [], []
        counter = 0
        for event in scene.events.values():
            n_emit = len(event)
            cached = mic_alias in getattr(event, "spatial_audio", {}).keys() and not ignore_cache
            if not cached:
                clip, clip_dtype = _clip_and_dtype(event, bool(ignore_cache), any(event is f for f in fresh))
                specs.append(_spec_of(event, clip, n_emit, counter, scene.ref_db))
                clips.append(clip)
                todo.append((event, counter, clip_dtype))
            counter += n_emit
        if not specs:
            continue
        pl = planning.plan_batch(specs, mic_ir.shape[0], mic_ir.shape[2], scene.sample_rate, lib=r.lib)
        # Only ENQUEUED here: nothing below waits for the GPU.  The finite check of the reference (librosa.util.valid_audio,
        # synthesize.py:603) is made where the results first meet the host: in generate_scene_audio_from_events (one combined
        # download of every statistic of the scene) or on the first read of event.spatial_audio[mic].
        res = r.render(pl, clips, ir_on_device.pop(mic_alias, mic_ir))
        for i, (event, em0, clip_dtype) in enumerate(todo):
            _publish(event, mic_alias, res, i, _reference_dtype(clip_dtype, len(event), mic_ir))
        _dry_batch(r, [(event, mic_ir[:, em0: em0 + len(event), :], i, em0) for i, (event, em0, _) in enumerate(todo)],
                   mic_alias, res)
    logger.info(f"Rendered scene audio in {(time() - start):.2f} seconds!")


# ----------------------------------------------------------------------------- mixdown
def _device_source(r: engine.Renderer, event, mic_alias: str):
    """(device buffer, offset, len, rows, scale buffer, scale index, RenderResult or None) of an event's render; uploads
    host arrays that did not come from this package (scale 1)."""
    sa = event.spatial_audio
    held = sa.device_source(mic_alias) if isinstance(sa, LazyAudioDict) else None
    if held is not None:
        res, idx = held
        ev = res.plan.events[idx]
        return res.spatial, int(ev["out_off"]), int(ev["len"]), res.plan.n_capsules, res.event_scale, idx, res
    arr = np.ascontiguousarray(event.spatial_audio[mic_alias], dtype=np.float32)
    return r.mem.upload(arr.reshape(-1)), 0, arr.shape[1], arr.shape[0], r.mem.upload(np.ones(1, np.float32)), 0, None


def generate_scene_audio_from_events(scene) -> None:
    """Ambience + additive mixdown of all events into ``scene.audio[mic]`` (reference synthesize.py:314-401).

    The mixdown is a tiled segmented sum on the GPU (deterministic, insertion order, float32 like the
    reference buffer).  ``event._spatial_audio_padded[mic]`` is produced lazily on access instead of
    allocating E full-scene buffers.
    """
    from .ambience import Ambience

    r = get_renderer()
    # phase 1: enqueue every microphone's mixdown, statistics and downloads; phase 2: ONE synchronisation for the whole scene
    queued = []
    pending = {}
    for mic_alias in scene.state.microphones.keys():
        events = list(scene.events.values())
        srcs = [_device_source(r, ev, mic_alias) for ev in events]
        channels = max(s[3] for s in srcs)
        duration = round(scene.duration * scene.sample_rate)
        amb_dev = []
        for ambience in scene.ambience.values() if len(scene.ambience) > 0 else []:
            if not isinstance(ambience, Ambience) and not hasattr(ambience, "load_ambience"):
                raise TypeError(f"Expected scene ambient noise to be of type Ambience, but got {type(ambience)}!")
            amb_dev.append(_ambience_on_device(r, ambience, (channels, duration)))
        # one mixdown launch per distinct source buffer (normally exactly one: the batch render)
        groups: Dict[int, List[int]] = {}
        for i, s in enumerate(srcs):
            groups.setdefault(id(s[0]), []).append(i)
        scene_dev = None
        first = True
        for idxs in groups.values():
            sub = [srcs[i] for i in idxs]
            mix = planning.plan_mixdown([events[i].scene_start for i in idxs], [events[i].scene_end for i in idxs],
                                        [s[2] for s in sub], [s[3] for s in sub], [s[1] for s in sub],
                                        [s[5] for s in sub], scene.duration, scene.sample_rate, channels, lib=r.lib)
            for k in mix.skipped:
                a, b = planning.event_slot(events[idxs[k]].scene_start, events[idxs[k]].scene_end, scene.sample_rate, duration)
                logger.warning(f"Skipping event due to invalid slice: start={a}, end={b}")
            fake = engine.RenderResult(plan=None, memory=r.mem, lib=r.lib, spatial=sub[0][0], event_scale=sub[0][4],
                                       event_stats=None, emitter_gain=None)
            pm = r.prepare_mixdown(mix, fake, amb_dev if first else [], scene=scene_dev)
            scene_dev = pm.run()
            first = False
            _attach_padded(events, idxs, mix, mic_alias, channels, duration)
        # librosa.util.valid_audio (synthesize.py:398,603) from device reductions (a host pass over the scene costs more than
        # rendering it): the scenes' statistics, every pending render's per-event statistics and the scenes themselves are
        # downloaded behind ONE synchronisation, each scene's DMA enqueued ahead of the small ones so nothing waits in front of it.
        for s_ in srcs:
            held_res = s_[6]
            if held_res is not None and not getattr(held_res, "_finite_ok", False):
                pending[id(held_res)] = held_res
        scene_stats_dev = r.row_stats(scene_dev, 1, channels * duration)
        if hasattr(r.mem, "download_async"):
            fetch = (r.mem.download_async(scene_dev), r.mem.download_async(scene_stats_dev))
        else:
            fetch = None
        queued.append((mic_alias, channels, duration, scene_dev, scene_stats_dev, fetch))
    if hasattr(r.mem, "download_async"):
        ev_t = {k: r.mem.download_async(v.event_stats) for k, v in pending.items()}
        r.mem.synchronize()
        ev_stats = {k: t.numpy() for k, t in ev_t.items()}
    else:
        ev_stats = {k: r.mem.download(v.event_stats) for k, v in pending.items()}
    for k, res_ in pending.items():
        res_.check_finite(ev_stats[k][: 4 * len(res_.plan.events)].reshape(-1, 4))
    for mic_alias, channels, duration, scene_dev, scene_stats_dev, fetch in queued:
        if fetch is not None:
            host, stats = fetch[0].numpy(), fetch[1].numpy().reshape(-1, 4)
        else:
            host, stats = r.mem.download(scene_dev), r.mem.download(scene_stats_dev).reshape(-1, 4)
        if stats[0, 2] > 0 or not np.isfinite(stats[0, 0]):
            raise ValueError("Audio buffer is not finite everywhere")
        host = host[: channels * duration].reshape(channels, duration)
        validate_shape(host.shape, (channels, duration))
        scene.audio[mic_alias] = host
        # the mixed scene stays in HBM too: Scene.generate encodes the WAV frames from it on the device (al_encode_frames)
        try:
            if not hasattr(scene, "_al_device_audio"):
                scene._al_device_audio = {}
            scene._al_device_audio[mic_alias] = (scene_dev, channels, duration, host)
        except AttributeError:
            pass


def encode_scene_frames(scene, mic_alias: str, subtype: str = "PCM_16") -> np.ndarray:
    """(T, C) frames of ``scene.audio[mic]`` as ``soundfile.write(audio.T, sr)`` stores them (core.py:1840-1847):
    int16 for ``PCM_16`` (soundfile's default subtype for WAV; lrintf(x * 32768) saturated to int16: libsndfile with clipping on, as python-soundfile sets it) or float32 for ``FLOAT``.
    Interleaving and quantisation run on the device, from the resident mix when ``scene.audio[mic]`` is still the
    array this package produced, else from an upload of it."""
    from . import _hip

    fmt, dtype = {"PCM_16": (_hip.FRAMES_PCM16, np.int16), "FLOAT": (_hip.FRAMES_F32, np.float32)}[subtype]
    r = get_renderer()
    host = scene.audio[mic_alias]
    held = getattr(scene, "_al_device_audio", {}).get(mic_alias)
    if held is not None and held[3] is host:
        dev, c, t = held[0], held[1], held[2]
    else:
        c, t = host.shape
        dev = r.mem.upload(np.ascontiguousarray(host, dtype=np.float32).reshape(-1))
    out = r.mem.empty(c * t, dtype)
    r.lib.call("al_encode_frames", r.mem.ptr(dev), c, t, fmt, r.mem.ptr(out), r.mem.stream())
    return r.mem.download(out)[: c * t].reshape(t, c)


def _ambience_on_device(r: engine.Renderer, ambience, shape):
    """(device noise, device float32[channels] of per-channel multipliers): load_ambience(normalize=True) x
    db_to_multiplier(ref_db, mean|noise|) (synthesize.py:350-356).  Device-drawn ambiences never meet the host: the noise
    stays un-normalised and the peak normalisation rides in the per-channel multiplier (Ambience.noise_and_scales_device)."""
    if hasattr(ambience, "noise_and_scales_device"):
        pair = ambience.noise_and_scales_device(r, shape)
        if pair is not None:
            return pair
    dev = ambience.load_ambience_device(r) if hasattr(ambience, "load_ambience_device") else None
    if dev is None:
        host = np.asarray(ambience.load_ambience(normalize=True))
        if host.shape != tuple(shape):
            raise ValueError(f"Scene ambient noise does not match expected shape. Expected {tuple(shape)}, but got {host.shape}.")
        if host.dtype == np.float64 and hasattr(r.mem, "upload_f64_as_f32"):
            # the reference's Ambience hands over float64 (737 MB at cfg2): cast by the thread pool under the DMA instead of a
            # single-threaded astype (60 ms) in front of it
            dev = r.mem.upload_f64_as_f32(host)
        else:
            dev = r.mem.upload(np.ascontiguousarray(host, dtype=np.float32).reshape(-1))
    elif tuple(ambience.device_shape) != tuple(shape):
        raise ValueError(f"Scene ambient noise does not match expected shape. Expected {tuple(shape)}, but got {tuple(ambience.device_shape)}.")
    if shape[0] <= 1024:
        # db_to_multiplier(ref_db, mean|noise|) from the per-channel statistics, on the stream: the host neither waits nor reads
        scales = r.mem.empty(shape[0])
        r.lib.call("al_ambience_scales", r.mem.ptr(r.row_stats(dev, shape[0], shape[1])), shape[0], shape[1], float(ambience.ref_db), 0,
                   r.mem.ptr(scales), r.mem.stream())
        return dev, scales
    n = shape[0] * shape[1]
    stats = r.mem.download(r.row_stats(dev, 1, n)).reshape(-1, 4)
    mult = min(db_to_multiplier(ambience.ref_db, stats[0, 0] / n), float(np.finfo(np.float32).max))   # silent noise: finite x 0 = 0
    return dev, r.mem.upload(np.full(shape[0], mult, dtype=np.float32))


def _attach_padded(events, idxs, mix, mic_alias, channels, duration) -> None:
    """event._spatial_audio_padded[mic] (synthesize.py:381-383) and the dry padded copy (386-395), lazily."""
    for k, i in enumerate(idxs):
        ev = events[i]
        if k in mix.skipped:
            continue
        a, b = planning.event_slot(ev.scene_start, ev.scene_end, _sr_of(ev), duration)
        ev._spatial_audio_padded = as_lazy(getattr(ev, "_spatial_audio_padded", None))

        # the closures live in the event's own dictionaries: a strong reference to the event would be a cycle that keeps the
        # event -- and through event.spatial_audio the whole render's device buffers -- alive until the garbage collector runs
        ev_ref = _weak(ev)

        def fetch(ev_ref=ev_ref, a=a, b=b):
            ev = ev_ref()
            full = np.zeros((channels, duration), dtype=np.float32)
            piece = pad_or_truncate_audio(ev.spatial_audio[mic_alias], b - a)
            full[: piece.shape[0], a:b] += piece
            return full

        LazyAudioDict.__setitem__(ev._spatial_audio_padded, mic_alias, fetch)
        dry = getattr(ev, "_spatial_audio_dry", None)
        if dry is not None and mic_alias in dry:     # kept lazy like the dry render itself: no download unless somebody reads it
            def fetch_dry(ev_ref=ev_ref, a=a, b=b):
                ev = ev_ref()
                line = np.zeros(duration, dtype=np.float32)
                line[a:b] += pad_or_truncate_audio(np.asarray(ev._spatial_audio_dry[mic_alias])[None, :], b - a)[0]
                return line

            ev._spatial_audio_dry_padded = as_lazy(getattr(ev, "_spatial_audio_dry_padded", None))
            LazyAudioDict.__setitem__(ev._spatial_audio_dry_padded, mic_alias, fetch_dry)


def _weak(obj):
    """weakref.ref(obj), or a strong stand-in for objects that cannot be weakly referenced (duck-typed events with __slots__)."""
    import weakref

    try:
        return weakref.ref(obj)
    except TypeError:
        return lambda: obj


def _sr_of(ev):
    return getattr(ev, "sample_rate", config.SAMPLE_RATE)


# ----------------------------------------------------------------------------- validation
def validate_scene(scene) -> None:
    """Pre-synthesis checks with the reference's messages (reference synthesize.py:681-739)."""
    if scene.state.num_emitters == 0:
        raise ValueError("WorldState has no emitters!")
    if len(scene.state.microphones) == 0:
        raise ValueError("WorldState has no microphones!")
    if len(scene.events) == 0:
        raise ValueError("Scene has no events!")
    total = 0
    for alias, ev in scene.events.items():
        try:
            total += len(ev)
        except ValueError:
            raise ValueError(f"Event with alias '{alias}' has no emitters registered. Has it been orphaned?")
    if not str(scene.state.name).upper() == "RLR":
        return
    ctx = scene.state.ctx
    if ctx.get_listener_count() == 0:
        raise ValueError("Ray-tracing engine has no listeners!")
    if ctx.get_source_count() == 0:
        raise ValueError("Ray-tracing engine has no sources!")
    counts = (total, scene.state.num_emitters, ctx.get_source_count())
    if not all(v == counts[0] for v in counts):
        raise ValueError(f"Mismatching number of emitters, events, and sources! Got {len(scene.events)} events, "
                         f"{scene.state.num_emitters} emitters, {ctx.get_source_count()} sources. Have any been orphaned?")
    capsules = sum(m.n_listeners for m in scene.state.microphones.values())
    if capsules != ctx.get_listener_count():
        raise ValueError(f"Mismatching number of microphones and listeners! Got {capsules} capsules, "
                         f"{ctx.get_listener_count()} listeners. Have any been orphaned?")


def __getattr__(name):
    # the reference keeps generate_dcase2024_metadata in synthesize.py (:742-878); here it is host bookkeeping outside the hot path
    # (audiblelight_amd/metadata.py), reachable under the reference's name without this module importing it
    if name in ("generate_dcase2024_metadata", "DCASE_2024_COLUMNS"):
        from . import metadata

        return getattr(metadata, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
