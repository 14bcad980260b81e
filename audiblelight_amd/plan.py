"""Host-side planning for the HIP synthesis path: pure index arithmetic, no audio arithmetic.

Turns "events of one microphone" (clip lengths, emitter columns, SNRs, trajectories) into the
tables the kernels read (``al_event`` / ``al_stream`` of include/audiblelight_hip.h), the
workspace sizes, and the mixdown slot/tile lists.  The arithmetic itself lives BEHIND the C ABI
(csrc/al_plan.cpp: al_plan_create, al_plan_emitter_parts, al_workspace_bytes, al_plan_mixdown,
al_interpolation_matrix), so a host in another language gets the same tables; this module is
the ctypes caller that wraps them as numpy arrays.  The samples never pass through here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import ctypes as ct

import numpy as np

from . import _hip, config
from ._hip import EVENT_DTYPE, MAX_LOG2_BLOCK, MIN_LOG2_BLOCK, STREAM_DTYPE


def _lib(lib=None):
    """The library a planning call goes through: the caller's (a Renderer built on its own Library -- host emulation, a
    sanitizer build, AUDIBLELIGHT_HIP_LIB -- plans with THAT library, whose al_plan_last_error is the one that holds the
    message) or the planner-only library (csrc/al_plan.cpp linked alone: no HIP, no torch, no GPU needed to plan)."""
    return lib if lib is not None else _hip.get_planner()


def _copy(ptr, n, dtype):
    """A numpy copy of ``n`` items of ``dtype`` behind a host pointer the planner library owns."""
    n = int(n)
    if n <= 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    return np.frombuffer(ct.string_at(ptr, nbytes), dtype=dtype).copy()


SPARSE_MAX_NJ = 6  # AL_SPARSE_MAX_NJ of include/audiblelight_hip.h
SPARSE_MAX_PARTITIONS = 24  # AL_SPARSE_MAX_PARTITIONS: beyond it flagged events go through the tile accumulate, which reads every partition


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def generate_interpolation_matrix(ir_times: np.ndarray, sr=config.SAMPLE_RATE, hop_size=config.HOP_SIZE,
                                  n_frames: Optional[int] = None, lib=None) -> np.ndarray:
    """Linear cross-fade weights W[frame, ir] between consecutive IRs of a moving source.

    Same contract as the reference's ``generate_interpolation_matrix`` (synthesize.py:148-181):
    IR ``l`` starts at frame ``round((t_l*sr + hop)/hop)`` and fades linearly into IR ``l+1``.
    """
    times = np.asarray(ir_times, dtype=np.float64)
    n = len(times)
    uniform = n >= 1 and times[0] == 0.0 and np.array_equal(times, np.linspace(0.0, times[-1], n))
    if not uniform:      # arbitrary IR times (the render path only ever asks for linspace(0, duration, n)): same rule, in numpy
        first = np.round((times * sr + hop_size) / hop_size)
        rows = int(first[-1]) if n_frames is None else int(n_frames)
        weights = np.zeros((rows, n))
        for l in range(n - 1):
            idx = np.arange(first[l], first[l + 1] + 1, dtype=int) - 1
            up = np.linspace(0.0, 1.0, len(idx))
            keep = (idx >= 0) & (idx < rows)
            weights[idx[keep], l] = 1.0 - up[keep]
            weights[idx[keep], l + 1] = up[keep]
        return weights
    lib = _lib(lib)
    duration = float(times[-1])
    rows = lib.call("al_interpolation_rows", n, duration, float(sr), int(hop_size)) if n_frames is None else int(n_frames)
    weights = np.zeros((max(rows, 0), n), dtype=np.float64)
    lib.call("al_interpolation_matrix", n, duration, float(sr), int(hop_size), int(rows), weights.ctypes.data)
    return weights


def stft_frame_count(n_samples: int, hop_size: int = config.HOP_SIZE, lib=None) -> int:
    """Frames the reference STFT produces for n_samples (synthesize.py:123)."""
    return int(_lib(lib).call("al_stft_frame_count", int(n_samples), int(hop_size)))


@dataclass
class EventSpec:
    """Shape-level description of one event at one microphone."""
    n_samples: int               # clip length La
    n_emitters: int              # len(event)
    snr: float
    emitter0: int = 0            # first IR column of this event (synthesize.py:662)
    is_moving: bool = False
    duration: Optional[float] = None   # seconds, Event.duration (moving events)
    gain: float = 1.0            # scalar folded into the clip (peak normalisation, FX gain/polarity)
    ref_db: float = config.DEFAULT_REF_DB
    stft_len: Optional[int] = None     # samples the STFT frame count is taken from (default n_samples)


@dataclass
class BatchPlan:
    log2_block: int
    n_capsules: int
    ir_len: int
    n_emitters: int
    events: np.ndarray           # EVENT_DTYPE
    streams: np.ndarray          # STREAM_DTYPE
    wtab: np.ndarray             # float32
    audio_offsets: np.ndarray    # int64 per event
    audio_floats: int
    spatial_floats: int
    xspec_blocks: int
    yspec_blocks: int
    n_partials: int              # entries of 4 floats
    hop: int = config.HOP_SIZE
    fft_size: int = config.FFT_SIZE
    _specs: Optional[list] = None        # what plan_batch was given
    _sample_rate: float = 0.0
    _win: int = config.WIN_SIZE
    _handle: object = None               # _PlanHandle: the C planner's object (al_plan *), shared by copies of this plan
    _library: object = None              # the _hip.Library that made (and frees) the handle; None = process default

    @property
    def block(self) -> int:
        return 1 << self.log2_block

    @property
    def n_partitions(self) -> int:
        return -(-self.ir_len // self.block) if self.n_emitters else 0

    @property
    def hspec_blocks(self) -> int:
        return self.n_emitters * self.n_capsules * self.n_partitions

    def _c_plan(self):
        """The C planner's own object for this batch (the tables above are copies of its arrays); lives as long as this plan."""
        if self._handle is None:
            if self._specs is None:
                raise ValueError("this BatchPlan was not made by plan_batch")
            self._handle = _PlanHandle(_create_c_plan(self._specs, self.n_capsules, self.ir_len, self._sample_rate, self.log2_block,
                                                      self.hop, self._win, self.fft_size, _lib(self._library)), _lib(self._library))
        return self._handle.ptr

    def emitter_parts(self) -> Optional[np.ndarray]:
        """al_batch.emitter_parts: per IR column, how many leading partitions can reach a block its event keeps (None: every
        partition of every IR).  pad_or_truncate_audio (synthesize.py:590) drops the convolution's tail from block n_blocks
        on, and partition p of an IR whose signal starts at block j_lo only feeds blocks >= j_lo + p.  Only the IRs of
        sliding-window moving events (al_event.reserved == 1: the one accumulate that honours it) get fewer than P; a
        column shared by several streams keeps the largest demand.  (csrc/al_plan.cpp: al_plan_emitter_parts.)"""
        if not len(self.events) or self.n_emitters <= 0:
            return None
        out = np.zeros(self.n_emitters, dtype=np.int32)
        have = _lib(self._library).call("al_plan_emitter_parts", self._c_plan(), out.ctypes.data)
        return out if have == 1 else None

    def max_nj_sliding(self) -> int:
        """Longest stream (in signal blocks) of the sliding-window moving events."""
        if not len(self.streams):
            return 0
        sliding = self.events["reserved"][self.streams["event"]] == 1
        return int(self.streams["n_j"][sliding].max()) if sliding.any() else 0

    @property
    def max_blocks(self) -> int:
        return int(self.events["n_blocks"].max()) if len(self.events) else 0

    @property
    def max_nj(self) -> int:
        return int(self.streams["n_j"].max()) if len(self.streams) else 0

    def chunks(self, chunk_events: Optional[int] = None) -> List[dict]:
        """Split the batch into runs of ``chunk_events`` consecutive events that share the global
        tables (al_batch.event0 / stream0 / emitter0 / *_block0, include/audiblelight_hip.h): al_plan_chunk."""
        n = len(self.events)
        step = n if not chunk_events or chunk_events <= 0 else int(chunk_events)
        if n == 0:
            return [dict(event0=0, n_events=0, stream0=0, n_streams=0, emitter0=0, n_emitters=0, xspec_block0=0, xspec_blocks=0,
                         yspec_block0=0, yspec_blocks=0, max_blocks=0, max_nj=0)]
        lib, handle, out = _lib(self._library), self._c_plan(), []
        for e0 in range(0, n, max(step, 1)):
            ch = _hip.AlChunk()
            lib.call("al_plan_chunk", handle, e0, min(step, n - e0), ct.byref(ch))
            out.append({name: int(getattr(ch, name)) for name, _ in _hip.AlChunk._fields_})
        return out

    def batch_flags(self, chunk: Optional[dict] = None) -> int:
        """al_batch.flags of a chunk (one of ``chunks()``; None = the whole plan): the library's dispatch policy --
        layout flags for the block size, accumulate flags for the chunk's event mix (al_plan_batch_flags; asked of the library
        that made this plan's handle)."""
        if not len(self.events):
            return 0
        flags = ct.c_int32(0)
        ch = None
        if chunk is not None:
            ch = _hip.AlChunk(**{name: int(chunk[name]) for name, _ in _hip.AlChunk._fields_})
        _lib(self._library).call("al_plan_batch_flags", self._c_plan(), ct.byref(ch) if ch is not None else None, ct.byref(flags))
        return int(flags.value)

    def workspace_bytes(self) -> int:
        """Bytes of the spectra workspaces + statistics of the batch as one chunk (al_workspace_bytes)."""
        b8 = self.block * 8
        return (self.hspec_blocks + self.xspec_blocks + self.yspec_blocks) * b8 + self.hspec_blocks * 4 \
            + self.n_emitters * 4 + self.n_partials * 16


def choose_log2_block(ir_len: int, max_clip: int, lib=None) -> int:
    """Largest block that keeps two workgroups resident per CU (B = 8192: 68 KiB of LDS each),
    shrunk for short inputs so the zero padding of the last block stays small (al_choose_log2_block)."""
    return int(_lib(lib).call("al_choose_log2_block", int(ir_len), int(max_clip)))


class _PlanHandle:
    """Owner of one ``al_plan *``: destroyed with the last BatchPlan that refers to it."""

    def __init__(self, ptr, lib):
        self.ptr, self.lib = ptr, lib

    def __del__(self):
        try:
            if self.ptr:
                self.lib.call("al_plan_destroy", self.ptr)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass
        self.ptr = None


def _create_c_plan(specs, n_capsules, ir_len, sample_rate, log2_block, hop, win, fft_size, lib):
    arr = (_hip.AlEventSpec * max(len(specs), 1))()
    for i, sp in enumerate(specs):
        arr[i] = _hip.AlEventSpec(n_samples=int(sp.n_samples), n_emitters=int(sp.n_emitters), emitter0=int(sp.emitter0),
                                  is_moving=1 if sp.is_moving else 0, snr=float(sp.snr), ref_db=float(sp.ref_db), gain=float(sp.gain),
                                  stft_len=int(sp.stft_len or 0), duration=float(sp.duration if sp.duration is not None else 0.0))
    handle = ct.c_void_p()
    try:
        lib.call("al_plan_create", arr, len(specs), int(n_capsules), int(ir_len), float(sample_rate), int(log2_block or 0), int(hop),
                    int(win), int(fft_size), ct.byref(handle))
    except _hip.HipError as exc:       # the reference's ValueErrors, with its messages (synthesize.py:565-584)
        raise ValueError(str(exc).split("): ", 1)[-1]) from None
    return handle


def plan_batch(specs: Sequence[EventSpec], n_capsules: int, ir_len: int, sample_rate: float,
               log2_block: Optional[int] = None, hop: int = config.HOP_SIZE, win: int = config.WIN_SIZE,
               fft_size: int = config.FFT_SIZE, lib=None) -> BatchPlan:
    """Build the kernel tables for a list of events sharing one IR tensor (one microphone): al_plan_create.
    ``lib``: the _hip.Library to plan with (a Renderer passes its own); default: the process-wide library."""
    if win != 2 * hop or fft_size < 2 * win - 1:
        raise ValueError("the HIP time-variant path needs win_size == 2*hop_size (sin^2 COLA) and fft_size >= 2*win_size-1")
    if log2_block is not None and not MIN_LOG2_BLOCK <= log2_block <= MAX_LOG2_BLOCK:
        raise ValueError(f"log2_block must be in [{MIN_LOG2_BLOCK}, {MAX_LOG2_BLOCK}]")
    specs = list(specs)
    library, lib = lib, _lib(lib)
    owner = _PlanHandle(_create_c_plan(specs, n_capsules, ir_len, sample_rate, log2_block, hop, win, fft_size, lib), lib)
    handle = owner.ptr
    info = _hip.AlPlanInfo()
    lib.call("al_plan_get_info", handle, ct.byref(info))
    events = _copy(lib.call("al_plan_events", handle), info.n_events, EVENT_DTYPE)
    streams = _copy(lib.call("al_plan_streams", handle), info.n_streams, STREAM_DTYPE)
    wtab = _copy(lib.call("al_plan_wtab", handle), info.wtab_floats, np.float32)
    offsets = _copy(lib.call("al_plan_audio_offsets", handle), info.n_events, np.int64)
    return BatchPlan(log2_block=info.log2_block, n_capsules=n_capsules, ir_len=ir_len, n_emitters=info.n_emitters,
                     events=events, streams=streams, wtab=wtab, audio_offsets=offsets, audio_floats=int(info.audio_floats),
                     spatial_floats=int(info.spatial_floats), xspec_blocks=int(info.xspec_blocks), yspec_blocks=int(info.yspec_blocks),
                     n_partials=int(info.n_partials), hop=hop, fft_size=fft_size, _specs=specs, _sample_rate=float(sample_rate), _win=win,
                     _handle=owner, _library=library)


# ----------------------------------------------------------------------------- mixdown
@dataclass
class MixPlan:
    n_capsules: int
    n_samples: int
    tile: int
    tile_ptr: np.ndarray
    tile_events: np.ndarray
    slot_src: np.ndarray
    slot_len: np.ndarray
    slot_start: np.ndarray
    slot_count: np.ndarray
    slot_rows: np.ndarray
    slot_event: np.ndarray
    skipped: List[int] = field(default_factory=list)

    @property
    def n_tiles(self) -> int:
        return -(-self.n_samples // self.tile)


def event_slot(scene_start: float, scene_end: float, sample_rate: float, n_scene: int):
    """Sample slot of an event in the scene timeline; Python round() like synthesize.py:361-362."""
    return max(0, round(scene_start * sample_rate)), min(round(scene_end * sample_rate), n_scene)


def plan_mixdown(starts: Sequence[float], ends: Sequence[float], lens: Sequence[int], rows: Sequence[int],
                 src_offsets: Sequence[int], event_index: Sequence[int], duration: float, sample_rate: float,
                 n_capsules: int, tile: int = 4096, lib=None) -> MixPlan:
    """Slots and per-tile event lists (events keep insertion order inside every tile): al_plan_mixdown."""
    n = len(starts)
    a = np.ascontiguousarray(starts, dtype=np.float64)
    b = np.ascontiguousarray(ends, dtype=np.float64)
    la = np.ascontiguousarray(lens, dtype=np.int32)
    rw = np.ascontiguousarray(rows, dtype=np.int32)
    so = np.ascontiguousarray(src_offsets, dtype=np.int64)
    ei = np.ascontiguousarray(event_index, dtype=np.int32)
    lib = _lib(lib)
    handle = ct.c_void_p()
    lib.call("al_plan_mixdown", a.ctypes.data, b.ctypes.data, la.ctypes.data, rw.ctypes.data, so.ctypes.data, ei.ctypes.data, n,
             float(duration), float(sample_rate), int(n_capsules), int(tile), ct.byref(handle))
    try:
        t = _hip.AlMixTables()
        lib.call("al_mix_plan_get", handle, ct.byref(t))
        ns = max(t.n_slots, 1)
        return MixPlan(n_capsules=t.n_capsules, n_samples=t.n_samples, tile=t.tile,
                       tile_ptr=_copy(t.tile_ptr, t.n_tiles + 1, np.int32), tile_events=_copy(t.tile_events, max(t.n_tile_events, 1), np.int32),
                       slot_src=_copy(t.slot_src, ns, np.int64), slot_len=_copy(t.slot_len, ns, np.int32),
                       slot_start=_copy(t.slot_start, ns, np.int32), slot_count=_copy(t.slot_count, ns, np.int32),
                       slot_rows=_copy(t.slot_rows, ns, np.int32), slot_event=_copy(t.slot_event, ns, np.int32),
                       skipped=[int(x) for x in _copy(t.skipped, t.n_skipped, np.int32)])
    finally:
        lib.call("al_mix_plan_destroy", handle)
