"""TEST INFRASTRUCTURE: the numpy planner of rounds 1-3, kept as the independent statement that the C planner behind the ABI
(audiblelight_amd/csrc/al_plan.cpp, called by audiblelight_amd/plan.py) is held to, table for table, on random scenes
(tests/test_host_logic.py).  The package never imports this module.

Host-side planning for the HIP synthesis path: pure index arithmetic, no audio arithmetic.

Turns "events of one microphone" (clip lengths, emitter columns, SNRs, trajectories) into the
tables the kernels read (``al_event`` / ``al_stream`` of include/audiblelight_hip.h), the
workspace sizes, and the mixdown slot/tile lists.  Everything here is cheap numpy on shapes;
the samples themselves never pass through this module.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from audiblelight_amd import config
from audiblelight_amd._hip import EVENT_DTYPE, MAX_LOG2_BLOCK, MIN_LOG2_BLOCK, STREAM_DTYPE


SPARSE_MAX_NJ = 6  # AL_SPARSE_MAX_NJ of include/audiblelight_hip.h
SPARSE_MAX_PARTITIONS = 24  # AL_SPARSE_MAX_PARTITIONS: beyond it flagged events go through the tile accumulate, which reads every partition


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def generate_interpolation_matrix(ir_times: np.ndarray, sr=config.SAMPLE_RATE, hop_size=config.HOP_SIZE,
                                  n_frames: Optional[int] = None) -> np.ndarray:
    """Linear cross-fade weights W[frame, ir] between consecutive IRs of a moving source.

    Same contract as the reference's ``generate_interpolation_matrix`` (synthesize.py:148-181):
    IR ``l`` starts at frame ``round((t_l*sr + hop)/hop)`` and fades linearly into IR ``l+1``.
    """
    first = np.round((np.asarray(ir_times, dtype=np.float64) * sr + hop_size) / hop_size)
    rows = int(first[-1]) if n_frames is None else int(n_frames)
    weights = np.zeros((rows, len(first)))
    for l in range(len(first) - 1):
        idx = np.arange(first[l], first[l + 1] + 1, dtype=int) - 1
        up = np.linspace(0.0, 1.0, len(idx))
        weights[idx, l] = 1.0 - up
        weights[idx, l + 1] = up
    return weights


def stft_frame_count(n_samples: int, hop_size: int = config.HOP_SIZE) -> int:
    """Frames the reference STFT produces for n_samples (synthesize.py:123)."""
    return 2 * int(np.ceil(n_samples / (2.0 * hop_size))) + 1


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

    @property
    def block(self) -> int:
        return 1 << self.log2_block

    @property
    def n_partitions(self) -> int:
        return -(-self.ir_len // self.block) if self.n_emitters else 0

    @property
    def hspec_blocks(self) -> int:
        return self.n_emitters * self.n_capsules * self.n_partitions

    def emitter_parts(self) -> Optional[np.ndarray]:
        """al_batch.emitter_parts: per IR column, how many leading partitions can reach a block its event keeps (None: every
        partition of every IR).  pad_or_truncate_audio (synthesize.py:590) drops the convolution's tail from block n_blocks
        on, and partition p of an IR whose signal starts at block j_lo only feeds blocks >= j_lo + p.  Only the IRs of
        sliding-window moving events (al_event.reserved == 1: the one accumulate that honours it) get fewer than P; a
        column shared by several streams keeps the largest demand."""
        P = self.n_partitions
        if (P <= 1 or P > SPARSE_MAX_PARTITIONS or not len(self.streams) or not len(self.events)
                or not (self.events["reserved"] == 1).any()):
            return None
        st, ev = self.streams, self.events
        real = (ev["n_streams"][st["event"]] > 0) & (st["emitter"] >= 0) & (st["emitter"] < self.n_emitters)
        reach = np.clip(ev["n_blocks"][st["event"]] - st["j_lo"], 0, P)
        want = np.where(ev["reserved"][st["event"]] == 1, np.where(st["n_j"] > 0, reach, 0), P).astype(np.int32)
        need = np.zeros(self.n_emitters, dtype=np.int32)
        np.maximum.at(need, st["emitter"][real], want[real])
        unused = np.ones(self.n_emitters, dtype=bool)
        unused[st["emitter"][real]] = False
        need[unused] = P
        return need if (need < P).any() else None

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
        tables (al_batch.event0 / stream0 / emitter0 / *_block0, include/audiblelight_hip.h)."""
        n = len(self.events)
        step = n if not chunk_events or chunk_events <= 0 else int(chunk_events)
        out = []
        for e0 in range(0, max(n, 1), max(step, 1)):
            e1 = min(e0 + step, n)
            ev = self.events[e0:e1]
            s0 = int(ev["stream0"][0]) if len(ev) else 0
            s1 = int(ev["stream0"][-1] + max(int(ev["n_streams"][-1]), 1)) if len(ev) else 0
            st = self.streams[s0:s1]
            conv = ev["n_streams"] > 0
            # streams of convolved events (pseudo-streams of zero-emitter events carry no spectra)
            is_conv = np.zeros(len(st), dtype=bool)
            for k in np.flatnonzero(conv):
                a = int(ev["stream0"][k]) - s0
                is_conv[a: a + int(ev["n_streams"][k])] = True
            cs = st[is_conv]
            em0 = int(cs["emitter"].min()) if len(cs) else 0
            em1 = int(cs["emitter"].max()) + 1 if len(cs) else 0
            x0 = int(cs["xspec_base"].min()) if len(cs) else 0
            x1 = int((cs["xspec_base"] + cs["n_j"]).max()) if len(cs) else 0
            y0 = int(ev["yspec_base"][conv].min()) if conv.any() else 0
            y1 = int((ev["yspec_base"][conv] + self.n_capsules * ev["n_blocks"][conv]).max()) if conv.any() else 0
            out.append(dict(event0=e0, n_events=e1 - e0, stream0=s0, n_streams=s1 - s0, emitter0=em0,
                            n_emitters=em1 - em0, xspec_block0=x0, xspec_blocks=x1 - x0, yspec_block0=y0,
                            yspec_blocks=y1 - y0, max_blocks=int(ev["n_blocks"].max()) if len(ev) else 0,
                            max_nj=int(st["n_j"].max()) if len(st) else 0))
        return out

    def workspace_bytes(self) -> int:
        b8 = self.block * 8
        return (self.hspec_blocks + self.xspec_blocks + self.yspec_blocks) * b8 + self.hspec_blocks * 4 \
            + self.n_emitters * 4 + self.n_partials * 16


def choose_log2_block(ir_len: int, max_clip: int) -> int:
    """Largest block that keeps two workgroups resident per CU (B = 8192: 68 KiB of LDS each),
    shrunk for short inputs so the zero padding of the last block stays small."""
    want = max(min(ir_len, max_clip), 1)
    lg = 13
    while lg > MIN_LOG2_BLOCK and (1 << lg) > 2 * want:
        lg -= 1
    return lg


def plan_batch(specs: Sequence[EventSpec], n_capsules: int, ir_len: int, sample_rate: float,
               log2_block: Optional[int] = None, hop: int = config.HOP_SIZE, win: int = config.WIN_SIZE,
               fft_size: int = config.FFT_SIZE) -> BatchPlan:
    """Build the kernel tables for a list of events sharing one IR tensor (one microphone)."""
    if win != 2 * hop or fft_size < 2 * win - 1:
        raise ValueError("the HIP time-variant path needs win_size == 2*hop_size (sin^2 COLA) and fft_size >= 2*win_size-1")
    if log2_block is None:
        log2_block = choose_log2_block(ir_len, max([s.n_samples for s in specs], default=1))
        # B = 16384 in the two situations al_plan_create (csrc/al_plan.cpp) switches to it
        p13 = -(-ir_len // 8192)
        any_moving = any(s.n_emitters > 1 for s in specs)
        if log2_block == 13 and not any_moving and 17 <= p13 <= 24:
            # (1) big batches of static events with 17..24 partitions of 8192
            if n_capsules * sum(-(-s.n_samples // 8192) for s in specs) >= 100000:
                log2_block = 14
        elif log2_block == 13 and any_moving:
            # (2) moving events that are off the sliding-window accumulate at B = 8192 but on it at B = 16384
            def all_sliding(lb):
                if -(-ir_len // (1 << lb)) > SPARSE_MAX_PARTITIONS:
                    return None
                pl = plan_batch(specs, n_capsules, ir_len, sample_rate, log2_block=lb, hop=hop, win=win, fft_size=fft_size)
                return pl if all(int(pl.events["reserved"][i]) == 1 for i, s in enumerate(specs) if s.n_emitters > 1) else None

            if all_sliding(13) is None and all_sliding(14) is not None:
                log2_block = 14
    if not MIN_LOG2_BLOCK <= log2_block <= MAX_LOG2_BLOCK:
        raise ValueError(f"log2_block must be in [{MIN_LOG2_BLOCK}, {MAX_LOG2_BLOCK}]")
    B = 1 << log2_block
    events = np.zeros(len(specs), dtype=EVENT_DTYPE)
    streams: List[tuple] = []
    wcols: List[np.ndarray] = []
    w_floats = 0
    audio_off = out_off = 0
    x_blocks = y_blocks = parts = 0
    n_emit_used = 0
    audio_offsets = np.zeros(len(specs), dtype=np.int64)
    for i, sp in enumerate(specs):
        La = int(sp.n_samples)
        if La <= 0:
            raise ValueError("event clip must have at least one sample")
        K = -(-La // B)
        ev = events[i]
        audio_offsets[i] = audio_off
        ev["audio_off"], ev["out_off"], ev["len"], ev["n_blocks"] = audio_off, out_off, La, K
        ev["snr"], ev["ref_db"] = sp.snr, sp.ref_db
        ev["stream0"], ev["yspec_base"], ev["part_base"] = len(streams), y_blocks, parts
        valid = La
        if sp.n_emitters == 0:
            ev["n_streams"] = 0
            streams.append((i, 0, 0, 0, 0, -1, 0, sp.gain))  # carries the gain only
        elif sp.n_emitters == 1:
            if sp.is_moving:
                raise ValueError("Moving Event has only one emitter!")
            ev["n_streams"] = 1
            streams.append((i, sp.emitter0, 0, K, x_blocks, -1, 0, sp.gain))
            x_blocks += K
        else:
            if not sp.is_moving:
                raise ValueError("Expected a moving event!")
            if sp.duration is None:
                raise ValueError("moving events need Event.duration")
            w = generate_interpolation_matrix(np.linspace(0, sp.duration, sp.n_emitters), sample_rate, hop)
            n_frames = min(stft_frame_count(sp.stft_len or La, hop), w.shape[0])
            valid = min(La, max(n_frames * hop - win, 0))
            ev["n_streams"] = sp.n_emitters
            first_stream = len(streams)
            for l in range(sp.n_emitters):
                col = w[:n_frames, l]
                nz = np.flatnonzero(col)
                if len(nz) == 0:
                    j_lo, n_j = 0, 0
                else:
                    t_lo = max(hop * (int(nz[0]) - 1), 0)
                    t_hi = min(hop * (int(nz[-1]) + 1), La)
                    j_lo = t_lo // B
                    j_hi = min(K - 1, -(-t_hi // B))
                    n_j = max(j_hi - j_lo + 1, 0) if t_hi > t_lo else 0
                streams.append((i, sp.emitter0 + l, j_lo, n_j, x_blocks, w_floats, n_frames, sp.gain * fft_size))
                x_blocks += n_j
                wcols.append(col.astype(np.float32))
                w_floats += n_frames
            mine = streams[first_stream:]
            starts_ok = all(a[2] <= b_[2] for a, b_ in zip(mine, mine[1:]) if a[3] > 0 and b_[3] > 0)
            if starts_ok and max(st_[3] for st_ in mine) <= SPARSE_MAX_NJ:
                ev["reserved"] = 1  # sliding-window accumulate (k_spectral_mac_moving)
        ev["valid_len"] = valid
        n_emit_used = max(n_emit_used, sp.emitter0 + sp.n_emitters)
        y_blocks += n_capsules * K if sp.n_emitters else 0
        parts += n_capsules * K
        audio_off += _round_up(La, 4)
        out_off += _round_up(n_capsules * La, 4)
    st = np.array(streams, dtype=STREAM_DTYPE) if streams else np.zeros(0, dtype=STREAM_DTYPE)
    wtab = np.concatenate(wcols) if wcols else np.zeros(1, dtype=np.float32)
    return BatchPlan(log2_block=log2_block, n_capsules=n_capsules, ir_len=ir_len, n_emitters=n_emit_used,
                     events=events, streams=st, wtab=wtab, audio_offsets=audio_offsets, audio_floats=max(audio_off, 4),
                     spatial_floats=max(out_off, 4), xspec_blocks=max(x_blocks, 1), yspec_blocks=max(y_blocks, 1),
                     n_partials=max(parts, 1), hop=hop, fft_size=fft_size)


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
                 n_capsules: int, tile: int = 4096) -> MixPlan:
    """Slots and per-tile event lists (events keep insertion order inside every tile)."""
    n_scene = round(duration * sample_rate)
    keep, skipped = [], []
    s_start, s_count = [], []
    for i, (t0, t1, la) in enumerate(zip(starts, ends, lens)):
        a, b = event_slot(t0, t1, sample_rate, n_scene)
        if b <= a:
            skipped.append(i)
            continue
        keep.append(i)
        s_start.append(a)
        s_count.append(min(b - a, int(la)))
    n_tiles = -(-n_scene // tile)
    lists: List[List[int]] = [[] for _ in range(n_tiles)]
    for slot, (a, cnt) in enumerate(zip(s_start, s_count)):
        for t in range(a // tile, min((a + cnt - 1) // tile, n_tiles - 1) + 1):
            lists[t].append(slot)
    ptr = np.zeros(n_tiles + 1, dtype=np.int32)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    flat = np.array([s for x in lists for s in x], dtype=np.int32) if ptr[-1] else np.zeros(1, dtype=np.int32)
    sel = np.array(keep, dtype=int)
    as_i32 = lambda seq: np.asarray(seq, dtype=np.int32)[sel] if len(sel) else np.zeros(1, dtype=np.int32)
    return MixPlan(n_capsules=n_capsules, n_samples=n_scene, tile=tile, tile_ptr=ptr, tile_events=flat,
                   slot_src=(np.asarray(src_offsets, dtype=np.int64)[sel] if len(sel) else np.zeros(1, dtype=np.int64)),
                   slot_len=as_i32(lens), slot_start=np.array(s_start or [0], dtype=np.int32),
                   slot_count=np.array(s_count or [0], dtype=np.int32), slot_rows=as_i32(rows),
                   slot_event=as_i32(event_index), skipped=skipped)
