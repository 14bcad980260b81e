"""Batch driver: keep one MI355X busy over many scenes.

The reference's dataset scripts render scenes in a serial loop (scripts/generate/benchmark.py:44-77,
scripts/seld/generate_dataset.py:96-260) and write one WAV per microphone (core.py:1840-1847).  Here the
stages of consecutive scenes overlap:

    scene i+1:  host planning + pinned staging + H2D copy      (copy stream)
    scene i  :  render + mixdown                                (compute stream)
    scene i-1:  D2H of scene.audio + WAV encode                 (download stream + writer thread)

Only plumbing lives here (torch streams/events, pinned buffers, a thread for file I/O); all arithmetic is
in the kernels behind the C ABI.
"""
from __future__ import annotations

import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np

from . import engine
from . import plan as planning


@dataclass
class SceneJob:
    """One microphone of one scene, described by arrays already in host memory."""
    specs: Sequence[planning.EventSpec]
    clips: Sequence[np.ndarray]
    irs: np.ndarray                 # (C, N_total, L) float32 or float64 (WorldState.get_irs() layout)
    starts: Sequence[float]
    ends: Sequence[float]
    duration: float
    sample_rate: int
    name: str = "scene"
    ambience: Sequence = ()         # [(device noise buffer, device scalar)] prepared by the caller, optional


@dataclass
class BatchReport:
    n_scenes: int = 0
    scene_seconds: float = 0.0
    wall_s: float = 0.0
    h2d_bytes: int = 0
    d2h_bytes: int = 0
    files: List[str] = field(default_factory=list)

    @property
    def scene_seconds_per_second(self) -> float:
        return self.scene_seconds / self.wall_s if self.wall_s > 0 else 0.0


class BatchDriver:
    """Pipelined rendering of a stream of SceneJobs on one GPU."""

    def __init__(self, renderer: Optional[engine.Renderer] = None, depth: int = 2):
        self.r = renderer or engine.Renderer()
        if not hasattr(self.r.mem, "torch"):
            raise RuntimeError("BatchDriver needs the torch/ROCm memory provider")
        self.torch = self.r.mem.torch
        self.depth = max(1, depth)
        dev = self.r.mem.device
        self.copy_stream = self.torch.cuda.Stream(device=dev)
        self.down_stream = self.torch.cuda.Stream(device=dev)
        self._pinned = {}   # (tag, dtype, numel) -> list of reusable pinned host tensors (page-locking is slow)
        workers = int(os.environ.get("AL_COPY_THREADS", "0")) or max(2, min(16, (os.cpu_count() or 2)))
        self._pool = ThreadPoolExecutor(max_workers=workers)

    def _parallel_copy(self, dst, src) -> None:
        """dst.copy_(src) split over the worker threads (a single thread moves ~8 GB/s, a scene's IRs are ~0.8 GB)."""
        n = dst.numel()
        parts = self._pool._max_workers
        step = -(-n // parts)
        if n < (1 << 22):
            dst.copy_(src)
            return
        futs = [self._pool.submit(lambda a, b: dst[a:b].copy_(src[a:b]), i, min(i + step, n)) for i in range(0, n, step)]
        for f in futs:
            f.result()

    def _pinned_buffer(self, tag: str, dtype, numel: int, slot: int):
        key = (tag, dtype, int(numel))
        pool = self._pinned.setdefault(key, [])
        while len(pool) <= slot:
            pool.append(self.torch.empty(int(numel), dtype=dtype).pin_memory())
        return pool[slot]

    # -- stage 1: host planning + asynchronous upload on the copy stream
    def _stage(self, job: SceneJob, slot: int = 0):
        torch, r = self.torch, self.r
        c, n, l = job.irs.shape
        pl = planning.plan_batch(job.specs, c, l, job.sample_rate)
        n_ev = len(job.clips)
        mix_plan = planning.plan_mixdown(job.starts, job.ends, [len(x) for x in job.clips], [c] * n_ev,
                                         pl.events["out_off"], list(range(n_ev)), job.duration, job.sample_rate, c)
        packed = torch.from_numpy(r.pack_audio(pl, job.clips))
        audio_host = self._pinned_buffer("audio", packed.dtype, packed.numel(), slot)
        audio_host.copy_(packed)
        flat = torch.from_numpy(np.ascontiguousarray(job.irs).reshape(-1))
        irs_host = self._pinned_buffer("irs", flat.dtype, flat.numel(), slot)
        self._parallel_copy(irs_host, flat)   # host memcpy into page-locked staging: the host-side cost of the boundary
        lp = (l + 3) // 4 * 4
        with torch.cuda.stream(self.copy_stream):
            irs_raw = irs_host.to(r.mem.device, non_blocking=True)
            if job.irs.dtype == np.float64 or lp != l:
                if job.irs.dtype != np.float64:
                    irs_raw = irs_raw.double()
                irs_dev = torch.empty(c * n * lp, dtype=torch.float32, device=r.mem.device)
                import ctypes as ct
                r.lib.call("al_pack_irs_f64", irs_raw.data_ptr(), irs_dev.data_ptr(), c * n, l, lp,
                           ct.c_void_p(self.copy_stream.cuda_stream))
            else:
                irs_dev = irs_raw
            audio_dev = audio_host.to(r.mem.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        h2d = irs_host.numel() * irs_host.element_size() + audio_host.numel() * 4
        return dict(job=job, plan=pl, mix=mix_plan, irs=irs_dev, strides=(n * lp, lp), audio=audio_dev, ready=ready,
                    keep=(audio_host, irs_host, irs_raw), h2d=h2d, slot=slot)

    # -- stage 2: kernels on the current (compute) stream
    def _render(self, st):
        torch, r = self.torch, self.r
        cur = torch.cuda.current_stream(r.mem.device)
        cur.wait_event(st["ready"])
        batch = r.prepare(st["plan"], st["job"].clips, st["irs"], st["strides"], audio_dev=st["audio"])
        res = batch.run()
        scene = r.prepare_mixdown(st["mix"], res, st["job"].ambience).run()
        done = torch.cuda.Event()
        done.record(cur)
        st.update(result=res, scene=scene, done=done, batch=batch)
        return st

    # -- stage 3: D2H on the download stream into pinned memory
    def _download(self, st):
        torch = self.torch
        c, t = st["mix"].n_capsules, st["mix"].n_samples
        host = self._pinned_buffer("scene", torch.float32, c * t, st.get("slot", 0))
        with torch.cuda.stream(self.down_stream):
            self.down_stream.wait_event(st["done"])
            host.copy_(st["scene"][: c * t], non_blocking=True)
            landed = torch.cuda.Event()
            landed.record(self.down_stream)
        st.update(host=host, landed=landed)
        return st

    def run(self, jobs: Iterable[SceneJob], output_dir: Optional[str] = None,
            on_scene: Optional[Callable[[str, np.ndarray], None]] = None, check_finite: bool = True) -> BatchReport:
        """Render all jobs; per scene either write ``<output_dir>/<name>.wav`` (float32, (T, C) interleaved like
        soundfile.write(audio.T) in core.py:1840-1847) or hand the (C, T) array to ``on_scene``."""
        rep = BatchReport()
        sink: "queue.Queue" = queue.Queue(maxsize=self.depth + 1)

        def writer():
            while True:
                item = sink.get()
                if item is None:
                    return
                st = item
                st["landed"].synchronize()
                if check_finite:
                    st["result"].check_finite()
                c, t = st["mix"].n_capsules, st["mix"].n_samples
                arr = st["host"].numpy().reshape(c, t)
                if output_dir is not None:
                    from scipy.io import wavfile

                    path = os.path.join(output_dir, f"{st['job'].name}.wav")
                    wavfile.write(path, st["job"].sample_rate, arr.T)
                    rep.files.append(path)
                if on_scene is not None:
                    on_scene(st["job"].name, arr)
                rep.d2h_bytes += arr.nbytes

        if output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
        th = threading.Thread(target=writer, daemon=True)
        th.start()
        t0 = time.perf_counter()
        staged = None
        it = iter(jobs)
        nxt = next(it, None)
        n_slots = self.depth + 2   # staging slots in flight: being filled, being rendered, being written
        index = 0
        if nxt is not None:
            staged = self._stage(nxt, index % n_slots)
        while staged is not None:
            cur = staged
            nxt = next(it, None)
            index += 1
            # upload of scene i+1 overlaps the render of scene i; a slot is reused only after its scene was written
            while nxt is not None and sink.qsize() > self.depth:
                time.sleep(0.0005)
            staged = self._stage(nxt, index % n_slots) if nxt is not None else None
            st = self._download(self._render(cur))
            rep.n_scenes += 1
            rep.scene_seconds += cur["job"].duration
            rep.h2d_bytes += cur["h2d"]
            sink.put(st)
        sink.put(None)
        th.join()
        self.torch.cuda.synchronize(self.r.mem.device)
        rep.wall_s = time.perf_counter() - t0
        return rep


# ----------------------------------------------------------------------------- many small scenes in one launch sequence
def merge_jobs(jobs: Sequence[SceneJob]):
    """Concatenate several scenes that share (C, Lir, sample rate) into ONE event batch (cfg4: BASELINE configs[3]).

    Small scenes are launch-bound when rendered one by one; their events are independent, so they can share one
    launch sequence: clips are concatenated, IR tensors are concatenated along the emitter axis and every event's
    ``emitter0`` is offset.  Returns (specs, clips, irs, event_ranges) with event_ranges[i] = (first, count) of scene i.
    """
    if not jobs:
        raise ValueError("no jobs")
    c, _, l = jobs[0].irs.shape
    sr = jobs[0].sample_rate
    specs, clips, ranges, col = [], [], [], 0
    for job in jobs:
        if job.irs.shape[0] != c or job.irs.shape[2] != l or job.sample_rate != sr:
            raise ValueError("merge_jobs needs equal capsule count, IR length and sample rate")
        ranges.append((len(specs), len(job.specs)))
        for sp, clip in zip(job.specs, job.clips):
            specs.append(planning.EventSpec(n_samples=sp.n_samples, n_emitters=sp.n_emitters, snr=sp.snr,
                                            emitter0=col + sp.emitter0, is_moving=sp.is_moving, duration=sp.duration,
                                            gain=sp.gain, ref_db=sp.ref_db, stft_len=sp.stft_len))
            clips.append(clip)
        col += job.irs.shape[1]
    irs = np.concatenate([np.asarray(j.irs, dtype=np.float32) for j in jobs], axis=1)
    return specs, clips, irs, ranges


def render_merged(renderer: engine.Renderer, jobs: Sequence[SceneJob]) -> List[np.ndarray]:
    """Render the scenes of ``merge_jobs`` with one render launch sequence and one mixdown launch per scene."""
    specs, clips, irs, ranges = merge_jobs(jobs)
    c, l = irs.shape[0], irs.shape[2]
    pl = planning.plan_batch(specs, c, l, jobs[0].sample_rate)
    res = renderer.render(pl, clips, irs)
    res.check_finite()
    out = []
    for job, (e0, n) in zip(jobs, ranges):
        mix = planning.plan_mixdown(job.starts, job.ends, [len(x) for x in job.clips], [c] * n,
                                    pl.events["out_off"][e0: e0 + n], list(range(e0, e0 + n)), job.duration,
                                    job.sample_rate, c)
        dev = renderer.mixdown(mix, res, job.ambience)
        out.append(renderer.mem.download(dev)[: c * mix.n_samples].reshape(c, mix.n_samples))
    return out
