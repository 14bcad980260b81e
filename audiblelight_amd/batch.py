"""Batch driver: keep one MI355X busy over many scenes.

The reference's dataset scripts render scenes in a serial loop (scripts/generate/benchmark.py:44-77,
scripts/seld/generate_dataset.py:96-260) and write one WAV per microphone (core.py:1840-1847).  Here the
stages of consecutive scenes overlap:

    scene i+2:  host planning, clips packed into page-locked memory   (planner thread)
    scene i+1:  H2D copy of clips and IRs                              (uploader thread, copy stream; PCIe-bound)
    scene i  :  render + mixdown + frame encoding into host memory     (calling thread, compute stream)
    scene i-1:  WAV files / callbacks                                  (writer threads)

Only plumbing lives here (torch streams/events, pinned buffers, threads for staging and file I/O); all arithmetic
is in the kernels behind the C ABI.
"""
from __future__ import annotations

import os
import queue
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np

from . import _hip, engine, switches
from . import plan as planning


@dataclass
class SceneJob:
    """One microphone of one scene, described by arrays already in host memory."""
    specs: Sequence[planning.EventSpec]
    clips: Sequence                 # float32 arrays or engine.ClipSource (device-resident / folded-FX clips)
    irs: np.ndarray                 # (C, N_total, L) float32 or float64 (WorldState.get_irs() layout)
    starts: Sequence[float]
    ends: Sequence[float]
    duration: float
    sample_rate: int
    name: str = "scene"
    ambience: Sequence = ()         # [(device noise (C * T floats), device float32[C] per-capsule multipliers)] prepared by the
                                    # caller, optional (a 1-element multiplier buffer is broadcast over the capsules)


@dataclass
class BatchReport:
    n_scenes: int = 0
    scene_seconds: float = 0.0
    wall_s: float = 0.0
    h2d_bytes: int = 0
    d2h_bytes: int = 0
    files: List[str] = field(default_factory=list)
    skipped: List[str] = field(default_factory=list)     # scenes left alone because their output exists
    latencies: dict = field(default_factory=dict)        # scene name -> seconds from staging to written / delivered
    host_s: dict = field(default_factory=dict)           # seconds by stage: plan / upload (feeder threads), wait_upload / enqueue / wait_writer (caller)

    @property
    def scene_seconds_per_second(self) -> float:
        return self.scene_seconds / self.wall_s if self.wall_s > 0 else 0.0


SUBTYPES = {"PCM_16": (_hip.FRAMES_PCM16, np.int16), "FLOAT": (_hip.FRAMES_F32, np.float32)}


class BatchDriver:
    """Pipelined rendering of a stream of SceneJobs on one GPU.

    Staging (profiles/r02_h2d_probe.txt): the caller's IR tensor goes to HBM straight from where it lies -- the runtime
    moves pageable memory at PCIe rate (56 GB/s), an extra pass through page-locked staging only adds host time; float64
    and odd row pitches are fixed up by a device kernel.  Clips (6 % of the bytes) are packed into a reused page-locked
    buffer.  The scene leaves as the file payload: ``al_encode_frames`` interleaves (and, for PCM_16, quantises) on the
    device, the download stream copies that into page-locked memory and a writer thread puts it on disk.
    """

    def __init__(self, renderer: Optional[engine.Renderer] = None, depth: int = 2, subtype: str = "PCM_16",
                 writers: int = 4):
        self.r = renderer or engine.Renderer()
        if not hasattr(self.r.mem, "torch"):
            raise RuntimeError("BatchDriver needs the torch/ROCm memory provider")
        if subtype not in SUBTYPES:
            raise ValueError(f"subtype must be one of {sorted(SUBTYPES)}")
        self.torch = self.r.mem.torch
        self.depth = max(1, depth)
        self.writers = max(1, writers)   # threads putting finished scenes on disk (one WAV of a 60 s / 32-capsule scene
        self.subtype = subtype           # takes longer to write than the scene takes to upload, render and download)
        dev = self.r.mem.device
        self.copy_stream = self.torch.cuda.Stream(device=dev)
        self.down_stream = self.torch.cuda.Stream(device=dev)
        self._pinned = {}   # (tag, dtype) -> {slot: reusable page-locked host tensor}
        self._slot_ready = {}   # slot -> H2D event of the scene that last used the slot's pinned clip buffer
        # default: the runtime's pageable copy (holds this thread); AL_H2D=async: page-lock in place + asynchronous DMA
        # AL_D2H=kernel: the encode / copy kernels store straight into page-locked host memory instead of a DMA on the
        # download stream; slower once uploads run from their own thread (the stores slow the concurrent H2D: 17.4 vs
        # 15.5 ms per cfg2 scene, profiles/r02_e2e_probe.txt)
        sw = switches.current()
        self.zero_copy_d2h = sw.d2h == "kernel"
        self.async_h2d = sw.h2d == "async"   # same rate measured (profiles/r02_e2e_probe.txt)
        # threads that cast float64 IR tensors to float32 in the planner stage (0: upload the float64 bytes, cast on the device)
        # 16: 17.2-19.1 ms per cfg2 scene against 19.5-27.8 with 8 and 16.0 for float32 IRs (profiles/r03i_f64_threads_ab.txt)
        self.cast_threads = engine.host_threads(sw.convert_threads or 16) if sw.f64_upload == "host" else 0
        self._cast_pool = None

    def _pinned_buffer(self, tag: str, dtype, numel: int, slot: int):
        """Reused page-locked host buffer of at least ``numel`` elements (a view of exactly ``numel``).  Page-locking costs
        tens of milliseconds, so buffers are kept per (tag, slot) and only ever grow, by at least a quarter: scenes of a
        dataset rarely have the same total clip length twice."""
        key = (tag, dtype)
        pool = self._pinned.setdefault(key, {})
        buf = pool.get(slot)
        if buf is None or buf.numel() < numel:
            cap = max(int(numel), 0 if buf is None else buf.numel() * 5 // 4)
            cap = (cap + 65535) // 65536 * 65536
            buf = pool[slot] = self.torch.empty(cap, dtype=dtype).pin_memory()
        return buf[: int(numel)]

    # -- stage 1a: host planning + clip packing into page-locked memory (CPU only)
    def _plan(self, job: SceneJob, slot: int = 0):
        torch, r = self.torch, self.r
        c, n, l = job.irs.shape
        pl = planning.plan_batch(job.specs, c, l, job.sample_rate, lib=r.lib)
        n_ev = len(job.clips)
        mix_plan = planning.plan_mixdown(job.starts, job.ends, [len(x) for x in job.clips], [c] * n_ev,
                                         pl.events["out_off"], list(range(n_ev)), job.duration, job.sample_rate, c, lib=r.lib)
        prev = self._slot_ready.get(slot)
        if prev is not None:
            prev.synchronize()   # the slot's pinned clip buffer may still be the source of an H2D copy in flight
        audio_host = self._pinned_buffer("audio", torch.float32, pl.audio_floats, slot)
        r.pack_audio(pl, job.clips, out=audio_host.numpy())
        st = dict(job=job, plan=pl, mix=mix_plan, audio_host=audio_host, slot=slot, t0=time.perf_counter(),
                  h2d=job.irs.nbytes + pl.audio_floats * 4)
        if job.irs.dtype == np.float64 and self.cast_threads > 0 and job.irs.size:
            # float64 IRs (the reference's get_irs() dtype): cast HERE, in the planner thread's time, by a pool of threads into
            # the slot's page-locked float32 buffer; the uploader then moves half the bytes, as one asynchronous DMA.  The cast
            # (11 ms for cfg2 alone, more beside the DMAs) runs beside the upload of the previous scene (14 ms) instead of doubling it.
            if self._cast_pool is None:
                from concurrent.futures import ThreadPoolExecutor

                self._cast_pool = ThreadPoolExecutor(self.cast_threads)
            flat = np.ascontiguousarray(job.irs).reshape(-1)
            host = self._pinned_buffer("irs", torch.float32, flat.size, slot)
            view = host.numpy()
            cut = np.linspace(0, flat.size, self.cast_threads + 1).astype(np.int64)
            list(self._cast_pool.map(lambda i: np.copyto(view[cut[i]: cut[i + 1]], flat[cut[i]: cut[i + 1]], casting="same_kind"),
                                     range(self.cast_threads)))
            st["irs_host_f32"] = host
            st["h2d"] = flat.size * 4 + pl.audio_floats * 4
        return st

    # -- stage 1b: H2D on the copy stream (PCIe-bound; the pageable IR copy holds the calling thread, not the GIL)
    def _upload(self, st, compute_stream=None):
        torch, r = self.torch, self.r
        job, pl, audio_host = st["job"], st["plan"], st["audio_host"]
        if compute_stream is None:
            compute_stream = torch.cuda.current_stream(r.mem.device)
        device_clips = [(int(off), engine.as_clip_source(clip)) for off, clip in zip(pl.audio_offsets, job.clips)
                        if engine.as_clip_source(clip).host is None]
        if device_clips:   # clips already in HBM were produced by FX kernels on the compute stream: order the copies behind them
            self.copy_stream.wait_stream(compute_stream)
        with torch.cuda.stream(self.copy_stream):
            audio_dev = audio_host.to(r.mem.device, non_blocking=True)
            for off, src in device_clips:
                audio_dev[off: off + len(src)] = src.device[: len(src)]
            release = [] if self.async_h2d else None
            if "irs_host_f32" in st:                     # cast by the planner: one asynchronous DMA out of page-locked memory
                c, n, l = job.irs.shape
                lp = (l + 3) // 4 * 4
                raw = st["irs_host_f32"].to(r.mem.device, non_blocking=True)
                if lp == l:
                    irs_dev, strides = raw, (n * lp, lp)
                else:
                    import ctypes as ct

                    irs_dev, strides = r.mem.empty(c * n * lp), (n * lp, lp)
                    r.lib.call("al_pack_irs_f32", r.mem.ptr(raw), r.mem.ptr(irs_dev), c * n, l, lp,
                               ct.c_void_p(self.copy_stream.cuda_stream))
                    raw.record_stream(self.copy_stream)
            else:
                irs_dev, strides = r.upload_irs(job.irs, async_release=release, host_cast=False)   # straight from the caller's memory
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self._slot_ready[st["slot"]] = ready
        st.update(irs=irs_dev, strides=strides, audio=audio_dev, ready=ready, release=release or [])
        return st

    def _stage(self, job: SceneJob, slot: int = 0):
        return self._upload(self._plan(job, slot))

    # -- stage 2: kernels on the current (compute) stream
    def _render(self, st):
        torch, r = self.torch, self.r
        cur = torch.cuda.current_stream(r.mem.device)
        cur.wait_event(st["ready"])
        # tensors made on the copy stream are used here: tell the allocator, or it may hand them out again too early
        st["irs"].record_stream(cur)
        st["audio"].record_stream(cur)
        batch = r.prepare(st["plan"], st["job"].clips, st["irs"], st["strides"], audio_dev=st["audio"])
        # every table upload (small blocking copies) BEFORE the first launch: a blocking copy behind the render kernels
        # on the same stream would hold this thread for the whole render
        mix = r.prepare_mixdown(st["mix"], batch.result(), st["job"].ambience)
        res = batch.run()
        scene = mix.run()
        st.update(result=res, scene=scene, batch=batch)
        return st

    # -- stage 3: encode on the device, D2H on the download stream into pinned memory
    def _download(self, st, want_frames: bool, want_scene: bool, subtype: str):
        import ctypes as ct

        torch, r = self.torch, self.r
        c, t = st["mix"].n_capsules, st["mix"].n_samples
        cur = torch.cuda.current_stream(r.mem.device)
        fmt, np_dtype = SUBTYPES[subtype]
        frames_dev = None
        if want_frames and not self.zero_copy_d2h:
            tdt = torch.int16 if np_dtype == np.int16 else torch.float32
            frames_dev = torch.empty(c * t, dtype=tdt, device=r.mem.device)
            r.lib.call("al_encode_frames", st["scene"].data_ptr(), c, t, fmt, frames_dev.data_ptr(),
                       ct.c_void_p(cur.cuda_stream))
        if self.zero_copy_d2h:
            # The kernels write straight into page-locked HOST memory (it is mapped into the GPU's address space): the
            # "download" is then PCIe stores issued by the CUs (16 bytes per lane).
            if want_frames:
                tdt = torch.int16 if np_dtype == np.int16 else torch.float32
                host = self._pinned_buffer("frames", tdt, c * t, st.get("out_slot", st["slot"]))
                r.lib.call("al_encode_frames", st["scene"].data_ptr(), c, t, fmt, host.data_ptr(), ct.c_void_p(cur.cuda_stream))
                st["frames"] = host
            if want_scene:
                host = self._pinned_buffer("scene", torch.float32, c * t, st.get("out_slot", st["slot"]))
                r.lib.call("al_wrap_copy", st["scene"].data_ptr(), c * t, host.data_ptr(), c * t, ct.c_void_p(cur.cuda_stream))
                st["host"] = host
            landed = torch.cuda.Event()
            landed.record(cur)
            st.update(landed=landed)
            return st
        done = torch.cuda.Event()
        done.record(cur)
        with torch.cuda.stream(self.down_stream):
            self.down_stream.wait_event(done)
            if want_frames:
                frames_dev.record_stream(self.down_stream)
                host = self._pinned_buffer("frames", frames_dev.dtype, c * t, st.get("out_slot", st["slot"]))
                host.copy_(frames_dev, non_blocking=True)
                st["frames"] = host
            if want_scene:
                st["scene"].record_stream(self.down_stream)
                host = self._pinned_buffer("scene", torch.float32, c * t, st.get("out_slot", st["slot"]))
                host.copy_(st["scene"][: c * t], non_blocking=True)
                st["host"] = host
            landed = torch.cuda.Event()
            landed.record(self.down_stream)
        st.update(landed=landed)
        return st

    def run(self, jobs: Iterable[SceneJob], output_dir: Optional[str] = None,
            on_scene: Optional[Callable[[str, np.ndarray], None]] = None, check_finite: bool = True,
            subtype: Optional[str] = None, skip_existing: bool = False,
            path_of: Optional[Callable[[SceneJob], str]] = None, copy_for_callback: bool = True,
            after_write: Optional[Callable[[SceneJob, float], None]] = None) -> BatchReport:
        """Render all jobs.  With ``output_dir`` every scene is written to ``<output_dir>/<name>.wav`` (or
        ``path_of(job)``): (T, C) interleaved frames like ``soundfile.write(audio.T, sr)`` (core.py:1840-1847), subtype
        ``PCM_16`` (soundfile's default for WAV) or ``FLOAT``; ``skip_existing`` leaves scenes whose file exists alone
        (scripts/generate/benchmark.py:54-55).  ``on_scene(name, array)`` receives a (C, T) float32 array of its own
        (a copy: the page-locked download buffers go back to a free list when the writer is done with them and are reused by
        later scenes; ``copy_for_callback=False`` hands out the view instead, for callbacks that consume it before returning).
        ``after_write(job, latency_s)`` runs in the writer thread once the job's file is on disk / its callback has returned
        (``render_dataset`` writes a scene's metadata from it).  A failure in the writer thread (non-finite audio, disk
        error, callback error) stops the run and is re-raised here."""
        subtype = subtype or self.subtype
        if subtype not in SUBTYPES:
            raise ValueError(f"subtype must be one of {sorted(SUBTYPES)}")
        rep = BatchReport()
        sink: "queue.Queue" = queue.Queue(maxsize=self.depth + 1)
        failure: List[BaseException] = []
        book, callback_lock = threading.Lock(), threading.Lock()
        written: List[tuple] = []
        want_frames, want_scene = output_dir is not None, on_scene is not None
        if path_of is None:
            path_of = lambda job: os.path.join(output_dir, f"{job.name}.wav")  # noqa: E731

        def write_one(st):
            try:
                write_one_locked_buffers(st)
            finally:
                free_out.put(st["out_slot"])   # only now may a later scene's D2H copy land in this page-locked buffer

        def write_one_locked_buffers(st):
            st["landed"].synchronize()
            for release in st.get("release", ()):   # the scene is out, so its H2D copy is long done: unpin the caller's IRs
                release()
            if check_finite:
                st["result"].check_finite()
            c, t = st["mix"].n_capsules, st["mix"].n_samples
            if want_frames:
                from scipy.io import wavfile

                path = path_of(st["job"])
                os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
                wavfile.write(path, int(st["job"].sample_rate), st["frames"].numpy().reshape(t, c))
                with book:
                    written.append((st["index"], path))
                    rep.d2h_bytes += st["frames"].numel() * st["frames"].element_size()
            if want_scene:
                view = st["host"].numpy().reshape(c, t)
                with callback_lock:   # callbacks never run concurrently, whatever the number of writer threads
                    on_scene(st["job"].name, view.copy() if copy_for_callback else view)
                with book:
                    rep.d2h_bytes += c * t * 4
            latency = time.perf_counter() - st["t0"]
            with book:
                rep.latencies[st["job"].name] = latency
            if after_write is not None:
                after_write(st["job"], latency)

        def writer():
            while True:
                item = sink.get()
                if item is None:
                    return
                if failure:
                    free_out.put(item["out_slot"])
                    continue          # keep draining so the producer never blocks on a full queue
                try:
                    write_one(item)
                except BaseException as exc:  # noqa: BLE001  (re-raised by run())
                    failure.append(exc)

        if output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
        threads = [threading.Thread(target=writer, daemon=True) for _ in range(self.writers if want_frames else 1)]
        for th in threads:
            th.start()
        t0 = time.perf_counter()
        # Page-locked output buffers are handed out from a FREE LIST and come back when the writer thread that holds one has
        # finished with it (writers complete out of order: a count bound alone does not say WHICH buffers are still being
        # read by a slow `wavfile.write`).  Enough for: one being rendered, `depth` + 1 queued, one per writer thread.
        # Clip staging buffers are guarded by their H2D event.
        out_slots, in_slots = self.depth + 2 + len(threads), 4
        free_out: "queue.Queue" = queue.Queue()
        for slot in range(out_slots):
            free_out.put(slot)
        compute = self.torch.cuda.current_stream(self.r.mem.device)
        planned: "queue.Queue" = queue.Queue(maxsize=1)
        uploaded: "queue.Queue" = queue.Queue(maxsize=1)
        stop = threading.Event()
        clock = {}
        source_failure: List[BaseException] = []

        def spend(key, dt):
            with book:
                clock[key] = clock.get(key, 0.0) + dt

        def hand_over(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def planner():          # the jobs iterable is consumed HERE: scene factories overlap with upload and render too
            try:
                index = 0
                for job in jobs:
                    if stop.is_set():
                        break
                    if skip_existing and want_frames and os.path.exists(path_of(job)):
                        rep.skipped.append(job.name)
                        continue
                    ta = time.perf_counter()
                    st = self._plan(job, index % in_slots)
                    st.update(index=index)
                    index += 1
                    spend("plan", time.perf_counter() - ta)
                    if not hand_over(planned, st):
                        break
            except BaseException as exc:  # noqa: BLE001  (re-raised by run() once the scenes already handed over are out)
                source_failure.append(exc)
            finally:
                hand_over(planned, None)

        def take(q):
            while not stop.is_set():
                try:
                    return q.get(timeout=0.05)
                except queue.Empty:
                    pass
            return None

        def uploader():
            try:
                while True:
                    st = take(planned)
                    if st is None:
                        break
                    ta = time.perf_counter()
                    self._upload(st, compute)
                    spend("upload", time.perf_counter() - ta)
                    if not hand_over(uploaded, st):
                        break
            except BaseException as exc:  # noqa: BLE001
                failure.append(exc)
                stop.set()
            finally:
                hand_over(uploaded, None)

        feeders = [threading.Thread(target=planner, daemon=True), threading.Thread(target=uploader, daemon=True)]
        for th in feeders:
            th.start()
        try:
            while not failure:
                ta = time.perf_counter()
                cur = take(uploaded)
                if cur is None:
                    break
                tb = time.perf_counter()
                cur["out_slot"] = None
                while cur["out_slot"] is None and not failure:   # a free page-locked output buffer (back-pressure on slow writers)
                    try:
                        cur["out_slot"] = free_out.get(timeout=0.05)
                    except queue.Empty:
                        pass
                if cur["out_slot"] is None:
                    break
                st = self._download(self._render(cur), want_frames, want_scene, subtype)   # enqueue only (asynchronous)
                tc = time.perf_counter()
                rep.n_scenes += 1
                rep.scene_seconds += cur["job"].duration
                rep.h2d_bytes += cur["h2d"]
                sink.put(st)          # blocks while the writers are `depth` + 1 scenes behind: back-pressure
                td = time.perf_counter()
                for key, dt in (("wait_upload", tb - ta), ("enqueue", tc - tb), ("wait_writer", td - tc)):
                    spend(key, dt)
        finally:
            stop.set()                # on failure: the feeders stop waiting on their queues
            while True:               # ... and whatever they had queued is dropped
                try:
                    uploaded.get_nowait()
                except queue.Empty:
                    break
            for th in feeders:
                th.join()
            for _ in threads:
                sink.put(None)
            for th in threads:
                th.join()
        self.torch.cuda.synchronize(self.r.mem.device)
        rep.wall_s = time.perf_counter() - t0
        rep.host_s.update(clock)
        rep.files.extend(path for _, path in sorted(written))   # job order, whichever writer thread finished first
        if failure or source_failure:
            raise (failure or source_failure)[0]
        return rep


# ----------------------------------------------------------------------------- dataset loop
def scene_jobs(scene, name: str, renderer: Optional[engine.Renderer] = None) -> List[SceneJob]:
    """One SceneJob per microphone of a Scene-like object (the attribute list of SURVEY.md 8a A15): what
    ``render_audio_for_all_scene_events`` + ``generate_scene_audio_from_events`` read from it
    (synthesize.py:613-677, 314-401), packaged for the pipelined driver."""
    from . import synthesize

    synthesize.validate_scene(scene)
    jobs = []
    n_scene = round(scene.duration * scene.sample_rate)
    for mic_alias, mic_ir in scene.state.get_irs().items():
        specs, clips, starts, ends, col = [], [], [], [], 0
        for event in scene.events.values():
            clip = synthesize._clip_of(event, False)
            specs.append(synthesize._spec_of(event, clip, len(event), col, scene.ref_db))
            clips.append(clip)
            starts.append(event.scene_start)
            ends.append(event.scene_end)
            col += len(event)
        ambience = []
        if len(getattr(scene, "ambience", {})) > 0:
            r = renderer or synthesize.get_renderer()
            ambience = [synthesize._ambience_on_device(r, a, (mic_ir.shape[0], n_scene)) for a in scene.ambience.values()]
        jobs.append(SceneJob(specs=specs, clips=clips, irs=np.asarray(mic_ir), starts=starts, ends=ends,
                             duration=scene.duration, sample_rate=scene.sample_rate, name=f"{name}/{mic_alias}",
                             ambience=ambience))
    return jobs


def render_dataset(scenes: Iterable, output_dir: str, skip_existing: bool = True, subtype: str = "PCM_16",
                   audio_fname: str = "audio_out", metadata_fname: str = "metadata_out", metadata_json: bool = True,
                   driver: Optional[BatchDriver] = None, rank: int = 0, world_size: int = 1,
                   metadata_dcase: bool = False) -> BatchReport:
    """The reference's dataset loops on the pipelined driver (scripts/generate/benchmark.py:35-82,
    scripts/generate/generate_with_random_events.py:222-238, scripts/seld/generate_dataset.py:96-260).

    ``scenes`` yields ``(name, scene)`` or ``(name, factory)`` pairs (a zero-argument factory is only called for
    scenes that are actually rendered).  Layout, as ``make_a_scene`` leaves it: ``<output_dir>/<name>/
    <audio_fname>_<mic>.wav`` ((T, C) frames, soundfile's default PCM_16 unless ``subtype="FLOAT"``) and
    ``<metadata_fname>.json`` = ``scene.to_dict()`` plus ``"time"`` (seconds from staging to the file on disk).
    ``skip_existing``: a scene whose folder holds its metadata file (written last, from the writer path, as soon as the
    scene's last microphone is on disk) is left alone (benchmark.py:54-55); a folder an interrupted run left without it is
    rendered again.  ``metadata_dcase``: also
    ``<metadata_fname>_<mic>.csv`` per microphone (``metadata.generate_dcase2024_metadata``; the events need class indices
    and emitter positions in their metadata).

    ``rank`` / ``world_size``: one process per GPU, every process handed the SAME scene stream; process ``rank`` renders
    scenes ``rank, rank + world_size, ...`` of it (``distributed.shard_stream``) into the shared ``output_dir``.  Scenes
    are independent, so there is no collective and no factory of another rank's scene is ever called."""
    import json

    from . import distributed

    scenes = distributed.shard_stream(scenes, rank, world_size)

    driver = driver or BatchDriver(subtype=subtype)
    os.makedirs(output_dir, exist_ok=True)
    meta = {}
    meta_lock = threading.Lock()
    skipped = []

    def finish_scene(name):
        """Metadata of one scene, written from the writer path the moment its last microphone is on disk (the reference
        writes it per scene inside make_a_scene): an interrupted run leaves complete scenes complete, and the JSON is the
        completion marker ``skip_existing`` looks for."""
        with meta_lock:
            d, latency, frames = meta[name]["dict"], meta[name]["latency"], meta[name]["dcase"]
            meta[name] = None        # only the small bookkeeping above outlives the scene: drop it too
        for mic, df in (frames or {}).items():
            df.to_csv(os.path.join(output_dir, name, f"{metadata_fname}_{mic}.csv"), sep=",", encoding="utf-8", header=None)
        if not metadata_json:
            return
        d["time"] = latency
        tmp = os.path.join(output_dir, name, f".{metadata_fname}.json.tmp")
        with open(tmp, "w") as fh:
            json.dump(d, fh, indent=4, ensure_ascii=False)
        os.replace(tmp, os.path.join(output_dir, name, f"{metadata_fname}.json"))

    def after_write(job, latency):
        name = job.name.rsplit("/", 1)[0]
        with meta_lock:
            entry = meta[name]
            entry["left"] -= 1
            entry["latency"] = max(entry["latency"], latency)
            last = entry["left"] == 0
        if last:
            finish_scene(name)

    def complete(folder):
        """A scene folder counts as done when its metadata file exists (written last); without ``metadata_json`` any
        existing folder does, as in the reference (benchmark.py:54-55)."""
        if not os.path.isdir(folder):
            return False
        return os.path.exists(os.path.join(folder, f"{metadata_fname}.json")) if metadata_json else True

    def jobs():
        for name, scene in scenes:
            folder = os.path.join(output_dir, name)
            if skip_existing and complete(folder):
                skipped.append(name)
                continue
            if callable(scene) and not hasattr(scene, "events"):
                scene = scene()
            os.makedirs(folder, exist_ok=True)
            per_mic = scene_jobs(scene, name, driver.r)
            frames = None
            if metadata_dcase:   # rows are tiny; computed while the scene object is still alive
                from . import metadata

                frames = metadata.generate_dcase2024_metadata(scene)
            # only the (small) metadata dictionary outlives the scene: a dataset run must not keep every clip alive
            with meta_lock:
                meta[name] = dict(dict=scene.to_dict() if hasattr(scene, "to_dict") else {}, left=len(per_mic), latency=0.0,
                                  dcase=frames)
            if not per_mic:          # a scene without microphones has nothing to render: its metadata is all there is
                finish_scene(name)
            yield from per_mic

    def path_of(job):
        name, mic = job.name.rsplit("/", 1)
        return os.path.join(output_dir, name, f"{audio_fname}_{mic}.wav")

    rep = driver.run(jobs(), output_dir=output_dir, subtype=subtype, path_of=path_of,
                     after_write=after_write if (metadata_json or metadata_dcase) else None)
    rep.skipped.extend(skipped)
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
    pl = planning.plan_batch(specs, c, l, jobs[0].sample_rate, lib=renderer.lib)
    res = renderer.render(pl, clips, irs)
    res.check_finite()
    out = []
    for job, (e0, n) in zip(jobs, ranges):
        mix = planning.plan_mixdown(job.starts, job.ends, [len(x) for x in job.clips], [c] * n,
                                    pl.events["out_off"][e0: e0 + n], list(range(e0, e0 + n)), job.duration,
                                    job.sample_rate, c, lib=renderer.lib)
        dev = renderer.mixdown(mix, res, job.ambience)
        out.append(renderer.mem.download(dev)[: c * mix.n_samples].reshape(c, mix.n_samples))
    return out
