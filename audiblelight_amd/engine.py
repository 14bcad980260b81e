"""Device orchestration of the HIP synthesis path: buffers, table upload, kernel launches.

PyTorch is used for plumbing only (HBM allocation, H2D/D2H copies, the current HIP stream);
all arithmetic happens in the kernels behind the C ABI (include/audiblelight_hip.h).
"""
from __future__ import annotations

import ctypes as ct
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _hip, switches
from .plan import BatchPlan, MixPlan


def _debug_flags() -> int:
    """Experimental kernel switches (AL_EXTRA_FLAGS, parsed once: switches.py), masked to the bits that select code paths
    with identical results; bit 0 (AL_FLAG_NO_IR_NORM) changes results and is never taken from it."""
    return switches.current().extra_flags & _hip.DEBUG_FLAG_MASK


# ----------------------------------------------------------------------------- host thread budget
_THREAD_CAP: Optional[int] = None


def set_thread_cap(n: Optional[int]) -> None:
    """Largest helper pool this process may start (IR cast threads, clip packers, the batch driver's cast pool): N ranks on one
    node each take ``distributed.host_share(...)["threads"]`` instead of sizing their pools for the whole machine."""
    global _THREAD_CAP
    _THREAD_CAP = None if n is None else max(1, int(n))


def host_threads(want: int) -> int:
    return max(1, int(want)) if _THREAD_CAP is None else max(1, min(int(want), _THREAD_CAP))


# ----------------------------------------------------------------------------- memory providers
class TorchMemory:
    """HBM through torch (ROCm).  Raises if no GPU is visible: there is no CPU path."""

    def __init__(self, device=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("audiblelight_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.torch = torch
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")

    def stream(self):
        return ct.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, n: int, dtype=np.float32):
        return self.torch.empty(max(int(n), 1), dtype=getattr(self.torch, np.dtype(dtype).name), device=self.device)

    def zeros(self, n: int, dtype=np.float32):
        return self.torch.zeros(max(int(n), 1), dtype=getattr(self.torch, np.dtype(dtype).name), device=self.device)

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        if arr.dtype.fields is not None:  # structured table -> raw bytes
            arr = arr.view(np.uint8)
        if not arr.flags.writeable:       # torch refuses to wrap read-only memory silently (cached tables are read-only)
            arr = arr.copy()
        return self.torch.from_numpy(arr).to(self.device, non_blocking=False)

    def upload_tables(self, arrays) -> list:
        """Several small host tables (structured or plain ndarrays) -> HBM through ONE page-locked arena and ONE asynchronous
        DMA; returns one device tensor per table (uint8 views for structured dtypes), each 16-byte aligned.  A scene needs a
        dozen of them (event / stream / mixdown tables): as separate pageable copies each one is a blocking round trip."""
        flat = []
        for a in arrays:
            a = np.ascontiguousarray(a)
            if a.dtype.fields is not None:
                a = a.view(np.uint8)
            flat.append(a.reshape(-1))
        offs, total = [], 0
        for a in flat:
            offs.append(total)
            total += (max(a.nbytes, 1) + 15) // 16 * 16

        def fill(view):
            raw = view.view(np.uint8)
            for off, a in zip(offs, flat):
                raw[off: off + a.nbytes] = a.view(np.uint8)

        dev = self.upload_staged((total + 3) // 4, fill, tag="tables").view(self.torch.uint8)
        out = []
        for off, a in zip(offs, flat):
            piece = dev[off: off + max(a.nbytes, 1)]
            out.append(piece if a.dtype == np.uint8 else piece[: a.nbytes].view(getattr(self.torch, a.dtype.name)))
        return out

    def download_async(self, buf):
        """Enqueue the D2H copy of a device buffer into page-locked memory on the current stream and return the host TENSOR
        at once; it holds the data only after the stream has been synchronised (the caller batches several of these behind
        one synchronisation)."""
        host = self.torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
        host.copy_(buf, non_blocking=True)
        return host

    def upload_staged(self, n: int, fill, tag: str = "audio") -> object:
        """float32[n] assembled by ``fill(host_view)`` in a REUSED page-locked staging buffer, then one asynchronous DMA:
        for inputs that have to be packed on the host anyway (the clips of a scene).  A fresh pageable buffer would cost
        a page fault per 4 KiB before the copy even starts.  One buffer per ``tag``."""
        if not hasattr(self, "_staging_by_tag"):
            self._staging_by_tag = {}
        slot = self._staging_by_tag.setdefault(tag, [None, None])   # [page-locked buffer that only grows, event of the last DMA out of it]
        host, last = slot
        if last is not None:
            last.synchronize()                   # the previous DMA out of this buffer must have finished
        if host is None or host.numel() < n:     # page-locking costs tens of ms: grow by at least a quarter, never per size
            cap = max(int(n), 0 if host is None else host.numel() * 5 // 4)
            host = slot[0] = self.torch.empty((cap + 65535) // 65536 * 65536, dtype=self.torch.float32, pin_memory=True)
        view = host[: int(n)]
        fill(view.numpy())
        dev = view.to(self.device, non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        slot[1] = ev
        return dev

    def upload_f64_as_f32(self, arr: np.ndarray, threads: Optional[int] = None):
        """float64 host array -> float32 in HBM (what the reference's ``WorldState.get_irs()`` hands over is float64): converted
        by a pool of threads (numpy releases the GIL while it casts) into a reused page-locked buffer, in chunks, each chunk
        on its way by DMA while the next one converts.  Half the PCIe bytes of uploading the float64 tensor and converting on
        the device: 22 ms instead of 28 ms for cfg2's 1.57 GB (profiles/r02_e2e_probe.txt)."""
        from concurrent.futures import ThreadPoolExecutor

        flat = np.ascontiguousarray(arr).reshape(-1)
        n = flat.size
        threads = host_threads(switches.current().convert_threads or threads or 8)
        if not hasattr(self, "_convert"):
            self._convert = dict(pool=None, threads=0, host=None, last=None)
        cv = self._convert
        if cv["pool"] is None or cv["threads"] != threads:
            cv["pool"], cv["threads"] = ThreadPoolExecutor(threads), threads
        if cv["last"] is not None:
            cv["last"].synchronize()                 # the previous DMA out of the staging buffer must have finished
        if cv["host"] is None or cv["host"].numel() < n:
            cap = max(n, 0 if cv["host"] is None else cv["host"].numel() * 5 // 4)
            cv["host"] = self.torch.empty((cap + 65535) // 65536 * 65536, dtype=self.torch.float32, pin_memory=True)
        host = cv["host"][:n]
        view = host.numpy()
        dev = self.torch.empty(n, dtype=self.torch.float32, device=self.device)
        n_chunks = max(1, min(8, n // (1 << 22)))
        bounds = np.linspace(0, n, n_chunks + 1).astype(np.int64)
        for a, b in zip(bounds[:-1], bounds[1:]):
            cut = np.linspace(a, b, threads + 1).astype(np.int64)
            list(cv["pool"].map(lambda i: np.copyto(view[cut[i]: cut[i + 1]], flat[cut[i]: cut[i + 1]], casting="same_kind"),
                                range(threads)))
            dev[a:b].copy_(host[a:b], non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        cv["last"] = ev
        return dev

    def upload_async(self, arr: np.ndarray):
        """Large caller-owned array -> HBM by DMA without holding the calling thread: the array's pages are page-locked in
        place (hipHostRegister), the copy is enqueued on the current stream, and the returned ``release()`` must be
        called once that copy has completed (it unregisters the pages).  Falls back to ``upload`` when registration fails."""
        arr = np.ascontiguousarray(arr)
        rt = self.torch.cuda.cudart()
        if int(rt.cudaHostRegister(arr.ctypes.data, arr.nbytes, 0)) != 0:
            return self.upload(arr), (lambda: None)
        src = self.torch.from_numpy(arr.reshape(-1))
        dev = self.torch.empty(src.numel(), dtype=src.dtype, device=self.device)
        dev.copy_(src, non_blocking=True)
        return dev, (lambda p=arr.ctypes.data, keep=arr: rt.cudaHostUnregister(p))

    def _helper(self):
        from concurrent.futures import ThreadPoolExecutor

        if getattr(self, "_beside_pool", None) is None:
            self._beside_pool = ThreadPoolExecutor(1, thread_name_prefix="al-upload")
        return self._beside_pool

    def upload_beside(self, arr: np.ndarray):
        """A large contiguous host array -> HBM while the calling thread goes on: the device buffer is allocated here, the copy
        itself is one ``hipMemcpyAsync`` + ``hipStreamSynchronize`` on a stream of its own, made from a helper thread
        through ctypes (which drops the GIL for the call; ``Tensor.to`` keeps it for a pageable source).  Returns
        (device buffer, future): the buffer holds the data once ``future.result()`` has returned.  For the one long blocking
        call of a scene -- 14 ms for cfg2's 0.8 GB pageable IR tensor -- beside the planning and clip packing."""
        if getattr(self, "_hiprt", None) is None:
            self._hiprt = ct.CDLL("libamdhip64.so")
            self._hiprt.hipMemcpyAsync.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_void_p]
            self._hiprt.hipStreamSynchronize.argtypes = [ct.c_void_p]
            self._hiprt.hipSetDevice.argtypes = [ct.c_int]
        flat = np.ascontiguousarray(arr).reshape(-1)
        dev = self.torch.empty(flat.size, dtype=getattr(self.torch, flat.dtype.name), device=self.device)
        if getattr(self, "_beside_stream", None) is None:
            self._beside_stream = self.torch.cuda.Stream(self.device)
        self._beside_stream.wait_stream(self.torch.cuda.current_stream(self.device))   # `dev` may be a block just released there
        side = self._beside_stream.cuda_stream
        index = self.device.index if self.device.index is not None else self.torch.cuda.current_device()

        def job(keep=flat):
            rt = self._hiprt
            # on a stream of its own: a copy on the caller's stream would hold that stream's queue lock for its whole length
            # and stall the caller's own (small, asynchronous) uploads behind it.  Complete on the host before anything reads it.
            err = rt.hipSetDevice(index) or rt.hipMemcpyAsync(dev.data_ptr(), keep.ctypes.data, keep.nbytes, 1, side) \
                or rt.hipStreamSynchronize(side)          # 1 = hipMemcpyHostToDevice
            if err:
                raise RuntimeError(f"background upload of {keep.nbytes} bytes failed with HIP error {err}")
            return dev

        return dev, self._helper().submit(job)

    def ptr(self, buf) -> int:
        return buf.data_ptr()

    def free_bytes(self) -> int:
        """HBM not in use: free on the device plus what torch's caching allocator holds but has not handed out."""
        free, _total = self.torch.cuda.mem_get_info(self.device)
        try:      # the allocator's own counters: torch.cuda.memory_reserved() flattens the whole statistics tree per call (0.1 ms)
            index = self.device.index if self.device.index is not None else self.torch.cuda.current_device()
            stats = self.torch._C._cuda_memoryStats(index)
            cached = stats["reserved_bytes"]["all"]["current"] - stats["allocated_bytes"]["all"]["current"]
        except (AttributeError, KeyError, TypeError, RuntimeError):
            cached = self.torch.cuda.memory_reserved(self.device) - self.torch.cuda.memory_allocated(self.device)
        return int(free + cached)

    def download(self, buf) -> np.ndarray:
        """Device buffer -> fresh ndarray.  Large buffers land in page-locked memory from torch's caching host allocator
        (55 GB/s instead of 12 GB/s into freshly faulted pageable memory, profiles/r02_h2d_probe.txt); the array keeps
        that block alive and the block returns to the cache when the array is dropped."""
        if buf.numel() * buf.element_size() < (1 << 22):
            return buf.cpu().numpy()
        host = self.torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
        host.copy_(buf, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        return host.numpy()

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


class _Arrival:
    """A future with a fixed value to hand out once it has completed."""

    def __init__(self, future, value):
        self.future, self.value = future, value

    def result(self):
        self.future.result()
        return self.value


_PACK_POOL = None


# ----------------------------------------------------------------------------- clip hand-over
@dataclass
class ClipSource:
    """One event's mono clip as the renderer takes it: in host memory or already in HBM, plus what is still to be
    applied to it ON THE DEVICE: the product of scalar FX (Gain, Invert) and the peak normalisation of
    ``Event.load_audio`` (event.py:529-536).  A plain ndarray means "finished clip" (prescale 1, no normalisation)."""
    host: Optional[np.ndarray] = None
    device: object = None          # device buffer (first ``n`` floats), used when ``host`` is None
    n: int = 0
    prescale: float = 1.0
    normalize: bool = False

    def __len__(self) -> int:
        return int(self.n if self.host is None else len(self.host))


def as_clip_source(clip) -> ClipSource:
    if isinstance(clip, ClipSource):
        return clip
    arr = np.ascontiguousarray(clip, dtype=np.float32)
    return ClipSource(host=arr, n=len(arr))


# ----------------------------------------------------------------------------- results
@dataclass
class RenderResult:
    plan: BatchPlan
    memory: object
    lib: _hip.Library
    spatial: object        # device: per event (C, len) UNSCALED convolution
    event_scale: object    # device float32[E]
    event_stats: object    # device float64[E, 4]: sum|x|, max|x|, non-finite count, db multiplier
    emitter_gain: object   # device float32[n_emitters]
    keep: tuple = ()       # buffers that must outlive the launches

    def scales(self) -> np.ndarray:
        return self.memory.download(self.event_scale)[: len(self.plan.events)].astype(np.float64)

    def stats(self) -> np.ndarray:
        return self.memory.download(self.event_stats)[: 4 * len(self.plan.events)].reshape(-1, 4)

    def check_finite(self, stats: Optional[np.ndarray] = None) -> None:
        """librosa.util.valid_audio of every event's render (synthesize.py:603) from the device's non-finite counts.
        ``stats``: the (E, 4) statistics already on the host (a caller that downloaded them with other results)."""
        if getattr(self, "_finite_ok", False):
            return
        bad = np.flatnonzero((self.stats() if stats is None else stats)[:, 2] > 0)
        if len(bad):
            raise ValueError(f"Audio buffer is not finite everywhere (events {bad.tolist()})")
        self._finite_ok = True

    def raw_spatial(self, i: int) -> np.ndarray:
        ev = self.plan.events[i]
        n = self.plan.n_capsules * int(ev["len"])
        flat = self.memory.download(self.spatial[int(ev["out_off"]): int(ev["out_off"]) + n])
        return flat.reshape(self.plan.n_capsules, int(ev["len"]))

    def scaled_copy(self, lo: int, n: int, scale_ptrs: Sequence[int], f64_scale_ptrs: Sequence[int] = ()):
        """A device copy of spatial[lo : lo + n] multiplied ON THE DEVICE by the float32 (``scale_ptrs``) and float64
        (``f64_scale_ptrs``) scalars behind those device addresses (al_scale_rows / al_scale_rows_f64: the float32 product the
        mixdown forms too): the host only copies and widens."""
        piece = self.spatial[lo: lo + n]
        tmp = piece.clone() if hasattr(piece, "clone") else np.array(piece)
        for p in scale_ptrs:
            self.lib.call("al_scale_rows", self.memory.ptr(tmp), n, p, self.memory.stream())
        for p in f64_scale_ptrs:
            self.lib.call("al_scale_rows_f64", self.memory.ptr(tmp), n, p, self.memory.stream())
        return tmp

    def spatial_audio(self, i: int, dtype=np.float64) -> np.ndarray:
        """event.spatial_audio[mic]: the scaled (C, La) render (synthesize.py:599,606), scaled by event_scale[i] on the device
        and widened to ``dtype`` after the copy (the reference's: float64, except float32 for a float32 clip tiled over the capsules
        or convolved with float32 IRs -- synthesize._reference_dtype)."""
        ev = self.plan.events[i]
        n = self.plan.n_capsules * int(ev["len"])
        dev = self.scaled_copy(int(ev["out_off"]), n, [self.memory.ptr(self.event_scale) + 4 * i])
        return self.memory.download(dev)[:n].reshape(self.plan.n_capsules, int(ev["len"])).astype(dtype)


class Renderer:
    """Launches the batch pipeline of include/audiblelight_hip.h on one GPU."""

    def __init__(self, lib: Optional[_hip.Library] = None, memory=None):
        self.lib = lib or _hip.get_library()
        self.mem = memory or TorchMemory()
        self._twiddles: Dict[int, object] = {}

    def twiddle(self, log2_block: int):
        if log2_block not in self._twiddles:
            nbytes = self.lib.call("al_twiddle_bytes", log2_block)
            buf = self.mem.empty(nbytes // 4, np.float32)
            self.lib.call("al_twiddle_init", self.mem.ptr(buf), log2_block, self.mem.stream())
            self._twiddles[log2_block] = buf
        return self._twiddles[log2_block]

    # -- IR upload: (C, N, L) any float dtype -> float32 device tensor with 4-float aligned rows
    def upload_irs(self, irs: np.ndarray, async_release: Optional[list] = None, host_cast: Optional[bool] = None):
        """``async_release``: a list; when given and the memory provider can do it, the H2D copy is an asynchronous DMA
        from the caller's page-locked-in-place array and a ``release`` callable is appended (call it after the copy).
        The caller's (C, N, L) tensor goes to HBM AS IT IS (no host-side copy, cast or padding pass: the H2D copy of
        pageable memory runs at PCIe rate, a fresh ``np.zeros`` + copy at a fifth of it, profiles/r02_h2d_probe.txt);
        rows whose length is not a multiple of 4 are re-pitched by a device kernel.  float64 (what ``WorldState.get_irs()``
        returns, worldstate.py:2183-2255) is cast to float32 by a pool of host threads into page-locked memory, chunk by
        chunk under the DMA of the previous chunk: 22 ms instead of 28 ms for cfg2's 1.57 GB (``host_cast=False`` /
        ``AL_F64_UPLOAD=device``: upload the float64 bytes and cast on the device; the pipelined batch driver casts in its
        planner stage instead, beside the previous scene's upload: profiles/r02_e2e_probe.txt)."""
        c, n, l = irs.shape
        lp = (l + 3) // 4 * 4
        if n == 0 or l == 0:
            return self.mem.zeros(max(c, 1) * lp), (lp, lp)
        if irs.dtype not in (np.float32, np.float64):
            irs = irs.astype(np.float32)
        if host_cast is None:
            host_cast = switches.current().f64_upload == "host"
        if irs.dtype == np.float64 and host_cast and hasattr(self.mem, "upload_f64_as_f32"):
            raw = self.mem.upload_f64_as_f32(irs)        # converted on the host by a thread pool: half the PCIe bytes
            if lp == l:
                return raw, (n * lp, lp)
            dev = self.mem.empty(c * n * lp)
            self.lib.call("al_pack_irs_f32", self.mem.ptr(raw), self.mem.ptr(dev), c * n, l, lp, self.mem.stream())
            return dev, (n * lp, lp)
        if async_release is not None and hasattr(self.mem, "upload_async"):
            raw, release = self.mem.upload_async(np.ascontiguousarray(irs).reshape(-1))
            async_release.append(release)
        else:
            raw = self.mem.upload(np.ascontiguousarray(irs).reshape(-1))
        if irs.dtype == np.float32 and lp == l:
            return raw, (n * lp, lp)
        dev = self.mem.empty(c * n * lp)
        self.lib.call("al_pack_irs_f64" if irs.dtype == np.float64 else "al_pack_irs_f32", self.mem.ptr(raw),
                      self.mem.ptr(dev), c * n, l, lp, self.mem.stream())
        return dev, (n * lp, lp)   # `raw` may be released: the allocator orders its reuse behind the kernel (same stream)

    def upload_irs_beside(self, irs):
        """Start ``upload_irs`` on the memory provider's helper thread and return at once: an object whose ``result()`` gives
        ``(device buffer, ir_strides)`` -- pass it to ``prepare`` / ``render`` as ``irs`` -- or None when this tensor has to
        go the inline way.  Taken for float32 tensors of 8 MiB and more with 4-float rows (the layout the kernels read as it is): one pageable copy
        through ctypes, 26.9 -> 25.6 ms per cfg2 ``Scene.generate()`` (profiles/r03h_dropin_beside_ab.txt)."""
        mem = self.mem
        if not (isinstance(irs, np.ndarray) and irs.ndim == 3 and irs.size > 0 and hasattr(mem, "upload_beside")):
            return None
        c, n, l = irs.shape
        if irs.dtype == np.float32 and l % 4 == 0 and irs.nbytes >= switches.current().beside_min_bytes and not getattr(mem, "_no_beside", False):
            try:
                dev, arrived = mem.upload_beside(irs)
            except (OSError, AttributeError):      # no libamdhip64.so under that name for ctypes: the inline copy is always there
                mem._no_beside = True
                return None
            return _Arrival(arrived, (dev, (n * l, l)))
        return None   # float64: the host-cast pipeline already overlaps cast and DMA; beside the clip packing it is SLOWER
                      # (29.4 vs 28.0 ms per cfg2 scene, profiles/r03h_dropin_beside_ab.txt); ragged rows: re-pitched by a kernel

    def pack_audio(self, plan: BatchPlan, clips: Sequence, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Host clips at their 4-float aligned offsets (device-resident clips are copied in HBM by ``prepare``)."""
        host = np.zeros(plan.audio_floats, dtype=np.float32) if out is None else out
        todo = [(int(off), as_clip_source(clip).host) for off, clip in zip(plan.audio_offsets, clips)
                if as_clip_source(clip).host is not None]
        if sum(len(a) for _, a in todo) < (1 << 22):
            for off, a in todo:
                host[off: off + len(a)] = a
            return host
        # tens of megabytes of clips (cfg2: 49 MB): a few threads (numpy releases the GIL while it copies) instead of one
        global _PACK_POOL
        if _PACK_POOL is None:
            from concurrent.futures import ThreadPoolExecutor

            _PACK_POOL = ThreadPoolExecutor(host_threads(4))

        def put(item):
            off, a = item
            host[off: off + len(a)] = a

        list(_PACK_POOL.map(put, todo))
        return host

    def auto_chunk_events(self, plan: BatchPlan) -> Optional[int]:
        """Events per chunk when the spectra workspace of the whole batch (H, X, Y: ``plan.workspace_bytes()``) would not fit
        the budget -- ``AL_WORKSPACE_GB`` or 40 % of the HBM that is free right now -- else None (one chunk, the fast path:
        chunking does not reduce time, profiles/r01_chunk_sweep.txt).  The chunks reuse ONE workspace, so a scene of any
        number of events renders in bounded memory beside its inputs and outputs."""
        budget = switches.current().workspace_gb
        if budget is not None:
            budget = float(budget) * 1e9
        elif hasattr(self.mem, "free_bytes"):
            budget = 0.4 * self.mem.free_bytes()
        else:
            return None
        total, n = plan.workspace_bytes(), len(plan.events)
        if total <= budget or n <= 1:
            return None
        return max(1, int(n * budget / total))

    def prepare(self, plan: BatchPlan, clips: Sequence[np.ndarray], irs, ir_strides=None,
                chunk_events: Optional[int] = None, normalize_irs: bool = True, lanes: int = 1,
                audio_dev=None, emitter_parts: Optional[np.ndarray] = None, spectra_workspaces: bool = True) -> "PreparedBatch":
        """Upload inputs + tables and allocate every workspace/output buffer of one batch.
        ``irs`` is a (C, N, L) ndarray or a device buffer from upload_irs (then pass ``ir_strides``).
        ``chunk_events``: run the batch as chunks of that many events over a reused spectra workspace.
        ``lanes``: number of workspaces / HIP streams the chunks alternate over (chunk i runs on lane i % lanes),
        so the HBM-bound accumulate of one chunk can overlap the instruction-bound transforms of another.
        ``emitter_parts``: the caller's own al_batch.emitter_parts (int32 per IR column; 0 = "energies only": the forward transform
        reads that IR for normalize_irs and neither transforms nor stores it) instead of the planner's.
        ``spectra_workspaces=False``: a batch that will only run the IR energy pass and the level law (the general STFT path of
        synthesize._render_moving_general): the H / X / Y spectra workspaces are one block each instead of the batch's size."""
        mem = self.mem
        on_its_way = irs if hasattr(irs, "result") else None      # from upload_irs_beside: waited for below, after the staging
        if ir_strides is None and on_its_way is None:   # a host tensor (with ir_strides: a device buffer from upload_irs / ingest.pack_ragged_irs)
            irs, ir_strides = self.upload_irs(irs)
        B = plan.block
        if chunk_events is None:
            chunk_events = self.auto_chunk_events(plan)
        chunks = plan.chunks(chunk_events)
        P, C = plan.n_partitions, plan.n_capsules
        h_blocks = max(max(c["n_emitters"] for c in chunks) * C * P, 1)
        x_blocks = max(max(c["xspec_blocks"] for c in chunks), 1)
        y_blocks = max(max(c["yspec_blocks"] for c in chunks), 1)
        if not spectra_workspaces:
            h_blocks = x_blocks = y_blocks = 1
        lanes = max(1, min(int(lanes), len(chunks)))
        sources = [as_clip_source(c) for c in clips]
        if audio_dev is None:
            if hasattr(mem, "upload_staged"):
                audio_dev = mem.upload_staged(plan.audio_floats, lambda view: self.pack_audio(plan, sources, out=view))
            else:
                audio_dev = mem.upload(self.pack_audio(plan, sources))
            for off, src in zip(plan.audio_offsets, sources):   # clips that are already in HBM (device FX chain): D2D
                if src.host is None:
                    audio_dev[int(off): int(off) + len(src)] = src.device[: len(src)]
        fold = any(src.normalize or src.prescale != 1.0 for src in sources)
        tables = [plan.events, plan.streams if len(plan.streams) else np.zeros(1, dtype=_hip.STREAM_DTYPE), plan.wtab]
        if fold:
            tables += [np.array([src.prescale for src in sources], dtype=np.float32),
                       np.array([1 if src.normalize else 0 for src in sources], dtype=np.int32)]
        sw = switches.current()      # parsed once per process (switches.py): nothing on this path reads the environment
        parts = plan.emitter_parts() if sw.trim_partitions else None
        if emitter_parts is not None:
            parts = np.ascontiguousarray(emitter_parts, dtype=np.int32)
        if parts is not None:        # al_batch.emitter_parts: IR partitions that cannot reach a kept block are not transformed
            tables.append(parts)
        tabs = mem.upload_tables(tables) if hasattr(mem, "upload_tables") else [mem.upload(t) for t in tables]
        parts_dev = tabs.pop() if parts is not None else None
        if on_its_way is not None:
            irs, ir_strides = on_its_way.result()
        bufs = dict(
            audio=audio_dev, ir=irs, events=tabs[0], streams=tabs[1], wtab=tabs[2], twiddle=self.twiddle(plan.log2_block),
            ir_energy=mem.empty(plan.hspec_blocks), emitter_gain=mem.empty(plan.n_emitters),
            # one extra, all-zero block behind each spectra workspace: the LDS-DMA accumulate reads the rows past an odd partition count from it
            hspec=mem.empty((h_blocks + 1) * B * 2), xspec=mem.empty((x_blocks + 1) * B * 2),
            yspec=mem.empty(y_blocks * B * 2), spatial=mem.empty(plan.spatial_floats),
            partials=mem.empty(plan.n_partials * 4), event_stats=mem.empty(len(plan.events) * 4, np.float64),
            event_scale=mem.empty(len(plan.events)))
        if fold:
            bufs["clip_scale"] = mem.empty(len(plan.events))
        if parts_dev is not None:
            bufs["emitter_parts"] = parts_dev
        ptrs = {k: mem.ptr(v) for k, v in bufs.items()}
        extra = [dict(hspec=mem.empty((h_blocks + 1) * B * 2), xspec=mem.empty((x_blocks + 1) * B * 2),
                      yspec=mem.empty(y_blocks * B * 2)) for _ in range(lanes - 1)]
        for ws in [bufs] + extra:
            ws["hspec"][h_blocks * B * 2:] = 0
            ws["xspec"][x_blocks * B * 2:] = 0
        lane_ptrs = [ptrs] + [dict(ptrs, **{k: mem.ptr(v) for k, v in e.items()}) for e in extra]
        bufs["_lanes"] = extra
        descs = [_hip.AlBatch(
            log2_block=plan.log2_block, n_capsules=C, n_events=c["n_events"], n_streams=c["n_streams"],
            n_emitters=c["n_emitters"], ir_len=plan.ir_len, ir_stride_c=ir_strides[0], ir_stride_n=ir_strides[1],
            n_partitions=P, max_blocks=c["max_blocks"], max_nj=c["max_nj"], hop=plan.hop, event0=c["event0"],
            stream0=c["stream0"], emitter0=c["emitter0"], xspec_block0=c["xspec_block0"],
            yspec_block0=c["yspec_block0"], flags=(0 if normalize_irs else _hip.FLAG_NO_IR_NORM) | _debug_flags(),
            xspec_zero_block=x_blocks, hspec_zero_block=h_blocks, **lane_ptrs[i % lanes])
            for i, c in enumerate(chunks)]
        # Which kernels run -- layout flags for the block size, accumulate flags for the chunk's event mix -- is the LIBRARY's
        # decision (al_plan_batch_flags, csrc/al_plan.cpp): a C host gets the same dispatch from the same plan
        # (tests/c_caller/render_planned.c).  The A/B switches of switches.py can force other paths.
        for desc, chunk in zip(descs, chunks):
            desc.flags |= plan.batch_flags(chunk)
            if sw.forces_dispatch:
                desc.flags = self._forced_dispatch(desc, plan, sw)
        if fold:   # A13 on the device: peak normalisation + folded scalar FX, no clip statistics cross PCIe
            pre, mode = tabs[3], tabs[4]
            for desc in descs:
                self.lib.call("al_clip_scales", ct.byref(desc), mem.ptr(pre), mem.ptr(mode), mem.stream())
            bufs["_clip_tables"] = (pre, mode)
        batch = PreparedBatch(self, plan, bufs, descs, lanes)
        batch.energy_only = not spectra_workspaces
        return batch

    def _forced_dispatch(self, desc, plan: BatchPlan, sw) -> int:
        """al_batch.flags with the A/B switches applied on top of the library's policy (tests and measurements only)."""
        lb, P = plan.log2_block, plan.n_partitions
        layout = _hip.FLAG_SPLIT_SPECTRA | _hip.FLAG_QUAD_SPECTRA
        accumulate = _hip.FLAG_STATIC_MAC | _hip.FLAG_ONLY_STATIC
        flags = desc.flags & ~layout
        quad16 = lb == 14 and sw.quad16      # AL_QUAD16=0: the one- / two-transform kernels at B = 16384
        want_split = sw.split if sw.split is not None else (lb == 13 or quad16)
        if want_split and lb >= 11:
            flags |= _hip.FLAG_SPLIT_SPECTRA | (_hip.FLAG_QUAD_SPECTRA if quad16 else 0)
        if not sw.static_mac or (sw.static_mac_max_p is not None and P > sw.static_mac_max_p):
            flags &= ~accumulate
        return flags

    def render(self, plan: BatchPlan, clips: Sequence[np.ndarray], irs, ir_strides=None,
               stages: Optional[Sequence[str]] = None, chunk_events: Optional[int] = None,
               normalize_irs: bool = True) -> RenderResult:
        """prepare + run stages 1-6 for one batch."""
        return self.prepare(plan, clips, irs, ir_strides, chunk_events, normalize_irs).run(stages)

    def prepare_mixdown(self, mix: MixPlan, result: RenderResult, ambience: Sequence = (), scene=None) -> "PreparedMix":
        """``scene``: an existing (C*T) device buffer to accumulate into (else a new buffer is made)."""
        mem = self.mem
        n = mix.n_capsules * mix.n_samples
        host_tabs = (mix.tile_ptr, mix.tile_events, mix.slot_src, mix.slot_len, mix.slot_start, mix.slot_count, mix.slot_rows,
                     mix.slot_event)
        tabs = mem.upload_tables(host_tabs) if hasattr(mem, "upload_tables") else [mem.upload(x) for x in host_tabs]
        # every ambience comes with ONE multiplier per capsule (k_mixdown reads ambience_scale[c], al_axpy_rows a[row]):
        # a shorter buffer would be read out of bounds, so a single scalar is broadcast and anything else is refused
        checked = []
        for noise, scales in ambience:
            have = int(scales.numel()) if hasattr(scales, "numel") else int(np.size(scales))
            if have == 1 and mix.n_capsules > 1:
                wide = mem.empty(mix.n_capsules)
                wide[: mix.n_capsules] = scales[0]
                scales = wide
            elif have < mix.n_capsules:
                raise ValueError(f"ambience multipliers: {have} values for {mix.n_capsules} capsules (one per capsule, or one scalar)")
            checked.append((noise, scales))
        ambience = checked
        # exactly one ambience and a fresh scene buffer (the normal case): added inside the mixdown kernel
        fused = list(ambience) if (len(ambience) == 1 and scene is None) else []
        ambience = [] if fused else list(ambience)
        accumulate = bool(ambience) or scene is not None
        zero_first = scene is None and bool(ambience)
        scene = mem.empty(n) if scene is None else scene
        p = mem.ptr
        desc = _hip.AlMix(n_capsules=mix.n_capsules, n_samples=mix.n_samples, tile=mix.tile, n_tiles=mix.n_tiles,
                          accumulate=1 if accumulate else 0, tile_ptr=p(tabs[0]), tile_events=p(tabs[1]),
                          slot_src=p(tabs[2]), slot_len=p(tabs[3]), slot_start=p(tabs[4]), slot_count=p(tabs[5]),
                          slot_rows=p(tabs[6]), slot_event=p(tabs[7]), spatial=p(result.spatial),
                          event_scale=p(result.event_scale), scene=p(scene),
                          ambience=p(fused[0][0]) if fused else None, ambience_scale=p(fused[0][1]) if fused else None)
        return PreparedMix(self, mix, desc, scene, list(ambience), tabs + [result] + fused, zero_first)

    # -- A11
    def mixdown(self, mix: MixPlan, result: RenderResult, ambience: Sequence = ()):
        """scene (C, T) float32 on device.  ``ambience`` = [(device noise (C*T floats), device float32[C] multipliers)]."""
        return self.prepare_mixdown(mix, result, ambience).run()

    def row_stats(self, x_dev, rows: int, cols: int):
        n = self.lib.call("al_row_stats_partials", rows, cols)
        partials = self.mem.empty(n)
        out = self.mem.empty(rows * 4, np.float64)
        self.lib.call("al_row_stats", self.mem.ptr(x_dev), rows, cols, self.mem.ptr(partials), self.mem.ptr(out),
                      self.mem.stream())
        return out


class PreparedBatch:
    """One batch with every buffer resident in HBM; ``run()`` only enqueues kernels."""

    STAGES = ("al_forward_spectra", "al_emitter_gains", "al_spectral_mac", "al_block_synthesis", "al_event_levels")

    def __init__(self, renderer: Renderer, plan: BatchPlan, bufs: dict, descs: List[_hip.AlBatch], lanes: int = 1):
        self.renderer, self.plan, self.bufs, self.descs, self.lanes = renderer, plan, bufs, descs, lanes
        self._streams = None

    def _run_lanes(self) -> RenderResult:
        """Chunk i on HIP stream i % lanes (torch streams); joins back into the current stream."""
        torch = self.renderer.mem.torch
        lib = self.renderer.lib
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=self.renderer.mem.device) for _ in range(self.lanes)]
        cur = torch.cuda.current_stream(self.renderer.mem.device)
        for st in self._streams:
            st.wait_stream(cur)
        for i, desc in enumerate(self.descs):
            for name in self.stage_names(i):
                lib.call(name, ct.byref(desc), ct.c_void_p(self._streams[i % self.lanes].cuda_stream))
        for st in self._streams:
            cur.wait_stream(st)
        return self.result()

    ENERGY_ONLY_STAGES = ("al_ir_spectra", "al_emitter_gains", "al_emitter_norm_sums", "al_emitter_gains_from_sums", "al_event_levels_from_stats")
    energy_only = False      # prepare(spectra_workspaces=False): the H / X / Y workspaces are one block each

    def _check_stage(self, name: str) -> None:
        if self.energy_only and name not in self.ENERGY_ONLY_STAGES:
            raise RuntimeError(f"{name}: this batch was prepared without spectra workspaces (spectra_workspaces=False): only the IR "
                               f"energy pass and the level law may run on it")

    def run(self, stages: Optional[Sequence[str]] = None) -> RenderResult:
        for name in (self.STAGES if stages is None else stages):
            self._check_stage(name)
        if self.lanes > 1 and stages is None and hasattr(self.renderer.mem, "torch"):
            return self._run_lanes()
        lib, stream = self.renderer.lib, self.renderer.mem.stream()
        for i, desc in enumerate(self.descs):
            for name in (self.stage_names(i) if stages is None else stages):
                lib.call(name, ct.byref(desc), stream)
        return self.result()

    def stage_names(self, chunk: int = 0) -> Sequence[str]:
        """The C-ABI calls one pass of a chunk makes, in order (bench.py times them one by one)."""
        return self.STAGES

    def run_stage(self, name: str, chunk: int = 0) -> None:
        self._check_stage(name)
        self.renderer.lib.call(name, ct.byref(self.descs[chunk]), self.renderer.mem.stream())

    def result(self) -> RenderResult:
        b = self.bufs
        return RenderResult(plan=self.plan, memory=self.renderer.mem, lib=self.renderer.lib, spatial=b["spatial"],
                            event_scale=b["event_scale"], event_stats=b["event_stats"],
                            emitter_gain=b["emitter_gain"],
                            # launches on the current stream are ordered against the allocator's reuse of the spectra
                            # workspace, so a finished single-lane batch need not be kept alive by its result (it would
                            # pin 5-13 GB per microphone for as long as an event keeps its render); side streams do
                            keep=(self,) if self.lanes > 1 else ())


class CapturedScene:
    """A prepared batch (+ optional mixdown) recorded once into a HIP graph and replayed with one launch.

    Small scenes are launch-bound (seven kernels of a few microseconds each); replaying a graph removes the
    per-kernel launch cost.  The recorded launches read the same device tables and buffers every time, so new inputs
    are written INTO those buffers (``PreparedBatch.bufs``) between replays.  Needs the torch/ROCm memory provider.
    """

    def __init__(self, batch: "PreparedBatch", mix: Optional["PreparedMix"] = None):
        mem = batch.renderer.mem
        if not hasattr(mem, "torch"):
            raise RuntimeError("HIP graph capture needs the torch/ROCm memory provider")
        if batch.lanes > 1:
            raise ValueError("capture a single-lane batch")
        torch = mem.torch
        self.batch, self.mix = batch, mix
        cur = torch.cuda.current_stream(mem.device)
        side = torch.cuda.Stream(device=mem.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # one eager pass off the default stream, as capture requires
            self._enqueue()
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._enqueue()

    def _enqueue(self):
        self.batch.run()
        if self.mix is not None:
            self.mix.run()

    def replay(self):
        """Enqueue the whole scene on the current stream; returns (RenderResult, scene buffer or None)."""
        self.graph.replay()
        return self.batch.result(), (self.mix.scene if self.mix is not None else None)


class PreparedMix:
    def __init__(self, renderer: Renderer, mix: MixPlan, desc: _hip.AlMix, scene, ambience, keep, zero_first=False):
        self.renderer, self.mix, self.desc, self.scene, self.ambience, self.keep = renderer, mix, desc, scene, ambience, keep
        self.zero_first = zero_first

    def retarget(self, scene) -> "PreparedMix":
        """The same mixdown (same device tables, same inputs) writing into another (C * T) float32 device buffer: a batch of
        scenes that reuses one set of resident inputs gives every scene its own output without a device-to-device copy."""
        desc = type(self.desc).from_buffer_copy(self.desc)   # a ctypes structure copied by value
        desc.scene = self.renderer.mem.ptr(scene)
        return PreparedMix(self.renderer, self.mix, desc, scene, self.ambience, self.keep, self.zero_first)

    def run(self):
        mem, lib = self.renderer.mem, self.renderer.lib
        stream = mem.stream()
        n = self.mix.n_capsules * self.mix.n_samples
        if self.ambience:
            if self.zero_first:
                self.scene.zero_() if hasattr(self.scene, "zero_") else self.scene.fill(0)
            for noise, a_dev in self.ambience:
                lib.call("al_axpy_rows", mem.ptr(self.scene), mem.ptr(noise), mem.ptr(a_dev), self.mix.n_capsules,
                         self.mix.n_samples, stream)
        lib.call("al_mixdown", ct.byref(self.desc), stream)
        return self.scene
