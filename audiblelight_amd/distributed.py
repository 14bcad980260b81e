"""Multi-GPU batch rendering: independent scenes sharded over one process per GPU.

Scenes never exchange data while rendering (the reference's dataset scripts are a serial loop over
independent scenes, scripts/generate/benchmark.py:44-77), so the only collective is the optional
gather of finished ``scene.audio`` buffers to one rank at the end: every peer sends over its own
xGMI link into the root (RCCL ``gather``; ``backend="nccl"`` is RCCL on ROCm, ``"gloo"`` on CPU tests).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Static round-robin ownership: item i belongs to rank i % world_size."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_items, world_size))


def shard_stream(items, rank: int, world_size: int):
    """The same round-robin ownership for a stream of unknown length (a generator of scenes): yields items
    ``rank, rank + world_size, ...`` and drops the others unevaluated."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    for i, item in enumerate(items):
        if i % world_size == rank:
            yield item


def host_share(local_rank: int, local_world: int, device_index: Optional[int] = None, pin: bool = True) -> dict:
    """This rank's share of the host: N ranks on one node must not each size their helper pools (IR cast threads, clip packers,
    the batch driver's planner / uploader / writer) for the whole machine.  Returns {"cpus": usable CPUs of this rank,
    "threads": the cap for any one pool, "numa_node", "pinned", "threads_pinned"}.  With ``pin`` the process is restricted
    (sched_setaffinity) to the CPUs local to its GPU's NUMA node (sysfs ``local_cpulist`` of the GPU's PCI function) when the node
    can be found -- H2D staging and page-locked buffers then live next to the GPU -- else to an equal contiguous slice of the usable
    CPUs.  The mask is applied to EVERY thread that exists at the time (/proc/self/task: sched_setaffinity(0, ...) alone would only
    move the calling thread; torch / OpenMP workers, RCCL proxies and upload helpers started earlier would keep the whole machine),
    threads created later inherit it.  Best effort: any failure leaves the affinity alone and only the cap applies."""
    import os

    try:
        usable = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = list(range(os.cpu_count() or 1))
    local_world = max(int(local_world), 1)
    out = {"cpus": len(usable), "threads": max(1, len(usable) // local_world), "numa_node": None, "pinned": False}
    mine: List[int] = []
    if device_index is not None:
        try:
            import torch

            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            base = f"/sys/bus/pci/devices/{bdf}"
            node = int(open(f"{base}/numa_node").read())
            cpus = _parse_cpulist(open(f"{base}/local_cpulist").read())
            mine = [c for c in cpus if c in set(usable)]
            out["numa_node"] = node if node >= 0 else None
        except Exception:  # noqa: BLE001 -- no sysfs entry, no such attribute: fall through to the equal slice
            mine = []
    if mine:
        # several GPUs usually share one NUMA node: their ranks share that node's CPUs (the scheduler balances them inside the
        # set), each with its pools capped at the per-rank share
        out["threads"] = max(1, min(out["threads"], len(mine)))
    else:
        per = max(1, len(usable) // local_world)
        mine = usable[(local_rank % local_world) * per: (local_rank % local_world) * per + per] or usable
    if pin and local_world > 1:
        try:
            os.sched_setaffinity(0, mine)
            out["pinned"], out["cpus"] = True, len(mine)
        except (AttributeError, OSError):
            return out
        moved = 0
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            tids = []
        for tid in tids:          # on Linux sched_setaffinity takes a thread id: every thread of this process, not only the caller
            try:
                os.sched_setaffinity(tid, mine)
                moved += 1
            except OSError:       # a thread that exited meanwhile
                pass
        out["threads_pinned"] = moved
    return out


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def init_process_group(backend: Optional[str] = None, timeout_s: Optional[float] = None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract).

    ``timeout_s`` (default: AL_DIST_TIMEOUT_S, else 120 s) bounds every collective: a rank that dies leaves its siblings blocked
    for that long instead of the backend's own default (RCCL: ten minutes)."""
    import os
    from datetime import timedelta

    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device(f"cuda:{local}")
    if timeout_s is None:
        timeout_s = float(os.environ.get("AL_DIST_TIMEOUT_S", "120"))
    dist.init_process_group(backend, timeout=timedelta(seconds=timeout_s), **kwargs)
    return dist


def gather_buffers(local: Dict[int, "object"], n_items: int, dst: int = 0, device=None, to_host: bool = True):
    """Collect per-scene (C, T) float32 buffers on rank ``dst``.

    ``local`` maps scene index -> torch tensor (device or CPU) or ndarray owned by this rank under
    ``shard_indices``.  Buffers may differ in shape between scenes: shapes travel first (one tiny all_gather),
    then every scene moves with one point-to-point send at its exact size (no padding to the round's maximum);
    with RCCL each peer's send uses its own xGMI link into the root.  ``to_host=False`` leaves the collected
    buffers on the root's device (torch tensors) instead of copying each one to a numpy array.
    """
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    backend = dist.get_backend()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))

    def as_tensor(x):
        t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
        return t.to(dev, dtype=torch.float32).contiguous()

    mine = shard_indices(n_items, rank, world)
    rounds = max(-(-n_items // world), 1)
    shapes = torch.zeros((rounds, 2), dtype=torch.int64, device=dev)
    for r, idx in enumerate(mine):
        shapes[r, 0], shapes[r, 1] = local[idx].shape[0], local[idx].shape[1]
    all_shapes = [torch.zeros_like(shapes) for _ in range(world)]
    dist.all_gather(all_shapes, shapes)
    all_shapes = [t.cpu() for t in all_shapes]
    out: Dict[int, object] = {}
    ops, keep = [], []
    if rank == dst:
        for idx in range(n_items):
            owner, r = idx % world, idx // world
            c, t = int(all_shapes[owner][r, 0]), int(all_shapes[owner][r, 1])
            if owner == dst:
                out[idx] = as_tensor(local[idx]).reshape(c, t)
            else:
                out[idx] = torch.empty((c, t), dtype=torch.float32, device=dev)
                if c * t:
                    ops.append(dist.P2POp(dist.irecv, out[idx].view(-1), owner))
    else:
        for idx in mine:
            flat = as_tensor(local[idx]).reshape(-1)
            keep.append(flat)
            if flat.numel():
                ops.append(dist.P2POp(dist.isend, flat, dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    if not to_host:
        return out
    return {i: v.cpu().numpy() for i, v in out.items()}


def render_and_gather_overlapped(render_fn: Callable[[int], "object"], n_items: int, shape, dst: int = 0, device=None,
                                 producer_stream=None):
    """Render this rank's scenes and collect ALL ``n_items`` (C, T) float32 buffers on rank ``dst`` WHILE the rendering goes
    on: a scene is sent the moment its kernels are enqueued (the send is stream-ordered behind them), instead of all of them
    after the last one (``gather_buffers``).  All scenes have the same ``shape`` (BASELINE configs[3]: a batch of equal scenes).

    The root posts its receives up front, one grouped launch per ROUND (round r = scenes r * world .. r * world + world - 1,
    one from every peer, each arriving over that peer's own xGMI link); a pair of ranks matches sends and receives in issue
    order, and every peer sends its scenes in ascending order.  ``render_fn(i)`` returns scene i's buffer on this rank's device
    (a tensor that stays valid until this function returns).  Returns {index: tensor} on ``dst``, None elsewhere.

    Stream contract: RCCL orders a send behind the work ALREADY ENQUEUED ON TORCH'S CURRENT STREAM, nothing else.  Every render
    path of this package enqueues there (TorchMemory.stream() is the current stream; multi-lane batches and graph replays join
    back into it).  A ``render_fn`` that renders on a stream of its own passes it as ``producer_stream`` (a torch.cuda.Stream):
    the current stream then waits for it before every send."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    backend = dist.get_backend()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    c, t = int(shape[0]), int(shape[1])
    out: Dict[int, object] = {}
    reqs, keep = [], []
    if rank == dst:
        for r in range(-(-n_items // world)):
            ops = []
            for owner in range(world):
                idx = r * world + owner
                if idx >= n_items or owner == dst:
                    continue
                out[idx] = torch.empty((c, t), dtype=torch.float32, device=dev)
                ops.append(dist.P2POp(dist.irecv, out[idx].view(-1), owner))
            if ops:
                reqs += dist.batch_isend_irecv(ops)
    for idx in shard_indices(n_items, rank, world):
        buf = render_fn(idx)
        if producer_stream is not None and backend == "nccl":
            torch.cuda.current_stream(dev).wait_stream(producer_stream)
        buf = (torch.from_numpy(np.ascontiguousarray(buf)) if isinstance(buf, np.ndarray) else buf).to(dev, dtype=torch.float32)
        if tuple(buf.shape) != (c, t):
            raise ValueError(f"scene {idx} has shape {tuple(buf.shape)}, the overlapped gather was set up for {(c, t)}")
        if rank == dst:
            out[idx] = buf
        else:
            flat = buf.contiguous().view(-1)
            keep.append(flat)
            reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, dst)])
    for req in reqs:
        req.wait()
    return out if rank == dst else None


def render_scenes(n_scenes: int, render_fn: Callable[[int], "object"], gather: bool = True, dst: int = 0):
    """Render scenes [0, n_scenes) across the process group.

    ``render_fn(i)`` renders scene i on this rank's GPU and returns its (C, T) float32 buffer (device
    tensor or ndarray).  Returns {index: ndarray} on ``dst`` when ``gather`` else this rank's own results.
    """
    import torch.distributed as dist

    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    local = {i: render_fn(i) for i in shard_indices(n_scenes, rank, world)}
    if not gather or world == 1:
        return {i: (v if isinstance(v, np.ndarray) else v.cpu().numpy()) for i, v in local.items()} if world == 1 else local
    return gather_buffers(local, n_scenes, dst=dst)


# ----------------------------------------------------------------------------- one scene, capsules sharded
def capsule_slice(n_capsules: int, rank: int, world_size: int) -> slice:
    """Contiguous capsule rows owned by ``rank`` (IR bytes dominate and are capsule-separable)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(n_capsules, world_size)   # balanced: the first `rem` ranks own one row more
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def prepare_capsule_sharded(renderer, specs, clips, irs_local: np.ndarray, total_capsules: int, sample_rate: float,
                            log2_block: Optional[int] = None):
    """Upload this rank's capsule rows + the (replicated) clips and allocate the workspaces of ``run_capsule_sharded``."""
    import torch.distributed as dist

    from . import plan as planning

    if dist.is_initialized() and dist.get_world_size() > total_capsules:
        # every rank would have to join the collectives with an empty batch; refuse up front on ALL ranks instead
        raise ValueError(f"capsule sharding needs world_size <= capsules ({dist.get_world_size()} > {total_capsules})")
    if irs_local.shape[0] == 0:
        raise ValueError("this rank owns no capsule rows")
    if log2_block is None:
        # the planner's automatic block size depends on the batch's (capsule, block) row count: taken from the WHOLE scene, so that
        # uneven shards cannot land on different sides of its threshold and every rank renders through the same transforms as a
        # single-GPU render of the scene (rank 0's re-render check compares against exactly that)
        log2_block = planning.plan_batch(specs, total_capsules, irs_local.shape[2], sample_rate, lib=renderer.lib).log2_block
    pl = planning.plan_batch(specs, irs_local.shape[0], irs_local.shape[2], sample_rate, log2_block=log2_block, lib=renderer.lib)
    return renderer.prepare(pl, clips, irs_local)


def run_capsule_sharded(renderer, batch, total_capsules: int, timers: Optional[dict] = None):
    """One pass of a prepared capsule-sharded batch: only enqueues (kernels and two stream-ordered EXCHANGES on DEVICE arrays =
    three all-reduce calls: the IR norm sums (SUM), then the level statistics as one SUM and one MAX call, since a collective has
    one reduction operator; nothing is copied to the host and nothing waits).  Every ``al_*`` launch goes to torch's current
    stream and ``dist.all_reduce`` (async_op=False) makes that stream wait for the collective, so kernels and collectives are
    ordered on the device without a host synchronisation.  ``timers``: a dict that receives ``(start, end)`` event
    pairs around each collective (``make_event()`` from the caller under key "make_event"), for bench.py."""
    import ctypes as ct

    import torch
    import torch.distributed as dist

    lib, stream = renderer.lib, renderer.mem.stream()
    desc = batch.descs[0]
    sharded = dist.is_initialized() and dist.get_world_size() > 1

    def as_tensor(buf):   # device tensors as they are; host-emulated memory (CPU tests) through a torch view
        return torch.from_numpy(buf) if isinstance(buf, np.ndarray) else buf

    def timed(key, fn):
        if timers is None:
            return fn()
        a, b = timers["make_event"](), timers["make_event"]()
        a.record()
        fn()
        b.record()
        timers.setdefault(key, []).append((a, b))

    # 1. emitter gains are a mean over ALL capsules (normalize_irs, synthesize.py:404-428): all-reduce the norm sums
    lib.call("al_ir_spectra", ct.byref(desc), stream)
    lib.call("al_emitter_norm_sums", ct.byref(desc), stream)
    if sharded:
        timed("allreduce_ir_norms", lambda: dist.all_reduce(as_tensor(batch.bufs["emitter_gain"]), op=dist.ReduceOp.SUM))
    lib.call("al_emitter_gains_from_sums", ct.byref(desc), int(total_capsules), stream)
    for name in ("al_signal_spectra", "al_spectral_mac", "al_block_synthesis", "al_event_stats"):
        lib.call(name, ct.byref(desc), stream)
    # 2. per-event level statistics {sum|x|, max|x|, non-finite count} over all capsules (synthesize.py:594-599)
    if sharded:
        def exchange():
            t = as_tensor(batch.bufs["event_stats"]).view(-1, 4)
            sums, peak = t[:, [0, 2]].contiguous(), t[:, 1].contiguous()
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            dist.all_reduce(peak, op=dist.ReduceOp.MAX)
            t[:, 0], t[:, 2], t[:, 1] = sums[:, 0], sums[:, 1], peak

        timed("allreduce_event_levels", exchange)
    lib.call("al_event_levels_from_stats", ct.byref(desc), int(total_capsules), stream)
    return batch.result()


def render_capsule_sharded(renderer, specs, clips, irs_local: np.ndarray, total_capsules: int, sample_rate: float,
                           log2_block: Optional[int] = None):
    """Render one microphone of one scene with this rank's capsules only.

    Every rank convolves all events against its own capsule rows of the IR tensor (clips are replicated: small).
    The per-event level (one scalar from sum|x| and max|x| over ALL capsules, synthesize.py:594-599) is the
    only coupling: E x {sum|x|, #non-finite} (one SUM all-reduce) and E x max|x| (one MAX all-reduce) between block
    synthesis and the level law (plus the per-emitter IR norm sums of normalize_irs, synthesize.py:404-428, one SUM
    all-reduce before the accumulate): three small all-reduce calls per scene.
    Returns the RenderResult of the local capsules (event_scale identical on every rank).
    """
    batch = prepare_capsule_sharded(renderer, specs, clips, irs_local, total_capsules, sample_rate, log2_block)
    return run_capsule_sharded(renderer, batch, total_capsules)
