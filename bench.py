#!/usr/bin/env python
"""Headline benchmark: rendered scene-seconds per second (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2]

One "step" = one full pass of the hot path over one synthetic scene of the chosen config (default
cfg2: 60 s scene, 32 capsules, 64 static events, 2 s RIRs, 48 kHz): the equivalents of
render_audio_for_all_scene_events + generate_scene_audio_from_events with clips, IRs and tables
already resident in HBM.  For N > 1 every rank renders its own scene (weak scaling, no data-path
collective; the end-of-job gather is timed separately and rank 0 re-renders two peers' scenes to
check the gathered buffers bit for bit); rank 0 prints ONE JSON line.

Timing: W warm-up steps, then `--repeats` (default 3) repeats of EXACTLY K steps, each repeat bracketed by
barrier + synchronize on both sides, max over ranks per repeat; `ms_per_step` / `value` come from the MEDIAN
repeat (all repeats are in the JSON), so the driver's short `--steps 20` run does not hang on one outlier.
Nothing but the launches is inside those regions; the per-kernel HIP-event durations come from one more
pass of K steps right behind them.

Other modes: `--total-scenes S` (cfg4: S scenes split over the ranks, every one rendered and ALL of them
gathered on rank 0: strong scaling), `--shard capsules` (cfg5: ONE scene, its capsules split over the
ranks, two exchanges (three all-reduce calls) of per-emitter / per-event scalars inside the step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

XGMI_LINK_GBS = 153.0     # one xGMI link, MI355X_MICROARCH.md (7 links per GPU, point-to-point)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md "Chip-level parameters")
METRIC = "rendered scene-seconds/sec @48kHz, 32-ch mic, 64 events, 2s RIR; 1/2/4/8 GPU"


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """(usable physical cores, how that was found).  The CPUs this process may run on (cgroup / affinity mask, not the machine's
    count) divided by the SMT width the kernel reports for cpu0 (thread_siblings_list)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    smt, how = 1, "no SMT information: every usable CPU counted as a core"
    try:
        with open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list") as fh:
            sib = fh.read().strip()
        n = 0
        for part in sib.split(","):
            lo, _, hi = part.partition("-")
            n += (int(hi) - int(lo) + 1) if hi else 1
        smt = max(n, 1)
        how = f"{usable} usable CPUs (sched_getaffinity) / {smt} hardware threads per core (cpu0 thread_siblings_list = {sib})"
    except (OSError, ValueError):
        pass
    return max(usable // smt, 1), how


def under_profiler() -> bool:
    """rocprofv3 (or any tool that preloads a library which initialises the GPU before main() runs): a process in that state
    must not start children -- on this pool an exec from a GPU-initialised process takes the machine down."""
    env = os.environ
    return ("rocprof" in env.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in env)
            or "rocprofiler" in env.get("HSA_TOOLS_LIB", "").lower())


def cpu_quota():
    """What the container is ALLOWED to use, beside the CPUs it can see: cgroup v2 cpu.max ("max 100000" = no limit,
    "<quota> <period>" = quota / period CPUs) or the v1 pair; plus the memory limit.  A quota below the visible CPU count turns
    "one process per physical core" into time-slicing, which is the first thing to rule out when per-process times explode."""
    out = {"cpu_max": None, "cpus_allowed_by_quota": None, "memory_max": None, "visible_cpus": os.cpu_count()}
    try:
        with open("/proc/self/cgroup") as fh:
            rel = [ln.strip().split(":", 2)[2] for ln in fh if ln.startswith("0::")]
        roots = ["/sys/fs/cgroup" + (rel[0] if rel else ""), "/sys/fs/cgroup"]
    except OSError:
        roots = ["/sys/fs/cgroup"]
    for root in roots:
        try:
            raw = open(os.path.join(root, "cpu.max")).read().split()
            out["cpu_max"] = " ".join(raw)
            out["cpus_allowed_by_quota"] = None if raw[0] == "max" else float(raw[0]) / float(raw[1])
            break
        except (OSError, ValueError, IndexError):
            continue
    if out["cpu_max"] is None:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cpu_max"] = f"{q} {p_} (cgroup v1)"
            out["cpus_allowed_by_quota"] = None if q < 0 else q / p_
        except (OSError, ValueError):
            pass
    for root in roots:
        try:
            out["memory_max"] = open(os.path.join(root, "memory.max")).read().strip()
            break
        except OSError:
            continue
    return out


def _pressure():
    """/proc/pressure/{cpu,memory} "some avg10" (percent of time some task waited for the resource), None where PSI is off."""
    out = {}
    for res in ("cpu", "memory", "io"):
        try:
            line = open(f"/proc/pressure/{res}").readline()
            out[res] = float(line.split("avg10=")[1].split()[0])
        except (OSError, IndexError, ValueError):
            out[res] = None
    return out


def committed_workers_sweep(config):
    """The worker-count sweep of the all-cores leg as measured ONCE on a pool box (bench.py --cpu-workers-sweep, committed under
    profiles/): why this leg runs one process per CPU of the cgroup quota and not per visible core.  Not re-measured per run."""
    path = os.path.join(ROOT, "profiles", "r05b_cpu_workers_sweep.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get("config") != config:
        return None
    return {"source": "profiles/r05b_cpu_workers_sweep.json (one pool box, 6 of 64 events per scene, not re-measured in this run)",
            "cpu_quota_there": rec["cpu_quota"]["cpus_allowed_by_quota"], "knee_workers": rec["knee_workers"],
            "scene_seconds_per_s_by_workers": {str(p_["workers"]): round(p_["scene_seconds_per_s"], 1) for p_ in rec["workers_sweep"]
                                               if p_["padded_copies"] and not p_["malloc_huge_pages"]},
            "without_padded_copies": {str(p_["workers"]): round(p_["scene_seconds_per_s"], 1) for p_ in rec["workers_sweep"]
                                      if not p_["padded_copies"]}}


def _cpu_scene_worker(job):
    """ONE whole scene through the oracle in a worker process, the way the reference runs it (synthesize.py:613-677 then
    :314-401): every event rendered, added into the float32 scene buffer, and the reference's per-event full-scene padded copy
    (:381-383) built and dropped.  Input generation is outside the timing.  max_events / n_irs_cap bound the sample for the
    configurations whose full scene takes a core minutes or hours (cfg3, cfg5): then the time is scaled in events x IRs."""
    config, index, scale, max_events, n_irs_cap = job[:5]
    keep_padded = job[5] if len(job) > 5 else True
    from audiblelight_amd import synthetic
    from oracle import synth_oracle as orc

    over = {"E": max_events} if max_events else {}
    sc = synthetic.make_scene(config, scene_index=index, scale=scale, **over)
    gains = sc.gain_db
    t0 = time.perf_counter()
    n_scene = round(sc.duration * sc.sr)
    scene = np.zeros((sc.n_capsules, n_scene), dtype=np.float32)
    work = 0
    for i, sp in enumerate(sc.specs):
        n_used = min(sp.n_emitters, n_irs_cap) if sp.is_moving else sp.n_emitters
        h = sc.irs[:, sp.emitter0: sp.emitter0 + n_used, :].astype(np.float64)
        clip = sc.clips[i]
        if gains is not None:
            clip = orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(clip, gains[i]))).astype(np.float32)
        x = orc.render_event(clip, h, sp.snr, sp.ref_db, sp.is_moving, sp.duration, sc.sr)["spatial"]
        a, b = orc.event_slot(sc.starts[i], sc.ends[i], sc.sr, n_scene)
        if b > a:
            piece = orc.fit_length(x, b - a)
            scene[:, a:b] += piece
            if keep_padded:
                padded = np.zeros_like(scene)            # event._spatial_audio_padded[mic]
                padded[:, a:b] += piece
                del padded
        work += max(n_used, 1)
    orc.check_audio(scene)
    return time.perf_counter() - t0, work


def cpu_baseline_all_cores(config, scale, workers, how, full_events, full_work, n_irs_cap=8, max_events=None, keep_padded=True):
    """the oracle on every physical core of the host, ONE PROCESS PER SCENE (scenes are independent; the
    reference's dataset loop is serial, scripts/generate/benchmark.py:44-77): `workers` different scenes rendered at once,
    mixdown and per-event padded copies included.  rate = scenes x scene seconds / wall time of the slowest worker."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    from audiblelight_amd import synthetic

    cfg = synthetic.CONFIGS[config]
    if max_events is None:
        # measured in full once (profiles/r04a_bench_cfg2_allcores_full.json: 128 cores x one whole cfg2 scene each took 400-536 s
        # per scene against 14 s for one process alone -- 128 processes share the memory system -- i.e. 14.3 scene-s/s); the
        # default run renders 6 of the 64 events per scene, every core busy for the whole sample, and scales by the event count
        max_events = {"cfg2": 6, "cfg3": 1, "cfg4": 6, "cfg5": 2}.get(config, 0)
    max_events = min(max_events, cfg["E"]) if max_events else 0
    duration = cfg["T"] * scale
    jobs = [(config, 10_000 + i, scale, max_events, n_irs_cap, keep_padded) for i in range(workers)]
    before = _pressure()
    # spawn, not fork: a broken worker raises instead of hanging.  Started BEFORE this process initialises the GPU.
    with ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as pool:
        t0 = time.perf_counter()
        results = list(pool.map(_cpu_scene_worker, jobs))
        wall_with_inputs = time.perf_counter() - t0
    slowest = max(t for t, _ in results)
    sample_work = results[0][1]
    extrapolated = sample_work < full_work
    per_scene_s = slowest * full_work / max(sample_work, 1)
    return dict(value=workers * duration / per_scene_s, value_extrapolated=bool(extrapolated), unit="scene-seconds/s", cores=workers,
                host_cpus=os.cpu_count(), kind="port",
                cpu_model=cpu_model(), core_count_from=how, extrapolated=extrapolated, cpu_quota=cpu_quota(),
                workers_sweep=committed_workers_sweep(config),
                pressure_avg10_before_after={"before": before, "after": _pressure()}, per_event_padded_copies=bool(keep_padded),
                seconds_per_scene_per_core=[round(min(t for t, _ in results), 2), round(slowest, 2)],
                wall_s_including_input_generation=round(wall_with_inputs, 1),
                sample=f"{workers} different {config} scenes at once, one oracle process per scene on one physical core each: "
                       + (f"all {full_events} events" if not extrapolated else
                          f"{max_events} of {full_events} events" + (f" with {n_irs_cap} IRs each" if config == "cfg3" else "")
                          + ", scaled linearly in events x IRs (--cpu-scene-events 0 renders whole scenes: minutes)")
                       + " + float32 mixdown + the reference's per-event padded copies; rate from the slowest worker")


def cpu_workers_sweep(args):
    """The all-cores leg at several worker counts (one whole-scene oracle process per worker, `--cpu-scene-events` events each),
    with and without the per-event full-scene padded copy of synthesize.py:381-383.  Reports the aggregate rate per point, the
    per-worker seconds, the container's CPU quota and the knee: the first count whose aggregate rate gains less than half of
    what a linear continuation from the previous point would."""
    from audiblelight_amd import synthetic

    cores, how = physical_cores()
    cfg = synthetic.CONFIGS[args.config]
    counts = [int(x) for x in args.cpu_workers_sweep.split(",") if x.strip()]
    events = None if args.cpu_scene_events < 0 else (args.cpu_scene_events or cfg["E"])
    points = []
    # third variant, at the largest count only: glibc malloc asking for transparent huge pages (GLIBC_TUNABLES, inherited by the
    # spawned workers) -- every event allocates and frees hundreds of MB of float64 temporaries, i.e. ~10^5 fresh 4 KiB page
    # faults per event per process; if the per-process slowdown is the kernel's page allocator under 128 faulting processes,
    # 2 MiB pages remove it
    variants = [(True, False, counts), (False, False, counts), (True, True, counts[-1:])]
    for padded, thp, which in variants:
        for n in which:
            saved = os.environ.get("GLIBC_TUNABLES")
            if thp:
                os.environ["GLIBC_TUNABLES"] = "glibc.malloc.hugetlb=1"
            try:
                rec = cpu_baseline_all_cores(args.config, args.scale, n, how, cfg["E"], cfg["E"] * cfg["N"], max_events=events,
                                             keep_padded=padded)
            finally:
                if thp and saved is None:
                    os.environ.pop("GLIBC_TUNABLES", None)
                elif thp:
                    os.environ["GLIBC_TUNABLES"] = saved
            points.append({"workers": n, "padded_copies": padded, "malloc_huge_pages": thp, "scene_seconds_per_s": rec["value"],
                           "seconds_per_scene_per_worker_min_max": rec["seconds_per_scene_per_core"], "wall_s": rec["wall_s_including_input_generation"],
                           "pressure": rec["pressure_avg10_before_after"]["after"], "sample": rec["sample"]})
    knee = None
    with_pad = [p_ for p_ in points if p_["padded_copies"] and not p_["malloc_huge_pages"]]
    for a, b in zip(with_pad, with_pad[1:]):
        linear = a["scene_seconds_per_s"] * b["workers"] / a["workers"]
        if b["scene_seconds_per_s"] - a["scene_seconds_per_s"] < 0.5 * (linear - a["scene_seconds_per_s"]):
            knee = b["workers"]
            break
    return {"workers_sweep": points, "cpu_quota": cpu_quota(), "knee_workers": knee, "physical_cores": cores, "core_count_from": how,
            "cpu_model": cpu_model(), "config": args.config, "scale": args.scale}


def oracle_clip(scene, i):
    """The clip the oracle convolves: cfg5's raw clips go through its [Gain, Invert] chain + peak normalisation."""
    from oracle import synth_oracle as orc

    if scene.gain_db is None:
        return scene.clips[i]
    return orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(scene.clips[i], scene.gain_db[i]))).astype(np.float32)


PARITY_TOL = 1e-4   # BASELINE.json north_star / SURVEY 8(d) "Parity metric": relative RMS AND max-abs / max|ref|, fp32 vs the fp64 oracle


def parity_record(got, ref, events, note):
    """Both halves of the contract's parity bound between a GPU scene buffer and the oracle's (C, T) float32 scene."""
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err2, ref2, worst, peak = 0.0, 0.0, 0.0, 0.0
    for c in range(ref.shape[0]):          # row by row: no third scene-sized temporary in float64
        r64, d = ref[c].astype(np.float64), got[c].astype(np.float64) - ref[c]
        err2 += float(np.dot(d, d))
        ref2 += float(np.dot(r64, r64))
        worst, peak = max(worst, float(np.max(np.abs(d)))), max(peak, float(np.max(np.abs(r64))))
    rel = (err2 / ref2) ** 0.5 if ref2 > 0 else float(err2 > 0)
    mx = worst / peak if peak > 0 else float(worst > 0)
    return {"rel_rms": rel, "max_abs_over_peak": mx, "rows": int(ref.shape[0]), "samples": int(ref.shape[1]), "events": int(events),
            "tol": PARITY_TOL, "ok": bool(rel <= PARITY_TOL and mx <= PARITY_TOL and np.isfinite(rel) and np.isfinite(mx)),
            "reference": "oracle/synth_oracle.py (float64 restatement of synthesize.py:613-677 + :314-401, pinned to the reference's goldens)",
            "note": note}


def sampled_row(e: int, n_capsules: int) -> int:
    """The capsule row of event e that meets the oracle when the event is not compared in full (a fixed pseudo-random choice)."""
    return int((e * 2654435761 + 40503) % (1 << 32) % n_capsules)


def oracle_row_samples(scene, result, events, ir_rows_of, gains_of):
    """ONE pseudo-random capsule row of every event in `events`, the GPU's UNSCALED render (downloaded alone: an event is 25-50 MB)
    against the float64 oracle's row of the same convolution (synthesize.py:71-106 static, :277-310 moving) -- the level law on top
    of it needs every row, and is held by the A9 invariant checked here too from the device's own statistics (synthesize.py:594-599):
    mean|scale_e * x_e| = 10^((ref_db + snr) / 20).  So every event of a scene the oracle cannot render in full still meets it.
    ir_rows_of(e, c) -> (N_e, Lir) IR rows of capsule c; gains_of(e) -> (N_e,) normalize_irs gains over ALL capsules.
    Returns the record merged into `parity` (worst row: both halves of the bound)."""
    from scipy.signal import fftconvolve

    from oracle import synth_oracle as orc

    t0 = time.perf_counter()
    worst_rms, worst_max, worst_level, rows = 0.0, 0.0, 0.0, {}
    scales, stats = result.scales(), result.stats()
    for e in events:
        sp = scene.specs[e]
        c = sampled_row(e, scene.n_capsules)
        clip = np.asarray(oracle_clip(scene, e), dtype=np.float64)
        h = np.asarray(ir_rows_of(e, c), dtype=np.float64) * np.asarray(gains_of(e), dtype=np.float64)[:, None]
        if sp.is_moving:
            want = orc.fit_length(orc.convolve_moving(clip, h[None], sp.duration, scene.sr), sp.n_samples)[0]
        else:
            want = fftconvolve(clip, h[0])[: sp.n_samples]
        ev = result.plan.events[e]
        n = int(ev["len"])
        off = int(ev["out_off"]) + c * n
        got = np.asarray(result.memory.download(result.spatial[off: off + n]), dtype=np.float64)
        d = got - want
        rms, peak = float(np.sqrt(np.mean(want ** 2))), float(np.max(np.abs(want)))
        worst_rms = max(worst_rms, float(np.sqrt(np.mean(d ** 2))) / rms if rms > 0 else float(np.any(d != 0)))
        worst_max = max(worst_max, float(np.max(np.abs(d))) / peak if peak > 0 else float(np.any(d != 0)))
        level = scales[e] * stats[e, 0] / (scene.n_capsules * sp.n_samples) / 10 ** ((sp.ref_db + sp.snr) / 20)
        worst_level = max(worst_level, abs(float(level) - 1.0))
        rows[str(e)] = c
    ok = bool(worst_rms <= PARITY_TOL and worst_max <= PARITY_TOL and worst_level <= 1e-4 and np.isfinite([worst_rms, worst_max, worst_level]).all())
    return {"events": len(rows), "rel_rms_worst_row": worst_rms, "max_abs_over_peak_worst_row": worst_max,
            "level_invariant_worst_rel_err": worst_level, "ok": ok, "seconds": round(time.perf_counter() - t0, 1), "row_of_event": rows,
            "note": "one pseudo-random capsule row of every event NOT compared in full: the unscaled GPU row vs the float64 oracle's row "
                    "of the same convolution (normalize_irs gains over all capsules), plus the A9 level invariant from the device statistics"}


def merge_row_samples(parity, sampled):
    """`parity` (first n events, all rows x all samples) + `sampled` (one row of every other event): parity.events counts BOTH."""
    parity["events_in_full"] = parity["events"]
    parity["events"] = parity["events"] + sampled["events"]
    parity["rows_sampled"] = sampled
    parity["ok"] = bool(parity["ok"] and sampled["ok"])
    return parity


def oracle_partial_scene(scene, n_events, irs_of=None):
    """The oracle's float32 (C, T) mix of the first `n_events` events of a synthetic scene, every IR of every event (no cap).
    irs_of(e) -> (C, N_e, Lir) float array of event e's IR columns when the scene's tensor lives on the device."""
    from oracle import synth_oracle as orc

    spatials = []
    for i in range(n_events):
        sp = scene.specs[i]
        h = irs_of(i) if irs_of is not None else scene.irs[:, sp.emitter0: sp.emitter0 + sp.n_emitters, :]
        spatials.append(orc.render_event(oracle_clip(scene, i), np.asarray(h, dtype=np.float64), sp.snr, sp.ref_db, sp.is_moving,
                                         sp.duration, scene.sr)["spatial"])
    return orc.mix_scene(spatials, list(zip(scene.starts[:n_events], scene.ends[:n_events])), scene.duration, scene.sr,
                         keep_padded=False)["scene"]


def gpu_partial_scene(r, scene, pl, result, n_events):
    """The GPU's mixdown of the first `n_events` events alone (no ambience) as a host (C, T) float32 array: what the oracle's
    bounded sample is compared with when it does not cover the whole scene."""
    from audiblelight_amd import plan as planning

    idx = list(range(n_events))
    mp = planning.plan_mixdown(scene.starts[:n_events], scene.ends[:n_events], [len(c) for c in scene.clips[:n_events]],
                               [scene.n_capsules] * n_events, pl.events["out_off"][:n_events], idx, scene.duration, scene.sr,
                               scene.n_capsules, lib=r.lib)
    dev = r.prepare_mixdown(mp, result, []).run()
    return r.mem.download(dev)[: scene.n_capsules * mp.n_samples].reshape(scene.n_capsules, mp.n_samples)


def cpu_baseline(scene, n_events: int, n_irs_cap: int = 8, keep_scene: bool = False):
    """Time the float64 numpy/scipy oracle (kind "port") on a bounded sample of the same workload.  keep_scene: also return the
    oracle's mixed (C, T) float32 scene of those events (None when the IR cap changed a moving event): (record, scene)."""
    from oracle import synth_oracle as orc

    t0 = time.perf_counter()
    spatials, work, full_work = [], 0, 0
    for i in range(n_events):
        sp = scene.specs[i]
        n_used = min(sp.n_emitters, n_irs_cap) if sp.is_moving else sp.n_emitters
        h = scene.irs[:, sp.emitter0: sp.emitter0 + n_used, :].astype(np.float64)
        spatials.append(orc.render_event(oracle_clip(scene, i), h, sp.snr, sp.ref_db, sp.is_moving, sp.duration, scene.sr)["spatial"])
        work += n_used
    for sp in scene.specs:
        full_work += max(sp.n_emitters, 1)
    t_events = time.perf_counter() - t0
    t0 = time.perf_counter()
    mixed = orc.mix_scene(spatials, list(zip(scene.starts[:n_events], scene.ends[:n_events])), scene.duration, scene.sr,
                          keep_padded=True)
    t_mix = time.perf_counter() - t0
    ref_scene = mixed["scene"]
    del mixed, spatials
    total = t_events * full_work / max(work, 1) + t_mix * len(scene.specs) / n_events
    moving = any(sp.is_moving for sp in scene.specs)
    capped = any(sp.is_moving and sp.n_emitters > n_irs_cap for sp in scene.specs[:n_events])
    rec = dict(value=scene.duration / total, unit="scene-seconds/s", cores=1, kind="port", cpu_model=cpu_model(),
                extrapolated=n_events < len(scene.specs) or moving,
                sample=f"{n_events} of {len(scene.specs)} events of one {scene.name} scene"
                       + (f" with {n_irs_cap} of {scene.specs[0].n_emitters} IRs each" if moving else "")
                       + f" (oracle render_event + mixdown incl. per-event padded copies): {t_events + t_mix:.1f} s measured"
                       + (", scaled linearly in events x IRs" if n_events < len(scene.specs) or moving else ""))
    return (rec, None if capped else ref_scene) if keep_scene else rec


VALU_PEAK_TFLOPS = 157.3   # MI355X vector fp32 (MI355X_MICROARCH.md "Chip-level parameters": 256 CUs x 4 SIMD-32 x 2 flop x 2.4 GHz)
VALU_ISSUE_PER_S = 256 * 4 * 2.4e9 / 2.0   # wave64 VALU instructions the chip can issue per second: one per 2 cycles per SIMD-32
                                           # with two or more waves on the SIMD (same guide, `v_fma_f32 (wave64): 2 cyc`)


def stage_flops(pl, mix_plan):
    """Floating-point operations one step performs per C-ABI stage, counted from the PLAN (not the monolithic-FFT convention of
    SURVEY 8d's 107 GFLOP): a 2B-sample real transform = 2.5 N log2 N with N = 2B; one complex multiply-accumulate = 8 flops per
    bin; the transforms / products of trimmed IR partitions (al_batch.emitter_parts) and of blocks past a clip's end are not
    counted because they are not made."""
    import math

    B, C, P = pl.block, pl.n_capsules, pl.n_partitions
    fft = 2.5 * (2 * B) * math.log2(2 * B)
    parts = pl.emitter_parts() if len(pl.events) else None
    live = np.full(max(pl.n_emitters, 1), P, dtype=np.int64) if parts is None else parts.astype(np.int64)
    n_h = int(live[: pl.n_emitters].sum()) * C
    st, ev = pl.streams, pl.events
    conv = ev["n_streams"][st["event"]] > 0 if len(st) else np.zeros(0, bool)
    n_x = int(st["n_j"][conv].sum()) if len(st) else 0
    products = 0
    for s_ in st[conv] if len(st) else []:
        K, p_live = int(ev["n_blocks"][s_["event"]]), int(live[s_["emitter"]])
        j = s_["j_lo"] + np.arange(int(s_["n_j"]))
        products += int(np.clip(K - j, 0, p_live).sum())       # partitions p < p_live with j + p < K
    n_y = int((C * ev["n_blocks"][ev["n_streams"] > 0]).sum())
    mixed = int((mix_plan.slot_count.astype(np.int64) * mix_plan.slot_rows).sum()) if mix_plan is not None else 0
    return {"al_forward_spectra": (n_h + n_x) * fft, "al_spectral_mac": products * C * B * 8.0, "al_block_synthesis": n_y * fft,
            "al_mixdown": 2.0 * mixed}


def secondary_roofline(pl, mix_plan, kernel_ms, valu_insts, scene_traffic, algo_bytes, dominant):
    """SURVEY 8(d) "Bound", the honest ceiling beside the HBM fraction: achieved GFLOP/s per stage (plan-counted flops / live HIP
    event duration) against the vector fp32 peak, the fraction of the chip's VALU issue slots each stage uses (SQ_INSTS_VALU
    from the committed --pmc pass x 2 cycles per wave64 instruction; packed-f32 instructions occupy two slots, so this is a
    lower bound for the accumulate), and how many bytes the step moves per algorithmic byte."""
    flops = stage_flops(pl, mix_plan)
    gf = {k: v / (kernel_ms[k] * 1e-3) / 1e9 for k, v in flops.items() if kernel_ms.get(k)}
    out = {"gflops": gf.get(dominant), "gflops_by_stage": {k: round(v, 1) for k, v in gf.items()},
           "flops_per_step": float(sum(flops.values())), "valu_peak_gflops": VALU_PEAK_TFLOPS * 1e3,
           "valu_frac_of_peak": (gf.get(dominant) or 0.0) / (VALU_PEAK_TFLOPS * 1e3) or None,
           "valu_issue_frac": None, "valu_issue_frac_by_stage": None,
           "traffic_ratio": (scene_traffic / algo_bytes) if scene_traffic else None}
    if valu_insts:
        by = {k: valu_insts[k] / VALU_ISSUE_PER_S / (kernel_ms[k] * 1e-3) for k in valu_insts if kernel_ms.get(k) and valu_insts[k]}
        out["valu_issue_frac"] = by.get(dominant)
        out["valu_issue_frac_by_stage"] = {k: round(v, 4) for k, v in by.items()}
    return out


def load_pmc_traffic(config: str, log2_block: int):
    """(HBM bytes per launch per stage, note) from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json).
    The table carries the hash of the kernel sources it was measured on; a table from another build is refused."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_traffic.json missing"
    try:
        table = json.load(open(path))
    except Exception as exc:  # noqa: BLE001
        return None, f"unreadable pmc_traffic.json: {exc}"
    if table.get("source_hash") != source_hash():
        return None, (f"pmc_traffic.json was measured on kernel sources {table.get('source_hash')}, this build is "
                      f"{source_hash()}: re-run profiles/tools/collect_profiles.sh")
    return table.get(f"{config}/log2_block={log2_block}"), f"rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE, {table.get('collected', '')}"


def load_pmc_valu(config: str, log2_block: int):
    """SQ_INSTS_VALU per launch per stage from the same committed table (third --pmc pass), None when absent or stale."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None
    if table.get("source_hash") != source_hash():
        return None
    return table.get(f"{config}/log2_block={log2_block}/valu_insts")


def dropin_leg(scene, renderer, n: int, ctx=None):
    """The reference's own call sequence on this package's objects: a Scene built from host numpy clips + IRs,
    ``Scene.generate()`` = render_audio_for_all_scene_events + generate_scene_audio_from_events (core.py:1828-1847),
    ``scene.audio[mic]`` back as a host ndarray.  Synchronous, one scene at a time, PCIe both ways inside the timing."""
    from audiblelight_amd import augmentation as aug, core, synthesize as syn

    syn.set_renderer(renderer)

    def one():
        sc = core.Scene(scene.duration, core.StaticIRState({"mic000": scene.irs}), sample_rate=scene.sr, ref_db=-65)
        for i, (clip, sp) in enumerate(zip(scene.clips, scene.specs)):
            fx = [aug.Gain(scene.sr, gain_db=scene.gain_db[i]), aug.Invert(scene.sr)] if scene.gain_db is not None else []
            sc.add_event(core.Event(f"e{i}", clip, scene.sr, snr=sp.snr, scene_start=scene.starts[i], augmentations=fx))
        audio = sc.generate()["mic000"]
        assert audio.shape == (scene.n_capsules, round(scene.duration * scene.sr)) and audio.dtype == np.float32
        return audio

    world = ctx["world"] if ctx else 1
    try:
        for _ in range(2):
            one()
        if ctx:
            ctx["barrier"]()        # N > 1: every rank makes its calls at the same time (shared host DRAM / PCIe / NUMA)
        calls = []
        for _ in range(n):
            t0 = time.perf_counter()
            one()
            calls.append(time.perf_counter() - t0)
        if ctx:
            ctx["barrier"]()
        dt = float(np.median(calls))       # a synchronous host-side call sequence: the median call, the spread beside it
    finally:
        syn.set_renderer(None)
    out = {"value": scene.duration / dt, "unit": "scene-seconds/s", "ms_per_scene": dt * 1e3, "scenes": n,
           "ms_per_scene_min_max": [round(min(calls) * 1e3, 2), round(max(calls) * 1e3, 2)],
           "note": "Scene.generate() per scene, synchronous: host float32 clips + IR tensor in (H2D), render, mixdown, "
                   "scene.audio out as a host ndarray (D2H); NOT the headline value"}
    if world > 1:
        by_rank = ctx["all_ranks"](out["value"])
        out.update(value=float(np.sum(by_rank)), value_by_rank=[round(v, 1) for v in by_rank], ranks_concurrent=world,
                   ms_per_scene_this_rank=out["ms_per_scene"], ms_per_scene=scene.duration / float(np.sum(by_rank)) * 1e3,
                   note=out["note"] + "; every rank calls at the same time: `value` = sum of the ranks' own rates")
    return out


def switches_in_effect() -> dict:
    from audiblelight_amd import switches

    return switches.current().non_default()


def strip_c_comments(text: str) -> str:
    """C / C++ source without its comments and with every run of white space collapsed to one blank (string and character
    literals are kept as they are): two sources with the same result compile to the same code object."""
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch in "\"'":                                     # literal: copy up to the matching quote, escapes included
            j = i + 1
            while j < n and text[j] != ch:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i: j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            while j > 0 and text[j - 1] == "\\":            # a line comment continued by a backslash
                j = text.find("\n", j + 1)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(ch)
            i += 1
    return " ".join("".join(out).split())


def source_hash() -> str:
    """Hash of the kernel sources AS THE COMPILER SEES THEM (comments and white space stripped): ties profiles/pmc_traffic.json to the
    build it was measured on without a comment edit invalidating four configs x three PMC passes of GPU time."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "audiblelight_amd", "csrc")
    for name in sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h"))):
        h.update(name.encode())
        h.update(strip_c_comments(open(os.path.join(csrc, name), encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


def dist_timeout():
    """Timeout of every collective of a multi-rank run (both init_process_group calls; AL_DIST_TIMEOUT_S overrides): a rank that
    dies leaves its siblings blocked for THIS long, not for the backend's default (RCCL: ten minutes)."""
    from datetime import timedelta

    return timedelta(seconds=float(os.environ.get("AL_DIST_TIMEOUT_S", "120")))


def spawn_ranks(n: int, poll_s: float = 0.2, grace_s: float = 5.0) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process
    touches the GPU, with the same environment torch.distributed.run would give them; rank 0 prints the JSON line.

    The children are POLLED: the first one that exits non-zero (out of memory, a failed parity assert, a killed process) ends
    the job -- the others are terminated (SIGTERM, SIGKILL after `grace_s`; they are children of a parent that never touched
    the GPU, nothing is re-executed) and that exit code is returned, instead of waiting rank by rank for processes that sit in a
    collective until the backend's watchdog fires."""
    import signal
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # the build / GPU boxes export this already (their host driver only supports dmabuf IPC; without it RCCL fails with
        # `hipIpcGetMemHandle: invalid argument`): kept for environments assembled by hand
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    failed = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = next((c for c in codes if c not in (None, 0)), 0)
            if failed or all(c is not None for c in codes):
                break
            time.sleep(poll_s)
    except KeyboardInterrupt:
        failed = 130
    if failed:
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.send_signal(signal.SIGTERM)
        deadline = time.monotonic() + grace_s
        for p in alive:
            try:
                p.wait(timeout=max(deadline - time.monotonic(), 0.1))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        dead = [r for r, c in enumerate(codes) if c not in (None, 0)]
        print(f"bench.py: rank {dead[0] if dead else '?'} exited with code {failed}; terminated {len(alive)} remaining rank(s)",
              file=sys.stderr, flush=True)
        return failed if failed > 0 else 128 - failed     # a signal's negative return code as the shell would report it
    return 0


def make_timers(emulate, torch):
    class HostEvent:   # emulation only
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    def new_event():
        return HostEvent() if emulate else torch.cuda.Event(enable_timing=True)

    def device_sync():
        if not emulate:
            torch.cuda.synchronize()

    return new_event, device_sync


def timed_repeats(step, steps, warmup, repeats, barrier, device_sync, reduce_max):
    """W warm-up steps, then `repeats` x (barrier, K steps, synchronize, barrier); per repeat the MAX over ranks.
    Returns (seconds per repeat after the MAX, this rank's own seconds per repeat)."""
    for _ in range(warmup):
        step()
    out, own = [], []
    for _ in range(max(repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        for _k in range(steps):
            step()
        device_sync()
        elapsed = time.perf_counter() - t0
        barrier()
        own.append(elapsed)
        out.append(reduce_max(elapsed))
    return out, own


def make_renderer(emulate):
    from audiblelight_amd import engine

    if emulate:
        from audiblelight_amd import _hip
        from tests import hostemu

        return engine.Renderer(lib=_hip.Library(hostemu.build()), memory=hostemu.NumpyMemory())
    return engine.Renderer()   # raises without the HIP extension or a GPU: no fallback


def resident_scene(r, scene, args):
    """Inputs, tables and workspaces of one scene in HBM; returns (batch, mix, mix_plan, plan)."""
    from audiblelight_amd import plan as planning

    pl = planning.plan_batch(scene.specs, scene.n_capsules, scene.ir_len, scene.sr, log2_block=args.log2_block)
    if scene.irs is None:   # IR tensor drawn on the device (other_configs leg): handed over as a device buffer + strides
        c, n, l = scene.ir_shape
        assert l % 4 == 0
        batch = r.prepare(pl, scene.sources(), scene.irs_dev, ir_strides=(n * l, l), chunk_events=args.chunk_events, lanes=args.lanes)
    else:
        batch = r.prepare(pl, scene.sources(), scene.irs, chunk_events=args.chunk_events, lanes=args.lanes)
    n_ev = len(scene.clips)
    mix_plan = planning.plan_mixdown(scene.starts, scene.ends, [len(c) for c in scene.clips], [scene.n_capsules] * n_ev,
                                     pl.events["out_off"], list(range(n_ev)), scene.duration, scene.sr, scene.n_capsules)
    ambience = []
    if scene.ambience_beta is not None:  # cfg5: white ambience drawn and scaled on the device, once (inputs resident)
        from audiblelight_amd import ambience as amb
        from audiblelight_amd.synthesize import _ambience_on_device

        a = amb.Ambience(channels=scene.n_capsules, duration=scene.duration, alias="bench", noise=scene.ambience_beta,
                         ref_db=-65, sample_rate=scene.sr, rng="device")
        ambience = [_ambience_on_device(r, a, (scene.n_capsules, mix_plan.n_samples))]
    mix = r.prepare_mixdown(mix_plan, batch.result(), ambience)
    return batch, mix, mix_plan, pl


def scene_tensor(mix, scene, mix_plan, emulate, torch):
    t = mix.scene[: scene.n_capsules * mix_plan.n_samples].reshape(scene.n_capsules, -1)
    return torch.from_numpy(t) if emulate else t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps per repeat")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=3, help="repeats of the K timed steps; the median is reported")
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; reported in config)")
    ap.add_argument("--log2-block", type=int, default=None)
    ap.add_argument("--chunk-events", type=int, default=0, help="events per chunk (0 = whole scene in one batch)")
    ap.add_argument("--lanes", type=int, default=1, help="workspaces / HIP streams the chunks alternate over")
    ap.add_argument("--total-scenes", type=int, default=0, metavar="S",
                    help="batch mode (BASELINE configs[3]: --config cfg4 --total-scenes 256): one step = ALL S scenes, split over the "
                         "ranks; every scene is rendered and all of them are gathered on rank 0 after the timed region")
    ap.add_argument("--shard", default="scenes", choices=["scenes", "capsules"],
                    help="capsules: ONE scene per step, its capsule rows split over the ranks (SURVEY 8e row 2; cfg5 secondary)")
    ap.add_argument("--cpu-events", type=int, default=None,
                    help="events timed for the CPU baseline (0 = skip; default: the whole scene for cfg2 = 18 s on one core, "
                         "a bounded sample for the larger configurations)")
    ap.add_argument("--parity-events", type=int, default=-1, metavar="N",
                    help="events of the timed scene compared with the float64 oracle, all capsule rows x all samples (JSON `parity`; "
                         "exit code 1 above 1e-4).  -1: the whole scene for cfg2 / cfg4 (the oracle scene of the cpu_baseline leg is "
                         "reused), 1 event for cfg3, 2 for cfg5; 0: skip")
    ap.add_argument("--cpu-workers", type=int, default=-1, metavar="N",
                    help="also time the oracle on N host processes at once, one whole scene each (all-cores CPU figure; -1 = one per "
                         "PHYSICAL core, bounded by free memory; 0 = skip; skipped by itself under a profiler)")
    ap.add_argument("--cpu-workers-sweep", default="", metavar="N1,N2,...",
                    help="CPU-only mode (no GPU is touched): the all-cores oracle leg at each of these worker counts, with and without "
                         "the reference's per-event padded copies, one JSON line {workers_sweep, cpu_quota, knee}; what explains "
                         "profiles/r04a's 30x per-process slowdown at 128 workers")
    ap.add_argument("--cpu-scene-events", type=int, default=-1, metavar="N",
                    help="events of every scene the all-cores leg renders (-1: a bounded sample, 6 for cfg2; 0: the whole scene, "
                         "which takes 128 busy cores about nine minutes at cfg2)")
    ap.add_argument("--end-to-end", type=int, default=None, metavar="N",
                    help="also run N scenes through the pipelined batch driver from HOST buffers (PCIe-inclusive rate; with "
                         "--gpus > 1 every rank runs it at the same time; default 16 per rank in the headline mode, 0 otherwise)")
    ap.add_argument("--dropin", type=int, default=None, metavar="N",
                    help="also time N calls of Scene.generate() (the drop-in API: host numpy clips + IRs in, scene.audio "
                         "out, synchronous; every rank at the same time with --gpus > 1; default 8 per rank for static configs in the "
                         "headline mode, 0 otherwise)")
    ap.add_argument("--other-configs", type=int, default=1, metavar="0|1",
                    help="on the default single-GPU cfg2 run also time a few steps of cfg3, cfg4 and cfg5 (compact leg "
                         "`other_configs` of the JSON line; 0 = skip)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the scene as one HIP graph (engine.CapturedScene) instead of seven launches")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="time an RCCL gather of the rendered scenes to rank 0 after the timed region (default: on for N > 1)")
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    args = ap.parse_args()

    if args.cpu_workers_sweep:
        print(json.dumps(cpu_workers_sweep(args)), flush=True)
        return
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))   # nothing in this process has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # AL_BENCH_DEVICE pins every rank to ONE device index (records of the N > 1 control flow with the real kernels on a box with
    # fewer GPUs than ranks, profiles/tools/two_ranks_one_gpu_r06.sh; needs AL_DIST_BACKEND=gloo: RCCL refuses two ranks per device)
    device_index = int(os.environ.get("AL_BENCH_DEVICE", local_rank))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                 f"(python bench.py --gpus N spawns them itself)")
    if args.gather is None:
        args.gather = world > 1
    if args.cpu_events is None:
        args.cpu_events = {"cfg2": 64, "cfg3": 1, "cfg4": 32, "cfg5": 4}.get(args.config, 64)
    plain = args.total_scenes == 0 and args.shard == "scenes"
    emulate = os.environ.get("AL_BENCH_EMULATE") == "1"   # tests only: host-emulated kernels, numbers are NOT measurements
    if args.end_to_end is None:     # N > 1: every rank runs the PCIe-inclusive legs at the same time (shared host)
        args.end_to_end = 16 if (plain and not emulate) else 0    # (the batch driver needs page-locked memory and HIP streams)
    if args.dropin is None:
        args.dropin = 8 if plain else 0

    # SURVEY 8d (ii), the all-cores CPU figure: its worker processes are started HERE, before this process initialises the
    # GPU (a GPU-initialised process must not be the parent of an exec on this pool).  Under a profiler the preloaded tool
    # library has ALREADY initialised the GPU when main() starts: the leg is skipped and the JSON says so.
    all_cores = None
    if rank == 0 and world == 1 and plain and args.cpu_workers != 0 and not emulate:
        if under_profiler():
            all_cores = {"skipped": "profiler preload detected (LD_PRELOAD / ROCPROF* / ROCP_TOOL*): this process already holds "
                                    "the GPU, so it must not start worker processes"}
        else:
            from audiblelight_amd import synthetic

            cores, how = physical_cores()
            workers = cores if args.cpu_workers < 0 else args.cpu_workers
            # the container's CPU QUOTA, not the CPUs it can see, is what "all cores" can mean here: the pool's boxes show 256 CPUs
            # under cpu.max = 16 CPUs, and 128 runnable processes on a 16-CPU quota are throttled into a 30x per-process slowdown
            # (profiles/r05b_cpu_workers_sweep.json: the aggregate rate peaks AT the quota -- 44 scene-s/s with 16 workers -- and
            # falls to 15 with 128).  One worker per allowed CPU unless --cpu-workers names a count.
            quota = cpu_quota()["cpus_allowed_by_quota"]
            if args.cpu_workers < 0 and quota:
                workers = max(1, min(workers, int(quota)))
                how += f"; capped at the cgroup CPU quota of {quota:g} CPUs (cpu.max)"
            cfg = synthetic.CONFIGS[args.config]
            try:   # every worker holds one scene's IR tensor, three scene-sized buffers and one event's float64 temporaries:
                   # stay inside half the free memory (a box that runs out of memory dies without a message)
                avail = next(int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable"))
                e_eff = {"cfg3": 1, "cfg5": 2}.get(args.config, cfg["E"])
                per = args.scale * (cfg["C"] * e_eff * cfg["N"] * cfg["Lir"] * 4 + cfg["C"] * cfg["N"] * cfg["Lir"] * 8
                                    + 3 * cfg["C"] * cfg["T"] * cfg["sr"] * 4 + cfg["C"] * (cfg["La"] + cfg["Lir"]) * 60) + 5e8
                workers = max(1, min(workers, int(0.5 * avail / per)))
            except (OSError, StopIteration, ValueError):
                pass
            all_cores = cpu_baseline_all_cores(args.config, args.scale, workers, how, cfg["E"], cfg["E"] * cfg["N"],
                                               max_events=None if args.cpu_scene_events < 0 else (args.cpu_scene_events or cfg["E"]))

    import torch

    if not emulate:
        torch.cuda.set_device(device_index)
    backend = os.environ.get("AL_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only to exercise this path without N GPUs
    use_dist = world > 1 or os.environ.get("AL_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL init with one rank
    dist = None
    if use_dist:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device_index}"), timeout=dist_timeout())
        else:
            dist.init_process_group(backend, timeout=dist_timeout())
    new_event, device_sync = make_timers(emulate, torch)
    coll_dev = "cuda" if (backend == "nccl" and not emulate) else "cpu"
    # N ranks on one node: each takes its share of the host (helper pools capped at usable CPUs / local ranks, the process pinned
    # to the CPUs next to its GPU) instead of sizing planner / cast / pack pools for the whole machine N times over
    host_share = None
    if use_dist:
        from audiblelight_amd import distributed as _dist_mod, engine as _engine_mod

        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        host_share = _dist_mod.host_share(local_rank, local_world, None if emulate else device_index, pin=world > 1)
        _engine_mod.set_thread_cap(host_share["threads"])

    def barrier():
        device_sync()
        if use_dist:
            dist.barrier()
        device_sync()

    def reduce_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(x):
        """[x of rank 0, ..., x of rank N-1] on every rank: proof that the collective saw N ranks."""
        if not use_dist:
            return [x]
        t = [torch.zeros(1, device=coll_dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(t, torch.tensor([x], device=coll_dev, dtype=torch.float64))
        return [float(v.item()) for v in t]

    ctx = dict(args=args, rank=rank, world=world, emulate=emulate, torch=torch, dist=dist, use_dist=use_dist, backend=backend,
               all_cores=all_cores, host_share=host_share,
               new_event=new_event, device_sync=device_sync, barrier=barrier, reduce_max=reduce_max, all_ranks=all_ranks)
    if args.shard == "capsules":
        out = run_capsule_sharded_mode(ctx)
    elif args.total_scenes > 0:
        out = run_scene_batch_mode(ctx)
    else:
        out = run_scene_per_rank_mode(ctx)
    failed = bool(out.pop("_failed", False))
    if host_share is not None:
        threads = all_ranks(float(host_share["threads"]))
        nodes = all_ranks(float(-1 if host_share["numa_node"] is None else host_share["numa_node"]))
        out["host_share"] = dict(host_share, threads_by_rank=[int(x) for x in threads], numa_node_by_rank=[int(x) for x in nodes],
                                 note="helper pools of every rank capped at usable CPUs / local ranks; with N > 1 the process is pinned "
                                      "to the CPUs local to its GPU's NUMA node (audiblelight_amd/distributed.py::host_share)")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if failed:
        sys.exit(1)


def base_record(ctx, scene, pl, value, rep_s, steps, scaling, extra_config):
    args, world, emulate = ctx["args"], ctx["world"], ctx["emulate"]
    med = float(np.median(rep_s))
    return {
        "metric": METRIC, "value": value, "unit": "scene-seconds/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": med / steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if not emulate else "synthetic (HOST EMULATION of the kernels: not a measurement)",
        "timing": {"repeats": len(rep_s), "ms_per_step_each_repeat": [round(x / steps * 1e3, 4) for x in rep_s],
                   "reported": "median repeat; every repeat = exactly `steps` steps between barrier + synchronize, max over ranks",
                   "ms_per_step_by_rank_last_repeat": None},
        "config": dict({"workload": scene.describe(), "scale": args.scale, "log2_block": pl.log2_block,
                        "chunk_events": args.chunk_events, "lanes": args.lanes, "hip_graph": bool(args.graph),
                        "source_hash": source_hash(),
                        # every A/B switch of audiblelight_amd/switches.py that is NOT at its default ({} = the library's own
                        # dispatch policy, al_plan_batch_flags, everywhere): a stray AL_* variable cannot change the timed path unseen
                        "switches": switches_in_effect()}, **extra_config),
    }


def run_scene_per_rank_mode(ctx):
    """The headline: one scene per rank per step (weak scaling)."""
    args, rank, world, emulate, torch = ctx["args"], ctx["rank"], ctx["world"], ctx["emulate"], ctx["torch"]
    from audiblelight_amd import engine, synthetic

    scene = synthetic.make_scene(args.config, scene_index=rank, scale=args.scale)
    r = make_renderer(emulate)
    batch, mix, mix_plan, pl = resident_scene(r, scene, args)
    n_ev = len(scene.clips)
    stages = list(batch.stage_names()) + ["al_mixdown"]
    chunked = len(batch.descs) > 1
    captured = engine.CapturedScene(batch, mix) if args.graph else None

    def step(events=None):
        if captured is not None and events is None:
            captured.replay()
            return
        if chunked:
            batch.run()
            mix.run()
            return
        for i, name in enumerate(stages):
            if events is not None:
                events[i][0].record()
            if name == "al_mixdown":
                mix.run()
            else:
                batch.run_stage(name)
            if events is not None:
                events[i][1].record()

    rep_s, own_s = timed_repeats(step, args.steps, args.warmup, args.repeats, ctx["barrier"], ctx["device_sync"], ctx["reduce_max"])
    # one more pass of K steps with HIP events on the launch stream (torch's current stream is the stream every al_* call
    # is given) around every stage: the per-kernel durations, outside the regions `value` is computed from
    ev = [[(ctx["new_event"](), ctx["new_event"]()) for _ in stages] for _ in range(args.steps)]
    ctx["barrier"]()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(None if chunked else ev[k])
    ctx["device_sync"]()
    instrumented = time.perf_counter() - t0
    by_rank = ctx["all_ranks"](own_s[-1] / args.steps * 1e3)
    batch.result().check_finite()
    # parity of WHAT WAS TIMED: the scene buffer the timed steps wrote, on the host before any other leg runs
    want_parity = rank == 0 and world == 1 and args.parity_events != 0
    timed_scene_host = None
    if want_parity and scene.ambience_beta is None:
        timed_scene_host = np.array(r.mem.download(mix.scene)[: scene.n_capsules * mix_plan.n_samples].reshape(scene.n_capsules, -1))
    elapsed = float(np.median(rep_s))
    ms_per_step = elapsed / args.steps * 1e3
    if chunked:
        kernel_ms = {"al_render_batch+al_mixdown": ms_per_step}
    else:
        kernel_ms = {name: float(np.mean([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(args.steps)]))
                     for i, name in enumerate(stages)}
    dominant = max(kernel_ms, key=kernel_ms.get)
    algo_bytes = scene.algorithmic_bytes()
    achieved = algo_bytes / (kernel_ms[dominant] * 1e-3) / 1e9
    pmc, pmc_note = (load_pmc_traffic(scene.name, pl.log2_block) if args.scale == 1.0 else (None, "reduced scale"))
    out = base_record(ctx, scene, pl, world * args.steps * scene.duration / elapsed, rep_s, args.steps, "weak",
                      {"scenes_per_step_per_gpu": 1})
    out["timing"]["ms_per_step_by_rank_last_repeat"] = [round(x, 4) for x in by_rank]
    out["timing"]["ms_per_step_with_stage_events"] = instrumented / args.steps * 1e3
    # what one step IS, rank for rank: the same C-ABI calls in the same order whether N = 1 or N > 1 (every rank runs this very
    # function on its own scene), so the N = 1 line of a scaling sweep must equal the single-GPU bench line
    out["config"]["step_calls"] = stages if not chunked else ["al_render_batch (chunked)", "al_mixdown"]
    out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS,
                       # the same algorithmic bytes over the WHOLE step (all kernels of the scene), per GPU
                       "path_frac": algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "traffic": (pmc or {}).get(dominant),
                       # HBM rate on the bytes actually moved (PMC traffic / live duration): what the kernel is up against
                       "traffic_rate": ((pmc or {}).get(dominant) or 0) / (kernel_ms[dominant] * 1e-3) / 1e9 or None,
                       "traffic_note": pmc_note,
                       "scene_traffic": sum(v for v in (pmc or {}).values() if isinstance(v, (int, float))) or None,
                       "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kernel_ms,
                       # SURVEY 8(d) names TWO contracts.  `frac` / `path_frac` above are the scene-only one (inputs read once +
                       # scene.audio written once); the API also exposes every event's (C, La) render and the kernels DO write it,
                       # so the full-API contract adds those bytes once: both, so a reader can hold the path to either
                       "contracts": {
                           "scene_only": {"algorithmic_bytes": algo_bytes, "frac": achieved / HBM_PEAK_GBS,
                                          "path_frac": algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "full_api": {"algorithmic_bytes": scene.algorithmic_bytes_full_api(),
                                        "frac": scene.algorithmic_bytes_full_api() / (kernel_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "path_frac": scene.algorithmic_bytes_full_api() / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "reported_as_frac": "scene_only"},
                       "hbm_bytes_per_launch_pmc": pmc}
    if not chunked:
        out["roofline"].update(secondary_roofline(
            pl, mix_plan, kernel_ms, load_pmc_valu(scene.name, pl.log2_block) if args.scale == 1.0 else None,
            out["roofline"]["scene_traffic"], algo_bytes, dominant))
    if args.other_configs and world == 1 and not emulate and args.config == "cfg2" and args.scale == 1.0:
        out["other_configs"] = other_configs_leg(ctx, r)
        if any(not v.get("parity", {}).get("ok", True) for v in out["other_configs"].values()):
            out["_failed"] = True
    multi = ctx if ctx["use_dist"] else None
    if args.end_to_end > 0:
        out["end_to_end"] = end_to_end_leg(r, scene, args.end_to_end, multi)
    if args.dropin > 0 and not emulate and not any(sp.is_moving for sp in scene.specs):
        out["end_to_end_dropin"] = dropin_leg(scene, r, args.dropin, multi)
    if (args.gather and world > 1) or (ctx["use_dist"] and world == 1):
        out["gather"], ok = gather_and_validate(ctx, r, scene_tensor(mix, scene, mix_plan, emulate, torch),
                                                lambda peer: rerender(ctx, r, peer))
        gather_vs_render(out["gather"], ms_per_step, 1, scene, world)
        out["_failed"] = not ok
    if rank == 0 and world == 1:
        ref_scene, ref_events = None, 0
        if args.cpu_events > 0:
            ref_events = min(args.cpu_events, n_ev)
            out["cpu_baseline"], ref_scene = cpu_baseline(scene, ref_events, keep_scene=True)
        if ctx["all_cores"] is not None:
            out["cpu_baseline_all_cores"] = ctx["all_cores"]
            # said at the TOP level of the line: the all-cores figure of the default run renders a bounded sample of every scene and
            # scales it (a baseline only; the one-core `cpu_baseline` above renders the whole scene, extrapolated: false)
            out["cpu_baseline_all_cores_value_extrapolated"] = bool(ctx["all_cores"].get("extrapolated", False))
        if want_parity:
            # the oracle's scene from the cpu_baseline leg IS the reference (no second CPU pass) whenever that leg rendered the
            # events the parity leg asks for with every IR; else the oracle renders `parity_events` events here
            n_par = n_ev if args.parity_events < 0 else min(args.parity_events, n_ev)
            if args.parity_events < 0 and args.config != "cfg2":
                n_par = min({"cfg3": 1, "cfg4": n_ev, "cfg5": 2}.get(args.config, n_ev), n_ev)
            if ref_scene is None or ref_events != n_par:
                ref_scene = oracle_partial_scene(scene, n_par)
            if n_par == n_ev and timed_scene_host is not None:
                got, note = timed_scene_host, "the (C, T) scene buffer the timed steps wrote vs the oracle's mix of ALL events"
            else:
                got = gpu_partial_scene(r, scene, pl, batch.result(), n_par)
                note = (f"GPU mixdown of the first {n_par} of {n_ev} events of the timed render (no ambience) vs the oracle's mix of the "
                        "same events, every IR, all rows")
            out["parity"] = parity_record(got, ref_scene, n_par, note)
            if n_par < n_ev:     # every OTHER event of the timed render meets the oracle on one capsule row
                from oracle import synth_oracle as orc

                cols = lambda e: slice(scene.specs[e].emitter0, scene.specs[e].emitter0 + scene.specs[e].n_emitters)   # noqa: E731
                merge_row_samples(out["parity"], oracle_row_samples(
                    scene, batch.result(), range(n_par, n_ev), lambda e, c: scene.irs[c, cols(e), :],
                    lambda e: orc.emitter_gains(scene.irs[:, cols(e), :])))
            if not out["parity"]["ok"]:
                out["_failed"] = True
    return out


def other_configs_leg(ctx, r):
    """BASELINE configs[2..4] on this GPU, a few steps each, so the driver's own run times them too (their full parity tests
    are tests/test_gpu_full_size.py).  Same step as the headline: every stage of the scene + the mixdown, inputs resident; the
    6.3 GB IR tensors of cfg3 / cfg5 are drawn on the device (torch.randn, same law) instead of on one host core."""
    import argparse as ap

    from audiblelight_amd import synthetic

    torch, args = ctx["torch"], ctx["args"]
    out = {}
    for name, steps in (("cfg3", 10), ("cfg4", 40), ("cfg5", 8)):
        t_in = time.perf_counter()
        scene = synthetic.make_scene(name, scene_index=0, torch_device=f"cuda:{torch.cuda.current_device()}")
        sub = ap.Namespace(log2_block=None, chunk_events=0, lanes=1)
        batch, mix, mix_plan, pl = resident_scene(r, scene, sub)
        stages = list(batch.stage_names()) + ["al_mixdown"]

        def step(events=None):
            for i, st in enumerate(stages):
                if events is not None:
                    events[i][0].record()
                if st == "al_mixdown":
                    mix.run()
                else:
                    batch.run_stage(st)
                if events is not None:
                    events[i][1].record()

        for _ in range(4):      # the first passes over a freshly allocated multi-GB workspace run at a third of the speed
            step()
        reps = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _k in range(steps):
                step()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / steps)
        ev = [[(ctx["new_event"](), ctx["new_event"]()) for _ in stages] for _ in range(steps)]
        for k in range(steps):
            step(ev[k])
        torch.cuda.synchronize()
        batch.result().check_finite()
        kernel_ms = {st: float(np.mean([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(steps)])) for i, st in enumerate(stages)}
        dominant = max(kernel_ms, key=kernel_ms.get)
        ms = float(np.median(reps)) * 1e3
        algo = scene.algorithmic_bytes()
        pmc, _note = load_pmc_traffic(name, pl.log2_block)
        out[name] = {"workload": scene.describe(), "ms_per_step": ms, "value": scene.duration / (ms * 1e-3), "steps": steps,
                     "ms_per_step_each_repeat": [round(x * 1e3, 4) for x in reps], "dominant": dominant,
                     "frac": algo / (kernel_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, "path_frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes": algo, "traffic": sum(v for v in (pmc or {}).values() if isinstance(v, (int, float))) or None,
                     "full_api": {"algorithmic_bytes": scene.algorithmic_bytes_full_api(),
                                  "frac": scene.algorithmic_bytes_full_api() / (kernel_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "path_frac": scene.algorithmic_bytes_full_api() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
                     "inputs": "clips drawn on the host, IR tensor on the device (same law)",
                     "setup_s": None}
        sec = secondary_roofline(pl, mix_plan, kernel_ms, load_pmc_valu(name, pl.log2_block), out[name]["traffic"], algo, dominant)
        out[name].update({k: sec[k] for k in ("gflops", "valu_frac_of_peak", "valu_issue_frac", "traffic_ratio")})
        if args.parity_events != 0:
            # a bounded number of events of THIS render against the oracle, every IR, all capsule rows x all scene samples
            n_par = {"cfg3": 1, "cfg4": 4, "cfg5": 2}[name]
            c_, n_, l_ = scene.ir_shape
            ir_view = scene.irs_dev.reshape(c_, n_, l_)

            def irs_of(e, ir_view=ir_view, scene=scene):
                sp = scene.specs[e]
                return ir_view[:, sp.emitter0: sp.emitter0 + sp.n_emitters, :].cpu().numpy()

            t_par = time.perf_counter()
            ref = oracle_partial_scene(scene, n_par, irs_of)
            got = gpu_partial_scene(r, scene, pl, batch.result(), n_par)
            out[name]["parity"] = parity_record(got, ref, n_par, f"GPU mixdown of the first {n_par} of {len(scene.specs)} events of the timed "
                                                "render (no ambience) vs the oracle's mix of the same events, every IR, all rows")
            out[name]["parity"]["seconds"] = round(time.perf_counter() - t_par, 1)
            # ... and ONE capsule row of every other event of this render against the oracle (gains on the device in float64:
            # normalize_irs needs every capsule's norm, the oracle row only one capsule's samples)

            def gains_of(e, ir_view=ir_view, scene=scene):
                sp = scene.specs[e]
                norms = ir_view[:, sp.emitter0: sp.emitter0 + sp.n_emitters, :].double().pow(2).sum(dim=2).sqrt()      # (C, N)
                return (1.0 / (norms + float(np.finfo(np.float64).tiny)).mean(dim=0)).cpu().numpy()

            def ir_rows_of(e, c, ir_view=ir_view, scene=scene):
                sp = scene.specs[e]
                return ir_view[c, sp.emitter0: sp.emitter0 + sp.n_emitters, :].cpu().numpy()

            merge_row_samples(out[name]["parity"], oracle_row_samples(scene, batch.result(), range(n_par, len(scene.specs)), ir_rows_of, gains_of))
            del ref, got, ir_view
        del batch, mix, scene, ev
        torch.cuda.empty_cache()
        out[name]["setup_s"] = round(time.perf_counter() - t_in - 4 * steps * ms * 1e-3, 1)
    return out


def rerender(ctx, r, peer):
    """The scene rank `peer` renders, rendered here (same seed = same inputs; the path is deterministic)."""
    from audiblelight_amd import synthetic

    args = ctx["args"]
    sc = synthetic.make_scene(args.config, scene_index=peer, scale=args.scale)
    batch, mix, mix_plan, _ = resident_scene(r, sc, args)
    batch.run()
    mix.run()
    ctx["device_sync"]()
    return scene_tensor(mix, sc, mix_plan, ctx["emulate"], ctx["torch"])


def gather_and_validate(ctx, r, mine, rerender_fn, n_items=None, local=None):
    """End-of-job collection on rank 0 (point-to-point at exact sizes: every peer sends over its own xGMI link), timed
    after the timed region, and SELF-VALIDATED: rank 0 renders the scenes of up to two peers itself and compares them with
    what arrived, bit for bit.  Returns (json record, ok)."""
    from audiblelight_amd import distributed

    rank, world, torch = ctx["rank"], ctx["world"], ctx["torch"]
    local = {rank: mine} if local is None else local
    n_items = world if n_items is None else n_items
    ctx["barrier"]()
    g0 = time.perf_counter()
    got = distributed.gather_buffers(local, n_items, dst=0, to_host=False)
    ctx["device_sync"]()
    g_ms = ctx["reduce_max"]((time.perf_counter() - g0) * 1e3)
    seen = ctx["all_ranks"](float(rank))
    rec = {"ms": g_ms, "bytes_total": None, "backend": ctx["backend"], "ranks_seen": sorted(int(x) for x in seen),
           "note": "rendered (C, T) float32 scenes collected on rank 0's device after the timed region (each peer sends over "
                   "its own xGMI link); not part of `value`"}
    ok = True
    if rank == 0:
        ok = sorted(got) == list(range(n_items))
        rec["bytes_total"] = int(sum(v.numel() * 4 for v in got.values()))
        # what crossed the links: everything but rank 0's own items; every peer sends over ITS OWN xGMI link into the root, so the
        # per-link rate is the aggregate divided by the peers (to hold against ~153 GB/s per link, MI355X_MICROARCH.md)
        moved = int(sum(v.numel() * 4 for i, v in got.items() if i % world != 0))
        rec["bytes_over_links"] = moved
        rec["GBps_into_root"] = moved / (g_ms * 1e-3) / 1e9 if g_ms > 0 else None
        rec["GBps_per_peer_link"] = (moved / max(world - 1, 1)) / (g_ms * 1e-3) / 1e9 if (g_ms > 0 and world > 1) else None
        rec["xgmi_link_peak_GBps"] = XGMI_LINK_GBS
        rec["link_frac"] = rec["GBps_per_peer_link"] / XGMI_LINK_GBS if rec["GBps_per_peer_link"] else None
        checked = {}
        for item in sorted({i for i in (1, n_items - 1) if 0 < i < n_items and i % world != 0}):
            want = rerender_fn(item)
            same = tuple(got[item].shape) == tuple(want.shape) and bool(torch.equal(got[item].to(want.device), want))
            checked[str(item)] = same
            ok = ok and same
        rec["validated_against_local_rerender"] = checked
        rec["bit_exact"] = all(checked.values()) if checked else None
    return rec, ok


def gather_vs_render(rec, render_ms, scenes_per_rank, scene, world):
    """The collection held against the rendering it follows: a peer renders `scenes_per_rank` scenes in `render_ms` and then sends them
    over ITS OWN xGMI link.  render_ms_over_gather_ms < 1 means the links, not the kernels, set the pace of a job that collects
    every scene on one rank (cfg2: 369 MB per 2.7 ms scene = 137 GB/s per peer link against ~153 GB/s)."""
    if not rec.get("ms"):
        return
    scene_bytes = scene.n_capsules * round(scene.duration * scene.sr) * 4
    rec["render_ms"] = render_ms
    rec["render_ms_over_gather_ms"] = render_ms / rec["ms"]
    rec["link_GBps_needed_to_keep_up"] = scenes_per_rank * scene_bytes / (render_ms * 1e-3) / 1e9 if world > 1 else None
    rec["note"] += ("; link_GBps_needed_to_keep_up = one rank's rendered bytes / its render time: above xgmi_link_peak_GBps an overlapped "
                    "gather is link-bound, not render-bound")


def end_to_end_leg(r, scene, n, ctx=None):
    """The pipelined batch driver from HOST buffers (PCIe-inclusive, never `value`).  With N > 1 EVERY rank runs it at the same
    time -- each pass starts behind a barrier -- so the ranks contend for host DRAM, the PCIe root complexes and their NUMA nodes
    as a multi-GPU dataset job would (core.py:1828-1847 of the reference: IRs from the host in, scene.audio out): the record then
    carries per-rank rates, the aggregate over the slowest rank's wall time and the aggregate H2D + D2H rate."""
    from audiblelight_amd import batch as batch_mod

    world = ctx["world"] if ctx else 1
    barrier = ctx["barrier"] if ctx else (lambda: None)
    jobs = [batch_mod.SceneJob(specs=scene.specs, clips=scene.clips, irs=scene.irs, starts=scene.starts, ends=scene.ends,
                               duration=scene.duration, sample_rate=scene.sr, name=f"s{i}") for i in range(n)]
    per_scene_ambience = scene.ambience_beta is not None
    if per_scene_ambience:    # cfg5: every scene draws ITS OWN ambience on the device inside the pipeline (nothing is cached)
        from audiblelight_amd import ambience as amb
        from audiblelight_amd.synthesize import _ambience_on_device

        class FreshAmbience:
            def __init__(self, seed):
                self.seed = seed

            def __iter__(self):
                a = amb.Ambience(channels=scene.n_capsules, duration=scene.duration, alias="a", noise=scene.ambience_beta,
                                 ref_db=-65, sample_rate=scene.sr, rng="device", seed=self.seed)
                return iter([_ambience_on_device(r, a, (scene.n_capsules, round(scene.duration * scene.sr)))])

            def __len__(self):
                return 1

        for i, job in enumerate(jobs):
            job.clips = scene.sources()
            job.ambience = FreshAmbience(1000 + i)
    driver = batch_mod.BatchDriver(r)
    consume = lambda name, arr: None   # noqa: E731  (scene.audio delivered as a (C, T) float32 host array)
    driver.run((jobs * 2)[:6], on_scene=consume, copy_for_callback=False)   # warm-up: page-locks every staging slot once
    reps, walls = [], []
    for _ in range(3):
        barrier()
        rep = driver.run(jobs, on_scene=consume, copy_for_callback=False)
        reps.append(rep)
        walls.append(ctx["reduce_max"](rep.wall_s) if ctx else rep.wall_s)    # the slowest rank's wall time of THIS pass
    barrier()
    by_rank = [[v] for v in (r_.scene_seconds_per_second for r_ in reps)]
    if ctx and world > 1:
        by_rank = [ctx["all_ranks"](r_.scene_seconds_per_second) for r_ in reps]
    return end_to_end_record(scene, reps, walls, by_rank, world, per_scene_ambience)


def end_to_end_record(scene, reps, walls, by_rank, world, per_scene_ambience=False):
    """The JSON record of the PCIe-inclusive leg from this rank's BatchReports (`reps`), the slowest rank's wall time of every pass
    (`walls`) and every rank's own rate in every pass (`by_rank`): best pass by the slowest rank's wall time (host-side noise is
    large); `value` = ALL ranks' scene-seconds over that wall time."""
    best = min(range(len(reps)), key=lambda i: walls[i])
    rep = reps[best]
    out = {"value": world * rep.scene_seconds / walls[best], "unit": "scene-seconds/s", "scenes": world * rep.n_scenes,
           "passes": [round(world * r_.scene_seconds / w, 1) for r_, w in zip(reps, walls)],
           "h2d_bytes_per_scene": rep.h2d_bytes // max(rep.n_scenes, 1),
           "d2h_bytes_per_scene": rep.d2h_bytes // max(rep.n_scenes, 1),
           "ir_upload_bound": "one scene's IR tensor over PCIe at 56 GB/s = %.1f ms" % (scene.irs.nbytes / 56e9 * 1e3),
           "ms_per_scene": 1e3 * walls[best] / max(world * rep.n_scenes, 1),
           "ambience": "drawn per scene on the device (Philox), never on the host" if per_scene_ambience else None,
           "note": "host float32 clips+IRs -> H2D -> render -> D2H of scene.audio into page-locked host memory, "
                   "pipelined over scenes (audiblelight_amd/batch.py); PCIe-inclusive, NOT the headline value"}
    if world > 1:
        out.update(
            value_by_rank=[round(v, 1) for v in by_rank[best]], value_per_rank_mean=float(np.mean(by_rank[best])),
            ranks_concurrent=world,
            host_GBps_aggregate={"h2d": world * rep.h2d_bytes / walls[best] / 1e9, "d2h": world * rep.d2h_bytes / walls[best] / 1e9,
                                 "both": world * (rep.h2d_bytes + rep.d2h_bytes) / walls[best] / 1e9},
            note=out["note"] + "; every rank runs this leg AT THE SAME TIME (passes start behind a barrier): `value` = all ranks' scene-"
                               "seconds over the slowest rank's wall time of the best pass, value_by_rank = each rank's own rate in it")
    return out


def run_scene_batch_mode(ctx):
    """BASELINE configs[3]: a batch of S independent scenes split over the ranks (rank r owns scenes r, r + N, ...), every
    scene rendered in full, ALL of them gathered on rank 0 after the timed region.  One step = the whole batch: strong
    scaling.  The inputs of `--input-sets` distinct scenes per rank stay resident and are reused cyclically (generating
    256 x 220 MB of random inputs on the host would take minutes and measure nothing); every scene has its own output."""
    args, rank, world, emulate, torch = ctx["args"], ctx["rank"], ctx["world"], ctx["emulate"], ctx["torch"]
    from audiblelight_amd import distributed, synthetic

    r = make_renderer(emulate)
    mine = distributed.shard_indices(args.total_scenes, rank, world)
    n_sets = max(1, min(4, len(mine)))
    sets = []
    for j in range(n_sets):
        sc = synthetic.make_scene(args.config, scene_index=(mine[j] if j < len(mine) else rank), scale=args.scale)
        sets.append((sc,) + resident_scene(r, sc, args))
    scene0, pl0 = sets[0][0], sets[0][4]
    n_out = scene0.n_capsules * sets[0][3].n_samples
    outputs = [r.mem.empty(n_out) for _ in mine]     # one (C, T) buffer per scene of this rank: what the gather collects
    # every scene's mixdown writes straight into its own output buffer (same tables and inputs as its input set's mixdown)
    mixes = [sets[j % n_sets][2].retarget(out_buf) for j, out_buf in enumerate(outputs)]

    def step():
        for j in range(len(outputs)):
            sets[j % n_sets][1].run()
            mixes[j].run()

    rep_s, own_s = timed_repeats(step, args.steps, args.warmup, args.repeats, ctx["barrier"], ctx["device_sync"], ctx["reduce_max"])
    by_rank = ctx["all_ranks"](own_s[-1] / args.steps * 1e3)
    elapsed = float(np.median(rep_s))
    out = base_record(ctx, scene0, pl0, args.steps * args.total_scenes * scene0.duration / elapsed, rep_s, args.steps, "strong",
                      {"total_scenes": args.total_scenes, "scenes_this_rank": len(mine), "distinct_input_sets_per_rank": n_sets,
                       "note": "one step = the whole batch; every scene's mixdown writes into its own output buffer"})
    out["timing"]["ms_per_step_by_rank_last_repeat"] = [round(x, 4) for x in by_rank]
    algo = scene0.algorithmic_bytes() * args.total_scenes
    out["roofline"] = {"bound": "hbm", "kernel": "whole step (all scenes of the batch)", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "achieved": algo / elapsed * args.steps / 1e9 / world, "frac": algo / elapsed * args.steps / 1e9 / world / HBM_PEAK_GBS,
                       "traffic": None, "algorithmic_bytes_per_step": algo}
    if ctx["use_dist"]:
        as_t = (lambda b: torch.from_numpy(b[:n_out]).reshape(scene0.n_capsules, -1)) if emulate else \
            (lambda b: b[:n_out].reshape(scene0.n_capsules, -1))
        local = {idx: as_t(buf) for idx, buf in zip(mine, outputs)}

        def rerender_item(item):
            j = (item // world) % n_sets     # which input set the owner used for that scene
            owner = item % world
            sc = synthetic.make_scene(args.config, scene_index=distributed.shard_indices(args.total_scenes, owner, world)[j],
                                      scale=args.scale)
            batch, mix, mix_plan, _ = resident_scene(r, sc, args)
            batch.run()
            mix.run()
            ctx["device_sync"]()
            return scene_tensor(mix, sc, mix_plan, emulate, torch)

        out["gather"], ok = gather_and_validate(ctx, r, None, rerender_item, n_items=args.total_scenes, local=local)
        gather_vs_render(out["gather"], float(np.median(rep_s)) / args.steps * 1e3, len(mine), scene0, world)
        # ... and the same collection OVERLAPPED with one more step of rendering: every scene is sent the moment it is
        # enqueued, rank 0 has its receives posted up front (distributed.render_and_gather_overlapped).  gather_overlap_ms is what
        # the collection costs a step once it hides behind the rendering (max over ranks), against `ms` above for doing it afterwards.
        def render_into(idx):
            j = mine.index(idx)
            sets[j % n_sets][1].run()
            mixes[j].run()
            return as_t(outputs[j])

        ctx["barrier"]()
        t0 = time.perf_counter()
        got = distributed.render_and_gather_overlapped(render_into, args.total_scenes, (scene0.n_capsules, sets[0][3].n_samples))
        ctx["device_sync"]()
        over_ms = ctx["reduce_max"]((time.perf_counter() - t0) * 1e3)
        step_ms = float(np.median(rep_s)) / args.steps * 1e3
        rec = {"step_with_gather_ms": over_ms, "step_alone_ms": step_ms, "gather_overlap_ms": over_ms - step_ms}
        if rank == 0:
            same = sorted(got) == list(range(args.total_scenes))
            for item in sorted({i for i in (1, args.total_scenes - 1) if 0 < i < args.total_scenes and i % world != 0}):
                want = rerender_item(item)
                same = same and bool(torch.equal(got[item].to(want.device), want))
            rec["bit_exact"] = same
            ok = ok and same
        out["gather"]["overlapped"] = rec
        out["_failed"] = not ok
    return out


def run_capsule_sharded_mode(ctx):
    """SURVEY 8e row 2 / cfg5 secondary: ONE scene per step, rank r owns capsule rows capsule_slice(C, r, N) of every event;
    the only exchanges are two (three all-reduce calls: SUM, SUM, MAX) of per-emitter / per-event scalars on device arrays (timed separately); every
    rank mixes its own rows.  After the timed region the rows are gathered on rank 0 and compared with rank 0's own render
    of the WHOLE scene (float32 rounding of the level law's partial sums: tolerance, not bit-exactness)."""
    args, rank, world, emulate, torch = ctx["args"], ctx["rank"], ctx["world"], ctx["emulate"], ctx["torch"]
    from audiblelight_amd import distributed, plan as planning, synthetic

    scene = synthetic.make_scene(args.config, scene_index=0, scale=args.scale)    # the SAME scene on every rank
    r = make_renderer(emulate)
    C = scene.n_capsules
    sl = distributed.capsule_slice(C, rank, world)
    rows = sl.stop - sl.start
    batch = distributed.prepare_capsule_sharded(r, scene.specs, scene.sources(), np.ascontiguousarray(scene.irs[sl]), C, scene.sr,
                                                log2_block=args.log2_block)
    pl = batch.plan
    n_ev = len(scene.clips)
    mix_plan = planning.plan_mixdown(scene.starts, scene.ends, [len(c) for c in scene.clips], [rows] * n_ev, pl.events["out_off"],
                                     list(range(n_ev)), scene.duration, scene.sr, rows)
    mix = r.prepare_mixdown(mix_plan, batch.result(), [])
    timers = {"make_event": ctx["new_event"]}

    def step(t=None):
        distributed.run_capsule_sharded(r, batch, C, timers=t)
        mix.run()

    rep_s, own_s = timed_repeats(step, args.steps, args.warmup, args.repeats, ctx["barrier"], ctx["device_sync"], ctx["reduce_max"])
    ctx["barrier"]()
    for _ in range(args.steps):     # instrumented pass: the two exchanges between HIP events on the launch stream
        step(timers)
    ctx["device_sync"]()
    coll = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in timers.items() if k != "make_event"}
    by_rank = ctx["all_ranks"](own_s[-1] / args.steps * 1e3)
    elapsed = float(np.median(rep_s))
    out = base_record(ctx, scene, pl, args.steps * scene.duration / elapsed, rep_s, args.steps, "strong",
                      {"shard": "capsules", "capsules_this_rank": rows, "scenes_per_step": 1})
    out["timing"]["ms_per_step_by_rank_last_repeat"] = [round(x, 4) for x in by_rank]
    out["collectives_ms"] = dict(coll, note="mean device time per step between HIP events around each exchange (includes "
                                            "waiting for the slowest rank): per-emitter IR norm sums (SUM) before the accumulate, "
                                            "E x {sum|x|, #non-finite} (SUM) + E x max|x| (MAX) before the level law")
    algo = scene.algorithmic_bytes()
    out["roofline"] = {"bound": "hbm", "kernel": "whole step", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "achieved": algo / world / (elapsed / args.steps) / 1e9,
                       "frac": algo / world / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "algorithmic_bytes_per_launch": algo}
    if ctx["use_dist"]:
        mine = mix.scene[: rows * mix_plan.n_samples].reshape(rows, -1)
        mine = torch.from_numpy(mine) if emulate else mine
        ctx["barrier"]()
        g0 = time.perf_counter()
        got = distributed.gather_buffers({rank: mine}, world, dst=0, to_host=False)
        ctx["device_sync"]()
        g_ms = ctx["reduce_max"]((time.perf_counter() - g0) * 1e3)
        rec = {"ms": g_ms, "backend": ctx["backend"], "ranks_seen": sorted(int(x) for x in ctx["all_ranks"](float(rank)))}
        if rank == 0:
            whole = torch.cat([got[i] for i in range(world)], dim=0)
            single = synthetic.make_scene(args.config, scene_index=0, scale=args.scale)
            b1, m1, mp1, _ = resident_scene(r, single, argparse.Namespace(**dict(vars(args), chunk_events=0, lanes=1)))
            # resident_scene adds cfg5's ambience; the sharded step does not mix one: compare event mixes only
            m1 = r.prepare_mixdown(mp1, b1.result(), [])
            b1.run()
            m1.run()
            ctx["device_sync"]()
            ref = scene_tensor(m1, single, mp1, emulate, torch).to(whole.device)
            err = float(((whole - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()).item())
            rec.update(rows_total=int(whole.shape[0]), rel_rms_vs_single_gpu_render=err, within_tolerance=err < 1e-5)
            out["_failed"] = not (whole.shape == ref.shape and err < 1e-5)
        out["gather"] = rec
    return out


if __name__ == "__main__":
    main()
