#!/usr/bin/env python
"""Headline benchmark: rendered scene-seconds per second (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2]

One "step" = one full pass of the hot path over one synthetic scene of the chosen config (default
cfg2: 60 s scene, 32 capsules, 64 static events, 2 s RIRs, 48 kHz): the equivalents of
render_audio_for_all_scene_events + generate_scene_audio_from_events with clips, IRs and tables
already resident in HBM.  For N > 1 every rank renders its own scene (weak scaling, no data-path
collective; the optional end-of-job gather is timed separately); rank 0 prints ONE JSON line.
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


def _cpu_worker(job):
    """One event through the oracle in a worker process (all-cores leg of the CPU baseline)."""
    from oracle import synth_oracle as orc

    clip, h, snr, ref_db, moving, duration, sr = job
    t0 = time.perf_counter()
    orc.render_event(clip, h, snr, ref_db, moving, duration, sr)
    return time.perf_counter() - t0


def cpu_baseline_all_cores(scene, workers: int, n_irs_cap: int = 8):
    """the oracle on `workers` host processes, one event each (events are independent, so a
    scene's events spread over the cores); rate = workers events per wall time, scaled to the scene's event count."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    jobs, work, full = [], 0, 0
    for i in range(workers):
        sp = scene.specs[i % len(scene.specs)]
        n_used = min(sp.n_emitters, n_irs_cap) if sp.is_moving else sp.n_emitters
        h = scene.irs[:, sp.emitter0: sp.emitter0 + n_used, :].astype(np.float64)
        jobs.append((scene.clips[i % len(scene.clips)], h, sp.snr, sp.ref_db, sp.is_moving, sp.duration, scene.sr))
        work += max(n_used, 1)
    for sp in scene.specs:
        full += max(sp.n_emitters, 1)
    # spawn, not fork: this process has initialised the GPU; a broken worker raises instead of hanging
    with ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as pool:
        list(pool.map(_cpu_worker, jobs))               # warm: imports, page faults
        t0 = time.perf_counter()
        list(pool.map(_cpu_worker, jobs))
        wall = time.perf_counter() - t0
    total = wall * full / work
    return dict(value=scene.duration / total, unit="scene-seconds/s", cores=workers, kind="port", cpu_model=cpu_model(),
                sample=f"{workers} events of one {scene.name} scene rendered concurrently, one oracle process each "
                       f"({wall:.1f} s wall), scaled linearly to the scene's events x IRs; mixdown not included")


def cpu_baseline(scene, n_events: int, n_irs_cap: int = 8):
    """Time the float64 numpy/scipy oracle (kind "port") on a bounded sample of the same workload."""
    from oracle import synth_oracle as orc

    t0 = time.perf_counter()
    spatials, work, full_work = [], 0, 0
    for i in range(n_events):
        sp = scene.specs[i]
        n_used = min(sp.n_emitters, n_irs_cap) if sp.is_moving else sp.n_emitters
        h = scene.irs[:, sp.emitter0: sp.emitter0 + n_used, :].astype(np.float64)
        spatials.append(orc.render_event(scene.clips[i], h, sp.snr, sp.ref_db, sp.is_moving, sp.duration, scene.sr)["spatial"])
        work += n_used
    for sp in scene.specs:
        full_work += max(sp.n_emitters, 1)
    t_events = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.mix_scene(spatials, list(zip(scene.starts[:n_events], scene.ends[:n_events])), scene.duration, scene.sr,
                  keep_padded=True)
    t_mix = time.perf_counter() - t0
    total = t_events * full_work / max(work, 1) + t_mix * len(scene.specs) / n_events
    moving = any(sp.is_moving for sp in scene.specs)
    return dict(value=scene.duration / total, unit="scene-seconds/s", cores=1, kind="port", cpu_model=cpu_model(),
                sample=f"{n_events} of {len(scene.specs)} events of one {scene.name} scene"
                       + (f" with {n_irs_cap} of {scene.specs[0].n_emitters} IRs each" if moving else "")
                       + f" (oracle render_event + mixdown incl. per-event padded copies): {t_events + t_mix:.1f} s measured, "
                         f"scaled linearly in events x IRs")


def load_pmc_traffic(config: str, log2_block: int):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
        return table.get(f"{config}/log2_block={log2_block}")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; reported in config)")
    ap.add_argument("--log2-block", type=int, default=None)
    ap.add_argument("--chunk-events", type=int, default=0, help="events per chunk (0 = whole scene in one batch)")
    ap.add_argument("--lanes", type=int, default=1, help="workspaces / HIP streams the chunks alternate over")
    ap.add_argument("--cpu-events", type=int, default=16, help="events timed for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, metavar="N",
                    help="also time the oracle on N host processes at once (all-cores CPU figure; -1 = one per core, max 64)")
    ap.add_argument("--end-to-end", type=int, default=0, metavar="N",
                    help="also run N scenes through the pipelined batch driver from HOST buffers (PCIe-inclusive rate)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the scene as one HIP graph (engine.CapturedScene) instead of seven launches; the per-stage "
                         "times then come from a separate eager pass")
    ap.add_argument("--gather", action="store_true", help="also time an RCCL gather of the rendered scenes to rank 0")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("AL_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only to exercise this path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    from audiblelight_amd import engine, plan as planning, synthetic

    scene = synthetic.make_scene(args.config, scene_index=rank, scale=args.scale)
    r = engine.Renderer()
    pl = planning.plan_batch(scene.specs, scene.n_capsules, scene.ir_len, scene.sr, log2_block=args.log2_block)
    batch = r.prepare(pl, scene.clips, scene.irs, chunk_events=args.chunk_events, lanes=args.lanes)
    n_ev = len(scene.clips)
    mix_plan = planning.plan_mixdown(scene.starts, scene.ends, [len(c) for c in scene.clips], [scene.n_capsules] * n_ev,
                                     pl.events["out_off"], list(range(n_ev)), scene.duration, scene.sr, scene.n_capsules)
    ambience = []
    if scene.ambience_beta is not None:  # cfg5: white ambience generated on the device, once (inputs resident)
        from audiblelight_amd import ambience as amb
        from audiblelight_amd.synthesize import _ambience_on_device

        a = amb.Ambience(channels=scene.n_capsules, duration=scene.duration, alias="bench", noise=scene.ambience_beta,
                         ref_db=-65, sample_rate=scene.sr)
        ambience = [_ambience_on_device(r, a, (scene.n_capsules, mix_plan.n_samples))]
    mix = r.prepare_mixdown(mix_plan, batch.result(), ambience)
    stages = list(batch.STAGES) + ["al_mixdown"]
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

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in stages]
          for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(None if (chunked or captured is not None) else ev[k])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if captured is not None:  # per-stage durations from an eager pass outside the timed region
        for k in range(args.steps):
            step(ev[k])
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    batch.result().check_finite()

    if chunked:
        kernel_ms = {"al_render_batch+al_mixdown": elapsed / args.steps * 1e3}
    else:
        kernel_ms = {name: float(np.mean([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(args.steps)]))
                     for i, name in enumerate(stages)}
    dominant = max(kernel_ms, key=kernel_ms.get)
    algo_bytes = scene.algorithmic_bytes()
    achieved = algo_bytes / (kernel_ms[dominant] * 1e-3) / 1e9
    pmc = load_pmc_traffic(scene.name, pl.log2_block) if args.scale == 1.0 else None
    out = {
        "metric": METRIC,
        "value": world * args.steps * scene.duration / elapsed,
        "unit": "scene-seconds/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": scene.describe(), "scale": args.scale, "log2_block": pl.log2_block,
                   "scenes_per_step_per_gpu": 1, "chunk_events": args.chunk_events, "lanes": args.lanes,
                   "hip_graph": bool(args.graph)},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (pmc or {}).get(dominant),
                     # HBM rate on the bytes actually moved (PMC traffic / live duration): what the kernel is up against
                     "traffic_rate": ((pmc or {}).get(dominant) or 0) / (kernel_ms[dominant] * 1e-3) / 1e9 or None,
                     "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kernel_ms,
                     "hbm_bytes_per_launch_pmc": pmc},
    }
    if args.end_to_end > 0:
        from audiblelight_amd import batch as batch_mod

        jobs = [batch_mod.SceneJob(specs=scene.specs, clips=scene.clips, irs=scene.irs, starts=scene.starts, ends=scene.ends,
                                   duration=scene.duration, sample_rate=scene.sr, name=f"s{i}") for i in range(args.end_to_end)]
        driver = batch_mod.BatchDriver(r)
        driver.run(jobs[:3], on_scene=lambda name, arr: None)   # warm-up: page-locks the staging buffers once
        rep = driver.run(jobs, on_scene=lambda name, arr: None)
        out["end_to_end"] = {"value": rep.scene_seconds_per_second, "unit": "scene-seconds/s", "scenes": rep.n_scenes,
                             "h2d_bytes_per_scene": rep.h2d_bytes // max(rep.n_scenes, 1),
                             "d2h_bytes_per_scene": rep.d2h_bytes // max(rep.n_scenes, 1),
                             "note": "host float32 clips+IRs -> pinned staging -> H2D -> render -> D2H of scene.audio, pipelined "
                                     "over scenes (audiblelight_amd/batch.py); PCIe-inclusive, NOT the headline value"}
    if args.gather and world > 1:
        from audiblelight_amd import distributed

        torch.cuda.synchronize()
        g0 = time.perf_counter()
        scene_t = mix.scene[: scene.n_capsules * mix_plan.n_samples].reshape(scene.n_capsules, -1)
        distributed.gather_buffers({rank: scene_t}, world, dst=0)
        torch.cuda.synchronize()
        out["gather"] = {"ms": (time.perf_counter() - g0) * 1e3, "bytes_per_rank": int(scene_t.numel() * 4),
                         "note": "RCCL gather of one (C, T) float32 scene per rank to rank 0 (includes the D2H on the root)"}
    if rank == 0:
        if world == 1 and args.cpu_events > 0:
            out["cpu_baseline"] = cpu_baseline(scene, min(args.cpu_events, n_ev))
            if args.cpu_workers != 0:
                workers = min(os.cpu_count() or 1, 64) if args.cpu_workers < 0 else args.cpu_workers
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(scene, workers)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
