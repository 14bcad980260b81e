#!/usr/bin/env python
"""Headline benchmark: rendered scene-seconds per second on the cfg2 workload of BASELINE.json.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one 60 s scene (32 capsules, 64 events, 2 s RIRs,
48 kHz): render_audio_for_all_scene_events + generate_scene_audio_from_events equivalents, inputs
already resident in HBM.  For N > 1 every rank renders its own scenes (weak scaling, no data-path
collective); rank 0 prints ONE JSON line.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def cpu_baseline(scene, n_events: int):
    """Time the float64 numpy/scipy oracle (kind "port") on a bounded sample of the same workload."""
    from oracle import synth_oracle as orc

    t0 = time.perf_counter()
    spatials = []
    for i in range(n_events):
        sp = scene.specs[i]
        h = scene.irs[:, sp.emitter0: sp.emitter0 + sp.n_emitters, :].astype(np.float64)
        spatials.append(orc.render_event(scene.clips[i], h, sp.snr, sp.ref_db, sp.is_moving, sp.duration, scene.sr)["spatial"])
    t_events = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.mix_scene(spatials, list(zip(scene.starts[:n_events], scene.ends[:n_events])), scene.duration, scene.sr,
                  keep_padded=True)
    t_mix = time.perf_counter() - t0
    per_event = (t_events + t_mix) / n_events
    total = per_event * len(scene.specs)
    return dict(value=scene.duration / total, unit="scene-seconds/s", cores=1, kind="port",
                sample=f"{n_events} of {len(scene.specs)} events of one {scene.name} scene (render_event + mixdown with "
                       f"per-event padded copies), {t_events + t_mix:.1f} s measured, scaled linearly in events")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; reported in config)")
    ap.add_argument("--log2-block", type=int, default=None)
    ap.add_argument("--chunk-events", type=int, default=0, help="events per chunk (0 = whole scene in one batch)")
    ap.add_argument("--cpu-events", type=int, default=6, help="events timed for the CPU baseline (0 = skip)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    from audiblelight_amd import engine, plan as planning, synthetic

    scene = synthetic.make_scene(args.config, scene_index=rank, scale=args.scale)
    r = engine.Renderer()
    pl = planning.plan_batch(scene.specs, scene.n_capsules, scene.ir_len, scene.sr, log2_block=args.log2_block)
    batch = r.prepare(pl, scene.clips, scene.irs, chunk_events=args.chunk_events)
    mix_plan = planning.plan_mixdown(scene.starts, scene.ends, [len(c) for c in scene.clips],
                                     [scene.n_capsules] * len(scene.clips), pl.events["out_off"],
                                     list(range(len(scene.clips))), scene.duration, scene.sr, scene.n_capsules)
    mix = r.prepare_mixdown(mix_plan, batch.result())
    stages = list(batch.STAGES) + ["al_mixdown"]

    def step(events=None):
        if len(batch.descs) > 1:
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
        step(ev[k])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    batch.result().check_finite()

    if len(batch.descs) > 1:
        kernel_ms = {"al_render_batch+al_mixdown": elapsed / args.steps * 1e3}
    else:
        kernel_ms = {name: float(np.mean([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(args.steps)]))
                     for i, name in enumerate(stages)}
    dominant = max(kernel_ms, key=kernel_ms.get)
    algo_bytes = scene.algorithmic_bytes()
    achieved = algo_bytes / (kernel_ms[dominant] * 1e-3) / 1e9
    out = {
        "metric": "rendered scene-seconds/sec @48kHz, 32-ch mic, 64 events, 2s RIR",
        "value": world * args.steps * scene.duration / elapsed,
        "unit": "scene-seconds/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{scene.name}: 1 scene/GPU/step, {scene.n_capsules} capsules, {len(scene.specs)} static events, "
                               f"{scene.ir_len / scene.sr:g} s RIR, {scene.clips[0].size / scene.sr:g} s clips, "
                               f"{scene.duration:g} s scene @ {scene.sr} Hz",
                   "scale": args.scale, "log2_block": pl.log2_block, "scenes_per_step_per_gpu": 1,
                   "chunk_events": args.chunk_events},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kernel_ms},
    }
    if rank == 0:
        if world == 1 and args.cpu_events > 0:
            out["cpu_baseline"] = cpu_baseline(scene, min(args.cpu_events, len(scene.specs)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
