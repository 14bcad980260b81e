#!/usr/bin/env python
"""A dataset of synthetic scenes through the pipelined driver, on one GPU or one process per GPU.

    python examples/render_dataset.py out/ 16                                           # one GPU
    python examples/render_dataset.py out/ 256 --gpus 8                                 # 8 GPUs: starts one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/render_dataset.py out/ 256                                             # the same under a launcher

What the reference does in a serial loop (scripts/generate/benchmark.py:44-77, scripts/seld/generate_dataset.py:96-260):
one folder per scene with a WAV per microphone ((T, C) PCM_16 frames) and ``metadata_out.json``; folders that exist are
skipped, so an interrupted run can simply be started again.  Scenes are independent: the ranks share nothing but the
output directory (no collective).  Every rank walks the same scene stream and builds only its own scenes.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_scene(index: int):
    from audiblelight_amd import ambience, augmentation, core

    rng = np.random.default_rng(1234 + index)
    sr, n_caps, ir_len, duration = 24000, 4, 12000, 10.0
    decay = np.exp(-np.arange(ir_len) / (ir_len / 6.9))
    n_events = int(rng.integers(2, 6))
    irs = (rng.standard_normal((n_caps, n_events, ir_len)) * decay).astype(np.float32)
    scene = core.Scene(duration=duration, state=core.StaticIRState({"mic000": irs}), sample_rate=sr, ref_db=-65)
    for k in range(n_events):
        clip = rng.standard_normal(int(rng.uniform(1.0, 3.0) * sr)).astype(np.float32)
        fx = [augmentation.Gain(sr, gain_db=float(rng.uniform(-6, 0)))] if k % 2 else []
        scene.add_event(core.Event(f"event{k:03d}", clip, sr, snr=float(rng.uniform(5, 30)),
                                   scene_start=float(rng.uniform(0, duration - 3.0)), augmentations=fx))
    scene.add_ambience(ambience.Ambience(channels=n_caps, duration=duration, alias="white", noise="white", ref_db=-65,
                                         sample_rate=sr, seed=index))
    return scene


def spawn_ranks(n: int, argv) -> int:
    """One fresh child process per GPU, started BEFORE this process has touched the GPU (a process that has initialised the
    GPU must never exec another program); the children share nothing but the output directory, so there is no rendezvous."""
    import subprocess

    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        env.setdefault("AL_AMBIENCE_RNG", "device")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    codes = [p.wait() for p in procs]
    return next((c for c in codes if c), 0)


def main(out_dir="dataset_out", n_scenes="8", *flags):
    if "--gpus" in flags and "RANK" not in os.environ:
        n = int(flags[flags.index("--gpus") + 1])
        if n > 1:
            sys.exit(spawn_ranks(n, [out_dir, n_scenes]))
    # ambience noise drawn on the device (Philox) unless the caller asks for the reference's host PCG64 realisation
    os.environ.setdefault("AL_AMBIENCE_RNG", "device")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    from audiblelight_amd import batch

    scenes = ((f"scene_{i:04d}", (lambda i=i: make_scene(i))) for i in range(int(n_scenes)))
    t0 = time.perf_counter()
    rep = batch.render_dataset(scenes, out_dir, rank=rank, world_size=world)
    dt = time.perf_counter() - t0
    print(f"rank {rank}/{world}: {rep.n_scenes} microphone renders, {rep.scene_seconds:.0f} scene-seconds in {dt:.2f} s "
          f"({rep.scene_seconds / dt:.0f} scene-s/s), {len(rep.skipped)} scenes skipped (already on disk)")


if __name__ == "__main__":
    main(*sys.argv[1:])
