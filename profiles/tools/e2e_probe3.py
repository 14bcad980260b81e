"""Pipelined batch driver, cfg2: DMA download stream vs kernels writing into page-locked host memory; scene / frames output."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np, torch
from audiblelight_amd import batch as B, engine, synthetic
sc = synthetic.make_scene("cfg2")
r = engine.Renderer()
jobs = [B.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs.copy() if i else sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}") for i in range(4)]
os.makedirs("/tmp/e2e_out", exist_ok=True)
for mode in ("dma", "kernel", "dma", "kernel"):
    _sw.set_env("AL_D2H", mode)
    for what, kw in (("scene f32 to callback", dict(on_scene=lambda n, a: None, copy_for_callback=False)),
                     ("PCM_16 wav files", dict(output_dir="/tmp/e2e_out")), ("FLOAT wav files", dict(output_dir="/tmp/e2e_out", subtype="FLOAT"))):
      for writers in ((4, 8) if "output_dir" in kw else (1,)):
        import dataclasses
        dd = B.BatchDriver(r, writers=writers)
        named = lambda n: [dataclasses.replace(jobs[i % 4], name=f"s{i}") for i in range(n)]
        dd.run(named(10), **kw)   # every slot's page-locked buffers exist after this
        t0 = time.perf_counter(); rep = dd.run(named(32), **kw); w = time.perf_counter() - t0
        print(f"{mode:7s} {what:24s} writers={writers}", {k: round(v / 32 * 1e3, 2) for k, v in rep.host_s.items()}, "ms/scene total", round(w / 32 * 1e3, 2), flush=True)
