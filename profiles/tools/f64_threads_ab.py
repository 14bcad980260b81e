"""Batch driver on cfg2 scenes with float64 IR tensors (the reference's get_irs() dtype): how many cast threads in the planner stage?
Also float32 IRs for the PCIe-bound reference point."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np, torch
from audiblelight_amd import engine, synthetic, batch as B
sc = synthetic.make_scene("cfg2")
irs64 = sc.irs.astype(np.float64)
r = engine.Renderer()
def jobs(irs, n=16):
    return [B.SceneJob(specs=sc.specs, clips=sc.clips, irs=irs, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}") for i in range(n)]
def run(irs, label):
    drv = B.BatchDriver(r)
    drv.run(jobs(irs, 6), on_scene=lambda n, a: None, copy_for_callback=False)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); rep = drv.run(jobs(irs), on_scene=lambda n, a: None, copy_for_callback=False); best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best / 16 * 1e3:.1f} ms/scene", {k: round(v / 16 * 1e3, 1) for k, v in rep.host_s.items()}, flush=True)
run(sc.irs, "float32 IRs")
for threads in [int(x) for x in os.environ.get("CAST_THREADS", "4,8,12,16,24,32,8").split(",")]:
    _sw.set_env("AL_CONVERT_THREADS", str(threads))
    run(irs64, f"float64 IRs, {threads:2d} cast threads")
