import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np
from audiblelight_amd import core, engine, synthetic, synthesize as syn
scene = synthetic.make_scene("cfg2")
syn.set_renderer(engine.Renderer())
irs64 = scene.irs.astype(np.float64)
def one(irs):
    sc = core.Scene(scene.duration, core.StaticIRState({"mic000": irs}), sample_rate=scene.sr, ref_db=-65)
    for i, (clip, sp) in enumerate(zip(scene.clips, scene.specs)):
        sc.add_event(core.Event(f"e{i}", clip, scene.sr, snr=sp.snr, scene_start=scene.starts[i]))
    return sc.generate()["mic000"]
for threads in (8, 16, 4, 12, 8, 16, 24):
    _sw.set_env("AL_CONVERT_THREADS", str(threads))
    for _ in range(3): one(irs64)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); one(irs64); ts.append((time.perf_counter() - t0) * 1e3)
    print("Scene.generate(), float64 IRs,", threads, "cast threads: median", round(float(np.median(ts)), 2), "ms  min", round(min(ts), 2), flush=True)
