import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for name, irs in (("float32", scene.irs), ("float64", irs64)):
    for mode in ("beside", "inline"):
        if mode == "inline":
            keep = engine.Renderer.upload_irs_beside; engine.Renderer.upload_irs_beside = lambda self, a: None
        for _ in range(3): one(irs)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); one(irs); ts.append((time.perf_counter() - t0) * 1e3)
        print(name, mode, "per call ms:", [round(t, 1) for t in ts], "median", round(float(np.median(ts)), 2), flush=True)
        if mode == "inline": engine.Renderer.upload_irs_beside = keep
