import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audiblelight_amd import core, engine, synthetic, synthesize as syn
syn.set_renderer(engine.Renderer())
for cfg in ("cfg1", "cfg4"):
    scene = synthetic.make_scene(cfg)
    def one():
        sc = core.Scene(scene.duration, core.StaticIRState({"mic000": scene.irs}), sample_rate=scene.sr, ref_db=-65)
        for i, (clip, sp) in enumerate(zip(scene.clips, scene.specs)):
            sc.add_event(core.Event(f"e{i}", clip, scene.sr, snr=sp.snr, scene_start=scene.starts[i]))
        return sc.generate()["mic000"]
    for _ in range(5): one()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); one(); ts.append((time.perf_counter() - t0) * 1e3)
    print(cfg, "Scene.generate() median ms", round(float(np.median(ts)), 3), "min", round(min(ts), 3), flush=True)
    if cfg == "cfg1":
        pr = cProfile.Profile(); pr.enable()
        for _ in range(50): one()
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(25)
