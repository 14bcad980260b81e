"""Scene.generate() at cfg2 size, per-call wall time and where the host time of a call goes (cProfile over 8 calls)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from audiblelight_amd import core, engine, synthetic, synthesize as syn
scene = synthetic.make_scene("cfg2")
syn.set_renderer(engine.Renderer())
def one():
    sc = core.Scene(scene.duration, core.StaticIRState({"mic000": scene.irs}), sample_rate=scene.sr, ref_db=-65)
    for i, (clip, sp) in enumerate(zip(scene.clips, scene.specs)):
        sc.add_event(core.Event(f"e{i}", clip, scene.sr, snr=sp.snr, scene_start=scene.starts[i]))
    return sc.generate()["mic000"]
for _ in range(3): one()
times = []
pr = cProfile.Profile(); pr.enable()
for _ in range(12):
    t0 = time.perf_counter(); one(); times.append((time.perf_counter() - t0) * 1e3)
pr.disable()
print("per call ms:", [round(t, 1) for t in times])
print("reserved GB", torch.cuda.memory_reserved() / 1e9, "allocated GB", torch.cuda.memory_allocated() / 1e9)
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
