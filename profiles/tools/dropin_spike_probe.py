"""Where do the occasional +30 ms calls of Scene.generate() at cfg2 size go?  Times every TorchMemory method per call."""
import collections, gc, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from audiblelight_amd import core, engine, synthetic, synthesize as syn
scene = synthetic.make_scene("cfg2")
syn.set_renderer(engine.Renderer())
spent = collections.Counter()
def wrap(cls, name):
    real = getattr(cls, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return real(*a, **k)
        finally:
            spent[name] += time.perf_counter() - t0
    setattr(cls, name, timed)
for name in ("empty", "zeros", "upload", "upload_tables", "upload_staged", "download_async", "synchronize", "free_bytes", "upload_beside"):
    if hasattr(engine.TorchMemory, name):
        wrap(engine.TorchMemory, name)
def one():
    sc = core.Scene(scene.duration, core.StaticIRState({"mic000": scene.irs}), sample_rate=scene.sr, ref_db=-65)
    for i, (clip, sp) in enumerate(zip(scene.clips, scene.specs)):
        sc.add_event(core.Event(f"e{i}", clip, scene.sr, snr=sp.snr, scene_start=scene.starts[i]))
    return sc.generate()["mic000"]
for mode in ("gc as it comes", "gc.collect() after every call", "gc disabled + collect after every call"):
    if mode.startswith("gc disabled"): gc.disable()
    for _ in range(3): one()
    print("==", mode, flush=True)
    for i in range(24):
        spent.clear()
        r0, h0 = torch.cuda.memory_reserved(), torch.cuda.memory_stats().get("num_device_alloc", 0)
        counts0 = gc.get_count()
        t0 = time.perf_counter(); one(); dt = (time.perf_counter() - t0) * 1e3
        if "collect" in mode: gc.collect()
        flag = "  <-- SPIKE" if dt > 35 else ""
        if flag or i < 2:
            print(f"call {i:2d} {dt:6.1f} ms  gc counts before {counts0}  reserved {r0 / 1e9:.2f} -> {torch.cuda.memory_reserved() / 1e9:.2f} GB  "
                  f"device mallocs +{torch.cuda.memory_stats().get('num_device_alloc', 0) - h0}  "
                  + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in spent.most_common(5)) + flag, flush=True)
    gc.enable()
