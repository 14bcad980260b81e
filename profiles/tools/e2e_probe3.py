import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from audiblelight_amd import batch as B, engine, synthetic
sc = synthetic.make_scene("cfg2")
r = engine.Renderer()
jobs = [B.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs.copy() if i else sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}") for i in range(4)]
for mode in ("blocking", "async"):
    os.environ["AL_H2D"] = mode
    dd = B.BatchDriver(r)
    kw = dict(on_scene=lambda n, a: None, copy_for_callback=False)
    dd.run([jobs[i % 4] for i in range(6)], **kw)
    t0 = time.perf_counter(); rep = dd.run([jobs[i % 4] for i in range(16)], **kw); w = time.perf_counter() - t0
    print(mode, {k: round(v / 16 * 1e3, 2) for k, v in rep.host_s.items()}, "ms/scene total", round(w / 16 * 1e3, 2))
