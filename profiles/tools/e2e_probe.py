"""Where the host time of one pipelined scene goes (BatchDriver stages, cfg2)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from audiblelight_amd import batch as B, engine, synthetic, plan as planning

sc = synthetic.make_scene("cfg2")
r = engine.Renderer()
job = B.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name="s")
d = B.BatchDriver(r)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{label:28s} host {1e3*(t1-t0):7.2f} ms   +gpu drain {1e3*(t2-t1):7.2f} ms", flush=True); return out
for rep in range(4):
    print("--- pass", rep)
    T("plan_batch", lambda: planning.plan_batch(job.specs, 32, sc.ir_len, sc.sr))
    st = T("_stage", lambda: d._stage(job, rep % 5))
    st = T("_render", lambda: d._render(st))
    st = T("_download(scene)", lambda: d._download(st, False, True, "PCM_16"))
    T("landed.sync", lambda: st["landed"].synchronize())
    T("check_finite", lambda: st["result"].check_finite())
    del st
print(torch.cuda.memory_reserved() / 1e9, "GB reserved")
t0 = time.perf_counter(); rep = d.run([job] * 10, on_scene=lambda n, a: None, copy_for_callback=False); print("run x10:", (time.perf_counter() - t0) / 10 * 1e3, "ms/scene")
