"""Does the scene download overlap the next upload if it is ISSUED only once the render has finished (from the writer
thread), instead of being enqueued early behind an event?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np, torch
from audiblelight_amd import batch as B, engine, synthetic
sc = synthetic.make_scene("cfg2")
r = engine.Renderer()
jobs = [B.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs.copy() if i else sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}") for i in range(4)]
class LateDown(B.BatchDriver):
    def _download(self, st, want_frames, want_scene, subtype):
        torch_ = self.torch
        done = torch_.cuda.Event(); done.record()
        drv = self
        c, t = st["mix"].n_capsules, st["mix"].n_samples
        host = self._pinned_buffer("scene", torch_.float32, c * t, st["slot"])
        class Landed:
            def synchronize(_):
                done.synchronize()
                with torch_.cuda.stream(drv.down_stream):
                    host.copy_(st["scene"][: c * t], non_blocking=True)
                drv.down_stream.synchronize()
        st.update(landed=Landed(), host=host)
        return st
for mode in ("blocking", "async"):
    _sw.set_env("AL_H2D", mode)
    for cls in (B.BatchDriver, LateDown):
        dd = cls(r)
        kw = dict(on_scene=lambda n, a: None, copy_for_callback=False, check_finite=False)
        dd.run([jobs[i % 4] for i in range(6)], **kw)
        t0 = time.perf_counter(); rep = dd.run([jobs[i % 4] for i in range(16)], **kw); w = time.perf_counter() - t0
        print(mode, cls.__name__, {k: round(v / 16 * 1e3, 2) for k, v in rep.host_s.items()}, "ms/scene total", round(w / 16 * 1e3, 2), flush=True)
