"""Which concurrent activity slows the IR upload inside the pipelined driver?  Same H2D with the scene download and/or
the render switched off."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np, torch
from audiblelight_amd import batch as B, engine, synthetic
sc = synthetic.make_scene("cfg2")
r = engine.Renderer()
jobs = [B.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs.copy() if i else sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}") for i in range(4)]
_sw.set_env("AL_H2D", "blocking")
class NoDown(B.BatchDriver):
    def _download(self, st, want_frames, want_scene, subtype):
        ev = self.torch.cuda.Event(); ev.record(); st.update(landed=ev, host=self._pinned_buffer("scene", self.torch.float32, 32 * 2880000, 0)); return st
class NoRender(B.BatchDriver):
    def _render(self, st):
        if not hasattr(self, "_cached"):
            self._cached = super()._render(st)
        st.update(result=self._cached["result"], scene=self._cached["scene"]); return st
class Neither(NoRender, NoDown):
    pass
for cls in (B.BatchDriver, NoDown, NoRender, Neither):
    dd = cls(r)
    kw = dict(on_scene=lambda n, a: None, copy_for_callback=False, check_finite=False)
    dd.run([jobs[i % 4] for i in range(6)], **kw)
    t0 = time.perf_counter(); rep = dd.run([jobs[i % 4] for i in range(16)], **kw); w = time.perf_counter() - t0
    print(cls.__name__, {k: round(v / 16 * 1e3, 2) for k, v in rep.host_s.items()}, "ms/scene total", round(w / 16 * 1e3, 2), flush=True)
