"""cfg2 with the IR cut to fewer taps (P = 12 ... 2 partitions): per-stage ms, capsule-loop accumulate (static_mac=1) vs tile kernels."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np, torch
from audiblelight_amd import engine, synthetic, plan as planning
sc = synthetic.make_scene("cfg2")
r = engine.Renderer()
for taps in [int(t) for t in os.environ.get('TAPS', '96000,88000,80000,64000,56000,40000,24000,12000').split(',')]:
    irs = np.ascontiguousarray(np.concatenate([sc.irs, 0.05 * sc.irs, 0.01 * sc.irs], axis=2)[:, :, :taps])
    c, n, l = irs.shape
    for static in ("1", "0"):
        _sw.set_env("AL_STATIC_MAC", static)
        pl = planning.plan_batch(sc.specs, c, l, sc.sr)
        batch = r.prepare(pl, sc.clips, irs)
        for _ in range(3): batch.run()
        torch.cuda.synchronize()
        names = batch.stage_names(0)
        tot = {}
        for rep in range(20):
            for name in names:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); r.lib.call(name, __import__("ctypes").byref(batch.descs[0]), r.mem.stream()); b.record()
                torch.cuda.synchronize()
                tot[name] = tot.get(name, 0.0) + a.elapsed_time(b) / 20
        print(f"taps {taps} P {pl.n_partitions} static_mac={static}", {k: round(v, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()), 3), flush=True)
        del batch
