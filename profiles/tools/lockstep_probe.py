"""Long clips (K > 24 blocks: several k-tile pairs per (event, bin tile)): accumulate time per library variant."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from audiblelight_amd import engine, plan as planning
r = engine.Renderer()
rng = np.random.default_rng(0)
for La in (192000, 300000, 400000, 960000):
    C, E, Lir = 32, 64, 96000
    if La > 500000: E = 24
    clips = [rng.standard_normal(La).astype(np.float32) for _ in range(E)]
    irs = (rng.standard_normal((C, E, Lir), dtype=np.float32) * np.exp(-np.arange(Lir) / (Lir / 6.9)).astype(np.float32))
    specs = [planning.EventSpec(n_samples=La, n_emitters=1, snr=10.0, emitter0=e) for e in range(E)]
    pl = planning.plan_batch(specs, C, Lir, 48000)
    batch = r.prepare(pl, clips, irs)
    res = batch.run(); torch.cuda.synchronize()
    ref = r.mem.download(batch.bufs["spatial"])[:100000].copy()
    t = 0.0
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r.lib.call("al_spectral_mac", ctypes.byref(batch.descs[0]), r.mem.stream()); b.record()
        torch.cuda.synchronize(); t += a.elapsed_time(b) / 10
    batch.run(); torch.cuda.synchronize()
    same = np.array_equal(ref, r.mem.download(batch.bufs["spatial"])[:100000])
    print(os.path.basename(r.lib.path), f"La={La} K={int(pl.events['n_blocks'].max())} E={E} accumulate {t:.3f} ms, rerun identical: {same}", flush=True)
    del batch, irs, clips
