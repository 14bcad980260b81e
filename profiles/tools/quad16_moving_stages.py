import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from audiblelight_amd import engine, plan as planning
r = engine.Renderer()
rng = np.random.default_rng(0)
C, E, N, Lir = 32, 16, int(sys.argv[1]), int(sys.argv[2])
La, sr = 372000, 48000
clips = [rng.standard_normal(La).astype(np.float32) for _ in range(E)]
specs = [planning.EventSpec(n_samples=La, n_emitters=N, snr=10.0, emitter0=e * N, is_moving=True, duration=La / sr) for e in range(E)]
irs = torch.randn((C * E * N, Lir), device="cuda", dtype=torch.float32).reshape(-1)
for lb in (13, 14):
    pl = planning.plan_batch(specs, C, Lir, sr, log2_block=lb)
    batch = r.prepare(pl, clips, irs, ir_strides=(E * N * Lir, Lir))
    print("lb", lb, "P", pl.n_partitions, "reserved", pl.events["reserved"].tolist()[:4], "n_j", sorted(set(pl.streams["n_j"].tolist())), "parts", None if pl.emitter_parts() is None else np.bincount(pl.emitter_parts()).tolist(), "xspec blocks", pl.xspec_blocks, "yspec", pl.yspec_blocks, "hspec", pl.hspec_blocks)
    for _ in range(2): batch.run()
    torch.cuda.synchronize()
    for name in batch.stage_names():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): batch.run_stage(name)
        b.record(); torch.cuda.synchronize()
        print("   ", name, round(a.elapsed_time(b) / 3, 3))
    del batch; torch.cuda.empty_cache()
