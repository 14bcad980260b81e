"""cfg4 (BASELINE configs[3]: a batch of independent 30 s scenes, 32 events x 32 capsules x 1 s RIRs): does rendering M scenes per
launch sequence (batch.merge_jobs: events concatenated, IR columns offset, one mixdown per scene) beat M launch sequences of one
scene?  Inputs resident in HBM, kernels only, per-SCENE milliseconds; bit-identical scenes checked.
    python3 profiles/tools/merge_probe.py [cfg4] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
from audiblelight_amd import batch, engine, plan as planning, synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
r = engine.Renderer()
scenes = [synthetic.make_scene(cfg, scene_index=i) for i in range(8)]


def prepared(group):
    jobs = [batch.SceneJob(specs=s.specs, clips=s.clips, irs=s.irs, starts=s.starts, ends=s.ends, duration=s.duration,
                           sample_rate=s.sr, name="s") for s in group]
    specs, clips, irs, ranges = batch.merge_jobs(jobs)
    c, l = irs.shape[0], irs.shape[2]
    pl = planning.plan_batch(specs, c, l, group[0].sr, lib=r.lib)
    b = r.prepare(pl, clips, irs)
    mixes = []
    for s, (e0, n) in zip(group, ranges):
        mp = planning.plan_mixdown(s.starts, s.ends, [len(x) for x in s.clips], [c] * n, pl.events["out_off"][e0: e0 + n],
                                   list(range(e0, e0 + n)), s.duration, s.sr, c, lib=r.lib)
        mixes.append(r.prepare_mixdown(mp, b.result(), []))
    return pl, b, mixes


ref = None
for m in (1, 2, 4, 8):
    groups = [prepared(scenes[i: i + m]) for i in range(0, 8, m)]

    def step():
        for _, b, mixes in groups:
            b.run()
            for mx in mixes:
                mx.run()

    for _ in range(3):
        step()
    reps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / steps / 8 * 1e3)
    outs = [r.mem.download(mx.scene).copy() for _, _, mixes in groups for mx in mixes]
    if ref is None:
        ref = outs
    same = all(np.array_equal(a, b_) for a, b_ in zip(ref, outs))
    pl = groups[0][0]
    print(f"{cfg}: {m} scene(s) per launch sequence: {np.median(reps):.4f} ms per scene {['%.4f' % x for x in reps]}, B = {pl.block}, "
          f"events per batch {len(pl.events)}, scenes identical to the one-by-one render: {same}")
    del groups
    torch.cuda.empty_cache()
