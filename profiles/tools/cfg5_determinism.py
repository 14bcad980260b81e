"""Round 4: the full-size cfg5 scene (128 events x 64 capsules, 4 s RIRs, B = 16384 by the planner, chip fully loaded) rendered
repeatedly: a device-side checksum of the event audio (sum of the float bits as int64) must not change between renders."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from audiblelight_amd import engine, plan as planning, synthetic

r = engine.Renderer()
sc = synthetic.make_scene("cfg5", torch_device="cuda")
pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
c, n, l = sc.ir_shape
batch = r.prepare(pl, sc.sources(), sc.irs_dev, ir_strides=(n * l, l))
seen = {}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(N):
    res = batch.run()
    live = res.spatial[: pl.spatial_floats]
    key = (int(torch.sum(live.view(torch.int32), dtype=torch.int64).item()), int(torch.sum(live[::7].view(torch.int32), dtype=torch.int64).item()))
    seen[key] = seen.get(key, 0) + 1
print(f"cfg5 at log2_block {pl.log2_block}: {N} renders, {len(seen)} distinct checksum(s): {seen}")
assert len(seen) == 1
