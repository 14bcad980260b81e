import sys, numpy as np
sys.path.insert(0, "/root/repo")
from audiblelight_amd import _hip, engine, plan as planning
from tests import hostemu
r = engine.Renderer(lib=_hip.Library(sys.argv[1]), memory=hostemu.NumpyMemory())
rng = np.random.default_rng(0)
C, L, sr = 3, 700, 8000
specs, clips, irs, col = [], [], [], 0
for n, ne in ((1501, 1), (2100, 3), (333, 0), (1025, 1)):
    clips.append(rng.standard_normal(n).astype(np.float32))
    irs.append(rng.standard_normal((C, ne, L)).astype(np.float32))
    specs.append(planning.EventSpec(n_samples=n, n_emitters=ne, snr=10.0, emitter0=col, is_moving=ne > 1, duration=n / sr))
    col += ne
pl = planning.plan_batch(specs, C, L, sr, log2_block=10)
res = r.render(pl, clips, np.concatenate(irs, axis=1), chunk_events=2)
mix = planning.plan_mixdown([0.0, 0.1, 0.2, 0.05], [0.3, 0.4, 0.25, 0.2], [len(c) for c in clips], [C] * 4, pl.events["out_off"], [0, 1, 2, 3], 0.5, sr, C)
scene = r.mem.download(r.mixdown(mix, res))
print("asan run ok", float(np.abs(scene[: C * mix.n_samples]).sum()), res.scales())
