"""Round 4: whole batch (forward transforms, accumulate, synthesis, levels) at B = 8192 against B = 16384 with the quad-tile
transforms (csrc/al_quad16.h), over IR lengths of 13..37 partitions of 8192.  Static events, La=192000 @ 48 kHz;  python3 quad16_ir_sweep.py [C E La]  (default 32 32 192000)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from audiblelight_amd import engine, plan as planning

r = engine.Renderer()
rng = np.random.default_rng(0)
C, E, La = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 32, 192000)))
clips = [rng.standard_normal(La).astype(np.float32) for _ in range(E)]
specs = [planning.EventSpec(n_samples=La, n_emitters=1, snr=10.0, emitter0=e) for e in range(E)]
for Lir in [int(x) for x in os.environ.get("LIRS", "90000 100000 115000 131072 150000 172000 192000 230000 300000").split()]:
    irs = torch.randn((C * E, Lir), device="cuda", dtype=torch.float32)        # drawn on the device, as bench.py does
    irs *= torch.exp(-torch.arange(Lir, device="cuda", dtype=torch.float32) / (Lir / 6.9))[None, :]
    irs = irs.reshape(-1)
    row = []
    for lb in (13, 14):
        pl = planning.plan_batch(specs, C, Lir, 48000, log2_block=lb)
        batch = r.prepare(pl, clips, irs, ir_strides=(E * Lir, Lir))
        for _ in range(3):
            batch.run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                batch.run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
        row.append((lb, pl.n_partitions, best))
        del batch
        torch.cuda.empty_cache()
    print(f"Lir={Lir}: " + "  ".join(f"B=2^{lb} P={p}: {t:.3f} ms" for lb, p, t in row) + f"  ratio {row[1][2] / row[0][2]:.3f}", flush=True)
    del irs
